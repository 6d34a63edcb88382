// dsim_device.h — per-drone device arithmetic (fp32, one drone per lane) for gfx950.
//
// Everything here is small-vector arithmetic on registers: no MFMA (there is no
// dense contraction on this path), no LDS needed for the arithmetic itself.
// The functions mirror, one for one, the reference functions listed in
// SURVEY.md 8(a); each cites the reference lines it computes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dsim_math.h"

#define DSIM_MAX_ACT 6
#ifndef DSIM_HEXA_BASE_CONST64
#define DSIM_HEXA_BASE_CONST64 1
#endif
#define DSIM_DEV_KIND_QUAD 0
#define DSIM_DEV_KIND_HEXA 1
#define DSIM_DEV_KIND_HEXA_QUADLAW 2      // morphing-hexa physics, the quad law on six actuators (hexa_6DOF_simple.urdf)

// fp32 image of dsim_type_params (include/dronesim_amd.h), with the reciprocals
// the kernel wants.  Lives in device memory; with a homogeneous fleet the address
// is wave-uniform so every field is fetched by scalar loads into SGPRs.
struct DevType {
  int32_t kind, n_act;
  float mass, inv_mass;
  float J[3], invJ[3];
  float gyro[3];                              // ((Jz - Jy) / Jx, (Jx - Jz) / Jy, (Jy - Jx) / Jz): Euler's equations (bullet_step_body)
  float kf, km;
  float scale[DSIM_MAX_ACT], cnst[DSIM_MAX_ACT], pmin[DSIM_MAX_ACT], pmax[DSIM_MAX_ACT];
  float rpos[DSIM_MAX_ACT][3], raxis[DSIM_MAX_ACT][3], spin[DSIM_MAX_ACT];
  float rxa[DSIM_MAX_ACT][3];                 // rpos x raxis: torque about the COM per unit thrust of rotor j
  float spax[DSIM_MAX_ACT][3];                // spin_j raxis_j: reaction torque per unit rotor moment (hexa_wrench_noise)
  float rsum[3];                              // quads: sum of the four rotor positions (the lateral-noise lever, quad_wrench_noise)
  float alloc[DSIM_MAX_ACT][DSIM_MAX_ACT];    // quad: pinv(G1/0.05); hexa: M1 (u_opt = M1 v + M4 u0)
  float alloc2[DSIM_MAX_ACT][DSIM_MAX_ACT];   // hexa: M4
  float B[DSIM_MAX_ACT][DSIM_MAX_ACT];        // hexa: G1/0.05, for the active-set fallback
  float kp, kd, katt[3], krate[3];
  float g, clin, cang, maxv;
  float drag[3], gnd_coeff, prop_radius, gnd_hclip, dw[3];
  float reset_thrust, reset_cmd;
  float speed_limit;                          // MAX_SPEED_KMH * 1000/3600 (VelocityAviary.py:92-94)
  float coll_r, coll_below;                   // bounding cylinder of the collision shapes (ground-plane watch)
  float mu_plane;                             // DSIM_OPT_PLANE: Coulomb coefficient against the plane
  float base_off[3];                          // integrated COM -> the reported point (base link COM), body frame
  float watch_below;                          // coll_below seen from the reported point (ground-plane watch)
  float dyn_lever[2][4];                      // Physics.DYN: x / y torque per unit force of rotor i (the mixer of BaseAviary.py:1794-1803 times its lever)
  float weight;                               // Physics.DYN: GRAVITY = G M (BaseAviary.py:226)
  float nchol[22];                            // six-actuator types: Cholesky factor of the body-wrench covariance of the rotor noise, lower
                                              // triangle row by row, over 0.01 (hexa_wrench_z); [21] pads the doubles behind it
  // (at the end: in the middle of the table they moved every offset behind them, and four instances of k_step_runs came back
  // with a 36-byte scratch reservation)
  double raxis64[DSIM_MAX_ACT][3], rxa64[DSIM_MAX_ACT][3];   // (double)raxis, (double)rxa — the fp32 values, widened (hexa_wrench_base)
};

// The table is written once (dsim_create) and only read by kernels, which read it through the CONSTANT address space.  A
// load through a plain global pointer becomes a scalar load only when the compiler can show that nothing in the kernel wrote
// memory in front of it, and that search gives up in the large multi-law kernels: in k_control_runs the ~130 constants of a
// type arrived by VECTOR loads — 105 VGPRs, 4 waves per SIMD, against 57 for the same law alone in a kernel.  A load from
// the constant address space needs no such proof.
typedef const __attribute__((address_space(4))) DevType CDevType;
__device__ __forceinline__ CDevType& dev_type(const DevType* table, int k) {
  return *reinterpret_cast<CDevType*>(reinterpret_cast<uintptr_t>(table + k));
}

struct V3 { float x, y, z; };
struct Q4 { float x, y, z, w; };   // xyzw, w last (dronesim/utils/math.py:6,25,47)
struct M3 { float m[9]; };         // row-major, body -> world

__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
  return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

// p.getMatrixFromQuaternion / btMatrix3x3::setRotation (s = 2/|q|^2); C8
__device__ __forceinline__ M3 matrix_from_quat(Q4 q) {
  const float d = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
  const float s = 2.0f * DSIM_RCP(d);
  const float xs = q.x * s, ys = q.y * s, zs = q.z * s;
  const float wx = q.w * xs, wy = q.w * ys, wz = q.w * zs;
  const float xx = q.x * xs, xy = q.x * ys, xz = q.x * zs;
  const float yy = q.y * ys, yz = q.y * zs, zz = q.z * zs;
  M3 R;
  R.m[0] = 1.0f - (yy + zz); R.m[1] = xy - wz;          R.m[2] = xz + wy;
  R.m[3] = xy + wz;          R.m[4] = 1.0f - (xx + zz); R.m[5] = yz - wx;
  R.m[6] = xz - wy;          R.m[7] = yz + wx;          R.m[8] = 1.0f - (xx + yy);
  return R;
}
__device__ __forceinline__ V3 mul(const M3& R, V3 a) {   // R a
  return v3(R.m[0] * a.x + R.m[1] * a.y + R.m[2] * a.z, R.m[3] * a.x + R.m[4] * a.y + R.m[5] * a.z,
            R.m[6] * a.x + R.m[7] * a.y + R.m[8] * a.z);
}
__device__ __forceinline__ V3 mulT(const M3& R, V3 a) {  // R^T a
  return v3(R.m[0] * a.x + R.m[3] * a.y + R.m[6] * a.z, R.m[1] * a.x + R.m[4] * a.y + R.m[7] * a.z,
            R.m[2] * a.x + R.m[5] * a.y + R.m[8] * a.z);
}

// p.getEulerFromQuaternion (ZYX, gimbal clamp at |sarg| >= 0.99999); C8.
// Returns the angles the controller adds increments to (roll, pitch; yaw only when WANT_YAW)
// and the sines/cosines of all three that the G matrix needs (INDIControl.py:301-305).  The
// sines/cosines come straight from the quaternion products the angles are defined by
// (sin(atan2(a,b)) = a/hypot(a,b), sin(asin(s)) = s, ...), so no trig call is spent on them.
struct Euler { float roll, pitch, yaw; float sph, cph, sth, cth, sps, cps; };
template <bool WANT_YAW>
__device__ __forceinline__ Euler euler_from_quat(Q4 q) {
  const float sqx = q.x * q.x, sqy = q.y * q.y, sqz = q.z * q.z, squ = q.w * q.w;
  const float sarg = -2.0f * (q.x * q.z - q.w * q.y);
  Euler e;
  if (sarg <= -0.99999f || sarg >= 0.99999f) {
    const bool up = sarg > 0.0f;
    const float hs = up ? -q.x : q.x, hc = up ? q.y : -q.y;     // yaw = 2 atan2(hs, hc)
    const float n2 = hs * hs + hc * hc;
    const float in2 = n2 > 0.0f ? DSIM_RCP(n2) : 0.0f;
    e.roll = 0.0f; e.sph = 0.0f; e.cph = 1.0f;
    e.pitch = up ? DSIM_PI_2 : -DSIM_PI_2; e.sth = up ? 1.0f : -1.0f; e.cth = 0.0f;
    e.sps = 2.0f * hs * hc * in2;
    e.cps = n2 > 0.0f ? (hc * hc - hs * hs) * in2 : 1.0f;
    e.yaw = WANT_YAW ? 2.0f * dsim_atan2(hs, hc) : 0.0f;
  } else {
    const float ra = 2.0f * (q.y * q.z + q.w * q.x), rb = squ - sqx - sqy + sqz;
    const float ya = 2.0f * (q.x * q.y + q.w * q.z), yb = squ + sqx - sqy - sqz;
    const float ir = DSIM_RSQ(ra * ra + rb * rb), iy = DSIM_RSQ(ya * ya + yb * yb);
    e.roll = dsim_atan2(ra, rb); e.sph = ra * ir; e.cph = rb * ir;
    e.pitch = dsim_asin(sarg); e.sth = sarg; e.cth = DSIM_SQRT(fmaxf(1.0f - sarg * sarg, 0.0f));
    e.sps = ya * iy; e.cps = yb * iy;
    e.yaw = WANT_YAW ? dsim_atan2(ya, yb) : 0.0f;
  }
  return e;
}

// p.getQuaternionFromEuler (half-angle product, normalised); C8
__device__ __forceinline__ Q4 quat_from_euler(V3 e) {
  float sph, cph, sth, cth, sps, cps;
  dsim_sincos(0.5f * e.x, &sph, &cph);
  dsim_sincos(0.5f * e.y, &sth, &cth);
  dsim_sincos(0.5f * e.z, &sps, &cps);
  Q4 q;
  q.x = sph * cth * cps - cph * sth * sps;
  q.y = cph * sth * cps + sph * cth * sps;
  q.z = cph * cth * sps - sph * sth * cps;
  q.w = cph * cth * cps + sph * sth * sps;
  const float inv = DSIM_RSQ(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  q.x *= inv; q.y *= inv; q.z *= inv; q.w *= inv;
  return q;
}

// dronesim/utils/math.py:75-80 norm_ang
__device__ __forceinline__ float norm_ang(float x) {
  while (x > DSIM_PI) x -= 2.0f * DSIM_PI;
  while (x < -DSIM_PI) x += 2.0f * DSIM_PI;
  return x;
}

// ---------------------------------------------------------------------------
// rotor noise: Threefry4x32-12 + Box-Muller (definition mirrored by
// oracle/dsim_oracle.c:orc_noise_normals; the reference's own draws come from
// the unseeded global numpy RNG, BaseAviary.py:1518-1525, and cannot be replayed)
//
// Counter-based: block = Threefry4x32-12(key = seed, counter = (drone, sub-step)).  Threefry (Salmon et al.,
// "Parallel random numbers: as easy as 1, 2, 3", SC'11; 12 rounds = the Crush-resistant form with margin) is
// add / rotate / xor only — ~85 full-rate vector instructions per block.  The round-1 generator, Philox4x32-10,
// needs 40 32-bit multiplies (v_mul_hi_u32 / v_mul_lo_u32, quarter rate on CDNA) and made the 5-sub-step kernel
// VALU-bound on the noise alone.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return __builtin_rotateleft32(x, (uint32_t)r); }
__device__ __forceinline__ void threefry4x32_12(uint32_t x[4], uint32_t k0, uint32_t k1) {
  // key words 2, 3 are zero; ks[4] = parity constant ^ all key words
  const uint32_t ks[5] = {k0, k1, 0u, 0u, 0x1BD11BDAu ^ k0 ^ k1};
  x[0] += ks[0]; x[1] += ks[1]; x[2] += ks[2]; x[3] += ks[3];
#pragma unroll
  for (int r = 0; r < 12; ++r) {
    // rotation constants R_32x4 of Threefry4x32, period 8
    constexpr int R0[8] = {10, 11, 13, 23, 6, 17, 25, 18};
    constexpr int R1[8] = {26, 21, 27, 5, 20, 11, 10, 20};
    if ((r & 1) == 0) {
      x[0] += x[1]; x[1] = rotl32(x[1], R0[r & 7]) ^ x[0];
      x[2] += x[3]; x[3] = rotl32(x[3], R1[r & 7]) ^ x[2];
    } else {
      x[0] += x[3]; x[3] = rotl32(x[3], R0[r & 7]) ^ x[0];
      x[2] += x[1]; x[1] = rotl32(x[1], R1[r & 7]) ^ x[2];
    }
    if ((r & 3) == 3) {                       // key injection after every 4th round
      const int s = r / 4 + 1;
      x[0] += ks[s % 5]; x[1] += ks[(s + 1) % 5]; x[2] += ks[(s + 2) % 5]; x[3] += ks[(s + 3) % 5] + (uint32_t)s;
    }
  }
}
// 16 bits -> two normals of standard deviation SIGMA (Box-Muller): radius from the HIGH byte (u1 = (h + 1/2) / 256), angle from
// the LOW byte (u2 = (l + 1/2) / 256 revolutions) — the lattice points sit at the CENTRES of the 256 x 256 cells of the unit square
// (round 6; up to round 5 they sat at cell corners, u1 = (h + 1) / 256, u2 = l / 256, which put 3 / 256 of the mass of a normal at
// exactly 0: the directions with cos = 0 or sin = 0 and the radius 0 of u1 = 1).  8 + 8 bits per pair (round 4; 16 + 16 before): one
// Threefry block (128 bits) yields SIXTEEN normals — both sub-steps of a pair of consecutive quad sub-steps, or all twelve normals
// of a hexa sub-step; the generator was 85 of the 336 vector instructions a quad sub-step cost, and the examples' five sub-steps per
// control are bound by vector issue.  The lattice as a distribution (exact; oracle/dsim_oracle.c mirrors it,
// tests/test_noise_distribution.py measures it): mean 0, variance sigma^2 exactly (the radius is scaled by DSIM_BM8_CORR =
// 2 / E[-2 ln u1] = 1.001355: the 256-point mean of -2 ln u1 is 1.9973, not 2), no atoms (no draw is 0), |n| <= 3.535 sigma,
// kurtosis 2.977 instead of 3, the two normals of a pair uncorrelated (E cos sin = 0 over the 256 directions).  The rotor noise is a
// 0.01 N / 0.001 N m perturbation of a 1.8 N thrust (BaseAviary.py:1518-1521).  This is the lattice of the kernels that LOOP over
// several sub-steps; launches of one sub-step draw the fine lattice below (include/dronesim_amd.h: noise_seed).
// The hardware transcendentals take the angle in revolutions and log in base 2; the deviation is folded into the radius:
// sigma sqrt(-2 corr ln u1) = sqrt(-sigma^2 2 ln2 corr log2(u1)).  v_cvt_f32_ubyteN converts a byte of the word in ONE
// instruction; the half-cell offsets ride in the multiply-adds that scale the bytes.
#define DSIM_BM8_S2 1.3881727911541815f          // 2 ln 2 x DSIM_BM8_CORR (1.0013550008475642)
template <int SIGMA_E3, int HALF>      // the deviation in thousandths (10 = rotor force noise, 1 = rotor moment noise); which 16 bits of w
__device__ __forceinline__ void box_muller8(uint32_t w, float& n0, float& n1) {
  constexpr float S2 = (SIGMA_E3 * 1e-3f) * (SIGMA_E3 * 1e-3f) * DSIM_BM8_S2;
  const float h = (float)((w >> (HALF ? 24 : 8)) & 0xFFu);      // (selected as v_cvt_f32_ubyte3 / _ubyte1)
  const float l = (float)((w >> (HALF ? 16 : 0)) & 0xFFu);
  const float u1 = (h + 0.5f) * (1.0f / 256.0f);
  const float u2 = (l + 0.5f) * (1.0f / 256.0f);
  const float r = DSIM_SQRT(-S2 * __builtin_amdgcn_logf(u1));
  n0 = r * __builtin_amdgcn_cosf(u2);
  n1 = r * __builtin_amdgcn_sinf(u2);
}
// The FINE lattice: 32 bits -> two normals, radius from the HIGH 16 bits (u1 = (h + 1/2) / 65536), direction from the LOW 16
// (u2 = (l + 1/2) / 65536 revolutions): 2^32 distinct pairs, |n| <= sqrt(2 corr ln 131072) = 4.855 sigma, variance sigma^2 exactly
// (DSIM_BM16_CORR = 2 / mean(-2 ln u1) = 1.0000053), kurtosis 2.99982.  Evaluated, not tabulated (65 536 radii do not fit
// the LDS budget of the looped kernels).  Mirrored by oracle/dsim_oracle.c:orc_bm16.  The lattice of every launch of ONE sub-step
// (where the generator is free: those kernels are bound by HBM) and of DSIM_OPT_NOISE_FINE at any count.
#define DSIM_BM16_S2 1.3863016922764049f         // 2 ln 2 x DSIM_BM16_CORR (1.0000052883115735)
template <int SIGMA_E3>
__device__ __forceinline__ void box_muller16(uint32_t w, float& n0, float& n1) {
  constexpr float S2 = (SIGMA_E3 * 1e-3f) * (SIGMA_E3 * 1e-3f) * DSIM_BM16_S2;
  const float u1 = ((float)(w >> 16) + 0.5f) * (1.0f / 65536.0f);
  const float u2 = ((float)(w & 0xFFFFu) + 0.5f) * (1.0f / 65536.0f);
  const float r = DSIM_SQRT(-S2 * __builtin_amdgcn_logf(u1));
  n0 = r * __builtin_amdgcn_cosf(u2);
  n1 = r * __builtin_amdgcn_sinf(u2);
}
// the fine stream's blocks: counter word 3's top bit set (a domain of its own beside the default stream's blocks)
__device__ __forceinline__ void noise_block_fine(uint64_t seed, uint64_t drone, uint64_t blk, uint32_t c[4]) {
  c[0] = (uint32_t)drone; c[1] = (uint32_t)(drone >> 32); c[2] = (uint32_t)blk; c[3] = (uint32_t)(blk >> 32) | 0x80000000u;
  threefry4x32_12(c, (uint32_t)seed, (uint32_t)(seed >> 32));
}
// quad sub-step `sub`: block `sub`; words 0, 1 -> the four force normals, words 2, 3 -> the four moment normals
__device__ __forceinline__ void quad_normals_fine(uint64_t seed, uint64_t drone, uint64_t sub, float* out) {
  uint32_t c[4];
  noise_block_fine(seed, drone, sub, c);
  box_muller16<10>(c[0], out[0], out[1]); box_muller16<10>(c[1], out[2], out[3]);
  box_muller16<1>(c[2], out[4], out[5]); box_muller16<1>(c[3], out[6], out[7]);
}
// hexa sub-step `sub` on the fine lattice: block `sub`; words 0, 1, 2 -> the six normals of the body wrench (hexa_wrench_z), scaled
// by 0.01 like the force normals (the factor's rows are stored over 0.01)
__device__ __forceinline__ void hexa_z_fine(uint64_t seed, uint64_t drone, uint64_t sub, float* z) {
  uint32_t c[4];
  noise_block_fine(seed, drone, sub, c);
  box_muller16<10>(c[0], z[0], z[1]); box_muller16<10>(c[1], z[2], z[3]); box_muller16<10>(c[2], z[4], z[5]);
}
// normals for (drone, sub-step counter).  Block = Threefry4x32-12(key = seed, counter = (drone, block index)), block index =
// sub >> 1 for EVERY vehicle kind: the even sub-step takes words 0, 1 of the block, the odd one words 2, 3.
//   quad (8 normals per sub-step: out[0 .. 4) force noise ~ N(0, 0.01), out[4 .. 8) moment noise ~ N(0, 0.001), BaseAviary.py:1518-1521):
//         force normals from the first word of the pair, moment normals from the second
//   six-actuator kinds (BaseAviary.py:1429-1430 draws twelve normals, one force and one moment per rotor): the twelve enter the rigid
//         composite only through the body wrench they add up to, W = M n with M the 6 x 12 map of hexa_wrench — a Gaussian
//         6-vector of covariance M diag(sigma^2) M^T.  The stream draws THAT vector: six unit normals z (both halves of the first
//         word, the low half of the second; the high half is unused) and W = L z, L the Cholesky factor of the covariance
//         (DevType.nchol, hexa_wrench_z) — the same distribution of the wrench, hence of the flight, from half the random bits,
//         half the Box-Muller pairs and 21 instead of 54 multiply-adds per sub-step.  Per-rotor normals remain an INPUT
//         (dsim_step_args.noise_replay, the general kernels: hexa_wrench).
__device__ __forceinline__ void noise_block(uint64_t seed, uint64_t drone, uint64_t blk, uint32_t c[4]) {
  c[0] = (uint32_t)drone; c[1] = (uint32_t)(drone >> 32); c[2] = (uint32_t)blk; c[3] = (uint32_t)(blk >> 32);
  threefry4x32_12(c, (uint32_t)seed, (uint32_t)(seed >> 32));
}
// the eight normals of a quad sub-step from its half of the block (odd: the sub-step counter is odd; wave-uniform)
__device__ __forceinline__ void quad_normals_from_block(const uint32_t c[4], bool odd, float* out) {
  const uint32_t wf = odd ? c[2] : c[0], wm = odd ? c[3] : c[1];
  box_muller8<10, 0>(wf, out[0], out[1]); box_muller8<10, 1>(wf, out[2], out[3]);
  box_muller8<1, 0>(wm, out[4], out[5]); box_muller8<1, 1>(wm, out[6], out[7]);
}
// The same normals from LDS tables (the kernels that loop over several sub-steps: the vector pipe bounds them, and the four
// Box-Muller pairs of a sub-step are 16 transcendental instructions).  A pair is a function of two BYTES: 256 radii per
// deviation, 256 directions — tabulated once per workgroup with the very expressions of box_muller8, so that a table entry
// holds the bits the direct evaluation produces and the kernels with and without tables draw identical noise.
struct NoiseTab { float2 rad[256]; float2 cs[256]; };          // rad[h] = (radius at sigma .01, at sigma .001); cs[l] = (cos, sin)
__device__ __forceinline__ void noise_tab_init(NoiseTab& t, unsigned tid /* 0..255 */) {
  constexpr float S2a = (10 * 1e-3f) * (10 * 1e-3f) * DSIM_BM8_S2, S2b = (1 * 1e-3f) * (1 * 1e-3f) * DSIM_BM8_S2;
  const float b = (float)(tid & 0xFFu);
  const float L = __builtin_amdgcn_logf((b + 0.5f) * (1.0f / 256.0f));
  const float u2 = (b + 0.5f) * (1.0f / 256.0f);
  t.rad[tid & 255u] = make_float2(DSIM_SQRT(-S2a * L), DSIM_SQRT(-S2b * L));
  t.cs[tid & 255u] = make_float2(__builtin_amdgcn_cosf(u2), __builtin_amdgcn_sinf(u2));
}
template <bool MOMENT, int HALF>
__device__ __forceinline__ void box_muller8_tab(const NoiseTab& t, uint32_t w, float& n0, float& n1) {
  const float2 rr = t.rad[(w >> (HALF ? 24 : 8)) & 0xFFu];
  const float2 cs = t.cs[(w >> (HALF ? 16 : 0)) & 0xFFu];
  const float r = MOMENT ? rr.y : rr.x;
  n0 = r * cs.x;
  n1 = r * cs.y;
}
__device__ __forceinline__ void quad_normals_from_block_tab(const NoiseTab& t, const uint32_t c[4], bool odd, float* out) {
  const uint32_t wf = odd ? c[2] : c[0], wm = odd ? c[3] : c[1];
  box_muller8_tab<false, 0>(t, wf, out[0], out[1]); box_muller8_tab<false, 1>(t, wf, out[2], out[3]);
  box_muller8_tab<true, 0>(t, wm, out[4], out[5]); box_muller8_tab<true, 1>(t, wm, out[6], out[7]);
}
// the six wrench normals of a hexa sub-step from its half of the block (odd: the sub-step counter is odd; wave-uniform), directly
// and from the tables (the looped hexa kernels: without tables the generator was 113 of the 281 us five sub-steps of 4 194 304
// hexas took); scaled by 0.01 (the force tables)
__device__ __forceinline__ void hexa_z_from_block(const uint32_t c[4], bool odd, float* z) {
  const uint32_t wa = odd ? c[2] : c[0], wb = odd ? c[3] : c[1];
  box_muller8<10, 0>(wa, z[0], z[1]); box_muller8<10, 1>(wa, z[2], z[3]); box_muller8<10, 0>(wb, z[4], z[5]);
}
__device__ __forceinline__ void hexa_z_from_block_tab(const NoiseTab& t, const uint32_t c[4], bool odd, float* z) {
  const uint32_t wa = odd ? c[2] : c[0], wb = odd ? c[3] : c[1];
  box_muller8_tab<false, 0>(t, wa, z[0], z[1]); box_muller8_tab<false, 1>(t, wa, z[2], z[3]); box_muller8_tab<false, 0>(t, wb, z[4], z[5]);
}
// (dsim_noise_draw, the general kernels) out[2 NACT]: the normals of (drone, sub), scaled by their deviations; six-actuator kinds:
// out[0 .. 6) = 0.01 z, out[6 .. 12) = 0
template <int NACT>
__device__ __forceinline__ void noise_normals(uint64_t seed, uint64_t drone, uint64_t sub, float* out) {
  uint32_t c[4];
  noise_block(seed, drone, sub >> 1, c);
  if constexpr (NACT == 4) {
    quad_normals_from_block(c, (sub & 1ull) != 0, out);
  } else {
    hexa_z_from_block(c, (sub & 1ull) != 0, out);
#pragma unroll
    for (int j = 6; j < 12; ++j) out[j] = 0.0f;
  }
}

// ---------------------------------------------------------------------------
// rigid state + controller memory of one drone, in registers
// ---------------------------------------------------------------------------
struct Rigid { V3 pos; Q4 q; V3 vel; V3 w; };                 // w: WORLD-frame angular velocity
template <int NACT> struct CtrlMem { V3 last_vel; V3 last_rates; float last_thrust; float cmd[NACT]; };
struct Target { V3 pos, vel, acc; float yaw; };

// Ground-plane watch.  The reference loads plane.urdf with collisions on (BaseAviary.py:680); the flight kernels do not
// model the plane (DSIM_OPT_PLANE does, in its own kernel instances), so an Env.step that ends with the vehicle's
// collision cylinder at or below z = 0 is COUNTED: one atomic per wave that holds such a drone, on one of 64 counter shards (dsim_query sums them).
#define DSIM_GROUND_SHARDS 64
template <class DT>
__device__ __forceinline__ void ground_watch(DT& T, const Rigid& s, unsigned long long* counters, bool live = true) {
  const float r22 = 1.0f - 2.0f * (s.q.x * s.q.x + s.q.y * s.q.y);                     // body z . world z (unit q)
  const float reach = T.watch_below * fabsf(r22) + T.coll_r * DSIM_SQRT(fmaxf(1.0f - r22 * r22, 0.0f));
  const bool hit = live && T.coll_r > 0.0f && s.pos.z <= reach;
  const unsigned long long m = __ballot(hit);
  if (m != 0ULL && (int)(threadIdx.x & 63u) == __builtin_ctzll(m))
    atomicAdd(&counters[8 + (blockIdx.x & (DSIM_GROUND_SHARDS - 1))], (unsigned long long)__popcll(m));
}

// P1: CtrlAviary._preprocessAction, CtrlAviary.py:258-263
template <int NACT, class DT>
__device__ __forceinline__ void preprocess_action(DT& T, const float* a, float* clipped) {
#pragma unroll
  for (int j = 0; j < NACT; ++j) clipped[j] = clampf(a[j], T.pmin[j], T.pmax[j]);
}

// P2: BaseAviary._quad_copter_physics standard branch, BaseAviary.py:1487-1490, 1514-1543.
// Body-frame wrench about the COM = what the four link forces (applied at the prop
// links' inertial origins, LINK_FRAME) and the base torque add up to.
// nz: 8 scaled normals (f_noise[4] ~ N(0,.01), m_noise[4] ~ N(0,.001)) or nullptr.
template <class DT>
__device__ __forceinline__ void quad_wrench(DT& T, const float cmd[4], const float* nz, V3& F, V3& tau) {
  float f[4], t[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float rpm = T.scale[i] * cmd[i] + T.cnst[i];
    f[i] = rpm * rpm * T.kf + (nz ? nz[i] : 0.0f);
    t[i] = rpm * rpm * T.km + (nz ? nz[4 + i] : 0.0f);
  }
  const float fx = nz ? nz[0] : 0.0f, fy = nz ? nz[1] : 0.0f;   // x/y reuse noise 0,1 for every rotor (:1532)
  F = v3(4.0f * fx, 4.0f * fy, (f[0] + f[1]) + (f[2] + f[3]));
  tau = v3(nz ? nz[4] : 0.0f, nz ? nz[5] : 0.0f, (-t[0] + t[1]) + (-t[2] + t[3]));
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const V3 r = v3(T.rpos[i][0], T.rpos[i][1], T.rpos[i][2]);
    tau = tau + cross(r, v3(fx, fy, f[i]));
  }
}

// The same map split for the sub-step loop: the command is constant over the sub-steps of an Env.step, only the 8 normals
// change (BaseAviary.py:510-545 re-applies the same clipped action).  quad_wrench_base: the noise-free part, once per
// Env.step; quad_wrench_noise: what the normals of one sub-step add.  With f_i = f0_i + n_i and the lateral noise (fx, fy)
// shared by the four rotor links (:1532):
//   F   = (4 fx, 4 fy, sum f0_i + sum n_i)
//   tau = sum r_i x (0, 0, f0_i) + (0, 0, tz0)  +  sum r_i x (fx, fy, n_i) + (m0, m1, -m0 + m1 - m2 + m3)
//   sum r_i x (fx, fy, n_i) = (sum r_iy n_i - fy Rz,  fx Rz - sum r_ix n_i,  fy Rx - fx Ry),   R = sum r_i
// 22 vector instructions per sub-step instead of the 60 of the whole map.
// The base is evaluated ONCE per Env.step and then enters every sub-step: its rounding would add up coherently over the
// sub-steps (the four rotor moments, +-1.35 rad/s per sub-step each, cancel to ~1e-3: every product r x f is rounded at
// its own magnitude), so the two cancelling sums are accumulated in fp64 and rounded once — a dozen fp64 operations per
// Env.step.  The rotor forces themselves are the fp32 values the unsplit map uses.
struct QuadBase { float Fz; V3 tau; };
template <class DT>
__device__ __forceinline__ QuadBase quad_wrench_base(DT& T, const float cmd[4]) {
  double fz = 0.0, tx = 0.0, ty = 0.0;
  float tz = 0.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float rpm = T.scale[i] * cmd[i] + T.cnst[i];
    const float f = rpm * rpm * T.kf, t = rpm * rpm * T.km;
    fz += (double)f;
    tx += (double)T.rpos[i][1] * (double)f;              // (r x (0, 0, f)).x =  r_y f
    ty -= (double)T.rpos[i][0] * (double)f;              // (r x (0, 0, f)).y = -r_x f   (the levers as doubles in the table, as
                                                         // hexa_wrench_base has them: 24 vector instructions MORE here — not taken)
    tz += (i & 1) ? t : -t;                              // -t0 + t1 - t2 + t3 (:1527)
  }
  return QuadBase{(float)fz, V3{(float)tx, (float)ty, tz}};
}
template <class DT>
__device__ __forceinline__ void quad_wrench_noise(DT& T, const QuadBase& b, const float nz[8], V3& F, V3& tau) {
  const float fx = nz[0], fy = nz[1];
  F = v3(4.0f * fx, 4.0f * fy, b.Fz + ((nz[0] + nz[1]) + (nz[2] + nz[3])));
  const float sy = T.rpos[0][1] * nz[0] + T.rpos[1][1] * nz[1] + T.rpos[2][1] * nz[2] + T.rpos[3][1] * nz[3];
  const float sx = T.rpos[0][0] * nz[0] + T.rpos[1][0] * nz[1] + T.rpos[2][0] * nz[2] + T.rpos[3][0] * nz[3];
  tau = v3(b.tau.x + (nz[4] + (sy - fy * T.rsum[2])),
           b.tau.y + (nz[5] + (fx * T.rsum[2] - sx)),
           b.tau.z + (((-nz[4] + nz[5]) + (-nz[6] + nz[7])) + (fy * T.rsum[0] - fx * T.rsum[1])));
}

// P7: BaseAviary._groundEffect, BaseAviary.py:1648-1699 (formula; dead code in the fork): per rotor
// dF_i = kf rpm_i^2 GND_EFF_COEFF (PROP_RADIUS / (4 h_i))^2 along the link z axis at the rotor link,
// h_i = rotor height clipped below at GND_EFF_H_CLIP, only while |roll|, |pitch| < pi/2.
template <class DT>
__device__ __forceinline__ void ground_effect_quad(DT& T, const Rigid& s, const float cmd[4], V3& F, V3& tau) {
  const Q4 q = s.q;
  const float sarg = -2.0f * (q.x * q.z - q.w * q.y);
  const float rb = q.w * q.w - q.x * q.x - q.y * q.y + q.z * q.z;      // cos(roll) > 0  <=>  |roll| < pi/2
  if (!(fabsf(sarg) < 0.99999f && rb > 0.0f)) return;                   // gimbal branch reports |pitch| = pi/2
  const M3 R = matrix_from_quat(q);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float rpm = T.scale[i] * cmd[i] + T.cnst[i];
    const V3 r = v3(T.rpos[i][0], T.rpos[i][1], T.rpos[i][2]);
    const float h = fmaxf(s.pos.z + R.m[6] * r.x + R.m[7] * r.y + R.m[8] * r.z, T.gnd_hclip);
    const float ratio = T.prop_radius * DSIM_RCP(4.0f * h);
    const float dF = rpm * rpm * T.kf * T.gnd_coeff * ratio * ratio;
    F.z += dF;
    tau = tau + cross(r, v3(0.0f, 0.0f, dF));
  }
}

// P6: BaseAviary._drag, BaseAviary.py:1705-1732 (formula; dead code in the fork), restated literally:
// drag = R . (-DRAG_COEFF * sum(2 pi rpm / 60) * v_world), handed to Bullet as a LINK_FRAME force at the COM.
template <class DT>
__device__ __forceinline__ V3 drag_quad(DT& T, const Rigid& s, const float last_cmd[4]) {
  float w = 0.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i) w += (T.scale[i] * last_cmd[i] + T.cnst[i]) * (6.28318530717958647692f / 60.0f);
  const M3 R = matrix_from_quat(s.q);
  return mul(R, v3(-T.drag[0] * w * s.vel.x, -T.drag[1] * w * s.vel.y, -T.drag[2] * w * s.vel.z));
}

// DSIM_OPT_PLANE: contact with the ground plane z = 0 — the product-defined model documented at
// oracle/dsim_oracle.c:orc_plane_contact (DSIM_PLANE_POINTS body-fixed points on the rim of the collision cylinder's
// lower face, 45 degrees apart from body x; penetrating points driven to erp depth / dt, separated ones limited to
// gap / dt inside the 0.02 m margin; Coulomb pyramid on the world tangents; DSIM_PLANE_ITERS projected Gauss-Seidel
// sweeps on (v, w)).  Kept out of the flight kernels: only the k_step_plane / k_physics_plane / k_adaptor<.., true>
// instances compile it (launched when the option bit is set), and only lanes that hold a drone within the margin run
// the sweeps.  Registers: the impulses and 1 / K of the 24 rows stay live (48), the points are rebuilt from the face
// centre and the two scaled body axes.
#define DSIM_PLANE_ITERS 24
#define DSIM_PLANE_POINTS 8
struct SymM3 { float xx, xy, xz, yy, yz, zz; };
__device__ __forceinline__ V3 mul(const SymM3& A, V3 b) {
  return v3(A.xx * b.x + A.xy * b.y + A.xz * b.z, A.xy * b.x + A.yy * b.y + A.yz * b.z, A.xz * b.x + A.yz * b.y + A.zz * b.z);
}
template <class DT>
__device__ __forceinline__ void plane_contact(DT& T, float dt, const V3 pos, const Q4 q, V3& v, V3& w) {
  if (!(T.coll_r > 0.0f)) return;
  const M3 R = matrix_from_quat(q);
  const V3 a = v3(R.m[2], R.m[5], R.m[8]);                          // body z axis in the world
  const float sgn = a.z >= 0.0f ? 1.0f : -1.0f;                     // which face is the lower one
  const V3 c = (-sgn * T.coll_below) * a;                           // its centre, relative to the COM
  const V3 ex = T.coll_r * v3(R.m[0], R.m[3], R.m[6]), ey = T.coll_r * v3(R.m[1], R.m[4], R.m[7]);   // body x, y, scaled
  // world inverse inertia R diag(1 / J) R^T
  SymM3 Ji;
  {
    const float i0 = T.invJ[0], i1 = T.invJ[1], i2 = T.invJ[2];
    Ji.xx = R.m[0] * R.m[0] * i0 + R.m[1] * R.m[1] * i1 + R.m[2] * R.m[2] * i2;
    Ji.xy = R.m[0] * R.m[3] * i0 + R.m[1] * R.m[4] * i1 + R.m[2] * R.m[5] * i2;
    Ji.xz = R.m[0] * R.m[6] * i0 + R.m[1] * R.m[7] * i1 + R.m[2] * R.m[8] * i2;
    Ji.yy = R.m[3] * R.m[3] * i0 + R.m[4] * R.m[4] * i1 + R.m[5] * R.m[5] * i2;
    Ji.yz = R.m[3] * R.m[6] * i0 + R.m[4] * R.m[7] * i1 + R.m[5] * R.m[8] * i2;
    Ji.zz = R.m[6] * R.m[6] * i0 + R.m[7] * R.m[7] * i1 + R.m[8] * R.m[8] * i2;
  }
  constexpr float H = 0.70710678118654752440f;
  constexpr float CB[8] = {1.0f, H, 0.0f, -H, -1.0f, -H, 0.0f, H};
  constexpr float SB[8] = {0.0f, H, 1.0f, H, 0.0f, -H, -1.0f, -H};
  float gap[DSIM_PLANE_POINTS], lam[DSIM_PLANE_POINTS][3], iK[DSIM_PLANE_POINTS][3];
  bool any = false;
#pragma unroll
  for (int j = 0; j < DSIM_PLANE_POINTS; ++j) {
    const V3 r = c + (CB[j] * ex + SB[j] * ey);
    gap[j] = pos.z + r.z;
    any = any || gap[j] < 0.02f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const V3 ax = v3(k == 0 ? 1.0f : 0.0f, k == 1 ? 1.0f : 0.0f, k == 2 ? 1.0f : 0.0f);
      const V3 u = cross(mul(Ji, cross(r, ax)), r);
      iK[j][k] = DSIM_RCP(T.inv_mass + (k == 0 ? u.x : (k == 1 ? u.y : u.z)));
      lam[j][k] = 0.0f;
    }
  }
  if (!any) return;
  const float inv_dt = DSIM_RCP(dt);
  for (int it = 0; it < DSIM_PLANE_ITERS; ++it) {
#pragma unroll
    for (int j = 0; j < DSIM_PLANE_POINTS; ++j) {
      if (!(gap[j] < 0.02f)) continue;
      const V3 r = c + (CB[j] * ex + SB[j] * ey);
#pragma unroll
      for (int pass = 0; pass < 3; ++pass) {
        const int k = pass == 0 ? 2 : pass - 1;                     // normal (z) first, then the tangents x, y
        const V3 wr = cross(w, r);
        const float u = (k == 0 ? v.x + wr.x : (k == 1 ? v.y + wr.y : v.z + wr.z));
        float dl;
        if (k == 2) {
          const float target = gap[j] < 0.0f ? 0.2f * (-gap[j]) * inv_dt : -gap[j] * inv_dt;
          const float nl = fmaxf(0.0f, lam[j][2] + (target - u) * iK[j][2]);
          dl = nl - lam[j][2]; lam[j][2] = nl;
        } else {
          const float lim = T.mu_plane * lam[j][2];
          const float nl = clampf(lam[j][k] - u * iK[j][k], -lim, lim);
          dl = nl - lam[j][k]; lam[j][k] = nl;
        }
        const V3 imp = v3(k == 0 ? dl : 0.0f, k == 1 ? dl : 0.0f, k == 2 ? dl : 0.0f);
        v = v + T.inv_mass * imp;
        w = w + mul(Ji, cross(r, imp));
      }
    }
  }
}

// P4: one Bullet btMultiBody floating-base step [BULLET-INTERNAL, parity unpinned];
// restated step by step in oracle/dsim_oracle.c:orc_bullet_step.  PLANE: with the contact solve (DSIM_OPT_PLANE).
template <bool PLANE = false, class DT>
__device__ __forceinline__ void bullet_step(DT& T, float dt, Rigid& s, V3 F_body, V3 tau_body) {
  const M3 R = matrix_from_quat(s.q);
  const V3 wb = mulT(R, s.w);
  // linear: world-frame form of a_b + w_b x v_b (the m w x v bias cancels), |v_b| = |v|
  const float vn = DSIM_SQRT(dot(s.vel, s.vel));
  const V3 Fw = mul(R, F_body);
  const float dl = T.clin + T.clin * vn;
  V3 vdot = v3(Fw.x * T.inv_mass - dl * s.vel.x, Fw.y * T.inv_mass - dl * s.vel.y,
               Fw.z * T.inv_mass - T.g - dl * s.vel.z);
  // angular: alpha_b = J^-1 (tau - w x Jw) - c (1+|w|) w
  const float wn = DSIM_SQRT(dot(wb, wb));
  const V3 Jw = v3(T.J[0] * wb.x, T.J[1] * wb.y, T.J[2] * wb.z);
  const V3 gy = cross(wb, Jw);
  const float da = T.cang + T.cang * wn;
  const V3 ab = v3((tau_body.x - gy.x) * T.invJ[0] - da * wb.x, (tau_body.y - gy.y) * T.invJ[1] - da * wb.y,
                   (tau_body.z - gy.z) * T.invJ[2] - da * wb.z);
  const V3 wdot = mul(R, ab);
  // applyDeltaVeeMultiDof: += acc*dt, clamp every coordinate to +-maxCoordinateVelocity
  s.w = v3(clampf(s.w.x + wdot.x * dt, -T.maxv, T.maxv), clampf(s.w.y + wdot.y * dt, -T.maxv, T.maxv),
           clampf(s.w.z + wdot.z * dt, -T.maxv, T.maxv));
  s.vel = v3(clampf(s.vel.x + vdot.x * dt, -T.maxv, T.maxv), clampf(s.vel.y + vdot.y * dt, -T.maxv, T.maxv),
             clampf(s.vel.z + vdot.z * dt, -T.maxv, T.maxv));
  if (PLANE) plane_contact(T, dt, s.pos, s.q, s.vel, s.w);      // velocity-level contact solve, then the positions
  // stepPositionsMultiDof: semi-implicit position, exponential-map orientation
  s.pos = s.pos + dt * s.vel;
  float fAngle = DSIM_SQRT(dot(s.w, s.w));
  if (fAngle * dt > DSIM_PI_4) fAngle = DSIM_PI_4 * DSIM_RCP(dt);
  // h = half rotation angle <= pi/8: sin(h)/fAngle = (dt/2) sinc(h); truncation error < 2e-9.
  // (Bullet's own |w| < 0.001 Taylor branch is the first two terms of the same series.)
  const float h = 0.5f * fAngle * dt, h2 = h * h;
  const float sinc = 1.0f + h2 * (-1.0f / 6.0f + h2 * (1.0f / 120.0f + h2 * (-1.0f / 5040.0f)));
  const float cw = 1.0f + h2 * (-0.5f + h2 * (1.0f / 24.0f + h2 * (-1.0f / 720.0f + h2 * (1.0f / 40320.0f))));
  const float sc = 0.5f * dt * sinc;
  const float ax = s.w.x * sc, ay = s.w.y * sc, az = s.w.z * sc;
  Q4 q = s.q, n;   // n = dq * q
  n.w = cw * q.w - ax * q.x - ay * q.y - az * q.z;
  n.x = cw * q.x + ax * q.w + ay * q.z - az * q.y;
  n.y = cw * q.y + ay * q.w + az * q.x - ax * q.z;
  n.z = cw * q.z + az * q.w + ax * q.y - ay * q.x;
  const float inv = DSIM_RSQ(n.x * n.x + n.y * n.y + n.z * n.z + n.w * n.w);
  s.q = Q4{n.x * inv, n.y * inv, n.z * inv, n.w * inv};
}

// D1: BaseAviary._dynamics, BaseAviary.py:1767-1828 — the reference's OWN explicit rigid-body model (Physics.DYN), restated
// line for line in oracle/dsim_oracle.c:orc_dynamics and pinned there by tests/golden/dynamics.npz (the reference's function
// run on recorded inputs).  The command is constant over the sub-steps of an Env.step (BaseAviary.py:510-545), so thrust and
// the three mixer torques are formed once (DynBase; the x / y sums cancel to ~1e-3 of their terms: fp64, rounded once, as
// quad_wrench_base does); the sub-step is the state update.  rr = self.rpy_rates; s.w is not touched.
struct DynBase { float thrust; V3 tq; };
template <class DT>
__device__ __forceinline__ DynBase dyn_base(DT& T, const float cmd[4]) {
  double th = 0.0, tx = 0.0, ty = 0.0;
  float tz = 0.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float rpm = T.scale[i] * cmd[i] + T.cnst[i];              // the fork's PWM -> RPM map, BaseAviary.py:1487-1490
    const float f = rpm * rpm * T.kf, t = rpm * rpm * T.km;          // :1788, 1792
    th += (double)f;                                                 // :1789
    tx += (double)T.dyn_lever[0][i] * (double)f;                     // :1794-1803
    ty += (double)T.dyn_lever[1][i] * (double)f;
    tz += (i & 1) ? t : -t;                                          // :1793
  }
  return DynBase{(float)th, V3{(float)tx, (float)ty, tz}};
}
template <class DT>
__device__ __forceinline__ void dyn_substep(DT& T, float dt, const DynBase& b, Rigid& s, V3& rr) {
  const Euler e = euler_from_quat<true>(s.q);                        // self.rpy as _updateAndStoreKinematicInformation stores it, :729
  const Q4 q = s.q;                                                  // third column of getMatrixFromQuaternion(quat), :1786
  const float sc = 2.0f * DSIM_RCP(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  const V3 bz = v3((q.x * q.z + q.w * q.y) * sc, (q.y * q.z - q.w * q.x) * sc, 1.0f - (q.x * q.x + q.y * q.y) * sc);
  const V3 acc = v3(bz.x * b.thrust * T.inv_mass, bz.y * b.thrust * T.inv_mass, (bz.z * b.thrust - T.weight) * T.inv_mass);   // :1790-1791, 1807
  const V3 Jw = v3(T.J[0] * rr.x, T.J[1] * rr.y, T.J[2] * rr.z);
  const V3 gy = cross(rr, Jw);                                       // :1805
  const V3 rd = v3((b.tq.x - gy.x) * T.invJ[0], (b.tq.y - gy.y) * T.invJ[1], (b.tq.z - gy.z) * T.invJ[2]);   // :1806
  s.vel = s.vel + dt * acc;                                          // :1809
  rr = rr + dt * rd;                                                 // :1810
  s.pos = s.pos + dt * s.vel;                                        // :1811
  s.q = quat_from_euler(v3(e.roll + dt * rr.x, e.pitch + dt * rr.y, e.yaw + dt * rr.z));   // :1812, 1817
}
// what p.getBaseVelocity reports after a DYN step: the placeholder the reference stores (:1821-1826), or — the product's
// flyable deviation, DSIM_OPT_DYN_BODY_RATES — the world-frame image of the rates the model treats as body rates
__device__ __forceinline__ V3 dyn_reported_ang_vel(bool body_rates, Q4 q, V3 rr) {
  if (!body_rates) return v3(-1.0f, -1.0f, -1.0f);
  return mul(matrix_from_quat(q), rr);
}

// P4 over SEVERAL sub-steps (the kernels that loop: the examples fly five per control step, examples/fly_INDI.py:139-141, and
// those kernels are bound by vector issue).  The same step as bullet_step, arranged so that what the loop carries is what the
// loop needs:
//   * the angular velocity is carried in the BODY frame.  bullet_step turns the world-frame w into the body frame, adds the
//     body-frame acceleration's image R a_b dt to the world-frame w, and the next sub-step turns it back: with R' = dR(w') R and
//     dR(w') a rotation ABOUT w', R'^T w' = R^T w' = w_b + a_b dt — the body-frame rates simply accumulate, and R^T w, R a_b
//     and the world-frame update leave the loop (one R^T w in front of it, one R w_b behind it);
//   * the orientation increment is applied in the body frame, q (x) (a_b, c) = (R a_b, c) (x) q: the same rotation;
//   * sin(h) / |w| and cos(h) of the exponential map are even series in h: they need h^2 = (dt / 2)^2 w.w, not |w| (Bullet's
//     clamp of the rotation per step to pi / 4 is a minimum on h^2);
//   * the quaternion is normalised once behind the loop (matrix_from_quat divides by |q|^2 anyway, and scaling commutes with
//     the product): every sub-step of the oracle normalises, the results differ by roundings.
// Bullet clamps every WORLD coordinate of the angular velocity to +-maxCoordinateVelocity (applyDeltaVeeMultiDof): while
// |w_b| < maxv no coordinate can reach it; a lane beyond that (100 rad/s: a tumbling wreck) takes the world-frame detour.
// 27 + 8 + 10 of the ~300 vector instructions of a sub-step less (profiles/r05_sub5_*).
struct RigidB { V3 pos; Q4 q; V3 vel; V3 wb; float ww; };      // ww = wb . wb (the exponential map of one sub-step needs it, the damping of the next one too)
//   * the quaternion enters the loop normalised (the same rotation; the caller's may be any length: the reference's helpers
//     do not normalise) and stays within roundings of unit length through it, so that 2 / |q|^2 of the rotation matrix is one
//     Newton step from 1, 4 - 2 |q|^2 (error (1 - |q|^2)^2 ~ 1e-13), instead of a reciprocal per sub-step.
__device__ __forceinline__ RigidB body_begin(const Rigid& s) {
  const float inv = DSIM_RSQ(s.q.x * s.q.x + s.q.y * s.q.y + s.q.z * s.q.z + s.q.w * s.q.w);
  const Q4 q = Q4{s.q.x * inv, s.q.y * inv, s.q.z * inv, s.q.w * inv};
  const V3 wb = mulT(matrix_from_quat(q), s.w);
  return RigidB{s.pos, q, s.vel, wb, dot(wb, wb)};
}
// matrix_from_quat for a quaternion within roundings of unit length (the loop's)
__device__ __forceinline__ M3 matrix_from_near_unit_quat(Q4 q) {
  const float d = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
  const float s = __builtin_fmaf(-2.0f, d, 4.0f);
  const float xs = q.x * s, ys = q.y * s, zs = q.z * s;
  const float wx = q.w * xs, wy = q.w * ys, wz = q.w * zs;
  const float xx = q.x * xs, xy = q.x * ys, xz = q.x * zs;
  const float yy = q.y * ys, yz = q.y * zs, zz = q.z * zs;
  M3 R;
  R.m[0] = 1.0f - (yy + zz); R.m[1] = xy - wz;          R.m[2] = xz + wy;
  R.m[3] = xy + wz;          R.m[4] = 1.0f - (xx + zz); R.m[5] = yz - wx;
  R.m[6] = xz - wy;          R.m[7] = yz + wx;          R.m[8] = 1.0f - (xx + yy);
  return R;
}
__device__ __forceinline__ void body_end(const RigidB& b, Rigid& s) {
  const float inv = DSIM_RSQ(b.q.x * b.q.x + b.q.y * b.q.y + b.q.z * b.q.z + b.q.w * b.q.w);
  s.pos = b.pos; s.vel = b.vel;
  s.q = Q4{b.q.x * inv, b.q.y * inv, b.q.z * inv, b.q.w * inv};
  s.w = mul(matrix_from_quat(s.q), b.wb);
}
// R v for a quaternion within roundings of unit length, without the matrix: R = I + s (w [u]x + [u]x^2), u = (x, y, z), s = 2 / |q|^2
// (what btMatrix3x3::setRotation spells out entry by entry), so R v = v + s (w t + u x t) with t = u x v: 23 vector instructions
// against 26 for the matrix + 9 for the product — the looped step needs R for this ONE product only.
__device__ __forceinline__ V3 rotate_near_unit(Q4 q, V3 v) {
  const float d = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
  const float s = __builtin_fmaf(-2.0f, d, 4.0f);
  const V3 u = v3(q.x, q.y, q.z);
  const V3 t = cross(u, v);
  const V3 c = cross(u, t);
  return v3(__builtin_fmaf(s, __builtin_fmaf(q.w, t.x, c.x), v.x), __builtin_fmaf(s, __builtin_fmaf(q.w, t.y, c.y), v.y),
            __builtin_fmaf(s, __builtin_fmaf(q.w, t.z, c.z), v.z));
}
// clamp to [-m, m] in ONE instruction (v_med3_f32; the max / min pair of clampf is two: LLVM forms the median only from
// constants).  The same value for every finite input.
__device__ __forceinline__ float clamp_sym(float v, float m) { return __builtin_amdgcn_fmed3f(v, -m, m); }
template <class DT>
__device__ __forceinline__ void bullet_step_body(DT& T, float dt, RigidB& s, V3 F_body, V3 tau_body) {
  const float vn = DSIM_SQRT(dot(s.vel, s.vel));
  const V3 Fw = rotate_near_unit(s.q, F_body);
  const float dl = T.clin + T.clin * vn;
  const V3 vdot = v3(Fw.x * T.inv_mass - dl * s.vel.x, Fw.y * T.inv_mass - dl * s.vel.y, Fw.z * T.inv_mass - T.g - dl * s.vel.z);
  // angular: alpha_b = J^-1 (tau - w x J w) - c (1 + |w|) w, Euler's equations written out: (w x J w)_x / J_x = gyro_x w_y w_z with
  // gyro = ((J_z - J_y) / J_x, (J_x - J_z) / J_y, (J_y - J_x) / J_z) a per-type constant — 16 instructions instead of 22, and
  // the products of the rates enter unrounded by J
  const V3 wb = s.wb;
  const float wn = DSIM_SQRT(s.ww);
  const float da = T.cang + T.cang * wn;
  const V3 ab = v3(__builtin_fmaf(-da, wb.x, __builtin_fmaf(-T.gyro[0], wb.y * wb.z, tau_body.x * T.invJ[0])),
                   __builtin_fmaf(-da, wb.y, __builtin_fmaf(-T.gyro[1], wb.z * wb.x, tau_body.y * T.invJ[1])),
                   __builtin_fmaf(-da, wb.z, __builtin_fmaf(-T.gyro[2], wb.x * wb.y, tau_body.z * T.invJ[2])));
  V3 wn_b = wb + dt * ab;
  float ww = dot(wn_b, wn_b);
  if (!(ww < T.maxv * T.maxv)) {                           // (rare) a world coordinate may reach the clamp: the world-frame form
    const M3 R = matrix_from_near_unit_quat(s.q);
    const V3 ww_ = mul(R, wn_b);
    const V3 wc = v3(clampf(ww_.x, -T.maxv, T.maxv), clampf(ww_.y, -T.maxv, T.maxv), clampf(ww_.z, -T.maxv, T.maxv));
    wn_b = mulT(R, wc);
    ww = dot(wn_b, wn_b);
  }
  s.wb = wn_b; s.ww = ww;
  s.vel = v3(clamp_sym(s.vel.x + vdot.x * dt, T.maxv), clamp_sym(s.vel.y + vdot.y * dt, T.maxv), clamp_sym(s.vel.z + vdot.z * dt, T.maxv));
  s.pos = s.pos + dt * s.vel;
  // exponential map of w' dt in the body frame: h^2 = (dt / 2)^2 w'.w', clamped at (pi / 8)^2 (the rotation per step at pi / 4)
  const float h2_free = 0.25f * dt * dt * ww;
  const float h2 = fminf(h2_free, (0.5f * DSIM_PI_4) * (0.5f * DSIM_PI_4));
  const float sinc = 1.0f + h2 * (-1.0f / 6.0f + h2 * (1.0f / 120.0f + h2 * (-1.0f / 5040.0f)));
  const float cw = 1.0f + h2 * (-0.5f + h2 * (1.0f / 24.0f + h2 * (-1.0f / 720.0f)));      // (h^8 / 40320 <= 1.4e-8 at the clamp: below half an ulp of 1)
  const float sc = 0.5f * dt * sinc;
  const float ax = wn_b.x * sc, ay = wn_b.y * sc, az = wn_b.z * sc;
  const Q4 q = s.q;                                        // n = q (x) (a, cw)
  s.q = Q4{q.w * ax + q.x * cw + q.y * az - q.z * ay,
           q.w * ay - q.x * az + q.y * cw + q.z * ax,
           q.w * az + q.x * ay - q.y * ax + q.z * cw,
           q.w * cw - q.x * ax - q.y * ay - q.z * az};
  // (rare) the rotation clamp engaged: Bullet scales the UNCLAMPED w by sin(h_c) / w_c, so the increment (a, cw) has length
  // sqrt(cw^2 + (|w| / w_c)^2 sin^2 h_c) > 1 and Bullet normalises the product (pQuatUpdateFun: quat.normalize()).  The loop's
  // quaternion must stay within roundings of unit length (rotate_near_unit's 4 - 2 |q|^2), so this lane normalises here —
  // unreachable at 240 Hz (|w| <= 173 rad/s < pi / (4 dt) = 188), reachable through freq= below ~220 Hz.
#ifndef DSIM_AB_NO_CLAMP_RENORM        // (A/B knob of the build: tests/test_gpu_envelope.py fails its pi4 regimes without this branch)
  if (h2_free > (0.5f * DSIM_PI_4) * (0.5f * DSIM_PI_4)) {
    const float inv = DSIM_RSQ(s.q.x * s.q.x + s.q.y * s.q.y + s.q.z * s.q.z + s.q.w * s.q.w);
    s.q = Q4{s.q.x * inv, s.q.y * inv, s.q.z * inv, s.q.w * inv};
  }
#endif
}

// C4: INDIControl._INDIRateControl, INDIControl.py:413-490 (also the whole of RPYTAviary's action
// adaptor, RPYTAviary.py:181-193): body rates, finite-difference angular acceleration, the virtual
// control v, du = pinv(G1/0.05) v, cmd += du, clip.
template <int NACT = 4, class DT>
__device__ __forceinline__ void indi_rate(DT& T, float inv_dt, const Rigid& s, V3 rate_sp, float thrust,
                                          CtrlMem<NACT>& m) {
  const M3 R = matrix_from_quat(s.q);                                          // :428
  const V3 wb = mulT(R, s.w);                                                  // :430
  float v[4];
  v[0] = (rate_sp.x - wb.x) * T.krate[0] - (wb.x - m.last_rates.x) * inv_dt;   // :433-453
  v[1] = (rate_sp.y - wb.y) * T.krate[1] - (wb.y - m.last_rates.y) * inv_dt;
  v[2] = (rate_sp.z - wb.z) * T.krate[2] - (wb.z - m.last_rates.z) * inv_dt;
  v[3] = thrust - m.last_thrust;                                               // :454
  m.last_rates = wb;                                                           // :442
  m.last_thrust = thrust;                                                      // :455
  // (NACT = 6: hexa_6DOF_simple flies the quad law on six actuators — G1 is 4 x 6, pinv(G1 / 0.05) 6 x 4; a real quad's rows
  // 4, 5 of a six-wide table are zero, and so are its limits: its last two commands stay 0)
#pragma unroll
  for (int j = 0; j < NACT; ++j) {                                             // :459, 486-487
    const float du = T.alloc[j][0] * v[0] + T.alloc[j][1] * v[1] + T.alloc[j][2] * v[2] + T.alloc[j][3] * v[3];
    m.cmd[j] = clampf(m.cmd[j] + du, T.pmin[j], T.pmax[j]);
  }
}

// C2 + C3 + C4: INDIControl.computeControl for a quad, INDIControl.py:154-227.
// Returns pos_e and (WANT_YAW) yaw_e, the reference's 2nd and 3rd return values.
//
// Yaw: the reference builds target_euler.z = psi + norm_ang(psi* - psi) = psi* - 2 pi j.  Only
// sin/cos of HALF that angle enter the target quaternion, so j flips the sign of the whole
// quaternion, hence of quat_err, and quat_wrap_shortest (math.py:46-51) removes exactly that
// sign.  The attitude error therefore depends on psi* alone, and the kernel evaluates
// sincos(psi*/2) directly; psi itself (one more atan2) is computed only when yaw_e is wanted.
template <bool WANT_YAW, int NACT = 4, class DT>
__device__ __forceinline__ void indi_quad(DT& T, float dt, const Rigid& s, const Target& tg,
                                          CtrlMem<NACT>& m, V3& pos_e, float& yaw_e) {
  // ---- _INDIPositionControl, :278-296
  pos_e = tg.pos - s.pos;
  const float inv_dt = DSIM_RCP(dt);
  V3 a_e;
  a_e.x = clampf((pos_e.x * T.kp + tg.vel.x - s.vel.x) * T.kd + tg.acc.x - (s.vel.x - m.last_vel.x) * inv_dt, -6.0f, 6.0f);
  a_e.y = clampf((pos_e.y * T.kp + tg.vel.y - s.vel.y) * T.kd + tg.acc.y - (s.vel.y - m.last_vel.y) * inv_dt, -6.0f, 6.0f);
  a_e.z = clampf((pos_e.z * T.kp + tg.vel.z - s.vel.z) * T.kd + tg.acc.z - (s.vel.z - m.last_vel.z) * inv_dt, -6.0f, 6.0f);
  m.last_vel = s.vel;
  // ---- Euler angles and G, :301-333
  const Euler e = euler_from_quat<WANT_YAW>(s.q);
  const float sph = e.sph, cph = e.cph, sth = e.sth, cth = e.cth, sps = e.sps, cps = e.cps;
  // ---- pinv(G) . accel_e, :314-339.  G = [T u | T w | b] with b = R e_z the thrust direction, u = db/droll,
  // w = db/dpitch (the reference writes the nine entries out, :314-333).  u, w, b are mutually ORTHOGONAL with
  // |u| = |b| = 1, |w| = |cos roll|, so the inverse is the scaled transpose
  //     inc_roll = u.a / T,   inc_pitch = w.a / (T cos^2 roll),   inc_thrust = b.a
  // (np.linalg.pinv agrees to 4e-15 relative over 20 000 random attitudes).  Only the pitch row is singular at
  // roll = +-90 deg (det G = T^2 cos roll); its denominator is clamped there (finite output; numpy's pinv
  // switches to the minimum-norm solution on that measure-zero set).
  const float iT = 1.0f / 9.81f;
  const float u0 = cph * sps - sph * cps * sth, u1 = -sph * sps * sth - cps * cph, u2 = -cth * sph;
  const float w0 = cph * cps * cth, w1 = cph * sps * cth, w2 = -sth * cph;
  const float b0 = sph * sps + cph * cps * sth, b1 = cph * sps * sth - cps * sph, b2 = cph * cth;
  const float inc0 = (u0 * a_e.x + u1 * a_e.y + u2 * a_e.z) * iT;
  const float inc1 = (w0 * a_e.x + w1 * a_e.y + w2 * a_e.z) * iT * DSIM_RCP(fmaxf(cph * cph, 1e-28f));
  const float inc2 = b0 * a_e.x + b1 * a_e.y + b2 * a_e.z;
  const float thrust = m.last_thrust + inc2;                                 // :347
  float target_yaw = tg.yaw;                                                 // == psi + norm_ang(psi* - psi) mod 2 pi
  if (WANT_YAW) {
    const float yaw_inc = norm_ang(tg.yaw - e.yaw);                          // :341
    target_yaw = e.yaw + yaw_inc;                                            // :344-346
    yaw_e = target_yaw - e.yaw;                                              // :227
  }
  const V3 target_euler = v3(e.roll + inc0, e.pitch + inc1, target_yaw);
  // ---- _INDIAttitudeControl, :388-402
  const Q4 tq = quat_from_euler(target_euler);
  const Q4 q = s.q;
  float ew = q.w * tq.w + q.x * tq.x + q.y * tq.y + q.z * tq.z;             // quat_inv_comp, math.py:23-31
  float ex = q.w * tq.x - q.x * tq.w - q.y * tq.z + q.z * tq.y;
  float ey = q.w * tq.y + q.x * tq.z - q.y * tq.w - q.z * tq.x;
  float ez = q.w * tq.z - q.x * tq.y + q.y * tq.x - q.z * tq.w;
  if (ew < 0.0f) { ex = -ex; ey = -ey; ez = -ez; }                           // quat_wrap_shortest, math.py:46-51
  const V3 rate_sp = v3(T.katt[0] * ex, T.katt[1] * ey, T.katt[2] * ez);
  indi_rate<NACT>(T, inv_dt, s, rate_sp, thrust, m);
}

// ===========================================================================
// morphing hexa: physics P3 and the 6-DOF INDI law C5 with WLS allocation C6
// ===========================================================================

// P3: BaseAviary._morphing_hexa_physics, BaseAviary.py:1398-1403, 1429-1457: force [0,0,F_j] and
// torque [0,0,tau_j] in the tilted prop link frames (rigid composite body, see params.py).
// nz: 12 scaled normals (f_noise[6], m_noise[6]) or nullptr.
template <class DT>
__device__ __forceinline__ void hexa_wrench(DT& T, const float cmd[6], const float* nz, V3& F, V3& tau) {
  // F = sum_j f_j a_j ;  tau = sum_j f_j (r_j x a_j) + tq_j a_j   (a_j: rotor axis, r_j: lever arm; r_j x a_j is a
  // per-type constant, so the wrench is three 3x6 matrix-vector products instead of six cross products)
  F = v3(0, 0, 0); tau = v3(0, 0, 0);
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const float rpm = T.scale[j] * cmd[j] + T.cnst[j];
    const float f = rpm * rpm * T.kf + (nz ? nz[j] : 0.0f);
    const float tq = (rpm * rpm * T.km + (nz ? nz[6 + j] : 0.0f)) * T.spin[j];   // :1439-1440
    F = F + f * v3(T.raxis[j][0], T.raxis[j][1], T.raxis[j][2]);
    tau = tau + f * v3(T.rxa[j][0], T.rxa[j][1], T.rxa[j][2]) + tq * v3(T.raxis[j][0], T.raxis[j][1], T.raxis[j][2]);
  }
}

// the same split (see quad_wrench_base): rpm_j, hence f0_j and tq0_j, are constant over the sub-steps
struct HexaBase { V3 F, tau; };
template <class DT>
__device__ __forceinline__ HexaBase hexa_wrench_base(DT& T, const float cmd[6]) {
#ifdef DSIM_HEXA_BASE_FP32          // (A/B knob: what the fp64 sums cost)
  typedef float acc_t;
#else
  typedef double acc_t;
#endif
  acc_t F[3] = {0.0, 0.0, 0.0}, tau[3] = {0.0, 0.0, 0.0};      // (fp64 sums: see quad_wrench_base)
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const float rpm = T.scale[j] * cmd[j] + T.cnst[j];
    const float f = rpm * rpm * T.kf, tq = rpm * rpm * T.km * T.spin[j];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#if defined(DSIM_HEXA_BASE_FP32) || !DSIM_HEXA_BASE_CONST64
      F[k] += (acc_t)f * (acc_t)T.raxis[j][k];
      tau[k] += (acc_t)f * (acc_t)T.rxa[j][k] + (acc_t)tq * (acc_t)T.raxis[j][k];
#else
      F[k] += (double)f * T.raxis64[j][k];                 // (the constants as doubles in the table: 36 conversions less per launch)
      tau[k] += (double)f * T.rxa64[j][k] + (double)tq * T.raxis64[j][k];
#endif
    }
  }
  return HexaBase{V3{(float)F[0], (float)F[1], (float)F[2]}, V3{(float)tau[0], (float)tau[1], (float)tau[2]}};
}
template <class DT>
__device__ __forceinline__ void hexa_wrench_noise(DT& T, const HexaBase& b, const float nz[12], V3& F, V3& tau) {
  F = b.F; tau = b.tau;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    // ((n spin_j) a_j = n (spin_j a_j) bit for bit: spin_j = +-1; the product is a per-type constant, DevType.spax)
    F = F + nz[j] * v3(T.raxis[j][0], T.raxis[j][1], T.raxis[j][2]);
    tau = tau + nz[j] * v3(T.rxa[j][0], T.rxa[j][1], T.rxa[j][2]) + nz[6 + j] * v3(T.spax[j][0], T.spax[j][1], T.spax[j][2]);
  }
}

// The default noise streams of the six-actuator kinds (see noise_normals): W = L z added to the noise-free wrench, z = six normals of
// deviation 0.01, L = DevType.nchol (rows of the lower triangle: W = (F, tau)) — 21 multiply-adds.
template <class DT>
__device__ __forceinline__ void hexa_wrench_z(DT& T, const V3 F0, const V3 tau0, const float z[6], V3& F, V3& tau) {
  // (T may live in the constant address space: the factor is indexed through T, never through a plain pointer)
#define DSIM_L(k) T.nchol[k]
  F.x = __builtin_fmaf(DSIM_L(0), z[0], F0.x);
  F.y = __builtin_fmaf(DSIM_L(2), z[1], __builtin_fmaf(DSIM_L(1), z[0], F0.y));
  F.z = __builtin_fmaf(DSIM_L(5), z[2], __builtin_fmaf(DSIM_L(4), z[1], __builtin_fmaf(DSIM_L(3), z[0], F0.z)));
  tau.x = __builtin_fmaf(DSIM_L(9), z[3], __builtin_fmaf(DSIM_L(8), z[2], __builtin_fmaf(DSIM_L(7), z[1], __builtin_fmaf(DSIM_L(6), z[0], tau0.x))));
  tau.y = __builtin_fmaf(DSIM_L(14), z[4], __builtin_fmaf(DSIM_L(13), z[3], __builtin_fmaf(DSIM_L(12), z[2], __builtin_fmaf(DSIM_L(11), z[1],
          __builtin_fmaf(DSIM_L(10), z[0], tau0.y)))));
  tau.z = __builtin_fmaf(DSIM_L(20), z[5], __builtin_fmaf(DSIM_L(19), z[4], __builtin_fmaf(DSIM_L(18), z[3], __builtin_fmaf(DSIM_L(17), z[2],
          __builtin_fmaf(DSIM_L(16), z[1], __builtin_fmaf(DSIM_L(15), z[0], tau0.z))))));
#undef DSIM_L
}

// Drones whose WLS allocation leaves the closed-form first iteration are queued here and finished by
// a second, tiny kernel (k_wls_fallback): the fp64 active-set loop needs ~250 registers and 2 KB of
// scratch, which must not be charged to every lane of the step kernels.
struct FbEntry { long long drone; float v[6]; };
struct FbList { FbEntry* entries; unsigned long long* count; unsigned long long* counters; };

// C6 fallback: the full active-set loop of wls_alloc (dronesim/control/wls_alloc.py:125-350) for
// the rare drones whose first-iteration solution leaves the +-1.0-slackened box.  fp64: the rows
// of A are scaled by gamma*Wv up to 1e8.  lstsq by Householder QR (A_free always contains the
// identity rows, so it has full column rank).  Returns 0 ok, -1 "solution failed" (:350), -2 where
// the reference would raise.  The loop indexes its matrices and index sets dynamically (free set, pivots), so the
// working set lives in LDS (WlsWork, one per lane of k_wls_fallback) — in registers it would be 2 KB of scratch.
struct WlsWork {
  double A[12][6], Af[12][6], Q[12][6], d[12], rhs[12], u[6], u_opt[6], p[6], p_free[6], W[6], Lambda[6];
  double umin[6], umax[6];
  int free_index[6], lookup[6];
};
template <class DT>
__device__ __forceinline__ int wls_active_set(DT& T, const float v[6], const float umin_[6],
                                           const float umax_[6], float u_out[6], WlsWork& ws) {
  const double gam = 100000.0;
  const double Wv[6] = {1000, 1000, 0.1, 10, 10, 100};          // INDIControl_6DOF.py:614
  double (&A)[12][6] = ws.A, (&Af)[12][6] = ws.Af, (&Q)[12][6] = ws.Q;
  double (&d)[12] = ws.d, (&rhs)[12] = ws.rhs;
  double (&u)[6] = ws.u, (&u_opt)[6] = ws.u_opt, (&p)[6] = ws.p, (&p_free)[6] = ws.p_free, (&W)[6] = ws.W, (&Lambda)[6] = ws.Lambda;
  int (&free_index)[6] = ws.free_index, (&lookup)[6] = ws.lookup;
  double (&umin)[6] = ws.umin, (&umax)[6] = ws.umax;      // indexed by the pivot: in LDS with the rest
  for (int i = 0; i < 6; ++i) { umin[i] = (double)umin_[i]; umax[i] = (double)umax_[i]; }
  int n_free = 0, free_chk = -1, iter = 0, n_p_free = 6, id_alpha = 0;
  bool alpha_set = false;
  double alpha = 0.0;
  for (int i = 0; i < 6; ++i) { u[i] = ((double)umax[i] + (double)umin[i]) * 0.5; W[i] = 0.0; p_free[i] = 0.0; }
  for (int i = 0; i < 6; ++i) { lookup[i] = n_free; free_index[n_free++] = i; }
  for (int i = 0; i < 6; ++i) {
    d[i] = gam * Wv[i] * (double)v[i];
    for (int j = 0; j < 6; ++j) { A[i][j] = gam * Wv[i] * (double)T.B[i][j]; d[i] -= A[i][j] * u[j]; }
  }
  for (int i = 6; i < 12; ++i) {
    for (int j = 0; j < 6; ++j) A[i][j] = (j == i - 6) ? 1.0 : 0.0;   // Wu = 1, up = None -> b = 0
    d[i] = -u[i - 6];
  }
  while (iter < 100) {
    ++iter;
    for (int i = 0; i < 6; ++i) { p[i] = 0.0; u_opt[i] = u[i]; }
    if (free_chk != n_free) {
      for (int i = 0; i < 12; ++i) for (int j = 0; j < n_free; ++j) Af[i][j] = A[i][free_index[j]];
      free_chk = n_free;
    }
    if (n_free) {   // p_free = lstsq(Af[:, :n_free], d) by Householder QR on copies
      for (int i = 0; i < 12; ++i) { rhs[i] = d[i]; for (int j = 0; j < n_free; ++j) Q[i][j] = Af[i][j]; }
      for (int k = 0; k < n_free; ++k) {
        double nrm = 0.0;
        for (int i = k; i < 12; ++i) nrm += Q[i][k] * Q[i][k];
        nrm = sqrt(nrm);
        const double akk = Q[k][k];
        const double beta = akk >= 0.0 ? -nrm : nrm;
        const double v0 = akk - beta;                     // reflector v = (v0, Q[k+1..][k]); Q[k][k] <- beta
        double vtv = v0 * v0;
        for (int i = k + 1; i < 12; ++i) vtv += Q[i][k] * Q[i][k];
        if (vtv > 0.0) {
          for (int j = k + 1; j < n_free; ++j) {
            double dotv = v0 * Q[k][j];
            for (int i = k + 1; i < 12; ++i) dotv += Q[i][k] * Q[i][j];
            const double f = 2.0 * dotv / vtv;
            Q[k][j] -= f * v0;
            for (int i = k + 1; i < 12; ++i) Q[i][j] -= f * Q[i][k];
          }
          double dotv = v0 * rhs[k];
          for (int i = k + 1; i < 12; ++i) dotv += Q[i][k] * rhs[i];
          const double f = 2.0 * dotv / vtv;
          rhs[k] -= f * v0;
          for (int i = k + 1; i < 12; ++i) rhs[i] -= f * Q[i][k];
        }
        Q[k][k] = beta;
      }
      for (int k = n_free - 1; k >= 0; --k) {
        double acc = rhs[k];
        for (int j = k + 1; j < n_free; ++j) acc -= Q[k][j] * p_free[j];
        p_free[k] = acc / Q[k][k];
      }
      n_p_free = n_free;
    }
    for (int i = 0; i < n_free; ++i) { p[free_index[i]] = p_free[i]; u_opt[free_index[i]] += p_free[i]; }
    int n_inf = 0;
    for (int i = 0; i < 6; ++i)
      if (u_opt[i] >= ((double)umax[i] + 1.0) || u_opt[i] <= ((double)umin[i] - 1.0)) ++n_inf;
    if (n_inf == 0) {
      for (int i = 0; i < 6; ++i) { u[i] = u_opt[i]; Lambda[i] = 0.0; }
      for (int i = 0; i < 12; ++i) {
        for (int k = 0; k < n_free; ++k) d[i] -= Af[i][k] * p_free[k];
        for (int k = 0; k < 6; ++k) Lambda[k] += A[i][k] * d[i];
      }
      bool brk = true;
      for (int i = 0; i < 6; ++i) {
        Lambda[i] *= W[i];
        if (Lambda[i] < -1.1920929e-07) {
          brk = false; W[i] = 0.0;
          if (lookup[i] < 0) { lookup[i] = n_free; free_index[n_free++] = i; }
        }
      }
      if (brk) { for (int i = 0; i < 6; ++i) u_out[i] = (float)u[i]; return 0; }
    } else {
      alpha = INFINITY; alpha_set = true; id_alpha = 0;
    }
    if (!alpha_set) return -2;
    for (int i = 0; i < n_free; ++i) {
      const int id = free_index[i];
      double at;
      if (fabs(p[id]) > 1.1920929e-07) at = p[id] < 0 ? ((double)umin[id] - u[id]) / p[id] : ((double)umax[id] - u[id]) / p[id];
      else at = INFINITY;
      if (at < alpha) { alpha = at; id_alpha = id; }
    }
    for (int i = 0; i < 6; ++i) u[i] += alpha * p[i];
    const int k_len = n_free < n_p_free ? n_free : n_p_free;
    for (int i = 0; i < 12; ++i) for (int k = 0; k < k_len; ++k) d[i] -= Af[i][k] * alpha * p_free[k];
    W[id_alpha] = p[id_alpha] > 0 ? 1.0 : -1.0;
    --n_free;
    if (n_free < 0 || lookup[id_alpha] < 0) return -2;
    free_index[lookup[id_alpha]] = free_index[n_free];
    lookup[free_index[lookup[id_alpha]]] = lookup[id_alpha];
    lookup[id_alpha] = -1;
  }
  return -1;
}

// C5: INDIControl_6DOF.computeControl, INDIControl_6DOF.py:259-634.
// Infeasible first iterations are queued in `fb` (see FbList).
template <bool WANT_YAW, class DT>
__device__ __forceinline__ void indi_hexa(DT& T, float dt, const Rigid& s, const Target& tg,
                                          CtrlMem<6>& m, V3& pos_e, float& yaw_e, const FbList& fb, long long drone) {
  pos_e = tg.pos - s.pos;                                                     // :397
  const float inv_dt = DSIM_RCP(dt);
  V3 a_e;                                                                     // :399-413 (no target_acc)
  a_e.x = clampf((pos_e.x * T.kp + tg.vel.x - s.vel.x) * T.kd - (s.vel.x - m.last_vel.x) * inv_dt, -6.0f, 6.0f);
  a_e.y = clampf((pos_e.y * T.kp + tg.vel.y - s.vel.y) * T.kd - (s.vel.y - m.last_vel.y) * inv_dt, -6.0f, 6.0f);
  a_e.z = clampf((pos_e.z * T.kp + tg.vel.z - s.vel.z) * T.kd - (s.vel.z - m.last_vel.z) * inv_dt, -6.0f, 6.0f);
  m.last_vel = s.vel;
  const Euler e = euler_from_quat<WANT_YAW>(s.q);                             // :418
  const float sph = e.sph, cph = e.cph, sth = e.sth, cth = e.cth, sps = e.sps, cps = e.cps;
  // only the thrust increment survives (target_euler is forced to zero, :495): the third row of inv(G) is the
  // thrust direction b = R e_z itself (see indi_quad)
  const float inc2 = (sph * sps + cph * cps * sth) * a_e.x + (cph * sps * sth - cps * sph) * a_e.y + (cph * cth) * a_e.z;
  const float thrust = m.last_thrust + inc2;                                  // :492
  if (WANT_YAW) yaw_e = 0.0f - e.yaw;                                         // :336, target_euler = 0
  // attitude: target quaternion = identity -> quat_inv_comp(q, (0,0,0,1)) = (-x,-y,-z,w); no wrap (:543-545)
  const float ex0 = -s.q.x, ey0 = -s.q.y, ez = -s.q.z;
  const float ex = cps * ex0 + sps * ey0, ey = -sps * ex0 + cps * ey0;        // inv(R_psi) . att_err.xy, :549-557
  const M3 R = matrix_from_quat(s.q);                                         // :566
  const V3 wb = mulT(R, s.w);
  float v[6];
  v[0] = (T.katt[0] * ex - wb.x) * T.krate[0] - (wb.x - m.last_rates.x) * inv_dt;   // :560-592
  v[1] = (T.katt[1] * ey - wb.y) * T.krate[1] - (wb.y - m.last_rates.y) * inv_dt;
  v[2] = (T.katt[2] * ez - wb.z) * T.krate[2] - (wb.z - m.last_rates.z) * inv_dt;
  const V3 ab = mulT(R, a_e);                                                 // :589
  v[3] = ab.x; v[4] = ab.y; v[5] = ab.z;
  m.last_rates = wb;                                                          // :580
  m.last_thrust = thrust;                                                     // :598
  // WLS allocation, :607-628.  First iteration in closed form: u_opt = M1 v + M4 u0.
  float umin[6], umax[6], du[6];
  bool feasible = true;
#pragma unroll
  for (int j = 0; j < 6; ++j) { umin[j] = T.pmin[j] - m.cmd[j]; umax[j] = T.pmax[j] - m.cmd[j]; }
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < 6; ++i) acc += T.alloc[j][i] * v[i] + T.alloc2[j][i] * (0.5f * (umin[i] + umax[i]));
    du[j] = acc;
    feasible = feasible && !(acc >= umax[j] + 1.0f || acc <= umin[j] - 1.0f);   // wls_alloc.py:255-259
  }
  if (feasible) {
#pragma unroll
    for (int j = 0; j < 6; ++j) m.cmd[j] = clampf(m.cmd[j] + du[j], T.pmin[j], T.pmax[j]);   // :630-631
  } else if (drone >= 0) {   // cmd stays as it is; k_wls_fallback finishes this drone from (v, cmd)
    // (drone < 0: a lane that only keeps a mixed wave's control flow uniform, see k_step_mixed)
    const unsigned long long slot = atomicAdd(fb.count, 1ULL);
    FbEntry e;
    e.drone = drone;
#pragma unroll
    for (int j = 0; j < 6; ++j) e.v[j] = v[j];
    fb.entries[slot] = e;
  }
}
