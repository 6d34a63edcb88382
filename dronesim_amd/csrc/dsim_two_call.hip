// dsim_two_call.hip — the reference-shaped two-call loop: dsim_physics (Env.step), dsim_control / dsim_control2 (computeControl),
// dsim_step_adaptor (VelocityAviary / RPYTAviary), Physics.DYN (gfx950 only).
#include "dsim_kernels.h"

// ---- Env.step only ---------------------------------------------------------
template <bool NOISE, int NACT, bool PLANE = false, class DT>
__device__ __forceinline__ void physics_gen_body(DT& T, const StepK& a, long long i, const Addr& ad) {
  Rigid s;
  load_rigid(ad.sb, ad.sfs, ad.sl, s);
  float raw[NACT], cmd[NACT];
#pragma unroll
  for (int j = 0; j < NACT; ++j) raw[j] = a.action ? a.action[(long long)j * a.n_pad + i] : ldg<false>(ad.sb + (20 + j) * ad.sfs, ad.sl);
  preprocess_action<NACT>(T, raw, cmd);
  V3 ext = v3(0, 0, 0);
  if (a.ext_force) ext = v3(a.ext_force[i], a.ext_force[a.n_pad + i], a.ext_force[2 * a.n_pad + i]);
  if (NACT == 6 && T.kind != DSIM_DEV_KIND_QUAD) {
    if constexpr (NACT == 6) hexa_substeps<NOISE, true, false, PLANE>(T, a, i, s, cmd, a.step_index, ext, NOISE ? noise_id(a, i) : -1LL);
  } else {
    float prev[4];       // last_clipped_action of the previous step (drag of sub-step 0); this step's action without it
#pragma unroll
    for (int j = 0; j < 4; ++j) prev[j] = a.echo ? a.echo[(long long)j * a.n_pad + i] : cmd[j];
    quad_substeps<NOISE ? 2 : 0, NACT, true, 0, PLANE>(T, a, i, s, cmd, a.step_index, ext, prev, NOISE ? noise_id(a, i) : -1LL);
  }
  ground_watch(T, s, a.fb.counters, i < a.n);
  store_rigid(ad.sb, ad.sfs, ad.sl, s);
  if (a.echo) {
#pragma unroll
    for (int j = 0; j < NACT; ++j) a.echo[(long long)j * a.n_pad + i] = cmd[j];   // last_clipped_action, BaseAviary.py:545
  }
}
template <bool NOISE, bool UNIFORM, int NACT>
__global__ __launch_bounds__(256, DSIM_GEN_WAVES) void k_physics_gen(StepK a) {
  const long long i0 = (long long)blockIdx.x * 256;
  const unsigned p = tile_slot<UNIFORM>(a.type_id, i0, a.n_pad);
  const long long i = i0 + p;
  if (i >= a.n_pad) return;
  const Addr ad = make_addr(a, i0, p);
  DSIM_FOR_MY_TYPE(UNIFORM, a, i, (physics_gen_body<NOISE, NACT>(T, a, i, ad)));
}
template <bool NOISE, bool UNIFORM, int NACT>
__global__ __launch_bounds__(256, 1) void k_physics_plane(StepK a) {     // DSIM_OPT_PLANE (see k_step_plane)
  const long long i0 = (long long)blockIdx.x * 256;
  const unsigned p = tile_slot<UNIFORM>(a.type_id, i0, a.n_pad);
  const long long i = i0 + p;
  if (i >= a.n_pad) return;
  const Addr ad = make_addr(a, i0, p);
  DSIM_FOR_MY_TYPE(UNIFORM, a, i, (physics_gen_body<NOISE, NACT, true>(T, a, i, ad)));
}

// ---- computeControl only ----------------------------------------------------
template <int NACT, class DT>
__device__ __forceinline__ void control_gen_body(DT& T, const StepK& a, long long i, const Addr& ad) {
  Rigid s;
  CtrlMem<NACT> m;
  Target tg;
  load_rigid(ad.sb, ad.sfs, ad.sl, s);
  load_mem<NACT>(ad.sb, ad.sfs, ad.sl, m);
  load_target(ad.tb, ad.tfs, ad.tl, tg);
  V3 pos_e;
  float yaw_e = 0.0f;
  if (NACT == 6 && T.kind == DSIM_DEV_KIND_HEXA) {
    if constexpr (NACT == 6) {
      if (a.yaw_e_out) indi_hexa<true>(T, a.dt_ctrl, s, tg, m, pos_e, yaw_e, a.fb, i);
      else indi_hexa<false>(T, a.dt_ctrl, s, tg, m, pos_e, yaw_e, a.fb, i);
    }
  } else {
    if (a.yaw_e_out) indi_quad<true, NACT>(T, a.dt_ctrl, s, tg, m, pos_e, yaw_e);
    else indi_quad<false, NACT>(T, a.dt_ctrl, s, tg, m, pos_e, yaw_e);
  }
  store_mem<NACT>(ad.sb, ad.sfs, ad.sl, m);
  if (a.pos_e_out) {
    a.pos_e_out[i] = pos_e.x; a.pos_e_out[a.n_pad + i] = pos_e.y; a.pos_e_out[2 * a.n_pad + i] = pos_e.z;
  }
  if (a.yaw_e_out) a.yaw_e_out[i] = yaw_e;
  if (a.cmd_out) {
#pragma unroll
    for (int j = 0; j < NACT; ++j) a.cmd_out[(long long)j * a.n_pad + i] = m.cmd[j];
  }
}
// (per-lane types only: a homogeneous fleet is one run of k_control_runs)
template <int NACT>
__global__ __launch_bounds__(256, DSIM_GEN_WAVES) void k_control_gen(StepK a) {
  const long long i0 = (long long)blockIdx.x * 256;
  const unsigned p = tile_slot<false>(a.type_id, i0, a.n_pad);
  const long long i = i0 + p;
  if (i >= a.n_pad) return;
  const Addr ad = make_addr(a, i0, p);
  DSIM_FOR_MY_TYPE(false, a, i, (control_gen_body<NACT>(T, a, i, ad)));
}

// ---- the reference-shaped two-call loop, fast forms ------------------------------------------------------------------
// obs = env.step(action); action = ctrl.computeControlFromState(obs)  (examples/fly_INDI.py:223-239) is two entry
// points here, dsim_physics and dsim_control.  For a homogeneous quad fleet in whole 256-drone tiles both have a fast
// form with the fused kernel's addressing (scalar base + one lane offset, streaming accesses, no per-lane branches):
//   k_physics_fast  reads 13 rigid + 4 action floats, writes 13 rigid + 4 echoed action floats and, fused (OBS), the
//                   20-wide observation row of the NEW state (Env.step's return value, BaseAviary.py:547-555) —
//                   transposed through LDS so that the row-major [n][20] block of the tile is written linearly;
//   k_control_fast  reads 13 + 11 + 10, writes the 11 controller-memory floats (+ pos_e, yaw_e, and the command as a
//                   plain SoA array that the next dsim_physics takes as its action without a copy).
// 216 + 212 bytes per drone and iteration instead of the 480+ of physics_gen + observe + control_gen + copies.
#ifndef DSIM_PHYS_WAVES
#define DSIM_PHYS_WAVES DSIM_STEP_WAVES   // (79 VGPRs, 6 waves per SIMD.  Measured and rejected: 8 waves per SIMD — 64 VGPRs and
                                          // 12 B of scratch per lane, 320 against 327 us for the two-call loop, inside that box's run-to-run spread)
#endif
#ifndef DSIM_OBS_STREAM
#define DSIM_OBS_STREAM 1      // observation rows leave with the streaming hint when the state does (A/B knob of the build)
#endif
// The 20-wide observation rows of a whole-tile quad kernel (BaseAviary.py:780-790), see k_physics_fast: the wave's 64 rows
// through its private LDS block, out as five 16-byte stores per lane over consecutive addresses.
template <bool NT>
__device__ __forceinline__ void obs_rows20_out(vf4* rows, const StepK& a, long long i0, const Rigid& s, const float cmd[4]) {
  constexpr int W = 20;
  const Euler e = euler_from_quat<true>(s.q);                                        // BaseAviary.py:729
  const unsigned w = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  vf4* blk = rows + w * (64 * (W / 4));                    // the wave's 64 rows x 5 pieces
  vf4* r = blk + lane * (W / 4);
  r[0] = vf4{s.pos.x, s.pos.y, s.pos.z, s.q.x};
  r[1] = vf4{s.q.y, s.q.z, s.q.w, e.roll};
  r[2] = vf4{e.pitch, e.yaw, s.vel.x, s.vel.y};
  r[3] = vf4{s.vel.z, s.w.x, s.w.y, s.w.z};
  r[4] = vf4{cmd[0], cmd[1], cmd[2], cmd[3]};
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the wave's own LDS writes, then its own reads: in order
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const long long w0 = i0 + 64 * (long long)w;             // first row of this wave
  const long long left = a.n - w0;                         // rows of this wave that exist (the last tile may be ragged)
  vf4* dst = reinterpret_cast<vf4*>(a.obs_out + w0 * W);
#pragma unroll
  for (int k = 0; k < W / 4; ++k) {
    const unsigned p = (unsigned)k * 64u + lane;           // piece p of the block belongs to row p / 5
    const vf4 v = blk[p];
    if ((long long)(p / (W / 4)) < left) {
      if (NT && DSIM_OBS_STREAM) __builtin_nontemporal_store(v, dst + p); else dst[p] = v;
    }
  }
}
// ---- Physics.DYN ----------------------------------------------------------------------------------------------------------
// BaseAviary.step with PHYSICS == Physics.DYN (BaseAviary.py:510-545: the loop calls _dynamics(clipped_action, i) per drone
// and sub-step, :525-527, and skips p.stepSimulation, :541-543): the reference's own explicit model (dsim_device.h:dyn_substep)
// on quad types, any fleet size (ragged tails included), per-lane type ids of a table of quads by the waterfall.  One kernel
// family for both entry points:
//   CTRL = false  dsim_physics: Env.step — the action clipped (CtrlAviary.py:258-263) and echoed, the sub-steps, the 13
//                 rigid floats and the three rpy rates written back, optionally (OBS) the 20-wide rows Env.step returns
//   CTRL = true   dsim_step: the same followed by computeControl on the new state, as the example loop orders them
//                 (examples/fly_INDI.py:223-239); an explicit action serves the physics part only
// Reads 13 + 3 (+ 4 | + 11 + 10), writes 13 + 3 (+ 4 | + 11) floats per drone: bound by HBM like every other single-launch
// form; no noise (the model has none), no ground-plane watch (the pose is SET, :1814-1819: no engine step, no contact).
struct Cmd4 { float c0, c1, c2, c3; };
template <bool CTRL, bool NT, class DT>
__device__ __forceinline__ void dyn_body(DT& T, const StepK& a, long long i0, const Addr& ad, Rigid& s, Cmd4& cmd_out) {
  float cmd[4];
  // (per-drone arrays beside the state: wave-uniform base + the lane's byte offset, like the state's own accesses)
  const unsigned lo = 4u * threadIdx.x;
  float* const rb = a.dyn_rates + i0;
  load_rigid<NT>(ad.sb, ad.sfs, ad.sl, s);
  V3 rr = v3(ldg<NT>(rb, lo), ldg<NT>(rb + a.n_pad, lo), ldg<NT>(rb + 2 * a.n_pad, lo));     // self.rpy_rates, :1785
  CtrlMem<4> m;
  Target tg;
  if (CTRL) {
    load_mem<4, NT>(ad.sb, ad.sfs, ad.sl, m);
    load_target<NT>(ad.tb, ad.tfs, ad.tl, tg);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float raw = a.action ? ldg<NT>(a.action + (long long)j * a.n_pad + i0, lo)
                               : (CTRL ? m.cmd[j] : ldg<NT>(ad.sb + (20 + j) * ad.sfs, ad.sl));
    cmd[j] = clampf(raw, T.pmin[j], T.pmax[j]);                                               // CtrlAviary.py:258-263
  }
  const DynBase b = dyn_base(T, cmd);
  for (int k = 0; k < a.substeps; ++k) dyn_substep(T, a.dt_phys, b, s, rr);
  const V3 w_new = dyn_reported_ang_vel((a.options & DSIM_OPT_DYN_BODY_RATES) != 0, s.q, rr);  // :1821-1826
  if (a.substeps > 0) s.w = w_new;
  if (CTRL) {
    V3 pos_e;
    float yaw_e;
    indi_quad<false>(T, a.dt_ctrl, s, tg, m, pos_e, yaw_e);
  }
  const unsigned so = pin_lane_offset(ad.sl), lo2 = pin_lane_offset(lo);
  store_rigid<NT>(ad.sb, ad.sfs, so, s);
  stg<NT>(rb, lo2, rr.x); stg<NT>(rb + a.n_pad, lo2, rr.y); stg<NT>(rb + 2 * a.n_pad, lo2, rr.z);   // :1828
  if (CTRL) store_mem<4, NT>(ad.sb, ad.sfs, so, m);
  if (!CTRL && a.echo) {
#pragma unroll
    for (int j = 0; j < 4; ++j) stg<NT>(a.echo + (long long)j * a.n_pad + i0, lo2, cmd[j]);    // last_clipped_action, :545
  }
  cmd_out = Cmd4{cmd[0], cmd[1], cmd[2], cmd[3]};
}
// OBS (Env.step only): the 20-wide observation rows of the NEW state written by the same launch (obs_rows20_out, as
// k_physics_fast: the wave's 64 rows through its private LDS block, behind the type waterfall where the wave is whole again).
template <bool CTRL, bool NT, bool OBS = false>
__global__ __launch_bounds__(256, OBS ? 4 : DSIM_STEP_WAVES) void k_dyn(StepK a) {     // (OBS at 5 waves per SIMD: 12 B of scratch)
  __shared__ __attribute__((aligned(16))) vf4 rows[OBS ? 4 * 64 * 5 : 1];
  const long long i0 = (long long)blockIdx.x * 256;
  const long long i = i0 + threadIdx.x;
  if (i >= a.n_pad) return;                            // (n_pad is a multiple of 64: whole waves leave)
  const Addr ad = make_addr(a, i0, threadIdx.x);
  Rigid s_new;
  Cmd4 c_new;
  // one body for homogeneous and mixed quad fleets: the wave peels one type per turn (a homogeneous fleet: one turn), the
  // type's constants through the constant address space at a wave-uniform index (scalar loads)
  const int my_t = a.type_id ? (int)a.type_id[i] : 0;
  for (;;) {
    const int cur_t = __builtin_amdgcn_readfirstlane(my_t);
    if (my_t == cur_t) { dyn_body<CTRL, NT>(dev_type(a.types, cur_t), a, i0, ad, s_new, c_new); break; }
  }
  if (OBS) { const float cmd_new[4] = {c_new.c0, c_new.c1, c_new.c2, c_new.c3}; obs_rows20_out<NT>(rows, a, i0, s_new, cmd_new); }
}

// LOOP: the launch has SEVERAL sub-steps on the default noise lattice (the examples' five, examples/fly_INDI.py:139-141): the
// instance that carries the body-frame form of the step and the Box-Muller tables (quad_substeps: LOOPED), as k_step_fast's
// looped instances do — Env.step of 4 194 304 quads x 5 sub-steps was bound by vector issue on the single-sub-step body.
template <bool NOISE, bool NT, bool OBS, bool LOOP = false>
__global__ __launch_bounds__(256, DSIM_PHYS_WAVES) void k_physics_fast(StepK a) {
  constexpr int W = 20;
  // Observation rows: each wave owns 64 consecutive rows = 5 120 contiguous bytes of the row-major [n][20] output.  Lane r
  // writes ITS row to the wave's private LDS block as five 16-byte pieces (row stride 80 B: eight lanes cover the 32
  // banks exactly once), and the block goes out as five 16-byte stores per lane over consecutive addresses.  No
  // workgroup barrier — the block is the wave's own — and no index arithmetic per element (round 2: a __syncthreads,
  // twenty dword stores per lane and a division by W each; SQ_WAIT_ANY 0.36).
  __shared__ __attribute__((aligned(16))) vf4 rows[OBS ? 4 * 64 * (W / 4) : 1];
  constexpr bool TAB = NOISE && LOOP;
  __shared__ NoiseTab ntab_[TAB ? 1 : 0 + 1];
  const NoiseTab* const ntab = TAB ? &ntab_[0] : nullptr;
  if (TAB) noise_tab_init(ntab_[0], threadIdx.x);
  const DevType& T = a.types[0];
  const long long sfs = a.st.field_stride;
  const unsigned sl = 4u * kv_lane(a.st, threadIdx.x);
  const long long i0 = (long long)blockIdx.x * 256;
  float* const sb = a.st.base + kv_off(a.st, i0);
  const long long i = i0 + threadIdx.x;
  Rigid s;
  load_rigid<NT>(sb, sfs, sl, s);
  if (TAB) __syncthreads();
  float cmd[4];
  if (a.action_rows) {                // (wave-uniform) the action row-major [n][4] (DSIM_OPT_ACTION_ROWS): one 16-byte load per lane
    vf4 r = vf4{0.0f, 0.0f, 0.0f, 0.0f};
    if (i < a.n) {                                                       // (rows exist for real drones only)
      const vf4* ar = reinterpret_cast<const vf4*>(a.action) + i0;
      r = NT ? __builtin_nontemporal_load(ar + threadIdx.x) : ar[threadIdx.x];
    }
    cmd[0] = r.x; cmd[1] = r.y; cmd[2] = r.z; cmd[3] = r.w;
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      cmd[j] = a.action ? ldg<NT>(a.action + (long long)j * a.n_pad + i0, 4u * threadIdx.x) : ldg<NT>(sb + (20 + j) * sfs, sl);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) cmd[j] = clampf(cmd[j], T.pmin[j], T.pmax[j]);           // CtrlAviary.py:258-263
  if (NOISE && a.step_index_dev) a.step_index += *a.step_index_dev;
  if constexpr (LOOP) quad_substeps<NOISE ? 1 : 0, 4, false, 0, false, 0, true, TAB ? 1 : 0>(T, a, i, s, cmd, a.step_index, V3{-0.0f, -0.0f, -0.0f}, nullptr, -1, ntab);
  else quad_substeps<NOISE ? 1 : 0, 4, false, 0, false, 1>(T, a, i, s, cmd, a.step_index);      // (both noise lattices)
  ground_watch(T, s, a.fb.counters, i < a.n);
  const unsigned so = pin_lane_offset(sl);
  store_rigid<NT>(sb, sfs, so, s);
  if (a.echo) {
#pragma unroll
    for (int j = 0; j < 4; ++j) stg<NT>(a.echo + (long long)j * a.n_pad + i0, 4u * threadIdx.x, cmd[j]);   // BaseAviary.py:545
  }
  if (OBS) obs_rows20_out<NT>(rows, a, i0, s, cmd);
}

template <bool NT, bool WANT_YAW>
__global__ __launch_bounds__(256, DSIM_STEP_WAVES) void k_control_fast(StepK a) {
  const DevType& T = a.types[0];
  const long long sfs = a.st.field_stride, tfs = a.tg.field_stride;
  const unsigned sl = 4u * kv_lane(a.st, threadIdx.x), tl = 4u * kv_lane(a.tg, threadIdx.x);
  const long long i0 = (long long)blockIdx.x * 256;
  float* const sb = a.st.base + kv_off(a.st, i0);
  const float* const tb = a.tg.base + kv_off(a.tg, i0);
  Rigid s;
  CtrlMem<4> m;
  Target tg;
  load_rigid<NT>(sb, sfs, sl, s);
  load_mem<4, NT>(sb, sfs, sl, m);
  load_target<NT>(tb, tfs, tl, tg);
  V3 pos_e;
  float yaw_e = 0.0f;
  indi_quad<WANT_YAW>(T, a.dt_ctrl, s, tg, m, pos_e, yaw_e);
  const unsigned so = pin_lane_offset(sl);
  store_mem<4, NT>(sb, sfs, so, m);
  const unsigned lo = 4u * threadIdx.x;
  if (a.cmd_out) {
#pragma unroll
    for (int j = 0; j < 4; ++j) stg<NT>(a.cmd_out + (long long)j * a.n_pad + i0, lo, m.cmd[j]);
  }
  if (a.pos_e_out) {
    stg<NT>(a.pos_e_out + i0, lo, pos_e.x); stg<NT>(a.pos_e_out + a.n_pad + i0, lo, pos_e.y);
    stg<NT>(a.pos_e_out + 2 * a.n_pad + i0, lo, pos_e.z);
  }
  if (WANT_YAW) stg<NT>(a.yaw_e_out + i0, lo, yaw_e);
}

// ---- the same two-call loop for every other fleet kind: runs of one type --------------------------------------------------
// examples/fly_hexa_6DOF.py:214-221 is the same loop on the morphing hexa; BASELINE config 5 flies quads and hexas
// together with the neighbour-downwash term.  The fleet is stored as runs of one type each (dsim_step_args.runs: what
// CtrlAviary makes of an interleaved fleet; a homogeneous fleet is ONE run), a workgroup runs the Env.step / computeControl
// of the run it falls in — the single-type body, per-type constants in SGPRs, the fused kernels' scalar-base addressing —
// and the launch serves all runs (RunTab, as k_step_runs).  Runs may begin and end inside a tile: a lane outside
// [lo, last) computes nothing and stores nothing, the neighbouring run's workgroup takes it.
//   k_physics_runs  13 rigid + n_act action floats in (+ the body-frame force of the downwash term), 13 rigid + n_act
//                   echoed action floats and the observation row of the NEW state out (20 wide for a quad-only table, 22
//                   wide with a morphing hexa in it; BaseAviary.py:780-790); noise keyed by the caller's drone index;
//                   optionally the next neighbour grid filled from the new positions (bin_next).
//   k_control_runs  13 + (11 | 13) + 10 in, controller memory + command + pos_e + yaw_e out; a hexa whose first WLS
//                   iteration leaves the box is queued for k_wls_fallback exactly as in k_step_hexa.
// Observation rows: 88-byte rows are 8-byte but not 16-byte aligned, and a run boundary inside a wave splits the wave's
// block of rows at a row boundary — so the wave-private LDS transpose of k_physics_fast is done in 8-byte pieces here
// (every piece belongs to exactly one row): each lane writes its row as W / 2 pieces, the block leaves as W / 2 stores of
// 8 bytes per lane over consecutive addresses, and a piece is stored when its row is one of this run's.
typedef float vf2 __attribute__((ext_vector_type(2)));
#define DSIM_OBS_WMAX 22
#ifndef DSIM_ROWS16
#define DSIM_ROWS16 1          // whole blocks of rows leave in 16-byte pieces (A/B knob of the build)
#endif
// IO (DSIM_OPT_CALLER_IO): the action is gathered from, and rows / command / errors are scattered to, the CALLER's drone
// number io_id[i].  The drones of a run are spread over the caller's whole range (even index quad, odd index hexa ...), so one
// run alone fills every other 88-byte row and every other dword of the command arrays: partial memory bursts, a
// read-modify-write each (measured, 4 194 304 interleaved drones: Env.step 351 us with the runs served one after the other
// against 190 us for a fleet of one type).  The IO instances therefore serve the runs SIDE BY SIDE (RunTab.block_map): a
// workgroup works on the SAME stretch of two neighbouring tiles of the map, i.e. of two runs that cover the same stretch of
// the caller's range.  First form: 512 threads = two whole tiles, the scattered arrays written with the default cache
// policy so that the halves of a line meet in the XCD's L2 before they leave for memory (418-435 us per loop iteration,
// traffic 1.14 x).  Second form, below: the outputs are ASSEMBLED in LDS over a window of DSIM_IO_WIN caller indices that
// starts at the workgroup's smallest one, and leave as whole lines (16-byte pieces with the streaming hint, like the rows
// of a single-type fleet); a flag per window row says whether this workgroup produced it (a hole belongs to another
// workgroup and is not touched), and a drone whose index falls outside the window writes its outputs itself, as before.
// Correct for any order, fast where the types are mixed evenly — the interleaved fleets BASELINE config 5 describes:
// traffic 1.001 x algorithmic, and the smaller the workgroup the better (two barriers couple its waves; same-box A/B of the
// loop: 512 threads 365 us, 256: 359, 128 — one wave of either tile, a window of 128 rows = 88 whole lines: 355 us).
// t = the thread's index inside its tile.
#ifndef DSIM_IO_WG
#define DSIM_IO_WG 128         // threads per workgroup of the caller-order kernels: the same stretch of two neighbouring tiles of the map
#endif
#define DSIM_IO_WIN DSIM_IO_WG
#define DSIM_IO_PARTS (512 / DSIM_IO_WG)          // workgroups per pair of tiles
// map entry and index inside its tile of a thread: the workgroup's first half works on the pair's first tile
#define DSIM_IO_ENTRY() (2 * (int)(blockIdx.x / DSIM_IO_PARTS) + (int)(threadIdx.x / (DSIM_IO_WG / 2)))
#define DSIM_IO_T() ((blockIdx.x % DSIM_IO_PARTS) * (DSIM_IO_WG / 2) + (threadIdx.x % (DSIM_IO_WG / 2)))
struct IoWin { float* win; unsigned char* flags; int base; };
// The window starts at the smallest caller index among the workgroup's live lanes, rounded down to 4 (16-byte aligned rows
// of any width, whole 16-byte pieces of the per-field arrays).  Every wave leaves its minimum in LDS and clears its share of
// the flags BEFORE its arithmetic; the workgroup meets once behind it (io_window_base), fills the window, meets again and
// writes it out — two barriers at the end of the waves' lives, none in front of their loads.
__device__ __forceinline__ void io_window_min(int id_or_max, int* wmin, unsigned char* flags) {
  int m = id_or_max;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = min(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63u) == 0) wmin[threadIdx.x >> 6] = m;
  flags[threadIdx.x] = 0;
}
__device__ __forceinline__ int io_window_base(const int* wmin) {
  __syncthreads();
  int b = wmin[0];
#pragma unroll
  for (int q = 1; q < DSIM_IO_WG / 64; ++q) b = min(b, wmin[q]);
  return __builtin_amdgcn_readfirstlane(b) & ~3;
}
struct IoRow { vf2 pc[11]; int id; bool have; };            // a lane's observation row on its way to the window
struct IoCtl { float v[10]; int id; bool have; };            // a lane's command (6), position error (3), yaw error
template <bool HEXA, bool NOISE, bool NT, bool OBS, bool IO, bool S1>
__device__ __forceinline__ void physics_run_body(const StepK& a, const RunOf& ro, float* rows_wave, unsigned t, IoRow& io,
                                                 const NoiseTab* tab = nullptr) {
  constexpr int NA = HEXA ? 6 : 4;
  const long long i0 = ro.i0, i = i0 + t;
  const long long w0 = i0 + (long long)(t & ~63u);                     // first drone of this wave
  if (w0 >= ro.last || w0 + 64 <= ro.lo) return;                       // (wave-uniform) nothing of this run in the wave
  const bool live = i >= ro.lo && i < ro.last;
  const int W = HEXA ? 22 : a.obs_w;                                   // row width: 20 for a quad-only table, 22 with a hexa in it
  const unsigned lane = t & 63u;
  if (live) {
    CDevType& T = dev_type(a.types, ro.type);
    const long long sfs = a.st.field_stride;
    const unsigned sl = 4u * kv_lane(a.st, t);
    float* const sb = a.st.base + kv_off(a.st, i0);
    Rigid s;
    load_rigid<NT>(sb, sfs, sl, s);
    float cmd[NA];
    long long id = i;
    if (IO) id = a.io_id[i];
    if (IO && a.action) {                     // the action is indexed by the caller's drone number: a gather
#pragma unroll
      for (int j = 0; j < NA; ++j) cmd[j] = clampf(a.action[(long long)j * a.n_pad + id], T.pmin[j], T.pmax[j]);
    } else {
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        const float raw = a.action ? ldg<NT>(a.action + (long long)j * a.n_pad + i0, 4u * t) : ldg<NT>(sb + (20 + j) * sfs, sl);
        cmd[j] = clampf(raw, T.pmin[j], T.pmax[j]);                                     // CtrlAviary.py:258-263
      }
    }
    V3 ext = v3(0, 0, 0);
    if (a.ext_force) ext = v3(a.ext_force[i], a.ext_force[a.n_pad + i], a.ext_force[2 * a.n_pad + i]);
    unsigned long long step_index = a.step_index;
    if (NOISE && a.step_index_dev) step_index += *a.step_index_dev;
    const long long nid = NOISE ? noise_id(a, i) : -1LL;
    // (S1: one sub-step per Env.step — BASELINE's metric definition — compiled straight-line, as in the fused kernels)
    if constexpr (HEXA) hexa_substeps<NOISE, false, S1, false, !S1, (NOISE && !S1) ? 1 : 0>(T, a, i, s, cmd, step_index, ext, nid, tab);
    else quad_substeps<NOISE ? 1 : 0, 4, false, S1 ? 1 : 0, false, -1, !S1, (NOISE && !S1) ? 1 : 0>(T, a, i, s, cmd, step_index, ext, nullptr, nid, tab);
    ground_watch(T, s, a.fb.counters, i < a.n);
    const unsigned so = pin_lane_offset(sl);
    float* const sb2 = (DSIM_LATE_STORE_BASE && !S1) ? const_cast<float*>(opaque_after(sb, s.pos.x)) : sb;   // (k_step_hexa)
    store_rigid<NT>(sb2, sfs, so, s);
    if (a.echo) {                                                                       // BaseAviary.py:545
#pragma unroll
      for (int j = 0; j < NA; ++j) stg<NT>(a.echo + (long long)j * a.n_pad + i0, 4u * t, cmd[j]);
      if (!HEXA && W == 22) {                 // a quad of a table with a six-actuator type: its rows 4, 5 hold zeros
        stg<NT>(a.echo + 4LL * a.n_pad + i0, 4u * t, 0.0f); stg<NT>(a.echo + 5LL * a.n_pad + i0, 4u * t, 0.0f);
      }
    }
    if (a.bin.count && i < a.n) bin_entry(a.bin, s.pos.x, s.pos.y, s.pos.z, a.bin.local_offset + i);   // next step's grid
    if (OBS && (!IO || i < a.n)) {
      const Euler e = euler_from_quat<true>(s.q);                                       // BaseAviary.py:729
      vf2 pc[11];
      pc[0] = vf2{s.pos.x, s.pos.y}; pc[1] = vf2{s.pos.z, s.q.x}; pc[2] = vf2{s.q.y, s.q.z}; pc[3] = vf2{s.q.w, e.roll};
      pc[4] = vf2{e.pitch, e.yaw}; pc[5] = vf2{s.vel.x, s.vel.y}; pc[6] = vf2{s.vel.z, s.w.x}; pc[7] = vf2{s.w.y, s.w.z};
      pc[8] = vf2{cmd[0], cmd[1]}; pc[9] = vf2{cmd[2], cmd[3]};
      if constexpr (HEXA) pc[10] = vf2{cmd[4], cmd[5]}; else pc[10] = vf2{0.0f, 0.0f};
      const int hw = W >> 1;
      if (IO) {                               // the kernel puts it into the window, behind the workgroup's first barrier
#pragma unroll
        for (int k = 0; k < 11; ++k) io.pc[k] = pc[k];
        io.id = (int)id; io.have = true;
      } else {
        vf2* r = reinterpret_cast<vf2*>(rows_wave + lane * (unsigned)W);
#pragma unroll
        for (int k = 0; k < 11; ++k) if (k < hw) r[k] = pc[k];
      }
    }
  }
  if (OBS && !IO) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the wave's own LDS writes, then its own reads: in order
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const unsigned hw = (unsigned)W >> 1;                                               // pieces per row
    const long long r_lo = ro.lo > w0 ? ro.lo - w0 : 0;                                 // this run's rows of the wave's block
    const long long r_hi = min(min(ro.last, a.n) - w0, 64LL);
    const unsigned p_lo = (unsigned)r_lo * hw, p_hi = r_hi > 0 ? (unsigned)r_hi * hw : 0u;
    const vf2* blk = reinterpret_cast<const vf2*>(rows_wave);
    if (DSIM_ROWS16 && p_lo == 0u && p_hi == 64u * hw && ((uintptr_t)a.obs_out & 15u) == 0) {
      // the whole block is this run's (every wave but those at a run's two ends): 64 rows of 80 / 88 bytes are 320 / 352
      // 16-byte pieces behind a 16-byte aligned address (64 rows in front of every block), stored as such
      const vf4* blk4 = reinterpret_cast<const vf4*>(rows_wave);
      vf4* dst4 = reinterpret_cast<vf4*>(a.obs_out + w0 * W);
      const unsigned n4 = 16u * (unsigned)W;
      for (unsigned p = lane; p < n4; p += 64u) {
        const vf4 v = blk4[p];
        if (NT && DSIM_OBS_STREAM) __builtin_nontemporal_store(v, dst4 + p); else dst4[p] = v;
      }
    } else {
      vf2* dst = reinterpret_cast<vf2*>(a.obs_out + w0 * W);
      for (unsigned k = 0; k < hw; ++k) {
        const unsigned p = k * 64u + lane;
        if (p >= p_lo && p < p_hi) {
          const vf2 v = blk[p];
          if (NT && DSIM_OBS_STREAM) __builtin_nontemporal_store(v, dst + p); else dst[p] = v;
        }
      }
    }
  }
}
#ifndef DSIM_PRUNS_WAVES
#define DSIM_PRUNS_WAVES 4
#endif
#ifndef DSIM_CRUNS_WAVES
#define DSIM_CRUNS_WAVES 4
#endif
template <bool NOISE, bool NT, bool OBS, bool S1>
__global__ __launch_bounds__(256, S1 ? DSIM_PRUNS_WAVES : 3) void k_physics_runs(StepK a, RunTab rt) {
  __shared__ __attribute__((aligned(16))) float rows[OBS ? 4 * 64 * DSIM_OBS_WMAX : 2];   // per wave: 64 rows
  DSIM_RUN_OF_BLOCK(rt, ro, blockIdx.x);
  float* rw = rows + (OBS ? (threadIdx.x >> 6) * (64 * DSIM_OBS_WMAX) : 0);
  IoRow none;
  DSIM_NOISE_TAB(NOISE && !S1, 256);
  if (ro.hexa) physics_run_body<true, NOISE, NT, OBS, false, S1>(a, ro, rw, threadIdx.x, none, ntab);
  else physics_run_body<false, NOISE, NT, OBS, false, S1>(a, ro, rw, threadIdx.x, none, ntab);
}
// DSIM_OPT_CALLER_IO: the same stretch of two neighbouring tiles of the side-by-side map, the rows assembled over the window
template <bool NOISE, bool NT, bool OBS, bool S1>
__global__ __launch_bounds__(DSIM_IO_WG, S1 ? DSIM_PRUNS_WAVES : 3) void k_physics_runs_io(StepK a, RunTab rt) {
  __shared__ __attribute__((aligned(16))) float win[OBS ? DSIM_IO_WIN * DSIM_OBS_WMAX : 4];
  __shared__ unsigned char flags[DSIM_IO_WIN];
  __shared__ int wmin[DSIM_IO_WG / 64];
  DSIM_RUN_OF_BLOCK(rt, ro, DSIM_IO_ENTRY());
  const unsigned t = DSIM_IO_T();
  IoRow io;
  io.have = false;
  if (OBS) {
    const long long i = ro.i0 + t;
    io_window_min(i >= ro.lo && i < ro.last && i < a.n ? a.io_id[i] : 0x7fffffff, wmin, flags);
  }
  DSIM_NOISE_TAB(NOISE && !S1, DSIM_IO_WG);
  if (ro.hexa) physics_run_body<true, NOISE, NT, OBS, true, S1>(a, ro, nullptr, t, io, ntab);
  else physics_run_body<false, NOISE, NT, OBS, true, S1>(a, ro, nullptr, t, io, ntab);
  if (OBS) {
    const int base = io_window_base(wmin);
    const int hw = a.obs_w >> 1;
    if (io.have) {
      const unsigned slot = (unsigned)(io.id - base);                  // where this row goes: the window, or straight out
      vf2* r;
      if (slot < (unsigned)DSIM_IO_WIN) {
        r = reinterpret_cast<vf2*>(win + slot * (unsigned)a.obs_w);
#pragma unroll
        for (int k = 0; k < 11; ++k) if (k < hw) r[k] = io.pc[k];
        flags[slot] = 1;
      } else {
        vf2* g = reinterpret_cast<vf2*>(a.obs_out + (long long)io.id * a.obs_w);
#pragma unroll
        for (int k = 0; k < 11; ++k) if (k < hw) g[k] = io.pc[k];
      }
    }
    __syncthreads();
    // the window leaves in 16-byte pieces; a piece holds two 8-byte halves, each inside ONE row (rows are 80 / 88 bytes).
    // (x / W by multiply-shift: exact for x < 11 272 with these constants.)
    const unsigned W = (unsigned)a.obs_w, n4 = DSIM_IO_WIN * W / 4u, mul = W == 22u ? 2979u : 3277u;
    const vf4* win4 = reinterpret_cast<const vf4*>(win);
    float* const g = a.obs_out + (long long)base * W;
    for (unsigned p = threadIdx.x; p < n4; p += (unsigned)DSIM_IO_WG) {
      const unsigned x = 4u * p;
      const bool fa = flags[(x * mul) >> 16] != 0, fb = flags[((x + 2u) * mul) >> 16] != 0;
      const bool whole = __ballot(fa && fb) == ~0ULL;     // 1 KB of whole pieces: streaming; holes: default policy (they merge in L2)
      if (fa || fb) {
        const vf4 v = win4[p];
        if (fa && fb) {
          if (NT && DSIM_OBS_STREAM && whole) __builtin_nontemporal_store(v, reinterpret_cast<vf4*>(g + x)); else *reinterpret_cast<vf4*>(g + x) = v;
        } else if (fa) *reinterpret_cast<vf2*>(g + x) = vf2{v.x, v.y};
        else *reinterpret_cast<vf2*>(g + x + 2u) = vf2{v.z, v.w};
      }
    }
  }
}

template <int KIND, bool NT, bool WANT_YAW, bool IO>
__device__ __forceinline__ void control_run_body(const StepK& a, const RunOf& ro, unsigned t, IoCtl& io) {
  constexpr bool HEXA = KIND != DSIM_DEV_KIND_QUAD;            // six actuators
  constexpr int NA = HEXA ? 6 : 4;
  const long long i0 = ro.i0, i = i0 + t;
  if (i >= ro.last || i < ro.lo) return;
  CDevType& T = dev_type(a.types, ro.type);
  const long long sfs = a.st.field_stride, tfs = a.tg.field_stride;
  const unsigned sl = 4u * kv_lane(a.st, t), tl = 4u * kv_lane(a.tg, t);
  float* const sb = a.st.base + kv_off(a.st, i0);
  const float* const tb = a.tg.base + kv_off(a.tg, i0);
  Rigid s;
  CtrlMem<NA> m;
  Target tg;
  load_rigid<NT>(sb, sfs, sl, s);
  load_mem<NA, NT>(sb, sfs, sl, m);
  load_target<NT>(tb, tfs, tl, tg);
  V3 pos_e;
  float yaw_e = 0.0f;
  if constexpr (KIND == DSIM_DEV_KIND_HEXA) indi_hexa<WANT_YAW>(T, a.dt_ctrl, s, tg, m, pos_e, yaw_e, a.fb, i);
  else indi_quad<WANT_YAW, NA>(T, a.dt_ctrl, s, tg, m, pos_e, yaw_e);       // (NA = 6: hexa_6DOF_simple)
  const unsigned so = pin_lane_offset(sl);
  store_mem<NA, NT>(sb, sfs, so, m);
  if (IO) {                 // the outputs go to the caller's drone number: the kernel puts them into the window
#pragma unroll
    for (int j = 0; j < NA; ++j) io.v[j] = m.cmd[j];
    if (!HEXA) { io.v[4] = 0.0f; io.v[5] = 0.0f; }   // a quad of a table with a six-actuator type: its rows 4, 5 hold zeros
    io.v[6] = pos_e.x; io.v[7] = pos_e.y; io.v[8] = pos_e.z; io.v[9] = yaw_e;
    io.id = a.io_id[i]; io.have = true;
    return;
  }
  const unsigned lo4 = 4u * t;
  if (a.cmd_out) {
#pragma unroll
    for (int j = 0; j < NA; ++j) stg<NT>(a.cmd_out + (long long)j * a.n_pad + i0, lo4, m.cmd[j]);
    if (!HEXA && a.obs_w == 22) {             // a quad of a table with a six-actuator type: its rows 4, 5 hold zeros
      stg<NT>(a.cmd_out + 4LL * a.n_pad + i0, lo4, 0.0f); stg<NT>(a.cmd_out + 5LL * a.n_pad + i0, lo4, 0.0f);
    }
  }
  if (a.pos_e_out) {
    stg<NT>(a.pos_e_out + i0, lo4, pos_e.x); stg<NT>(a.pos_e_out + a.n_pad + i0, lo4, pos_e.y);
    stg<NT>(a.pos_e_out + 2 * a.n_pad + i0, lo4, pos_e.z);
  }
  if (WANT_YAW) stg<NT>(a.yaw_e_out + i0, lo4, yaw_e);
}
template <bool NT, bool WANT_YAW>
__global__ __launch_bounds__(256, DSIM_CRUNS_WAVES) void k_control_runs(StepK a, RunTab rt) {
  DSIM_RUN_OF_BLOCK(rt, ro, blockIdx.x);
  IoCtl none;
  if (ro.quadlaw6) control_run_body<DSIM_DEV_KIND_HEXA_QUADLAW, NT, WANT_YAW, false>(a, ro, threadIdx.x, none);
  else if (ro.hexa) control_run_body<DSIM_DEV_KIND_HEXA, NT, WANT_YAW, false>(a, ro, threadIdx.x, none);
  else control_run_body<DSIM_DEV_KIND_QUAD, NT, WANT_YAW, false>(a, ro, threadIdx.x, none);
}
template <bool NT, bool WANT_YAW>
__global__ __launch_bounds__(DSIM_IO_WG, DSIM_CRUNS_WAVES) void k_control_runs_io(StepK a, RunTab rt) {
  __shared__ float win[10 * DSIM_IO_WIN];     // field f of the window: win[f * DSIM_IO_WIN + slot]; 0-5 command, 6-8 pos_e, 9 yaw_e
  __shared__ unsigned char flags[DSIM_IO_WIN];
  __shared__ int wmin[DSIM_IO_WG / 64];
  DSIM_RUN_OF_BLOCK(rt, ro, DSIM_IO_ENTRY());
  const unsigned t = DSIM_IO_T();
  IoCtl io;
  io.have = false;
  {
    const long long i = ro.i0 + t;
    io_window_min(i >= ro.lo && i < ro.last ? a.io_id[i] : 0x7fffffff, wmin, flags);
  }
  if (ro.quadlaw6) control_run_body<DSIM_DEV_KIND_HEXA_QUADLAW, NT, WANT_YAW, true>(a, ro, t, io);
  else if (ro.hexa) control_run_body<DSIM_DEV_KIND_HEXA, NT, WANT_YAW, true>(a, ro, t, io);
  else control_run_body<DSIM_DEV_KIND_QUAD, NT, WANT_YAW, true>(a, ro, t, io);
  const int base = io_window_base(wmin);
  const int nc = a.obs_w - 16;                // rows of the command table: 4, or 6 with a six-actuator type in the fleet
  if (io.have) {
    const unsigned slot = (unsigned)(io.id - base);
    if (slot < (unsigned)DSIM_IO_WIN) {
#pragma unroll
      for (int f = 0; f < 10; ++f) win[f * DSIM_IO_WIN + slot] = io.v[f];
      flags[slot] = 1;
    } else {                                  // outside the window: straight out (default cache policy, see above)
      if (a.cmd_out) {
#pragma unroll
        for (int j = 0; j < 6; ++j) if (j < nc) a.cmd_out[(long long)j * a.n_pad + io.id] = io.v[j];
      }
      if (a.pos_e_out) {
#pragma unroll
        for (int j = 0; j < 3; ++j) a.pos_e_out[(long long)j * a.n_pad + io.id] = io.v[6 + j];
      }
      if (WANT_YAW) a.yaw_e_out[io.id] = io.v[9];
    }
  }
  __syncthreads();
  // window row threadIdx.x: whole lines of every output array when all 64 rows of the wave were produced here (then with
  // the streaming hint; a wave with holes leaves them to their owners and keeps the default policy, so that the parts of a
  // line still meet in the cache)
  const bool mine = flags[threadIdx.x] != 0;
  const bool whole = __ballot(mine) == ~0ULL;
  if (mine) {
    const unsigned lo4 = 4u * threadIdx.x;
#define DSIM_IO_OUT(PTR, F) do { float* ub_ = (PTR) + base; const float v_ = win[(F) * DSIM_IO_WIN + threadIdx.x];          \
                                 if (whole) stg<NT>(ub_, lo4, v_); else stg<false>(ub_, lo4, v_); } while (0)
    if (a.cmd_out) {
#pragma unroll
      for (int j = 0; j < 6; ++j) if (j < nc) DSIM_IO_OUT(a.cmd_out + (long long)j * a.n_pad, j);
    }
    if (a.pos_e_out) {
#pragma unroll
      for (int j = 0; j < 3; ++j) DSIM_IO_OUT(a.pos_e_out + (long long)j * a.n_pad, 6 + j);
    }
    if (WANT_YAW) DSIM_IO_OUT(a.yaw_e_out, 9);
#undef DSIM_IO_OUT
  }
}

// ---- Env.step of the alternate action adaptors (VelocityAviary / RPYTAviary) --------------
// control (inside _preprocessAction) on the CURRENT state, then the physics with the new command
template <int MODE, bool NOISE, bool PLANE, class DT>
__device__ __forceinline__ void adaptor_body(DT& T, const StepK& a, long long i, const Addr& ad) {
  Rigid s;
  CtrlMem<4> m;
  load_rigid(ad.sb, ad.sfs, ad.sl, s);
  load_mem<4>(ad.sb, ad.sfs, ad.sl, m);
  float v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = a.action[(long long)j * a.n_pad + i];
  if (MODE == DSIM_ADAPT_VELOCITY) {                       // VelocityAviary.py:241-262
    const float nrm = DSIM_SQRT(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    const float sc = nrm != 0.0f ? T.speed_limit * fabsf(v[3]) * DSIM_RCP(nrm) : 0.0f;
    Target tg;
    tg.pos = s.pos;                                        // "same as the current position"
    tg.vel = v3(sc * v[0], sc * v[1], sc * v[2]);
    tg.acc = v3(0, 0, 0);
    tg.yaw = euler_from_quat<true>(s.q).yaw;               // "keep current yaw" (state[9])
    V3 pos_e;
    float yaw_e;
    indi_quad<false>(T, a.dt_ctrl, s, tg, m, pos_e, yaw_e);
  } else {                                                 // RPYTAviary.py:184-191
    indi_rate<4>(T, DSIM_RCP(a.dt_ctrl), s, v3(v[0], v[1], v[2]), v[3], m);
  }
  quad_substeps<NOISE ? 1 : 0, 4, PLANE, 0, PLANE, 1>(T, a, i, s, m.cmd, a.step_index, V3{-0.0f, -0.0f, -0.0f}, nullptr,
                                                   NOISE ? noise_id(a, i) : -1LL);
  ground_watch(T, s, a.fb.counters, i < a.n);
  store_rigid(ad.sb, ad.sfs, ad.sl, s);
  store_mem<4>(ad.sb, ad.sfs, ad.sl, m);
  if (a.echo) {
#pragma unroll
    for (int j = 0; j < 4; ++j) a.echo[(long long)j * a.n_pad + i] = m.cmd[j];
  }
}
template <int MODE, bool NOISE, bool UNIFORM, bool PLANE = false>
__global__ __launch_bounds__(256, PLANE ? 1 : DSIM_GEN_WAVES) void k_adaptor(StepK a) {
  const long long i0 = (long long)blockIdx.x * 256;
  const unsigned p = tile_slot<UNIFORM>(a.type_id, i0, a.n_pad);
  const long long i = i0 + p;
  if (i >= a.n_pad) return;
  const Addr ad = make_addr(a, i0, p);
  DSIM_FOR_MY_TYPE(UNIFORM, a, i, (adaptor_body<MODE, NOISE, PLANE>(T, a, i, ad)));
}

// The same on a homogeneous quad fleet in whole tiles, as ONE launch that also returns Env.step's observation: the fused
// kernels' addressing (scalar base + one lane offset, streaming accesses, constants in SGPRs), the action taken as the
// caller holds it — StepK.action_rows: row-major [n][4] (VelocityAviary.py:221-264 / RPYTAviary.py:181-193 take one 4-vector per drone),
// one 16-byte load per lane — and the 20-wide rows of the NEW state written by the same launch (OBS).  Before: a transpose of
// the action (torch, 50 us), k_adaptor (147-160 us) and k_observe (125 us) per Env.step of 4 194 304 drones.
//   reads 24 state + 4 action floats, writes 24 state + 4 echoed command + 20 row floats: 304 bytes per drone-step
template <int MODE, bool NOISE, bool NT>
__global__ __launch_bounds__(256, DSIM_STEP_WAVES) void k_adaptor_fast(StepK a) {
  __shared__ __attribute__((aligned(16))) vf4 rows[4 * 64 * 5];
  const DevType& T = a.types[0];
  const long long sfs = a.st.field_stride;
  const unsigned sl = 4u * kv_lane(a.st, threadIdx.x);
  const long long i0 = (long long)blockIdx.x * 256;
  float* const sb = a.st.base + kv_off(a.st, i0);
  const long long i = i0 + threadIdx.x;
  Rigid s;
  CtrlMem<4> m;
  load_rigid<NT>(sb, sfs, sl, s);
  load_mem<4, NT>(sb, sfs, sl, m);
  float v[4];
  if (a.action_rows) {                                                   // (wave-uniform)
    vf4 r = vf4{0.0f, 0.0f, 0.0f, 0.0f};
    if (i < a.n) {                                                       // (rows exist for real drones only)
      const vf4* ar = reinterpret_cast<const vf4*>(a.action) + i0;
      r = NT ? __builtin_nontemporal_load(ar + threadIdx.x) : ar[threadIdx.x];
    }
    v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w;
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = ldg<NT>(a.action + (long long)j * a.n_pad + i0, 4u * threadIdx.x);
  }
  if (MODE == DSIM_ADAPT_VELOCITY) {                       // VelocityAviary.py:241-262
    const float nrm = DSIM_SQRT(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    const float sc = nrm != 0.0f ? T.speed_limit * fabsf(v[3]) * DSIM_RCP(nrm) : 0.0f;
    Target tg;
    tg.pos = s.pos;                                        // "same as the current position"
    tg.vel = v3(sc * v[0], sc * v[1], sc * v[2]);
    tg.acc = v3(0, 0, 0);
    tg.yaw = euler_from_quat<true>(s.q).yaw;               // "keep current yaw" (state[9])
    V3 pos_e;
    float yaw_e;
    indi_quad<false>(T, a.dt_ctrl, s, tg, m, pos_e, yaw_e);
  } else {                                                 // RPYTAviary.py:184-191
    indi_rate<4>(T, DSIM_RCP(a.dt_ctrl), s, v3(v[0], v[1], v[2]), v[3], m);
  }
  if (NOISE && a.step_index_dev) a.step_index += *a.step_index_dev;
  quad_substeps<NOISE ? 1 : 0>(T, a, i, s, m.cmd, a.step_index);
  ground_watch(T, s, a.fb.counters, i < a.n);
  const unsigned so = pin_lane_offset(sl);
  store_rigid<NT>(sb, sfs, so, s);
  store_mem<4, NT>(sb, sfs, so, m);
  if (a.echo) {
#pragma unroll
    for (int j = 0; j < 4; ++j) stg<NT>(a.echo + (long long)j * a.n_pad + i0, 4u * threadIdx.x, m.cmd[j]);
  }
  if (a.obs_out) obs_rows20_out<NT>(rows, a, i0, s, m.cmd);
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
// Physics.DYN (DSIM_OPT_DYN): what the mode does not combine with is refused, not dropped
int dyn_check(const dsim_ctx* ctx, const dsim_step_args* args, const StepK& a) {
  if (!args->dyn_rpy_rates) return DSIM_E_ARG;
  if (ctx->max_act != 4) return DSIM_E_UNSUPPORTED;             // both mixers of BaseAviary.py:1794-1803 read forces[0..3]
  if (args->options & (DSIM_OPT_DRAG | DSIM_OPT_GROUND | DSIM_OPT_PLANE | DSIM_OPT_CHAINED | DSIM_OPT_CALLER_IO | DSIM_OPT_ACTION_ROWS))
    return DSIM_E_UNSUPPORTED;
  if (args->ext_force || args->wp_table || a.n_steps > 1 || args->bin_next) return DSIM_E_UNSUPPORTED;   // (step_index_dev only moves the noise counter: no noise here)
  return DSIM_OK;
}

int dyn_launch(bool ctrl, const StepK& a, bool nt, hipStream_t st) {
  const dim3 g(grid_for(a.n_pad)), b(256);
  if (ctrl) { if (nt) hipLaunchKernelGGL((k_dyn<true, true>), g, b, 0, st, a); else hipLaunchKernelGGL((k_dyn<true, false>), g, b, 0, st, a); }
  else if (a.obs_out) { if (nt) hipLaunchKernelGGL((k_dyn<false, true, true>), g, b, 0, st, a); else hipLaunchKernelGGL((k_dyn<false, false, true>), g, b, 0, st, a); }
  else { if (nt) hipLaunchKernelGGL((k_dyn<false, true>), g, b, 0, st, a); else hipLaunchKernelGGL((k_dyn<false, false>), g, b, 0, st, a); }
  return (int)hipGetLastError();
}

extern "C" {

int dsim_physics(dsim_ctx* ctx, void* stream, int64_t n, dsim_view state, float* last_action_out,
                 const dsim_step_args* args) {
  StepK a;
  int rc = fill_stepk(ctx, n, state, nullptr, args, &a);
  if (rc) return rc;
  ctx->dw_prebin_valid = false;
  a.echo = last_action_out;
  if (args->options & DSIM_OPT_DYN) {
    rc = dyn_check(ctx, args, a);
    if (rc) return rc;
    if (args->obs_out && args->obs_width != 20) return DSIM_E_ARG;
    const bool obs_fused = args->obs_out && ((uintptr_t)args->obs_out & 15u) == 0;       // (16-byte stores of the rows)
    a.obs_out = obs_fused ? args->obs_out : nullptr;
    rc = dyn_launch(false, a, stream_policy(args, state.n_pad, args->obs_out ? 240.0 : 160.0), (hipStream_t)stream);
    if (rc) return rc;
    if (args->obs_out && !obs_fused) return observe_impl(ctx, stream, n, state, last_action_out, args->obs_out, 20, 0);
    return DSIM_OK;
  }
  const bool fine_slow = (args->noise_seed != 0 && !args->noise_replay && (a.options & DSIM_OPT_NOISE_FINE) && a.substeps > 1);
  const bool phys_opts = (args->options & (DSIM_OPT_DRAG | DSIM_OPT_GROUND | DSIM_OPT_PLANE)) != 0;
  if ((args->options & (DSIM_OPT_DRAG | DSIM_OPT_GROUND)) && ctx->max_act == 6) return DSIM_E_UNSUPPORTED;
  const int obs_w = 16 + ctx->max_act;
  if (args->obs_out && args->obs_width != obs_w) return DSIM_E_ARG;
  const bool noise = args->noise_seed != 0 || args->noise_replay != nullptr;
  const hipStream_t st_ = (hipStream_t)stream;
  if ((args->options & DSIM_OPT_CALLER_IO) && !args->drone_id) return DSIM_E_ARG;
  const bool arows = (args->options & DSIM_OPT_ACTION_ROWS) != 0;
  if (arows && (!args->action || ((uintptr_t)args->action & 15u))) return DSIM_E_ARG;
  if (args->type_id == nullptr && ctx->max_act == 4 && !args->noise_replay && !args->ext_force && !phys_opts &&
      (a.n_pad % 256) == 0 && !args->bin_next && !args->drone_id && !(args->options & DSIM_OPT_CALLER_IO)) {
    // homogeneous quad fleet in whole tiles: the fast form, observation fused (16-byte stores: any torch allocation is
    // aligned far beyond that; a misaligned caller buffer gets the rows from the observation kernel behind the step)
    const bool obs_fused = args->obs_out && ((uintptr_t)args->obs_out & 15u) == 0;
    a.obs_out = obs_fused ? args->obs_out : nullptr;
    const bool nt = stream_policy(args, state.n_pad, args->obs_out ? 216.0 : 136.0);
    const dim3 g((unsigned)(a.n_pad / 256)), b(256);
    const bool loop = a.substeps > 1 && !(noise && (a.options & DSIM_OPT_NOISE_FINE));      // (the looped instance: coarse lattice only)
#define DSIM_PHYS_CASE(N_, T_) do {                                                                                     \
      if (loop) { if (a.obs_out) hipLaunchKernelGGL((k_physics_fast<N_, T_, true, true>), g, b, 0, st_, a);             \
                  else hipLaunchKernelGGL((k_physics_fast<N_, T_, false, true>), g, b, 0, st_, a); }                    \
      else { if (a.obs_out) hipLaunchKernelGGL((k_physics_fast<N_, T_, true>), g, b, 0, st_, a);                        \
             else hipLaunchKernelGGL((k_physics_fast<N_, T_, false>), g, b, 0, st_, a); } } while (0)
    if (noise) { if (nt) DSIM_PHYS_CASE(true, true); else DSIM_PHYS_CASE(true, false); }
    else { if (nt) DSIM_PHYS_CASE(false, true); else DSIM_PHYS_CASE(false, false); }
#undef DSIM_PHYS_CASE
    if (args->obs_out && !obs_fused) return observe_impl(ctx, stream, n, state, last_action_out, args->obs_out, obs_w, 0);
    return (int)hipGetLastError();
  }
  if (arows) return DSIM_E_UNSUPPORTED;               // (every other kernel takes the action field-major)
  // Every other fleet kind on the fast form: runs of one type (dsim_step_args.runs), or a homogeneous fleet as ONE run —
  // morphing hexas, type-major quad + hexa fleets, fleets with the downwash force, ragged tails.  The observation rows are
  // written by the same launch; the new positions may fill the next neighbour grid (bin_next).
  {
    const dsim_type_run* runs = args->runs;
    int n_runs = args->n_runs;
    dsim_type_run whole;
    if (!(runs && n_runs > 0 && n_runs <= DSIM_MAX_TYPES) && args->type_id == nullptr) {
      whole.first = 0; whole.count = a.n_pad; whole.type = 0; whole._pad = 0;
      runs = &whole; n_runs = 1;
    }
    if (runs && n_runs > 0 && n_runs <= DSIM_MAX_TYPES && !args->noise_replay && !phys_opts && !fine_slow) {   // (k_physics_fast above carries
      RunTab rt;                                                                                              //  both lattices at any count)
      bool any_hexa = false;
      const int blocks = make_runtab(ctx, a.n_pad, runs, n_runs, &rt, &any_hexa);
      if (blocks < 0) return blocks;
      if (a.io_id) { rc = side_by_side_map(ctx, st_, runs, n_runs, &rt); if (rc) return rc; }
      if (a.io_id && args->obs_out && ((uintptr_t)args->obs_out & 15u)) return DSIM_E_ARG;   // (the window's 16-byte pieces)
      const bool obs_fused = args->obs_out && ((uintptr_t)args->obs_out & 7u) == 0;     // (8-byte pieces of the rows)
      a.obs_out = obs_fused ? args->obs_out : nullptr;
      const bool nt = stream_policy(args, state.n_pad, args->obs_out ? 240.0 : 152.0);
      bin_next_prepare(ctx, n, args, &a, st_);
      if (blocks > 0) {
        const dim3 g((unsigned)blocks), b(256);
#define DSIM_PRUNS_CASE2(N_, T_, S_) do {                                                                                      \
          if (a.io_id) { const dim3 g2((unsigned)((blocks + 1) / 2) * DSIM_IO_PARTS), b2(DSIM_IO_WG);                                                 \
                         if (a.obs_out) hipLaunchKernelGGL((k_physics_runs_io<N_, T_, true, S_>), g2, b2, 0, st_, a, rt);      \
                         else hipLaunchKernelGGL((k_physics_runs_io<N_, T_, false, S_>), g2, b2, 0, st_, a, rt); }              \
          else { if (a.obs_out) hipLaunchKernelGGL((k_physics_runs<N_, T_, true, S_>), g, b, 0, st_, a, rt);                  \
                 else hipLaunchKernelGGL((k_physics_runs<N_, T_, false, S_>), g, b, 0, st_, a, rt); } } while (0)
#define DSIM_PRUNS_CASE(N_, T_) do { if (a.substeps == 1) DSIM_PRUNS_CASE2(N_, T_, true); else DSIM_PRUNS_CASE2(N_, T_, false); } while (0)
        if (noise) { if (nt) DSIM_PRUNS_CASE(true, true); else DSIM_PRUNS_CASE(true, false); }
        else { if (nt) DSIM_PRUNS_CASE(false, true); else DSIM_PRUNS_CASE(false, false); }
#undef DSIM_PRUNS_CASE
#undef DSIM_PRUNS_CASE2
      }
      bin_next_commit(ctx, n, args, a);
      if (args->obs_out && !obs_fused) return observe_impl(ctx, stream, n, state, last_action_out, args->obs_out, obs_w, 0);
      return (int)hipGetLastError();
    }
  }
  if (a.io_id) return DSIM_E_UNSUPPORTED;            // the caller's numbering is served by the run kernels only
  const dim3 g(grid_for(a.n_pad));
  if (args->options & DSIM_OPT_PLANE) DSIM_LAUNCH_GEN_ANY(k_physics_plane, noise, ctx->max_act == 6, g, a, st_);
  else DSIM_LAUNCH_GEN_ANY(k_physics_gen, noise, ctx->max_act == 6, g, a, st_);
  if (args->obs_out)       // general fleets: the same rows by the observation kernel, behind the step on the stream
    return observe_impl(ctx, stream, n, state, last_action_out, args->obs_out, obs_w, 0);
  return (int)hipGetLastError();
}

int dsim_step_adaptor(dsim_ctx* ctx, void* stream, int64_t n, dsim_view state, const float* action,
                      int32_t mode, float* last_action_out, const dsim_step_args* args) {
  StepK a;
  if (!action || (mode != DSIM_ADAPT_VELOCITY && mode != DSIM_ADAPT_RPYT)) return DSIM_E_ARG;
  if (ctx && ctx->max_act == 6) return DSIM_E_UNSUPPORTED;
  if (args && (args->noise_replay || args->wp_table || args->ext_force ||
               (args->options & (DSIM_OPT_DRAG | DSIM_OPT_GROUND))))
    return DSIM_E_UNSUPPORTED;        // plain PYB physics (+ the plane): refuse what the adaptor kernels would silently drop
  int rc = fill_stepk(ctx, n, state, nullptr, args, &a);
  if (rc) return rc;
  if (args->options & (DSIM_OPT_CALLER_IO | DSIM_OPT_DYN)) return DSIM_E_UNSUPPORTED;   // (the adaptor envs fly Physics.PYB)
  ctx->dw_prebin_valid = false;
  a.action = action; a.echo = last_action_out;
  const bool noise = args->noise_seed != 0, uni = args->type_id == nullptr;
  const dim3 g(grid_for(a.n_pad)), b(256);
  const hipStream_t st_ = (hipStream_t)stream;
  const bool arows = (args->options & DSIM_OPT_ACTION_ROWS) != 0;
  if (args->obs_out && args->obs_width != 20) return DSIM_E_ARG;
  // (k_adaptor_fast carries both noise lattices: its sub-step loop is the non-looped form, dsim_kernels.h:quad_substeps)
  if (uni && (a.n_pad % 256) == 0 && !(args->options & DSIM_OPT_PLANE) && !args->drone_id &&
      (!arows || ((uintptr_t)action & 15u) == 0)) {
    // homogeneous quad fleet in whole tiles: ONE launch, the observation rows fused (16-byte stores; a misaligned caller
    // buffer gets them from the observation kernel behind the step), the action in either layout
    const bool obs_fused = args->obs_out && ((uintptr_t)args->obs_out & 15u) == 0;
    a.obs_out = obs_fused ? args->obs_out : nullptr;
    const bool nt = stream_policy(args, state.n_pad, args->obs_out ? 304.0 : 224.0);
    const dim3 gf((unsigned)(a.n_pad / 256));
#define DSIM_AF2(M_, N_) do { if (nt) hipLaunchKernelGGL((k_adaptor_fast<M_, N_, true>), gf, b, 0, st_, a);                   \
                              else hipLaunchKernelGGL((k_adaptor_fast<M_, N_, false>), gf, b, 0, st_, a); } while (0)
#define DSIM_AF1(M_) do { if (noise) DSIM_AF2(M_, true); else DSIM_AF2(M_, false); } while (0)
    if (mode == DSIM_ADAPT_VELOCITY) DSIM_AF1(DSIM_ADAPT_VELOCITY); else DSIM_AF1(DSIM_ADAPT_RPYT);
#undef DSIM_AF1
#undef DSIM_AF2
    if (args->obs_out && !obs_fused) return observe_impl(ctx, stream, n, state, last_action_out, args->obs_out, 20, 0);
    return (int)hipGetLastError();
  }
  if (arows) return DSIM_E_UNSUPPORTED;               // (the general kernels take the action field-major)
#define DSIM_ADAPT_CASE2(M_, P_)                                                                       \
  do { if (noise) hipLaunchKernelGGL((k_adaptor<M_, true, false, P_>), g, b, 0, st_, a);                \
       else hipLaunchKernelGGL((k_adaptor<M_, false, false, P_>), g, b, 0, st_, a); } while (0)       /* (one instance for homogeneous and mixed fleets: DSIM_LAUNCH_GEN_ANY) */
#define DSIM_ADAPT_CASE(M_) do { if (args->options & DSIM_OPT_PLANE) DSIM_ADAPT_CASE2(M_, true); else DSIM_ADAPT_CASE2(M_, false); } while (0)
  if (mode == DSIM_ADAPT_VELOCITY) DSIM_ADAPT_CASE(DSIM_ADAPT_VELOCITY); else DSIM_ADAPT_CASE(DSIM_ADAPT_RPYT);
#undef DSIM_ADAPT_CASE
#undef DSIM_ADAPT_CASE2
  if (args->obs_out)       // general fleets: the rows by the observation kernel, behind the step on the stream
    return observe_impl(ctx, stream, n, state, last_action_out, args->obs_out, 20, 0);
  return (int)hipGetLastError();
}

int dsim_control(dsim_ctx* ctx, void* stream, int64_t n, dsim_view state, dsim_view targets,
                 const dsim_step_args* args, float* pos_e_out, float* yaw_e_out) {
  return dsim_control2(ctx, stream, n, state, targets, args, pos_e_out, yaw_e_out, nullptr);
}

int dsim_control2(dsim_ctx* ctx, void* stream, int64_t n, dsim_view state, dsim_view targets,
                  const dsim_step_args* args, float* pos_e_out, float* yaw_e_out, float* cmd_out) {
  StepK a;
  if (args && args->wp_table) return DSIM_E_UNSUPPORTED;   // computeControl takes explicit targets
  int rc = fill_stepk(ctx, n, state, &targets, args, &a);
  if (rc) return rc;
  a.pos_e_out = pos_e_out; a.yaw_e_out = yaw_e_out; a.cmd_out = cmd_out;
  const dim3 g(grid_for(a.n_pad)), b(256);
  const hipStream_t st_ = (hipStream_t)stream;
  const bool uni = args->type_id == nullptr;
  if ((args->options & DSIM_OPT_CALLER_IO) && !args->drone_id) return DSIM_E_ARG;
  if (uni && ctx->max_act == 4 && (a.n_pad % 256) == 0 && a.tg.base && !a.io_id) {
    const bool nt = stream_policy(args, state.n_pad, 212.0);
    const dim3 gt((unsigned)(a.n_pad / 256));
    if (yaw_e_out) { if (nt) hipLaunchKernelGGL((k_control_fast<true, true>), gt, b, 0, st_, a);
                     else hipLaunchKernelGGL((k_control_fast<false, true>), gt, b, 0, st_, a); }
    else { if (nt) hipLaunchKernelGGL((k_control_fast<true, false>), gt, b, 0, st_, a);
           else hipLaunchKernelGGL((k_control_fast<false, false>), gt, b, 0, st_, a); }
    return (int)hipGetLastError();
  }
  {
    // every other fleet kind: runs of one type (or a homogeneous fleet as one run) on the single-type bodies (k_control_runs)
    const dsim_type_run* runs = args->runs;
    int n_runs = args->n_runs;
    dsim_type_run whole;
    if (!(runs && n_runs > 0 && n_runs <= DSIM_MAX_TYPES) && uni) {
      whole.first = 0; whole.count = a.n_pad; whole.type = 0; whole._pad = 0;
      runs = &whole; n_runs = 1;
    }
    if (runs && n_runs > 0 && n_runs <= DSIM_MAX_TYPES) {
      RunTab rt;
      bool any_hexa = false;
      const int blocks = make_runtab(ctx, a.n_pad, runs, n_runs, &rt, &any_hexa);
      if (blocks < 0) return blocks;
      if (any_hexa) {
        rc = fb_prepare(ctx, a.n_pad, st_);
        if (rc) return rc;
        a.fb.entries = ctx->d_fb;
      }
      if (a.io_id) { rc = side_by_side_map(ctx, st_, runs, n_runs, &rt); if (rc) return rc; }
      const bool nt = stream_policy(args, state.n_pad, 236.0);
      if (blocks > 0) {
        const dim3 gr((unsigned)blocks);
#define DSIM_CRUNS_CASE(T_, Y_) do {                                                                                         \
          if (a.io_id) hipLaunchKernelGGL((k_control_runs_io<T_, Y_>), dim3((unsigned)((blocks + 1) / 2) * DSIM_IO_PARTS), dim3(DSIM_IO_WG), 0, st_, a, rt); \
          else hipLaunchKernelGGL((k_control_runs<T_, Y_>), gr, b, 0, st_, a, rt); } while (0)
        if (yaw_e_out) { if (nt) DSIM_CRUNS_CASE(true, true); else DSIM_CRUNS_CASE(false, true); }
        else { if (nt) DSIM_CRUNS_CASE(true, false); else DSIM_CRUNS_CASE(false, false); }
#undef DSIM_CRUNS_CASE
      }
      if (any_hexa) fb_finish(ctx, a, st_);
      return (int)hipGetLastError();
    }
  }
  // what is left: per-lane types without usable runs (the caller's own order of a heterogeneous fleet)
  if (a.io_id) return DSIM_E_UNSUPPORTED;
  if (ctx->max_act == 6) {
    rc = fb_prepare(ctx, a.n_pad, st_);
    if (rc) return rc;
    a.fb.entries = ctx->d_fb;
    hipLaunchKernelGGL((k_control_gen<6>), g, b, 0, st_, a);
    fb_finish(ctx, a, st_);
  }
  else hipLaunchKernelGGL((k_control_gen<4>), g, b, 0, st_, a);
  return (int)hipGetLastError();
}

}  // extern "C"
