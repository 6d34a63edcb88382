// dsim_api.hip — context, reset, observation rows, trajectory sampler, deferred WLS fallbacks, noise draw, and the host-side
// helpers every entry point shares (include/dronesim_amd.h; gfx950 only).
#include "dsim_kernels.h"

__global__ void k_counter_add(unsigned long long* c, unsigned long long inc) { *c += inc; }

// dsim_noise_draw: the unit-variance normals of (drone, sub-step), through the very functions the step kernels call
struct NoiseK { long long n, n_pad; int n_act, substeps; unsigned long long seed, step_index; unsigned options; const int* drone_id; float* out; };
__global__ __launch_bounds__(256) void k_noise_draw(NoiseK a) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  const uint64_t key = (uint64_t)(a.drone_id ? (long long)a.drone_id[i] : i);
  for (int k = 0; k < a.substeps; ++k) {
    const uint64_t sub = a.step_index * (uint64_t)a.substeps + (uint64_t)k;
    float nz[12];
    const bool fine = (a.options & DSIM_OPT_NOISE_FINE) != 0;
    if (a.n_act == 4) { if (fine) quad_normals_fine(a.seed, key, sub, nz); else noise_normals<4>(a.seed, key, sub, nz); }
    else {                                 // six-actuator kinds: rows 0 .. 5 the six normals of the body wrench, rows 6 .. 11 zero
      if (fine) { hexa_z_fine(a.seed, key, sub, nz); for (int j = 6; j < 12; ++j) nz[j] = 0.0f; }
      else noise_normals<6>(a.seed, key, sub, nz);
    }
    // (the functions return the normals already scaled by their deviations: .01 on the force rows, .001 on the moment rows)
    for (int j = 0; j < 2 * a.n_act; ++j)
      a.out[((long long)k * 2 * a.n_act + j) * a.n_pad + i] = nz[j] * (j < a.n_act ? 100.0f : 1000.0f);
  }
}

// ---- deferred WLS fallbacks (hexa) -----------------------------------------------
// Launched behind every step of a fleet that holds a morphing hexa.  The queue is normally EMPTY: every workgroup
// then leaves after one scalar load (no fence, no ticket — 1.5-2 us of kernel boundary instead of the 4.7 us the
// unconditional fence-and-ticket epilogue of round 1 cost).  Otherwise the grid (sized for the chip by the host:
// up to one 64-lane workgroup per CU, 144 KB of LDS each) strides over the queue, and the last workgroup to finish
// empties it for the next step.
struct FbK { KView st; const DevType* types; const uint8_t* type_id; FbList fb; float* cmd_out; long long n_pad; const int* io_id; };
#define DSIM_FB_LANES 64
__global__ __launch_bounds__(DSIM_FB_LANES) void k_wls_fallback(FbK a) {
  const unsigned long long cnt = *a.fb.count;
  if (cnt == 0) return;                                   // wave-uniform: the queue length is final (previous kernel)
  __shared__ WlsWork work[DSIM_FB_LANES];
  for (unsigned long long e = (unsigned long long)blockIdx.x * DSIM_FB_LANES + threadIdx.x; e < cnt;
       e += (unsigned long long)gridDim.x * DSIM_FB_LANES) {
    const FbEntry en = a.fb.entries[e];
    const long long i = en.drone;
    const DevType& T = a.types[a.type_id ? a.type_id[i] : 0];
    float* p = a.st.base + kv_off(a.st, i);
    const long long fs = a.st.field_stride;
    float cmd[6], umin[6], umax[6], du[6];
    for (int j = 0; j < 6; ++j) { cmd[j] = p[(20 + j) * fs]; umin[j] = T.pmin[j] - cmd[j]; umax[j] = T.pmax[j] - cmd[j]; }
    const int rc = wls_active_set(T, en.v, umin, umax, du, work[threadIdx.x]);
    atomicAdd(&a.fb.counters[0], 1ULL);
    if (rc == 0) {
      for (int j = 0; j < 6; ++j) {
        const float c = clampf(cmd[j] + du[j], T.pmin[j], T.pmax[j]);
        p[(20 + j) * fs] = c;
        if (a.cmd_out) a.cmd_out[(long long)j * a.n_pad + (a.io_id ? (long long)a.io_id[i] : i)] = c;   // computeControl's first return value (dsim_control2)
      }
    }
    else atomicAdd(&a.fb.counters[1], 1ULL);   // the reference would raise here; cmd is left unchanged
  }
  // the last workgroup to finish empties the queue for the next step (no per-step memset on the stream)
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(&a.fb.counters[3], 1ULL) == (unsigned long long)gridDim.x - 1) {
      *a.fb.count = 0ULL;
      a.fb.counters[3] = 0ULL;
      __threadfence();
    }
  }
}

// ---- reset -------------------------------------------------------------------
struct ResetK {
  KView st;
  const DevType* types;
  const uint8_t* type_id;
  const float *pos, *rpy, *vel, *cmd;
  long long n_pad;
  int n_fields;
};
__global__ __launch_bounds__(256) void k_reset(ResetK a) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n_pad) return;
  const DevType& T = a.types[a.type_id ? a.type_id[i] : 0];
  const long long o = kv_off(a.st, i);
  Rigid s;
  s.pos = v3(a.pos[i], a.pos[a.n_pad + i], a.pos[2 * a.n_pad + i]);
  s.q = quat_from_euler(v3(a.rpy[i], a.rpy[a.n_pad + i], a.rpy[2 * a.n_pad + i]));   // BaseAviary.py:687
  s.vel = a.vel ? v3(a.vel[i], a.vel[a.n_pad + i], a.vel[2 * a.n_pad + i]) : v3(0, 0, 0);  // :695-705
  s.w = v3(0, 0, 0);
  store_rigid(a.st.base + o, a.st.field_stride, 0u, s);
  float* p = a.st.base + o;
  const long long fs = a.st.field_stride;
#pragma unroll
  for (int f = 13; f < 19; ++f) p[f * fs] = 0.0f;            // INDIControl.reset, INDIControl.py:125-130
  p[19 * fs] = T.reset_thrust;
  const int nact = a.n_fields - 20;
  for (int j = 0; j < nact; ++j)
    p[(20 + j) * fs] = a.cmd ? a.cmd[(long long)j * a.n_pad + i] : (j < T.n_act ? T.reset_cmd : 0.0f);
}

// ---- observation rows (BaseAviary.py:780-790) --------------------------------
struct ObsK { KView st; const float* last_action; float* out; long long n, n_pad; int width; int soa; };
__global__ __launch_bounds__(256) void k_observe(ObsK a) {
  // Row-major output [n][width] (the reference's per-drone vectors, BaseAviary.py:780-790): a lane that wrote its
  // own row would scatter 4-byte stores at a stride of `width` floats (measured 429 us for 4.2 M drones, 1.4 TB/s),
  // so the tile's 256 x width block — contiguous in the output — is transposed through LDS and written linearly.
  __shared__ float rows[256 * 23];      // rows padded to width + 1 floats
  const long long i0 = (long long)blockIdx.x * 256;
  const long long i = i0 + threadIdx.x;
  const int W = a.width;
  float v[22];
  if (i < a.n) {
    const long long o = kv_off(a.st, i);
    Rigid s;
    load_rigid(a.st.base + o, a.st.field_stride, 0u, s);
    const Euler e = euler_from_quat<true>(s.q);
    v[0] = s.pos.x; v[1] = s.pos.y; v[2] = s.pos.z;
    v[3] = s.q.x; v[4] = s.q.y; v[5] = s.q.z; v[6] = s.q.w;
    v[7] = e.roll; v[8] = e.pitch; v[9] = e.yaw;
    v[10] = s.vel.x; v[11] = s.vel.y; v[12] = s.vel.z;
    v[13] = s.w.x; v[14] = s.w.y; v[15] = s.w.z;
#pragma unroll
    for (int j = 0; j < 6; ++j)
      if (j < W - 16)
        v[16 + j] = a.last_action ? a.last_action[(long long)j * a.n_pad + i] : a.st.base[o + (20 + j) * a.st.field_stride];
    if (a.soa) {                     // field-major [width][n_pad] (log slabs): already coalesced
#pragma unroll
      for (int f = 0; f < 22; ++f) if (f < W) a.out[(long long)f * a.n_pad + i] = v[f];
    } else {
      // W is 20 or 22: odd multiples of 2 -> consecutive lanes hit banks 2 apart... pad rows to W + 1 floats
#pragma unroll
      for (int f = 0; f < 22; ++f) if (f < W) rows[threadIdx.x * (W + 1) + f] = v[f];
    }
  }
  if (a.soa) return;
  __syncthreads();
  const long long tile_rows = min((long long)256, a.n - i0);
  const int total = (int)tile_rows * W;
  float* dst = a.out + i0 * W;
  for (int k = threadIdx.x; k < total; k += 256) {
    const int r = k / W, f = k - r * W;
    dst[k] = rows[r * (W + 1) + f];
  }
}

// ---- trajectory sampler (trajGen.get_des_state + get_yaw) ------------------------------
struct TrajK {
  KView tg;
  const double* coeffs;   // [n_seg*10][3]
  const double* ts;       // [n_seg+1]
  double* t;              // [n_pad]
  double* yaw_state;      // SoA [3][n_pad]
  const float* offset;    // SoA [3][n_pad] or null
  long long n, n_pad;
  int n_seg;
  double dt_advance;
};
__global__ __launch_bounds__(256) void k_traj_sample(TrajK a) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  double t = a.t[i];
  const double t_end = a.ts[a.n_seg];
  if (t > t_end) t = t_end - 0.001;                                   // trajGen.py:110-111
  int seg = 0;
  for (int k = 0; k <= a.n_seg; ++k) if (t >= a.ts[k]) seg = k;       // :113
  if (seg >= a.n_seg) seg = a.n_seg - 1;
  t -= a.ts[seg];                                                      // :115
  double pw[10];
  pw[0] = 1.0;
#pragma unroll
  for (int j = 1; j < 10; ++j) pw[j] = pw[j - 1] * t;
  double out[9];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    double p = 0.0, v = 0.0, ac = 0.0;
#pragma unroll
    for (int j = 0; j < 10; ++j) {                                     // coeff @ polyder(t, k), :118-120
      const double c = a.coeffs[(seg * 10 + j) * 3 + d];
      p += c * pw[j];
      if (j >= 1) v += c * (double)j * pw[j - 1];
      if (j >= 2) ac += c * (double)(j * (j - 1)) * pw[j - 2];
    }
    out[d] = p; out[3 + d] = v; out[6 + d] = ac;
  }
  // get_yaw(vel[:2]), :128-143 — per-drone memory (yaw, heading)
  double yaw = a.yaw_state[i];
  const double hx = a.yaw_state[a.n_pad + i], hy = a.yaw_state[2 * a.n_pad + i];
  const double nv = sqrt(out[3] * out[3] + out[4] * out[4]);
  const double cx = out[3] / nv, cy = out[4] / nv;
  const double cosine = fmax(-1.0, fmin(hx * cx + hy * cy, 1.0));
  const double dyaw = acos(cosine);
  const double cr = hx * cy - hy * cx;
  yaw += (cr > 0.0 ? 1.0 : (cr < 0.0 ? -1.0 : 0.0)) * dyaw;
  if (yaw > 3.14159265358979323846) yaw -= 2.0 * 3.14159265358979323846;
  if (yaw < -3.14159265358979323846) yaw += 2.0 * 3.14159265358979323846;
  a.yaw_state[i] = yaw; a.yaw_state[a.n_pad + i] = cx; a.yaw_state[2 * a.n_pad + i] = cy;
  a.t[i] += a.dt_advance;
  float* q = a.tg.base + kv_off(a.tg, i);
  const long long fs = a.tg.field_stride;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    q[d * fs] = (float)out[d] + (a.offset ? a.offset[d * a.n_pad + i] : 0.0f);
    q[(3 + d) * fs] = (float)out[3 + d];
    q[(6 + d) * fs] = (float)out[6 + d];
  }
  q[9 * fs] = (float)yaw;
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
int make_kview(const dsim_view& v, int need_fields, KView* k, bool bcast) {
  if (!v.base) return DSIM_E_ARG;
  if (v.n_fields < need_fields) return DSIM_E_LAYOUT;
  k->base = v.base;
  k->field_stride = v.field_stride;
  k->block_stride = v.block_stride;
  if (bcast) { k->mask = 0; k->shift = 63; return DSIM_OK; }
  if (v.n_pad <= 0 || (v.n_pad & 63)) return DSIM_E_LAYOUT;
  if (v.block == v.n_pad) { k->mask = -1; k->shift = 63; return DSIM_OK; }
  if (v.block < 64 || (v.block & (v.block - 1)) || (v.n_pad % v.block)) return DSIM_E_LAYOUT;
  k->mask = v.block - 1;
  int sh = 0;
  while ((1LL << sh) < v.block) ++sh;
  k->shift = sh;
  if (v.field_stride < v.block || v.block_stride < v.field_stride * need_fields) return DSIM_E_LAYOUT;
  return DSIM_OK;
}

static void to_dev(const dsim_type_params& p, DevType* d) {
  memset(d, 0, sizeof(*d));
  d->kind = p.kind; d->n_act = p.n_act;
  d->mass = (float)p.mass; d->inv_mass = (float)(1.0 / p.mass);
  for (int k = 0; k < 3; ++k) {
    d->J[k] = (float)p.inertia[k]; d->invJ[k] = (float)(1.0 / p.inertia[k]);
    d->katt[k] = (float)p.att_gain[k]; d->krate[k] = (float)p.rate_gain[k];
    d->drag[k] = (float)p.drag_coeff[k]; d->dw[k] = (float)p.dw_coeff[k];
  }
  d->gyro[0] = (float)((p.inertia[2] - p.inertia[1]) / p.inertia[0]);
  d->gyro[1] = (float)((p.inertia[0] - p.inertia[2]) / p.inertia[1]);
  d->gyro[2] = (float)((p.inertia[1] - p.inertia[0]) / p.inertia[2]);
  d->kf = (float)p.kf; d->km = (float)p.km;
  for (int j = 0; j < DSIM_MAX_ACT; ++j) {
    d->scale[j] = (float)p.pwm2rpm_scale[j]; d->cnst[j] = (float)p.pwm2rpm_const[j];
    d->pmin[j] = (float)p.pwm_min[j]; d->pmax[j] = (float)p.pwm_max[j];
    d->spin[j] = (float)p.rotor_spin[j];
    for (int k = 0; k < 3; ++k) { d->rpos[j][k] = (float)p.rotor_pos[j][k]; d->raxis[j][k] = (float)p.rotor_axis[j][k]; }
    for (int k = 0; k < 3; ++k) d->spax[j][k] = d->spin[j] * d->raxis[j][k];       // (exact: the spins are +-1)
    const double* r = p.rotor_pos[j]; const double* ax = p.rotor_axis[j];
    d->rxa[j][0] = (float)(r[1] * ax[2] - r[2] * ax[1]);
    d->rxa[j][1] = (float)(r[2] * ax[0] - r[0] * ax[2]);
    d->rxa[j][2] = (float)(r[0] * ax[1] - r[1] * ax[0]);
    for (int k = 0; k < 3; ++k) { d->raxis64[j][k] = (double)d->raxis[j][k]; d->rxa64[j][k] = (double)d->rxa[j][k]; }
    if (j < 4) for (int k = 0; k < 3; ++k) d->rsum[k] += (float)r[k];
    for (int i = 0; i < DSIM_MAX_ACT; ++i) {
      d->alloc[j][i] = (float)p.alloc[j][i];
      d->alloc2[j][i] = (float)p.alloc2[j][i];
      d->B[j][i] = (float)(p.G1[j][i] / 0.05);          // INDIControl_6DOF.py:627: self.G1 / 0.05
    }
  }
  d->kp = (float)p.kp_pos; d->kd = (float)p.kd_pos;
  d->g = (float)p.gravity; d->clin = (float)p.lin_damping; d->cang = (float)p.ang_damping;
  d->maxv = (float)p.max_coord_vel;
  d->gnd_coeff = (float)p.gnd_eff_coeff; d->prop_radius = (float)p.prop_radius; d->gnd_hclip = (float)p.gnd_eff_h_clip;
  if (p.kind == DSIM_KIND_HEXA6DOF) { d->reset_thrust = 0.3f; d->reset_cmd = 0.5f; }   // INDIControl_6DOF.py:232-234
  d->speed_limit = (float)(p.max_speed_kmh * (1000.0 / 3600.0));
  d->coll_r = (float)p.collision_radius; d->coll_below = (float)p.collision_below;
  d->mu_plane = (float)p.contact_friction;
  for (int k = 0; k < 3; ++k) d->base_off[k] = (float)p.base_offset[k];
  d->watch_below = (float)(p.collision_below + p.base_offset[2]);    // (the offset of the shipped hexa is along body z)
  // Physics.DYN: the mixer of BaseAviary.py:1794-1803 as a lever per rotor
  const double lx = p.dyn_mixer == DSIM_DYN_MIXER_PLUS ? p.arm : p.arm / sqrt(2.0);
  const double mx_x[4] = {1, 1, -1, -1}, my_x[4] = {-1, 1, 1, -1}, mx_p[4] = {0, 1, 0, -1}, my_p[4] = {-1, 0, 1, 0};
  for (int i = 0; i < 4; ++i) {
    d->dyn_lever[0][i] = (float)((p.dyn_mixer == DSIM_DYN_MIXER_PLUS ? mx_p[i] : mx_x[i]) * lx);
    d->dyn_lever[1][i] = (float)((p.dyn_mixer == DSIM_DYN_MIXER_PLUS ? my_p[i] : my_x[i]) * lx);
  }
  d->weight = (float)(p.gravity * p.mass);
  // Six-actuator kinds: the Cholesky factor of the covariance of the body wrench that the per-rotor noise of
  // BaseAviary.py:1429-1457 adds up to (dsim_device.h:noise_normals, hexa_wrench_z).  W = M n, column j of M = (a_j, r_j x a_j) for the
  // force normal of rotor j (deviation 0.01), (0, spin_j a_j) for its moment normal (0.001), from the SAME fp32 constants
  // hexa_wrench multiplies; cov = M diag(sigma^2) M^T; L L^T = cov, stored over 0.01 (the normals arrive scaled by 0.01).
  if (p.n_act == 6) {
    double M[6][12], cov[6][6], L[6][6];
    for (int j = 0; j < 6; ++j)
      for (int k = 0; k < 3; ++k) {
        M[k][j] = 0.01 * (double)d->raxis[j][k]; M[3 + k][j] = 0.01 * (double)d->rxa[j][k];
        M[k][6 + j] = 0.0; M[3 + k][6 + j] = 0.001 * (double)d->spax[j][k];
      }
    for (int r = 0; r < 6; ++r)
      for (int c = 0; c < 6; ++c) { double acc = 0.0; for (int j = 0; j < 12; ++j) acc += M[r][j] * M[c][j]; cov[r][c] = acc; L[r][c] = 0.0; }
    for (int r = 0; r < 6; ++r)
      for (int c = 0; c <= r; ++c) {
        double acc = cov[r][c];
        for (int k = 0; k < c; ++k) acc -= L[r][k] * L[c][k];
        L[r][c] = r == c ? sqrt(acc > 0.0 ? acc : 0.0) : (L[c][c] > 0.0 ? acc / L[c][c] : 0.0);      // (a degenerate geometry: that direction gets no noise)
      }
    int q = 0;
    for (int r = 0; r < 6; ++r)
      for (int c = 0; c <= r; ++c) d->nchol[q++] = (float)(L[r][c] / 0.01);
  }
}

int fill_stepk(dsim_ctx* ctx, int64_t n, const dsim_view& state, const dsim_view* targets,
                      const dsim_step_args* args, StepK* a) {
  if (!ctx || !args || n <= 0 || n > state.n_pad) return DSIM_E_ARG;
  if (args->phys_substeps < 0 || !(args->dt_phys > 0) || !(args->dt_ctrl > 0)) return DSIM_E_ARG;
  if (ctx->n_types > 1 && !args->type_id) return DSIM_E_ARG;
  int rc = make_kview(state, 20 + ctx->max_act, &a->st);
  if (rc) return rc;
  if (targets && !args->wp_table) {
    const bool bc = (args->options & DSIM_OPT_BCAST_TGT) != 0;
    rc = make_kview(*targets, DSIM_NT, &a->tg, bc);
    if (rc) return rc;
    if (!bc && targets->n_pad != state.n_pad) return DSIM_E_LAYOUT;
  } else {
    memset(&a->tg, 0, sizeof(a->tg));
  }
  a->hexa_types = 0;
  for (int t = 0; t < ctx->n_types; ++t) a->hexa_types |= (ctx->h_types[t].kind != DSIM_KIND_QUAD ? 1u : 0u) << t;
  a->types = ctx->d_types; a->type_id = args->type_id; a->noise_replay = args->noise_replay;
  a->action = args->action; a->echo = nullptr; a->pos_e_out = nullptr; a->yaw_e_out = nullptr;
  a->cmd_out = nullptr; a->obs_out = nullptr; a->obs_w = 16 + ctx->max_act; a->n = n;
  a->fb.entries = ctx->d_fb; a->fb.count = ctx->d_counters + 2; a->fb.counters = ctx->d_counters;
  a->n_pad = state.n_pad; a->first = 0; a->seed = args->noise_seed;
  a->wp_table = args->wp_table; a->wp_counter = args->wp_counter; a->wp_offset = args->wp_offset;
  a->n_wp = args->n_wp; a->n_steps = args->n_steps > 1 ? args->n_steps : 1;
  a->ext_force = args->ext_force; a->step_index_dev = (const unsigned long long*)args->step_index_dev;
  if (a->wp_table && (!a->wp_counter || a->n_wp < 1)) return DSIM_E_ARG;
  a->step_index = args->step_index;
  a->substeps = args->phys_substeps; a->dt_phys = args->dt_phys; a->dt_ctrl = args->dt_ctrl;
  if ((args->options & DSIM_OPT_NOISE_FINE) && (args->options & DSIM_OPT_NOISE_COARSE)) return DSIM_E_ARG;
  a->options = resolve_noise_lattice(args->options, args->phys_substeps);
  memset(&a->bin, 0, sizeof(a->bin));
  a->lo = 0; a->last = a->n_pad; a->run_type = 0;
  a->drone_id = args->drone_id;
  a->io_id = (args->options & DSIM_OPT_CALLER_IO) ? args->drone_id : nullptr;
  a->action_rows = (args->options & DSIM_OPT_ACTION_ROWS) ? 1 : 0;     // (honoured by the entry points that check it)
  a->dyn_rates = args->dyn_rpy_rates;
  return DSIM_OK;
}

// The deferred-fallback queue is the one ctx-owned buffer that depends on the fleet size: it is
// (re)allocated when a larger hexa fleet is first seen, never per call afterwards.
int fb_prepare(dsim_ctx* ctx, long long n_pad, hipStream_t st) {
  if (ctx->max_act != 6) return DSIM_OK;
  if (ctx->fb_cap < n_pad) {
    hipError_t e = hipStreamSynchronize(st);
    if (e != hipSuccess) return (int)e;
    if (ctx->d_fb) (void)hipFree(ctx->d_fb);
    ctx->d_fb = nullptr; ctx->fb_cap = 0;
    e = hipMalloc((void**)&ctx->d_fb, sizeof(FbEntry) * n_pad);
    if (e != hipSuccess) return (int)e;
    ctx->fb_cap = n_pad;
  }
  return DSIM_OK;   // the queue length is reset by k_wls_fallback itself
}

void fb_finish(dsim_ctx* ctx, const StepK& a, hipStream_t st) {
  if (ctx->max_act != 6) return;
  if (a.options & DSIM_OPT_DEFER_FALLBACK) return;          // the caller launches dsim_wls_fallback itself
  FbK f;
  f.st = a.st; f.types = a.types; f.type_id = a.type_id; f.fb = a.fb;
  f.cmd_out = a.cmd_out; f.n_pad = a.n_pad; f.io_id = a.io_id;
  f.fb.entries = ctx->d_fb;
  // one workgroup per 64 possible entries, at most one per CU (each holds 144 KB of LDS): a start-up transient that
  // queues a large part of a big fleet is worked off by the whole chip, an empty queue costs one scalar load per group
  const long long groups = (a.n_pad + DSIM_FB_LANES - 1) / DSIM_FB_LANES;
  hipLaunchKernelGGL(k_wls_fallback, dim3((unsigned)(groups < ctx->n_cu ? groups : ctx->n_cu)), dim3(DSIM_FB_LANES), 0, st, f);
}

// Lays the runs of a type-major fleet out over the workgroups of ONE launch (RunTab): run r takes the whole 256-drone
// tiles from the one that holds its first drone to the one that holds its last.  Returns the number of workgroups, or a
// negative error code.  n_runs <= DSIM_MAX_TYPES.
int make_runtab(const dsim_ctx* ctx, long long n_pad, const dsim_type_run* runs, int n_runs, RunTab* rt, bool* any_hexa) {
  memset(rt, 0, sizeof(*rt));
  *any_hexa = false;
  if (n_runs < 1 || n_runs > DSIM_MAX_TYPES) return DSIM_E_ARG;
  int blocks = 0;
  for (int r = 0; r < DSIM_MAX_TYPES; ++r) {
    rt->blk0[r] = blocks;
    if (r >= n_runs) continue;
    const dsim_type_run& run = runs[r];
    if (run.first < 0 || run.count < 0 || run.first + run.count > n_pad || run.type < 0 || run.type >= ctx->n_types) return DSIM_E_ARG;
    rt->first[r] = run.first & ~255LL; rt->lo[r] = run.first; rt->last[r] = run.first + run.count; rt->type[r] = run.type;
    const int kind = ctx->h_types[run.type].kind;
    if (kind != DSIM_KIND_QUAD) rt->hexa_mask |= 1u << r;
    if (kind == DSIM_KIND_HEXA_QUADLAW) rt->quadlaw6_mask |= 1u << r;
    if (kind == DSIM_KIND_HEXA6DOF) *any_hexa = true;                 // (the WLS fallback queue is the 6-DOF law's)
    blocks += run.count > 0 ? (int)((rt->last[r] - rt->first[r] + 255) / 256) : 0;
  }
  rt->blk0[DSIM_MAX_TYPES] = blocks;
  return blocks;
}

// RunTab.block_map: the workgroups of the runs dealt side by side, one tile at a time to the run that is furthest behind
// (progress = tiles served / tiles of the run), so that every run sweeps the caller's index range at the same pace.  Kept
// by the ctx and re-made only when the runs change: that rare path waits for the whole DEVICE (a launch of this ctx on
// another stream may still read the old table) and may allocate — so the first DSIM_OPT_CALLER_IO call with a new set of
// runs must not sit inside a stream capture (include/dronesim_amd.h); the upload is ordered on the caller's stream.
int side_by_side_map(dsim_ctx* ctx, hipStream_t st, const dsim_type_run* runs, int n_runs, RunTab* rt) {
  const int blocks = rt->blk0[DSIM_MAX_TYPES] + (rt->blk0[DSIM_MAX_TYPES] & 1);      // (two entries per workgroup: an odd count is padded)
  rt->block_map = nullptr;
  if (blocks < 2) return DSIM_OK;
  bool same = ctx->d_block_map && ctx->block_map_blocks == blocks && ctx->block_map_runs == n_runs;
  for (int r = 0; same && r < n_runs; ++r)
    same = ctx->block_map_key[r].first == runs[r].first && ctx->block_map_key[r].count == runs[r].count && ctx->block_map_key[r].type == runs[r].type;
  if (!same) {
    if (ctx->block_map_cap < blocks) {
      hipError_t e = hipDeviceSynchronize();                   // (a launch in flight, on any stream, may still read the old table)
      if (e != hipSuccess) return (int)e;
      if (ctx->d_block_map) (void)hipFree(ctx->d_block_map);
      free(ctx->h_block_map);
      ctx->d_block_map = nullptr; ctx->h_block_map = nullptr; ctx->block_map_cap = 0;
      ctx->h_block_map = (int*)malloc(sizeof(int) * (size_t)blocks);
      if (!ctx->h_block_map) return (int)hipErrorOutOfMemory;
      e = hipMalloc((void**)&ctx->d_block_map, sizeof(int) * (size_t)blocks);
      if (e != hipSuccess) return (int)e;
      ctx->block_map_cap = blocks;
    } else {
      hipError_t e = hipDeviceSynchronize();
      if (e != hipSuccess) return (int)e;
    }
    int next[DSIM_MAX_TYPES], total[DSIM_MAX_TYPES];
    for (int r = 0; r < DSIM_MAX_TYPES; ++r) { next[r] = 0; total[r] = r < n_runs ? rt->blk0[r + 1] - rt->blk0[r] : 0; }
    // workgroup w serves entries 2 w and 2 w + 1: one tile of each of the two runs that are furthest behind (progress =
    // tiles served / tiles of the run), so that every run sweeps the caller's index range at the same pace and the two
    // tiles of a workgroup cover the same stretch of it
    for (int b = 0; b < blocks; ++b) {
      int pick = -1;
      for (int r = 0; r < n_runs; ++r) {
        if (next[r] >= total[r]) continue;
        if (pick < 0 || (long long)next[r] * total[pick] < (long long)next[pick] * total[r]) pick = r;
      }
      ctx->h_block_map[b] = pick < 0 ? -8 : ((next[pick]++ << 3) | pick);              // (-8: tile -1, nothing to serve)
    }
    hipError_t e = hipMemcpyAsync(ctx->d_block_map, ctx->h_block_map, sizeof(int) * (size_t)blocks, hipMemcpyHostToDevice, st);
    if (e != hipSuccess) return (int)e;
    ctx->block_map_blocks = blocks; ctx->block_map_runs = n_runs;
    for (int r = 0; r < n_runs; ++r) ctx->block_map_key[r] = runs[r];
  }
  rt->block_map = ctx->d_block_map;
  return DSIM_OK;
}

int observe_impl(dsim_ctx* ctx, void* stream, int64_t n, dsim_view state, const float* last_action,
                        float* obs_out, int32_t obs_width, int soa) {
  if (!ctx || !obs_out || n <= 0 || n > state.n_pad) return DSIM_E_ARG;
  if (obs_width < 16 || obs_width > 16 + DSIM_MAX_ACT || 20 + (obs_width - 16) > state.n_fields) return DSIM_E_ARG;
  ObsK a;
  int rc = make_kview(state, 20 + ctx->max_act, &a.st);
  if (rc) return rc;
  a.last_action = last_action; a.out = obs_out; a.n = n; a.n_pad = state.n_pad; a.width = obs_width; a.soa = soa;
  hipLaunchKernelGGL(k_observe, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

extern "C" {

int dsim_abi_version(void) { return DSIM_ABI_VERSION; }

const char* dsim_strerror(int code) {
  switch (code) {
    case DSIM_OK: return "ok";
    case DSIM_E_ARG: return "dsim: null or inconsistent argument";
    case DSIM_E_LAYOUT: return "dsim: view violates the blocked-SoA layout contract";
    case DSIM_E_NODEVICE: return "dsim: no HIP device (gfx950 required; there is no CPU fallback)";
    case DSIM_E_TYPES: return "dsim: bad type table";
    case DSIM_E_UNSUPPORTED: return "dsim: unsupported configuration";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "dsim: unknown error";
  }
}

int dsim_create(dsim_ctx** out, int device, const dsim_type_params* types, int n_types) {
  if (!out || !types) return DSIM_E_ARG;
  if (n_types < 1 || n_types > DSIM_MAX_TYPES) return DSIM_E_TYPES;
  int max_act = 4;
  for (int t = 0; t < n_types; ++t) {
    if (!(types[t].mass > 0)) return DSIM_E_TYPES;
    if (types[t].kind == DSIM_KIND_QUAD) { if (types[t].n_act != 4) return DSIM_E_TYPES; }
    else if (types[t].kind == DSIM_KIND_HEXA6DOF || types[t].kind == DSIM_KIND_HEXA_QUADLAW) {
      if (types[t].n_act != 6) return DSIM_E_TYPES;
      max_act = 6;
    }
    else return DSIM_E_TYPES;
    for (int k = 0; k < 3; ++k) if (!(types[t].inertia[k] > 0)) return DSIM_E_TYPES;
  }
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count == 0 || device < 0 || device >= count) return DSIM_E_NODEVICE;
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return (int)e;
  dsim_ctx* c = new (std::nothrow) dsim_ctx;
  if (!c) return (int)hipErrorOutOfMemory;
  c->device = device; c->n_types = n_types; c->max_act = max_act; c->d_types = nullptr; c->d_counters = nullptr;
  c->d_fb = nullptr; c->fb_cap = 0; c->dw_ws = nullptr; c->dw_cells = 0; c->dw_parity = 0; c->dw_mode = 0;
  c->n_cu = 256; c->dw_prebin = false; c->dw_prebin_valid = false; c->dw_prebin_n = 0; c->dw_prebin_off = 0;
  c->dw_prebin_geo[0] = c->dw_prebin_geo[1] = c->dw_prebin_geo[2] = 0.0f;
  c->dw_prebin_nx = c->dw_prebin_ny = 0; c->dw_local_m = 0; c->dw_prebin_kind = 0; c->dw_reuses = 0; c->h_keep_fb = nullptr; c->d_keep_fb = nullptr; c->dw_keep_ws = nullptr; c->dw_keep_cells = c->dw_keep_n = 0;
  c->dw_keep_geo[0] = c->dw_keep_geo[1] = c->dw_keep_geo[2] = c->dw_keep_geo[3] = 0.0f; c->dw_keep_nx = c->dw_keep_ny = 0; c->dwh_parity = 0; c->dwh_ws = nullptr; c->dwh_cells = 0;
  c->d_bounds = nullptr;
  c->d_block_map = nullptr; c->h_block_map = nullptr; c->block_map_cap = 0; c->block_map_blocks = 0; c->block_map_runs = 0;
  { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && v > 0) c->n_cu = v; }
  // the device table always holds DSIM_MAX_TYPES entries, the unused ones zero: a kernel instance compiled for four types may
  // load the constants of a type no drone has (k_step_mixed4 serves tables of three types with its four-type instance)
  DevType h[DSIM_MAX_TYPES];
  memset(h, 0, sizeof(h));
  for (int t = 0; t < n_types; ++t) { c->h_types[t] = types[t]; to_dev(types[t], &h[t]); }
  e = hipMalloc((void**)&c->d_types, sizeof(DevType) * DSIM_MAX_TYPES);
  if (e == hipSuccess) e = hipMemcpy(c->d_types, h, sizeof(DevType) * DSIM_MAX_TYPES, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc((void**)&c->d_counters, sizeof(unsigned long long) * (8 + DSIM_GROUND_SHARDS));
  if (e == hipSuccess) e = hipMemset(c->d_counters, 0, sizeof(unsigned long long) * (8 + DSIM_GROUND_SHARDS));
  if (e == hipSuccess) e = hipMalloc((void**)&c->d_bounds, sizeof(unsigned) * 8);
  if (e == hipSuccess) {
    // two ints of host memory the REUSE queries of the neighbour downwash report into (dsim_downwash_keep_stats): best effort —
    // without it the statistics read as "nothing known"
    void* hp = nullptr;
    void* dp = nullptr;
    if (hipHostMalloc(&hp, 4 * sizeof(int), hipHostMallocMapped) == hipSuccess) {
      memset(hp, 0, 4 * sizeof(int));
      if (hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess) { c->h_keep_fb = (volatile int*)hp; c->d_keep_fb = (int*)dp; }
      else (void)hipHostFree(hp);
    }
    (void)hipGetLastError();
  }
  if (e == hipSuccess) {
    const unsigned init[8] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u, 0u, 0u, 0u};     // min keys, max keys, ticket
    e = hipMemcpy(c->d_bounds, init, sizeof(init), hipMemcpyHostToDevice);
  }
  if (e != hipSuccess) {
    if (c->d_types) (void)hipFree(c->d_types);
    if (c->d_counters) (void)hipFree(c->d_counters);
    if (c->d_bounds) (void)hipFree(c->d_bounds);
    delete c;
    return (int)e;
  }
  *out = c;
  return DSIM_OK;
}

int dsim_destroy(dsim_ctx* ctx) {
  if (!ctx) return DSIM_E_ARG;
  hipError_t e = hipFree(ctx->d_types);
  (void)hipFree(ctx->d_counters);
  (void)hipFree(ctx->d_bounds);
  if (ctx->h_keep_fb) (void)hipHostFree((void*)ctx->h_keep_fb);
  if (ctx->d_fb) (void)hipFree(ctx->d_fb);
  if (ctx->d_block_map) (void)hipFree(ctx->d_block_map);
  free(ctx->h_block_map);
  delete ctx;
  return (int)e;
}

int dsim_dev_alloc(dsim_ctx* ctx, int64_t bytes, void** out) {
  if (!ctx || !out || bytes <= 0) return DSIM_E_ARG;
  *out = nullptr;
  hipError_t e = hipSetDevice(ctx->device);
  if (e == hipSuccess) e = hipMalloc(out, (size_t)bytes);
  return (int)e;
}

int dsim_dev_free(dsim_ctx* ctx, void* ptr) {
  (void)ctx;                                          // (may be NULL: a block may outlive the ctx it was allocated through)
  return ptr ? (int)hipFree(ptr) : DSIM_OK;           // (hipFree waits for the work that may still use the block)
}

int dsim_query(dsim_ctx* ctx, void* stream, int32_t what, int64_t* value_out) {
  if (!ctx || !value_out || what < 0 || what > DSIM_Q_DW_MOVERS) return DSIM_E_ARG;
  unsigned long long h[8 + DSIM_GROUND_SHARDS];
  hipError_t e = hipMemcpyAsync(h, ctx->d_counters, sizeof(h), hipMemcpyDeviceToHost, (hipStream_t)stream);
  if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  if (what == DSIM_Q_GROUND_CONTACTS) {
    unsigned long long sum = 0;
    for (int k = 0; k < DSIM_GROUND_SHARDS; ++k) sum += h[8 + k];
    *value_out = (int64_t)sum;
  } else if (what == DSIM_Q_HALO_OVERFLOW) {
    *value_out = (int64_t)h[4];
  } else if (what == DSIM_Q_DW_REUSES) {
    *value_out = (int64_t)ctx->dw_reuses;
  } else if (what == DSIM_Q_DW_MOVERS) {
    *value_out = (int64_t)h[5];
  } else {
    *value_out = (int64_t)h[what];
  }
  return DSIM_OK;
}

int dsim_reset(dsim_ctx* ctx, void* stream, int64_t n, dsim_view state, const float* init_pos,
               const float* init_rpy, const float* init_vel, const float* init_cmd, const uint8_t* type_id) {
  if (!ctx || !init_pos || !init_rpy || n <= 0 || n > state.n_pad) return DSIM_E_ARG;
  if (ctx->n_types > 1 && !type_id) return DSIM_E_ARG;
  ResetK a;
  int rc = make_kview(state, 20 + ctx->max_act, &a.st);
  if (rc) return rc;
  ctx->dw_prebin_valid = false;
  a.types = ctx->d_types; a.type_id = type_id;
  a.pos = init_pos; a.rpy = init_rpy; a.vel = init_vel; a.cmd = init_cmd;
  a.n_pad = state.n_pad; a.n_fields = 20 + ctx->max_act;
  hipLaunchKernelGGL(k_reset, dim3(grid_for(a.n_pad)), dim3(256), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

int dsim_reserve(dsim_ctx* ctx, void* stream, int64_t n_pad) {
  if (!ctx || n_pad <= 0) return DSIM_E_ARG;
  return fb_prepare(ctx, n_pad, (hipStream_t)stream);
}

int dsim_wls_fallback(dsim_ctx* ctx, void* stream, int64_t n, dsim_view state, const uint8_t* type_id, float* cmd_out) {
  if (!ctx || n <= 0 || n > state.n_pad) return DSIM_E_ARG;
  if (ctx->max_act != 6) return DSIM_OK;                     // no morphing hexa in the table: nothing is ever queued
  if (ctx->n_types > 1 && !type_id) return DSIM_E_ARG;
  StepK a;
  memset(&a, 0, sizeof(a));
  int rc = make_kview(state, 20 + ctx->max_act, &a.st);
  if (rc) return rc;
  rc = fb_prepare(ctx, state.n_pad, (hipStream_t)stream);
  if (rc) return rc;
  a.types = ctx->d_types; a.type_id = type_id; a.n_pad = state.n_pad; a.cmd_out = cmd_out;
  a.fb.entries = ctx->d_fb; a.fb.count = ctx->d_counters + 2; a.fb.counters = ctx->d_counters;
  fb_finish(ctx, a, (hipStream_t)stream);
  return (int)hipGetLastError();
}

int dsim_noise_draw(dsim_ctx* ctx, void* stream, int64_t n, int64_t n_pad, int32_t n_act, uint64_t noise_seed, uint64_t step_index,
                    int32_t substeps, uint32_t options, const int32_t* drone_id, float* out) {
  if (!ctx || !out || n <= 0 || n > n_pad || (n_act != 4 && n_act != 6) || substeps < 1 || noise_seed == 0) return DSIM_E_ARG;
  NoiseK a;
  a.n = n; a.n_pad = n_pad; a.n_act = n_act; a.substeps = substeps; a.seed = noise_seed; a.step_index = step_index;
  if ((options & DSIM_OPT_NOISE_FINE) && (options & DSIM_OPT_NOISE_COARSE)) return DSIM_E_ARG;
  a.options = resolve_noise_lattice(options, substeps); a.drone_id = drone_id; a.out = out;
  hipLaunchKernelGGL(k_noise_draw, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

int dsim_counter_add(dsim_ctx* ctx, void* stream, uint64_t* counter, uint64_t inc) {
  if (!ctx || !counter) return DSIM_E_ARG;
  hipLaunchKernelGGL(k_counter_add, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long*)counter,
                     (unsigned long long)inc);
  return (int)hipGetLastError();
}

int dsim_observe(dsim_ctx* ctx, void* stream, int64_t n, dsim_view state, const float* last_action,
                 float* obs_out, int32_t obs_width) {
  return observe_impl(ctx, stream, n, state, last_action, obs_out, obs_width, 0);
}

int dsim_observe_soa(dsim_ctx* ctx, void* stream, int64_t n, dsim_view state, const float* last_action,
                     float* obs_out, int32_t obs_width) {
  return observe_impl(ctx, stream, n, state, last_action, obs_out, obs_width, 1);
}

int dsim_traj_sample(dsim_ctx* ctx, void* stream, int64_t n, const double* coeffs, const double* ts,
                     int32_t n_seg, double* t, double dt_advance, double* yaw_state, const float* offset,
                     dsim_view targets_out) {
  if (!ctx || !coeffs || !ts || !t || !yaw_state || n <= 0 || n > targets_out.n_pad || n_seg < 1) return DSIM_E_ARG;
  TrajK a;
  int rc = make_kview(targets_out, DSIM_NT, &a.tg);
  if (rc) return rc;
  a.coeffs = coeffs; a.ts = ts; a.t = t; a.yaw_state = yaw_state; a.offset = offset;
  a.n = n; a.n_pad = targets_out.n_pad; a.n_seg = n_seg; a.dt_advance = dt_advance;
  hipLaunchKernelGGL(k_traj_sample, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

}  // extern "C"
