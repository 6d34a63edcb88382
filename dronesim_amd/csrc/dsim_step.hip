// dsim_step.hip — dsim_step: Env.step + computeControl in ONE launch (the hot path of examples/fly_INDI.py:217-245), every
// kernel family that serves it (gfx950 only).
#include "dsim_kernels.h"

// ---- fused Env.step + computeControl (the hot path) -----------------------
// Fast form: homogeneous quad fleet, action = the controller's stored cmd, whole 256-drone tiles
// (the launcher hands ragged tails and every other configuration to the general kernel below).
// No per-lane branches and no bounds checks, so every access keeps the scalar-base + 32-bit
// lane-offset form (one VGPR of addressing for all 58 accesses).
//
// (Measured and rejected on MI355X, 4.2 M drones: a persistent grid-stride form that prefetches
// the next tile into registers, 266 vs 176 us, and two tiles per workgroup with both tiles' loads
// issued up front, 190 vs 163 us — fewer, fatter waves hide HBM latency worse than 4 waves/SIMD of
// this short kernel; forcing 4 waves/SIMD by spilling also lost, 176 vs 172 us.)
// EXT = waypoint-table targets and/or several steps per launch; the plain single-step kernel is
// compiled without that generality (it would cost the hot kernel registers: 128 + spills vs 121).
// CH = DSIM_OPT_CHAINED: last_vel / last_rates are recomputed from the rigid state the previous step
// stored (they are functions of it) instead of being read, and are not written: 184 B/drone-step.
// ACT = an explicit action for the physics part (dsim_step_args.action: the first iteration of the example loop,
// or a caller that overrides the controller): four more loads, clipped as CtrlAviary._preprocessAction does; the
// controller memory keeps its own cmd.  A template flag so that the plain form does not even test the pointer.
template <bool NOISE, bool NT, bool EXT, bool CH = false, int SUB = 0, bool ACT = false>
__global__ __launch_bounds__(256, EXT ? 3 : DSIM_STEP_WAVES) void k_step_fast(StepK a) {
  const DevType& T = a.types[0];
  const long long sfs = a.st.field_stride, tfs = a.tg.field_stride;
  const unsigned sl = 4u * kv_lane(a.st, threadIdx.x), tl = 4u * kv_lane(a.tg, threadIdx.x);   // bytes
  const long long i0 = (long long)blockIdx.x * 256;                   // wave-uniform
  float* const sb = a.st.base + kv_off(a.st, i0);                     // scalar bases
  const float* const tb = a.tg.base + kv_off(a.tg, i0);
  Rigid s;
  CtrlMem<4> m;
  Target tg;
  // several sub-steps per launch (the examples' setting: vector-issue bound): the Box-Muller pairs from LDS tables
  constexpr bool TAB = NOISE && (EXT || SUB != 1);
  __shared__ NoiseTab ntab_[TAB ? 1 : 0 + 1];
  const NoiseTab* const ntab = TAB ? &ntab_[0] : nullptr;
  if (TAB) noise_tab_init(ntab_[0], threadIdx.x);
  load_rigid<NT>(sb, sfs, sl, s);
  // (Measured and rejected, round 5: the loads only the control law needs — 7 controller-memory floats, 10 targets — issued BEHIND
  // the sub-step loop of the looped instances instead of in front of it: 80 -> 74 VGPRs, still 6 waves per SIMD, 166.9 against
  // 164.6 us for five sub-steps; forced to 7 waves (72 VGPRs, 16 B of scratch) 171.5 us; issued at the top of the last sub-step
  // the compiler peels that iteration: 96 VGPRs and scratch.)
  load_mem<4, NT, CH>(sb, sfs, sl, m);
  if (TAB) __syncthreads();
  if (CH) { m.last_vel = s.vel; m.last_rates = mulT(matrix_from_quat(s.q), s.w); }
  const long long i = i0 + threadIdx.x;
  if (NOISE && a.step_index_dev) a.step_index += *a.step_index_dev;    // wave-uniform scalar load
  V3 pos_e;
  float yaw_e;
  if (!EXT) {
    load_target<NT>(tb, tfs, tl, tg);
    if (ACT) {
      float act[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) act[j] = clampf(a.action[(long long)j * a.n_pad + i], T.pmin[j], T.pmax[j]);   // CtrlAviary.py:258-263
      quad_substeps<NOISE ? 1 : 0, 4, false, SUB, false, -1, SUB == 0, TAB ? 1 : 0>(T, a, i, s, act, a.step_index, V3{-0.0f, -0.0f, -0.0f}, nullptr, -1, ntab);
    } else {
      quad_substeps<NOISE ? 1 : 0, 4, false, SUB, false, -1, SUB == 0, TAB ? 1 : 0>(T, a, i, s, m.cmd, a.step_index, V3{-0.0f, -0.0f, -0.0f}, nullptr, -1, ntab);    // stored cmd is already clipped
    }
    if (SUB == 1) ground_watch(T, s, a.fb.counters, i < a.n);     // (the single-sub-step instances: see the end of the kernel)
    indi_quad<false>(T, a.dt_ctrl, s, tg, m, pos_e, yaw_e);
  } else {
    int wp = 0;
    if (a.wp_table) wp = a.wp_counter[i]; else load_target<NT>(tb, tfs, tl, tg);
    for (int k = 0; k < a.n_steps; ++k) {
      if (a.wp_table) waypoint_target(a, i, wp, tg);
      // (wave-uniform: several sub-steps take the body-frame loop, as the looped plain instances do — at BASELINE's literal sizes
      // these launches are one wave per SIMD and their duration IS their instruction count; one sub-step keeps the world-frame step)
      if (a.substeps > 1) quad_substeps<NOISE ? 1 : 0, 4, false, 0, false, -1, true, TAB ? 1 : 0>(T, a, i, s, m.cmd, a.step_index + k, V3{-0.0f, -0.0f, -0.0f}, nullptr, -1, ntab);
      else quad_substeps<NOISE ? 1 : 0, 4, false, 0, false, -1, false, TAB ? 1 : 0>(T, a, i, s, m.cmd, a.step_index + k, V3{-0.0f, -0.0f, -0.0f}, nullptr, -1, ntab);
      ground_watch(T, s, a.fb.counters, i < a.n);
      indi_quad<false>(T, a.dt_ctrl, s, tg, m, pos_e, yaw_e);
      wp = waypoint_next(wp, a.n_wp);
    }
    if (a.wp_table) a.wp_counter[i] = wp;
  }
  const unsigned so = pin_lane_offset(sl);
  store_rigid<NT>(sb, sfs, so, s);
  store_mem<4, NT, CH>(sb, sfs, so, m);
  // (the looped instances, at the very end: the counter's atomic between the physics and the law is a memory write in front of
  // the law's type constants, which then arrive by VECTOR loads — 28 VGPRs of constants and a vmcnt(0) in the middle of the
  // kernel, which a fleet of one wave per SIMD waits out in full: 4 096 quads x 5 sub-steps 6.37 -> 6.14 us per launch, 4 194 304
  // 166.4 -> 163.7 us settled.  The single-sub-step instances keep the watch where it was: the headline kernel, on its memory
  // floor with the constants in VGPRs, measured 153-157 us there and 161-162 us with the watch at the end — same box, two
  // processes each, profiles/r05_ab_ground_watch_at_the_end.txt)
  if (!EXT && SUB != 1) ground_watch(T, s, a.fb.counters, i < a.n);
}

// The same fast form for a homogeneous morphing-hexa fleet (6-DOF INDI, first WLS iteration in closed form,
// infeasible drones queued for k_wls_fallback): whole tiles, stored cmd as the action, one Env.step per
// launch.  Compiled apart from the mixed-fleet kernel, whose quad branch and per-lane options cost it
// registers (177-252 VGPRs, 2 waves/SIMD).
template <bool NOISE, bool NT, bool S1, bool ACT = false>
__global__ __launch_bounds__(256, DSIM_HEXA_WAVES) void k_step_hexa(StepK a) {
  const DevType& T = a.types[0];
  const long long sfs = a.st.field_stride, tfs = a.tg.field_stride;
  const unsigned sl = 4u * kv_lane(a.st, threadIdx.x), tl = 4u * kv_lane(a.tg, threadIdx.x);
  const long long i0 = (long long)blockIdx.x * 256;
  float* const sb = a.st.base + kv_off(a.st, i0);
  const float* const tb = a.tg.base + kv_off(a.tg, i0);
  constexpr bool TAB = NOISE && !S1;          // several sub-steps per launch: the Box-Muller pairs from LDS tables (k_step_fast)
  __shared__ NoiseTab ntab_[TAB ? 1 : 0 + 1];
  const NoiseTab* const ntab = TAB ? &ntab_[0] : nullptr;
  if (TAB) noise_tab_init(ntab_[0], threadIdx.x);
  Rigid s;
  CtrlMem<6> m;
  Target tg;
  load_rigid<NT>(sb, sfs, sl, s);
  load_mem<6, NT>(sb, sfs, sl, m);
  load_target<NT>(tb, tfs, tl, tg);
  if (TAB) __syncthreads();
  const long long i = i0 + threadIdx.x;
  if (NOISE && a.step_index_dev) a.step_index += *a.step_index_dev;
  V3 pos_e;
  float yaw_e;
  if (ACT) {
    float act[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) act[j] = clampf(a.action[(long long)j * a.n_pad + i], T.pmin[j], T.pmax[j]);
    hexa_substeps<NOISE, false, S1, false, !S1, TAB ? 1 : 0>(T, a, i, s, act, a.step_index, V3{-0.0f, -0.0f, -0.0f}, -1, ntab);
  } else {
    hexa_substeps<NOISE, false, S1, false, !S1, TAB ? 1 : 0>(T, a, i, s, m.cmd, a.step_index, V3{-0.0f, -0.0f, -0.0f}, -1, ntab);
  }
  indi_hexa<false>(T, a.dt_ctrl, s, tg, m, pos_e, yaw_e, a.fb, i);
  // The looped instances store through a base the compiler cannot see through, made behind the sub-steps (opaque_after): left to
  // itself it keeps the 23 field addresses it formed for the loads (SGPR pairs) alive across the sub-step loop, runs out of SGPRs
  // inside it and parks 46 of them in VGPR lanes — 92 v_writelane / v_readlane of the ~2 100 vector instructions a looped launch
  // executes; formed again behind the loop they are 46 scalar adds.
  float* const sb2 = (DSIM_LATE_STORE_BASE && (!S1 || DSIM_LATE_STORE_BASE_S1)) ? const_cast<float*>(opaque_after(sb, s.pos.x)) : sb;
  const unsigned so = pin_lane_offset(sl);
  store_rigid<NT>(sb2, sfs, so, s);
  store_mem<6, NT>(sb2, sfs, so, m);
  ground_watch(T, s, a.fb.counters, i < a.n);       // (at the very end: between the physics and the law it cost 44 VGPRs)
}

// ends a chained sequence: last_vel / last_rates back into the state block
struct MatK { KView st; long long n_pad; };
__global__ __launch_bounds__(256) void k_materialize(MatK a) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n_pad) return;
  float* p = a.st.base + kv_off(a.st, i);
  const long long fs = a.st.field_stride;
  Rigid s;
  load_rigid(p, fs, 0u, s);
  const V3 wb = mulT(matrix_from_quat(s.q), s.w);
  p[13 * fs] = s.vel.x; p[14 * fs] = s.vel.y; p[15 * fs] = s.vel.z;
  p[16 * fs] = wb.x; p[17 * fs] = wb.y; p[18 * fs] = wb.z;
}

// Type-major storage (dsim_step_args.runs): a run of one type is stepped by the single-type law of its kind,
// the fast form (no partition, no waterfall, per-type constants in SGPRs); ext = optional downwash force.
// ACT: an explicit action for the physics part (dsim_step_args.action: the first iteration of the example loop), clipped as
// CtrlAviary._preprocessAction does; the controller memory keeps its own cmd (k_step_runs only: a template flag, as in k_step_fast)
// KIND: DSIM_DEV_KIND_* of the run's type — 2 = morphing-hexa physics with the quad law on its six actuators
template <int KIND, bool NOISE, bool NT, bool S1, bool ACT = false>
__device__ __forceinline__ void run_body(const StepK& a, long long i0, long long lo, long long last, int run_type,
                                         const NoiseTab* tab = nullptr) {
  constexpr bool HEXA = KIND != DSIM_DEV_KIND_QUAD;            // six actuators, morphing-hexa physics
  const long long i = i0 + threadIdx.x;
  if (i >= last || i < lo) return;          // (a run may begin and end inside a tile: the neighbouring run's lanes take the rest)
  // (the constant address space — dsim_device.h, as in the two-call run kernels — costs THIS body SGPR spills and a scratch
  // reservation: k_step_runs 166.9 against 162.7 us on the interleaved fleet, same-box A/B; and it buys a fleet of one wave per
  // SIMD, which waits out every vector load of a constant in full, nothing either: config 5's chain 45.4 us both ways, round 5)
  const DevType& T = a.types[run_type];
  const long long sfs = a.st.field_stride, tfs = a.tg.field_stride;
  const unsigned sl = 4u * kv_lane(a.st, threadIdx.x), tl = 4u * kv_lane(a.tg, threadIdx.x);
  float* const sb = a.st.base + kv_off(a.st, i0);
  const float* const tb = a.tg.base + kv_off(a.tg, i0);
  constexpr int NA = HEXA ? 6 : 4;
  Rigid s;
  CtrlMem<NA> m;
  Target tg;
  load_rigid<NT>(sb, sfs, sl, s);
  load_mem<NA, NT>(sb, sfs, sl, m);
  load_target<NT>(tb, tfs, tl, tg);
  V3 ext = v3(0, 0, 0);
  if (a.ext_force) ext = v3(a.ext_force[i], a.ext_force[a.n_pad + i], a.ext_force[2 * a.n_pad + i]);
  unsigned long long step_index = a.step_index;
  if (NOISE && a.step_index_dev) step_index += *a.step_index_dev;
  V3 pos_e;
  float yaw_e;
  const long long nid = NOISE ? noise_id(a, i) : -1LL;
  float act[NA];
#pragma unroll
  for (int j = 0; j < NA; ++j) act[j] = ACT ? clampf(a.action[(long long)j * a.n_pad + i], T.pmin[j], T.pmax[j]) : m.cmd[j];   // CtrlAviary.py:258-263
  if constexpr (HEXA) {
    hexa_substeps<NOISE, false, S1, false, !S1, (NOISE && !S1 && !ACT) ? 1 : 0>(T, a, i, s, act, step_index, ext, nid, tab);      // (the tables exist where the kernels make them: DSIM_NOISE_TAB)
    if constexpr (KIND == DSIM_DEV_KIND_HEXA) indi_hexa<false>(T, a.dt_ctrl, s, tg, m, pos_e, yaw_e, a.fb, i);
    else indi_quad<false, 6>(T, a.dt_ctrl, s, tg, m, pos_e, yaw_e);
  } else {
    quad_substeps<NOISE ? 1 : 0, 4, false, S1 ? 1 : 0, false, -1, !S1, (NOISE && !S1 && !ACT) ? 1 : 0>(T, a, i, s, act, step_index, ext, nullptr, nid, tab);
    indi_quad<false>(T, a.dt_ctrl, s, tg, m, pos_e, yaw_e);
  }
  const unsigned so = pin_lane_offset(sl);
  float* const sb2 = (DSIM_LATE_STORE_BASE && !S1) ? const_cast<float*>(opaque_after(sb, s.pos.x)) : sb;   // (k_step_hexa: the field addresses formed again behind the loop)
  store_rigid<NT>(sb2, sfs, so, s);
  store_mem<NA, NT>(sb2, sfs, so, m);
  ground_watch(T, s, a.fb.counters, i < a.n);
  // (measured and dropped: reserving the slot of the next grid right behind the physics, so that the atomic's round trip
  // rides under the control law — 45.4 against 45.7 us for the config-5 chain, and 36 bytes of scratch in two instances)
  // (the quad law on six actuators has no registers left for the refreshing form of kept lists: 20 bytes of scratch)
  if (a.bin.count && i < a.n) bin_entry<KIND != DSIM_DEV_KIND_HEXA_QUADLAW>(a.bin, s.pos.x, s.pos.y, s.pos.z, a.bin.local_offset + i);   // next step's grid
}
template <int KIND, bool NOISE, bool NT, bool S1>
__global__ __launch_bounds__(256, KIND ? DSIM_HEXA_WAVES : DSIM_STEP_WAVES) void k_step_run(StepK a) {
  DSIM_NOISE_TAB(NOISE && !S1, 256);
  run_body<KIND, NOISE, NT, S1>(a, a.first + (long long)blockIdx.x * 256, a.lo, a.last, a.run_type, ntab);
}
template <bool NOISE, bool NT, bool S1, bool ACT>
__global__ __launch_bounds__(256, 3) void k_step_runs(StepK a, RunTab rt) {
  DSIM_RUN_OF_BLOCK(rt, ro, blockIdx.x);
  DSIM_NOISE_TAB(NOISE && !S1 && !ACT, 256);       // (the explicit-action instances: one step of an example loop; with the tables they spill)
  if (ro.hexa) run_body<DSIM_DEV_KIND_HEXA, NOISE, NT, S1, ACT>(a, ro.i0, ro.lo, ro.last, ro.type, ntab);
  else run_body<DSIM_DEV_KIND_QUAD, NOISE, NT, S1, ACT>(a, ro.i0, ro.lo, ro.last, ro.type, ntab);
}

extern "C" {

int dsim_step(dsim_ctx* ctx, void* stream, int64_t n, dsim_view state, dsim_view targets,
              const dsim_step_args* args) {
  StepK a;
  int rc = fill_stepk(ctx, n, state, &targets, args, &a);
  if (rc) return rc;
  if (args->options & DSIM_OPT_DYN) {
    rc = dyn_check(ctx, args, a);
    if (rc) return rc;
    if (!a.tg.base) return DSIM_E_ARG;
    ctx->dw_prebin_valid = false;
    return dyn_launch(true, a, stream_policy(args, state.n_pad, 256.0), (hipStream_t)stream);
  }
  if (args->options & DSIM_OPT_CALLER_IO) return DSIM_E_UNSUPPORTED;     // (dsim_physics / dsim_control2 only)
  ctx->dw_prebin_valid = false;      // the positions move: a grid binned before this call is stale (bin_next_commit re-validates)
  const bool noise = args->noise_seed != 0 || args->noise_replay != nullptr;
  const bool uni = args->type_id == nullptr;
  const bool six = ctx->max_act == 6;
  const hipStream_t st_ = (hipStream_t)stream;
  const dim3 b(256);
  long long first = 0;
  // the fine noise lattice (resolved by fill_stepk): carried by every instance but the looped fast ones (quad_substeps)
  const bool fine = noise && !args->noise_replay && (a.options & DSIM_OPT_NOISE_FINE) != 0;
  const bool fine_slow = fine && a.substeps > 1;        // several sub-steps per launch on the fine lattice: the general kernels
  const bool phys_opts = (args->options & (DSIM_OPT_DRAG | DSIM_OPT_GROUND | DSIM_OPT_PLANE)) != 0 || fine_slow;
  const bool plane = (args->options & DSIM_OPT_PLANE) != 0;
  if ((args->options & (DSIM_OPT_DRAG | DSIM_OPT_GROUND)) && six)
    return DSIM_E_UNSUPPORTED;                          // the add-on formulas are written for the four-rotor links
  if ((args->options & DSIM_OPT_CHAINED) && (!uni || six || args->action || args->noise_replay || args->ext_force ||
                                             phys_opts || (state.n_pad % 256)))
    return DSIM_E_UNSUPPORTED;                          // chained stepping is a fast-path-only mode
  const bool runs_ok = !args->noise_replay && !a.wp_table && a.n_steps == 1 && !phys_opts && a.tg.base &&
                       !(args->options & DSIM_OPT_CHAINED);
  const bool plain = runs_ok && !args->action;
  const dsim_type_run* runs = args->runs;
  int n_runs = args->n_runs;
  dsim_type_run whole;
  bool any_quadlaw6 = false;             // a DSIM_KIND_HEXA_QUADLAW type in the table: served by the per-run kernels (k_step_run)
  for (int t = 0; t < ctx->n_types; ++t) any_quadlaw6 |= ctx->h_types[t].kind == DSIM_KIND_HEXA_QUADLAW;
  if (!(runs && n_runs > 0) && uni && runs_ok && (args->ext_force || (any_quadlaw6 && !args->action))) {
    // a homogeneous fleet with an external (downwash) force, or of hexa_6DOF_simple: ONE run of its only type — the
    // single-type kernel with the force input and the fused neighbour-grid binning, instead of the general kernel
    whole.first = 0; whole.count = a.n_pad; whole.type = 0; whole._pad = 0;
    runs = &whole; n_runs = 1;
  }
  // (an explicit action — the first iteration of the example loop, fly_INDI.py:214 — is served by the ACT instances of the
  // one-launch form; beyond DSIM_MAX_TYPES runs it goes to the general kernel.  The ACT instances exist with the default cache
  // policy only: an explicit action is ONE step of a loop, the streaming hint would buy it nothing and cost twelve instances)
  if (runs && n_runs > 0 && runs_ok && (!args->action || (n_runs <= DSIM_MAX_TYPES && !any_quadlaw6))) {
    // type-major storage: one single-type launch per run
    const bool nt = stream_policy(args, state.n_pad, 240.0);
    bool any_hexa = false;
    bin_next_prepare(ctx, n, args, &a, st_);
    for (int r = 0; r < n_runs; ++r) {
      const dsim_type_run& run = runs[r];
      if (run.first < 0 || run.count < 0 || run.first + run.count > a.n_pad || run.type < 0 || run.type >= ctx->n_types)
        return DSIM_E_ARG;
      any_hexa |= ctx->h_types[run.type].kind == DSIM_KIND_HEXA6DOF;
    }
    if (any_hexa) {
      rc = fb_prepare(ctx, a.n_pad, st_);
      if (rc) return rc;
      a.fb.entries = ctx->d_fb;
    }
    // (measured on MI355X, 50 % quads + 50 % hexas: 65 536 drones 12.0 us against 9.5 + 9.0 us for two dependent launches;
    // 4 194 304 drones 160.1 against 165.2 us — the launch boundary between the runs costs more than the registers the
    // second law adds (83 VGPRs, 5 waves per SIMD, against 74 and 6): one launch is the default at every size for ONE sub-step.
    // With several sub-steps the launch is bound by vector issue and the looped instances differ more: k_step_runs 102 VGPRs, 4 waves
    // per SIMD, against 76 / 6 (quads) and 89 / 5 (hexas) for the single-law k_step_run — a fleet that fills the chip then takes
    // one launch per run (round 6: 4 194 304 interleaved quads + hexas x 5 sub-steps))
    const bool per_run_pays = a.substeps > 1 && a.n_pad >= (1LL << 20) && !args->action;
    const bool one_launch = !any_quadlaw6 && !per_run_pays;
    if (n_runs <= DSIM_MAX_TYPES && (n_runs >= 2 || args->action) && one_launch) {
      // several runs (or an explicit action): one launch for all of them (k_step_runs)
      RunTab rt;
      const int blocks = make_runtab(ctx, a.n_pad, runs, n_runs, &rt, &any_hexa);
      if (blocks < 0) return blocks;
      if (blocks > 0) {
        const dim3 g((unsigned)blocks);
#define DSIM_RUNS_CASE2(S_, A_)                                                                              \
  do { if (noise) { if (nt) hipLaunchKernelGGL((k_step_runs<true, !A_, S_, A_>), g, b, 0, st_, a, rt);     \
                    else hipLaunchKernelGGL((k_step_runs<true, false, S_, A_>), g, b, 0, st_, a, rt); }     \
       else { if (nt) hipLaunchKernelGGL((k_step_runs<false, !A_, S_, A_>), g, b, 0, st_, a, rt);          \
              else hipLaunchKernelGGL((k_step_runs<false, false, S_, A_>), g, b, 0, st_, a, rt); } } while (0)
#define DSIM_RUNS_CASE(S_) do { if (args->action) DSIM_RUNS_CASE2(S_, true); else DSIM_RUNS_CASE2(S_, false); } while (0)
        if (a.substeps == 1) DSIM_RUNS_CASE(true); else DSIM_RUNS_CASE(false);
#undef DSIM_RUNS_CASE
#undef DSIM_RUNS_CASE2
      }
      if (any_hexa) fb_finish(ctx, a, st_);
      bin_next_commit(ctx, n, args, a);
      return (int)hipGetLastError();
    }
#define DSIM_RUN_CASE2(H_, S_)                                                                        \
  /* (the quad law on six actuators — hexa_6DOF_simple.urdf — with the default cache policy only: four streaming instances less) */ \
  do { constexpr bool T_ok = (H_) != DSIM_DEV_KIND_HEXA_QUADLAW;                                                                   \
       if (noise) { if (nt) hipLaunchKernelGGL((k_step_run<H_, true, T_ok, S_>), g, b, 0, st_, a);    \
                    else hipLaunchKernelGGL((k_step_run<H_, true, false, S_>), g, b, 0, st_, a); }    \
       else { if (nt) hipLaunchKernelGGL((k_step_run<H_, false, T_ok, S_>), g, b, 0, st_, a);         \
              else hipLaunchKernelGGL((k_step_run<H_, false, false, S_>), g, b, 0, st_, a); } } while (0)
#define DSIM_RUN_CASE(H_) do { if (a.substeps == 1) DSIM_RUN_CASE2(H_, true); else DSIM_RUN_CASE2(H_, false); } while (0)
    for (int r = 0; r < n_runs; ++r) {
      const dsim_type_run& run = runs[r];
      if (run.count == 0) continue;
      // the launch covers whole 256-drone tiles from the one that holds the run's first drone; lanes outside
      // [lo, last) retire, so two runs may share a tile (each launch takes its own lanes of it)
      a.first = run.first & ~255LL; a.lo = run.first; a.last = run.first + run.count; a.run_type = run.type;
      const dim3 g(grid_for(a.last - a.first));
      const int kind = ctx->h_types[run.type].kind;
      if (kind == DSIM_KIND_HEXA6DOF) DSIM_RUN_CASE(DSIM_DEV_KIND_HEXA);
      else if (kind == DSIM_KIND_HEXA_QUADLAW) DSIM_RUN_CASE(DSIM_DEV_KIND_HEXA_QUADLAW);
      else DSIM_RUN_CASE(DSIM_DEV_KIND_QUAD);
    }
#undef DSIM_RUN_CASE
#undef DSIM_RUN_CASE2
    if (any_hexa) fb_finish(ctx, a, st_);
    bin_next_commit(ctx, n, args, a);
    return (int)hipGetLastError();
  }
  const bool multi = a.wp_table != nullptr || a.n_steps > 1;
  if (uni && !six && !(args->action && multi) && !args->noise_replay && !args->ext_force && !phys_opts) {      // (fine_slow is a phys_opt: a fine launch that comes here has ONE sub-step)
    // fast path over the whole 256-drone tiles (an explicit action: the ACT instances of the plain form)
    const bool nt = stream_policy(args, state.n_pad, 232.0);
    const long long tiles = a.n_pad / 256;
    if (tiles > 0) {
      const dim3 g((unsigned)tiles);
      const bool ext = multi;
      const bool ch = (args->options & DSIM_OPT_CHAINED) != 0;
#define DSIM_FAST_CASE(N_, T_)                                                                      \
  do { if (ext) { if (ch) hipLaunchKernelGGL((k_step_fast<N_, T_, true, true>), g, b, 0, st_, a);   \
                  else hipLaunchKernelGGL((k_step_fast<N_, T_, true, false>), g, b, 0, st_, a); }   \
       else { if (args->action) { if (a.substeps == 1) hipLaunchKernelGGL((k_step_fast<N_, false, false, false, 1, true>), g, b, 0, st_, a); \
                                  else hipLaunchKernelGGL((k_step_fast<N_, false, false, false, 0, true>), g, b, 0, st_, a); } \
              else if (ch && a.substeps == 1) hipLaunchKernelGGL((k_step_fast<N_, T_, false, true, 1>), g, b, 0, st_, a); \
              else if (ch) hipLaunchKernelGGL((k_step_fast<N_, T_, false, true>), g, b, 0, st_, a); \
              else if (a.substeps == 1) hipLaunchKernelGGL((k_step_fast<N_, T_, false, false, 1>), g, b, 0, st_, a); \
              else hipLaunchKernelGGL((k_step_fast<N_, T_, false, false>), g, b, 0, st_, a); } } while (0)
      if (noise) { if (nt) DSIM_FAST_CASE(true, true); else DSIM_FAST_CASE(true, false); }
      else { if (nt) DSIM_FAST_CASE(false, true); else DSIM_FAST_CASE(false, false); }
#undef DSIM_FAST_CASE
      first = tiles * 256;
    }
  }
  bool fb_open = false;
  if (uni && six && ctx->h_types[0].kind == DSIM_KIND_HEXA6DOF && !args->noise_replay && !args->ext_force &&
      !a.wp_table && a.n_steps == 1 && a.n_pad >= 256 && !phys_opts) {
    const long long tiles = a.n_pad / 256;
    const bool nt = stream_policy(args, state.n_pad, 248.0);
    rc = fb_prepare(ctx, a.n_pad, st_);
    if (rc) return rc;
    a.fb.entries = ctx->d_fb;
    fb_open = true;
    const dim3 g((unsigned)tiles);
#define DSIM_HEXA_CASE2(S_, A_)                                                                     \
  do { if (noise) { if (nt) hipLaunchKernelGGL((k_step_hexa<true, !A_, S_, A_>), g, b, 0, st_, a);  \
                    else hipLaunchKernelGGL((k_step_hexa<true, false, S_, A_>), g, b, 0, st_, a); }  \
       else { if (nt) hipLaunchKernelGGL((k_step_hexa<false, !A_, S_, A_>), g, b, 0, st_, a);       \
              else hipLaunchKernelGGL((k_step_hexa<false, false, S_, A_>), g, b, 0, st_, a); } } while (0)
#define DSIM_HEXA_CASE(S_) do { if (args->action) DSIM_HEXA_CASE2(S_, true); else DSIM_HEXA_CASE2(S_, false); } while (0)
    if (a.substeps == 1) DSIM_HEXA_CASE(true); else DSIM_HEXA_CASE(false);
#undef DSIM_HEXA_CASE
#undef DSIM_HEXA_CASE2
    first = tiles * 256;
    if (first >= a.n_pad) fb_finish(ctx, a, st_);
  }
  if (first < a.n_pad)     // ragged tail, or everything when the fast path does not apply (dsim_step_mixed.hip)
    return step_general(ctx, n, state, targets, args, a, first, fb_open, st_);
  return (int)hipGetLastError();
}

int dsim_materialize(dsim_ctx* ctx, void* stream, int64_t n, dsim_view state) {
  if (!ctx || n <= 0 || n > state.n_pad) return DSIM_E_ARG;
  MatK a;
  int rc = make_kview(state, 20 + ctx->max_act, &a.st);
  if (rc) return rc;
  a.n_pad = state.n_pad;
  hipLaunchKernelGGL(k_materialize, dim3(grid_for(a.n_pad)), dim3(256), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

}  // extern "C"
