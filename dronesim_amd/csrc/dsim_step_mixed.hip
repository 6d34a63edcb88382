// dsim_step_mixed.hip — the part of dsim_step that no single-type fast path serves: the general kernels (ragged tails, options,
// replayed noise, more than four types per lane) and the LDS-staged kernels of mixed fleets kept in the caller's own order
// (gfx950 only).
#include "dsim_kernels.h"

// FULL = false: the lean form for plain stepping of mixed fleets (stored cmd as the action, no
// noise replay, no waypoint table, one Env.step per launch) — the options cost registers.
template <bool NOISE, int NACT, bool FULL, bool PLANE = false, class DT>
__device__ __forceinline__ void step_gen_body(DT& T, const StepK& a, long long i, const Addr& ad) {
  Rigid s;
  CtrlMem<NACT> m;
  Target tg;
  load_rigid(ad.sb, ad.sfs, ad.sl, s);
  load_mem<NACT>(ad.sb, ad.sfs, ad.sl, m);
  int wp = 0;
  if (FULL && a.wp_table) wp = a.wp_counter[i]; else load_target(ad.tb, ad.tfs, ad.tl, tg);
  V3 ext = v3(0, 0, 0);
  if (a.ext_force) ext = v3(a.ext_force[i], a.ext_force[a.n_pad + i], a.ext_force[2 * a.n_pad + i]);
  const int n_steps = FULL ? a.n_steps : 1;
  for (int k = 0; k < n_steps; ++k) {
    float act[NACT];
#pragma unroll
    for (int j = 0; j < NACT; ++j) act[j] = m.cmd[j];
    if (FULL && a.action && k == 0) {    // an explicit action applies to the first Env.step only
#pragma unroll
      for (int j = 0; j < NACT; ++j) act[j] = a.action[(long long)j * a.n_pad + i];
      preprocess_action<NACT>(T, act, act);   // the stored cmd is already clipped (INDIControl.py:487)
    }
    if (FULL && a.wp_table) waypoint_target(a, i, wp, tg);
    V3 pos_e;
    float yaw_e;
    if (NACT == 6 && T.kind != DSIM_DEV_KIND_QUAD) {     // wave-uniform branch: morphing-hexa physics (both hexa kinds)
      if constexpr (NACT == 6) {
        hexa_substeps<NOISE, FULL, false, PLANE>(T, a, i, s, act, a.step_index + k, ext, NOISE ? noise_id(a, i) : -1LL);
        ground_watch(T, s, a.fb.counters, i < a.n);
        if (T.kind == DSIM_DEV_KIND_HEXA) indi_hexa<false>(T, a.dt_ctrl, s, tg, m, pos_e, yaw_e, a.fb, i);
        else indi_quad<false, 6>(T, a.dt_ctrl, s, tg, m, pos_e, yaw_e);         // hexa_6DOF_simple: the quad law on six actuators
      }
    } else {
      quad_substeps<NOISE ? (FULL ? 2 : 1) : 0, NACT, FULL, 0, PLANE>(T, a, i, s, act, a.step_index + k, ext, nullptr,
                                                                      NOISE ? noise_id(a, i) : -1LL);
      ground_watch(T, s, a.fb.counters, i < a.n);
      indi_quad<false, NACT>(T, a.dt_ctrl, s, tg, m, pos_e, yaw_e);
    }
    wp = waypoint_next(wp, a.n_wp);
  }
  if (FULL && a.wp_table) a.wp_counter[i] = wp;
  store_rigid(ad.sb, ad.sfs, ad.sl, s);
  store_mem<NACT>(ad.sb, ad.sfs, ad.sl, m);
}
// The full-option body with the in-kernel noise holds both laws, the replay and waypoint paths and the add-on terms:
// compiled for 2 waves/SIMD (256 VGPRs) it spills 200-380 B of scratch per lane, and the scratch traffic (2.8 x the
// state's bytes) costs more than the lost occupancy; those instances take the whole register file instead.
template <bool NOISE, bool UNIFORM, int NACT>
__global__ __launch_bounds__(256, NOISE ? 1 : DSIM_GEN_WAVES) void k_step_gen(StepK a) {
  const long long i0 = a.first + (long long)blockIdx.x * 256;
  const unsigned p = tile_slot<UNIFORM>(a.type_id, i0, a.n_pad);
  const long long i = i0 + p;
  if (i >= a.n_pad) return;
  if (a.step_index_dev) a.step_index += *a.step_index_dev;
  const Addr ad = make_addr(a, i0, p);
  DSIM_FOR_MY_TYPE(UNIFORM, a, i, (step_gen_body<NOISE, NACT, true>(T, a, i, ad)));
}
// DSIM_OPT_PLANE: the full-option body with the ground-plane contact solve between the velocity and the position
// update of every sub-step (dsim_device.h:plane_contact).  A landing / take-off configuration, not a flight one:
// these instances take the whole register file rather than spill.
template <bool NOISE, bool UNIFORM, int NACT>
__global__ __launch_bounds__(256, 1) void k_step_plane(StepK a) {
  const long long i0 = a.first + (long long)blockIdx.x * 256;
  const unsigned p = tile_slot<UNIFORM>(a.type_id, i0, a.n_pad);
  const long long i = i0 + p;
  if (i >= a.n_pad) return;
  if (a.step_index_dev) a.step_index += *a.step_index_dev;
  const Addr ad = make_addr(a, i0, p);
  DSIM_FOR_MY_TYPE(UNIFORM, a, i, (step_gen_body<NOISE, NACT, true, true>(T, a, i, ad)));
}
template <bool NOISE, bool UNIFORM, int NACT>
__global__ __launch_bounds__(256, DSIM_GEN_WAVES) void k_step_lean(StepK a) {
  const long long i0 = a.first + (long long)blockIdx.x * 256;
  const unsigned p = tile_slot<UNIFORM>(a.type_id, i0, a.n_pad);
  const long long i = i0 + p;
  if (i >= a.n_pad) return;
  if (a.step_index_dev) a.step_index += *a.step_index_dev;
  const Addr ad = make_addr(a, i0, p);
  DSIM_FOR_MY_TYPE(UNIFORM, a, i, (step_gen_body<NOISE, NACT, false>(T, a, i, ad)));
}

// ---- mixed fleets kept in the caller's own order (storage = "caller") ----------------------------------------------------------
// A tile is partitioned by type so that every wave runs ONE law in uniform control flow, and staged through LDS so that HBM
// only ever sees whole lines.  Two forms serve the product: k_step_mixed4 (wave-tiled layout: two waves per tile, LDS-DMA
// staging) and k_step_mixed3 (any other layout: row DMAs).  Round 1's VGPR-staged form and round 2's persistent LDS-DMA ring
// were measured slower and live in the git history (tools/variants/ up to round 5).
// select-the-r-th-set-bit: lane r of a compute wave finds the r-th drone of its type in the tile's ballot masks
__device__ __forceinline__ unsigned nth_set_bit64(unsigned long long m, unsigned r) {     // position of the r-th (0-based) set bit
  unsigned pos = 0;
  unsigned w = (unsigned)m;
  unsigned c = (unsigned)__popc(w);
  if (r >= c) { r -= c; pos = 32; w = (unsigned)(m >> 32); }
#pragma unroll
  for (int sh = 16; sh >= 1; sh >>= 1) {
    const unsigned lo = w & ((1u << sh) - 1u);
    c = (unsigned)__popc(lo);
    if (r >= c) { r -= c; pos += sh; w >>= sh; } else { w = lo; }
  }
  return pos;
}
// LDS image of one 64-drone block: the block's rows as they lie in the wave-tiled state / target arrays
// ([F][64] floats, field rows contiguous), so that a 16-byte-per-lane DMA moves four rows at once.  Both row groups
// are padded to a multiple of four rows: the last DMA of each group (rows 24-25 / 8-9) runs with ALL lanes active, its
// upper half re-reading the same two rows into the padding.  (An exec-masked DMA under `if (lane < 32)` is a hazard:
// the LDS destination of an LDS-DMA is wave-uniform (M0), and the compiler's tail merging of the two sides of such a
// branch produced ONE instruction with a per-lane "uniform" destination resolved by v_readfirstlane — half the wave's
// rows landed in the wrong place.  No DMA in this file sits under a per-lane branch.)
struct Stage64 { float st[DSIM_NF_HEXA + 2][64]; float tg[DSIM_NT + 2][64]; };       // 28 + 12 rows = 10 KB
// the 7 + 3 DMAs of 1 KB that bring one 64-drone block (26 state rows, 10 target rows) into a Stage64
template <int AUX>
__device__ __forceinline__ void dma_block64(const float* state_block, const float* target_block, Stage64& dst, unsigned lane) {
  const float* sp = state_block + 4 * lane;            // 16 bytes per lane
  const float* tp = target_block + 4 * lane;
  const unsigned fold = 4 * (lane & 31u);              // last DMA of a group: lanes 32..63 re-read what lanes 0..31 read
  float* ls = &dst.st[0][0];
  float* lt = &dst.tg[0][0];
#pragma unroll
  for (int q = 0; q < 6; ++q) __builtin_amdgcn_global_load_lds(sp + 256 * q, ls + 256 * q, 16, 0, AUX);
  __builtin_amdgcn_global_load_lds(state_block + 256 * 6 + fold, ls + 256 * 6, 16, 0, AUX);        // rows 24, 25 (+ padding)
  __builtin_amdgcn_global_load_lds(tp, lt, 16, 0, AUX);
  __builtin_amdgcn_global_load_lds(tp + 256, lt + 256, 16, 0, AUX);
  __builtin_amdgcn_global_load_lds(target_block + 512 + fold, lt + 512, 16, 0, AUX);                // rows 8, 9 (+ padding)
}
template <bool HEXA, bool NOISE, bool S1, class DT>
__device__ __forceinline__ void staged_body2(DT& T, const StepK& a, long long i, Stage64* tile, unsigned d,
                                             bool active) {
  constexpr int NA = HEXA ? 6 : 4;
  float (*st)[64] = tile[d >> 6].st;
  float (*tt)[64] = tile[d >> 6].tg;
  const unsigned c = d & 63u;
  Rigid s;
  CtrlMem<NA> m;
  Target tg;
  s.pos = v3(st[0][c], st[1][c], st[2][c]);
  s.q = Q4{st[3][c], st[4][c], st[5][c], st[6][c]};
  s.vel = v3(st[7][c], st[8][c], st[9][c]);
  s.w = v3(st[10][c], st[11][c], st[12][c]);
  m.last_vel = v3(st[13][c], st[14][c], st[15][c]);
  m.last_rates = v3(st[16][c], st[17][c], st[18][c]);
  m.last_thrust = st[19][c];
#pragma unroll
  for (int j = 0; j < NA; ++j) m.cmd[j] = st[20 + j][c];
  tg.pos = v3(tt[0][c], tt[1][c], tt[2][c]);
  tg.vel = v3(tt[3][c], tt[4][c], tt[5][c]);
  tg.acc = v3(tt[6][c], tt[7][c], tt[8][c]);
  tg.yaw = tt[9][c];
  V3 ext = v3(0, 0, 0);
  if (a.ext_force) ext = v3(a.ext_force[i], a.ext_force[a.n_pad + i], a.ext_force[2 * a.n_pad + i]);
  V3 pos_e;
  float yaw_e;
  if constexpr (HEXA) {
    hexa_substeps<NOISE, false, S1, false, !S1>(T, a, i, s, m.cmd, a.step_index, ext);
    indi_hexa<false>(T, a.dt_ctrl, s, tg, m, pos_e, yaw_e, a.fb, active ? i : -1LL);
  } else {
    quad_substeps<NOISE ? 1 : 0, 4, false, S1 ? 1 : 0, false, -1, !S1>(T, a, i, s, m.cmd, a.step_index, ext);
    indi_quad<false>(T, a.dt_ctrl, s, tg, m, pos_e, yaw_e);
  }
  ground_watch(T, s, a.fb.counters, active && i < a.n);      // (behind the law: in front of it it costs registers)
  if (!active) return;
  st[0][c] = s.pos.x; st[1][c] = s.pos.y; st[2][c] = s.pos.z;
  st[3][c] = s.q.x; st[4][c] = s.q.y; st[5][c] = s.q.z; st[6][c] = s.q.w;
  st[7][c] = s.vel.x; st[8][c] = s.vel.y; st[9][c] = s.vel.z;
  st[10][c] = s.w.x; st[11][c] = s.w.y; st[12][c] = s.w.z;
  st[13][c] = m.last_vel.x; st[14][c] = m.last_vel.y; st[15][c] = m.last_vel.z;
  st[16][c] = m.last_rates.x; st[17][c] = m.last_rates.y; st[18][c] = m.last_rates.z;
  st[19][c] = m.last_thrust;
#pragma unroll
  for (int j = 0; j < NA; ++j) st[20 + j][c] = m.cmd[j];
}
#define DSIM_MIXED2_TYPES 4            // the launcher takes this form for tables of up to four types
// ---- mixed fleets, third form: one tile per workgroup, LDS-DMA staging, partition by ballots ---------------------------
// The ring above keeps a tile per workgroup in flight at all times, but its 37 KB of LDS leave a CU only 12 waves, and
// with two barriers per tile three waves per SIMD cannot keep the vector pipe busy: it measured SLOWER (227 us) than the
// first form (211 us) at 4 194 304 drones.  What the first form lacks is waves, not prefetch depth: this form keeps its
// one-tile-per-workgroup shape (the hardware overlaps workgroups) and removes what limits their number and speed —
//   * staging in NATURAL drone order by LDS-DMA (no VGPR round trip, no staging ds_writes): 18 KB instead of 27.6 KB
//     per workgroup, so a CU holds 8 of them instead of 5;
//   * the partition by type needs no LDS table and no barrier (every wave ballots the tile's type ids itself and finds
//     its drones by select-the-r-th-set-bit, as in the ring): two barriers per tile instead of three.
// TILED: wave-tiled layout (rows of a block contiguous) -> 10 DMAs of 1 KB per half; otherwise 36 row DMAs of 256 B.
template <bool NOISE, bool NT, int WT, bool S1, bool TILED>
__global__ __launch_bounds__(64 * WT, S1 ? 4 : 3) void k_step_mixed3(StepK a) {
  constexpr int TILE = 128;
  __shared__ __attribute__((aligned(16))) Stage64 tile[2];                  // [half]: 20 KB
  const unsigned t = threadIdx.x, w = t >> 6, lane = t & 63;
  const long long i0 = a.first + (long long)blockIdx.x * TILE;
  if (NOISE && a.step_index_dev) a.step_index += *a.step_index_dev;
  constexpr int AUX = NT ? 2 : 0;
  const long long ih = i0 + 64 * (long long)w;
  if (w < 2 && ih < a.n_pad) {                                              // each natural wave brings its own half in
    if (TILED) {
      dma_block64<AUX>(a.st.base + (ih >> 6) * a.st.block_stride, a.tg.base + (ih >> 6) * a.tg.block_stride, tile[w], lane);
    } else {
      const long long il = ih + lane;
      const float* sp = a.st.base + kv_off(a.st, il);
      const float* tp = a.tg.base + kv_off(a.tg, il);                       // (a broadcast row: kv_off = 0 for every lane)
      const long long sfs = a.st.field_stride, tfs = a.tg.field_stride;
      // (all 26 rows for every lane — quads' rows 24, 25 are unused words of the 26-field state: no DMA under a
      // per-lane branch, see Stage64)
#pragma unroll
      for (int f = 0; f < 26; ++f) __builtin_amdgcn_global_load_lds(sp + f * sfs, &tile[w].st[f][0], 4, 0, AUX);
#pragma unroll
      for (int f = 0; f < 10; ++f) __builtin_amdgcn_global_load_lds(tp + f * tfs, &tile[w].tg[f][0], 4, 0, AUX);
    }
  }
  // ---- partition (overlaps the DMAs): every wave ballots both halves itself; the masks are wave-uniform (SGPRs)
  const int t0 = (i0 + lane < a.n_pad) ? min((int)a.type_id[i0 + lane], DSIM_MAX_TYPES - 1) : DSIM_MAX_TYPES;
  const int t1 = (i0 + 64 + lane < a.n_pad) ? min((int)a.type_id[i0 + 64 + lane], DSIM_MAX_TYPES - 1) : DSIM_MAX_TYPES;
  int wave_t = -1;
  unsigned d = 0;
  bool active = false;
  unsigned acc_w = 0;
#pragma unroll
  for (int ty = 0; ty < DSIM_MIXED2_TYPES; ++ty) {
    const unsigned long long m0 = __ballot(t0 == ty), m1 = __ballot(t1 == ty);
    const unsigned c0 = (unsigned)__popcll(m0), tot = c0 + (unsigned)__popcll(m1), nw = (tot + 63) >> 6;
    if (w >= acc_w && w < acc_w + nw) {                             // wave-uniform: this wave runs type ty
      wave_t = ty;
      const unsigned r = (w - acc_w) * 64 + lane;
      active = r < tot;
      const unsigned rr = active ? r : 0u;
      d = rr < c0 ? nth_set_bit64(m0, rr) : 64u + nth_set_bit64(m1, rr - c0);
    }
    acc_w += nw;
  }
  wave_t = __builtin_amdgcn_readfirstlane(wave_t);
  __builtin_amdgcn_s_waitcnt(0x0f70);                               // vmcnt(0): this wave's DMAs have landed
  __syncthreads();
  if (wave_t >= 0) {
    const long long i = i0 + d;
    CDevType& T = dev_type(a.types, wave_t);     // (constant address space, dsim_device.h: 122-156 -> 97-102 VGPRs, 238 -> 210 us)
    if (T.kind == DSIM_DEV_KIND_HEXA) staged_body2<true, NOISE, S1>(T, a, i, tile, d, active);
    else staged_body2<false, NOISE, S1>(T, a, i, tile, d, active);
  }
  __syncthreads();
  if (t < TILE && i0 + t < a.n_pad) {
    const bool nat_hexa = (a.hexa_types >> min((int)a.type_id[i0 + t], DSIM_MAX_TYPES - 1)) & 1u;   // (re-read: not kept live)
    float* sp = a.st.base + kv_off(a.st, i0 + t);
    const long long sfs = a.st.field_stride;
    float (*rows)[64] = tile[w].st;
#pragma unroll
    for (int f = 0; f < 24; ++f) stg<NT>(sp + f * sfs, 0u, rows[f][lane]);
    if (nat_hexa) { stg<NT>(sp + 24 * sfs, 0u, rows[24][lane]); stg<NT>(sp + 25 * sfs, 0u, rows[25][lane]); }
    if (a.bin.count && i0 + t < a.n)
      bin_entry(a.bin, rows[0][lane], rows[1][lane], rows[2][lane], a.bin.local_offset + i0 + t);
  }
}

// ---- mixed fleets, fourth form: TWO waves per 128-drone tile ------------------------------------------------------------
// Counters of the third form at 4 194 304 drones (profiles/r02_mixed_summary.json): waves parked 70 % of their cycles,
// vector ALU 16 % — a latency-bound kernel, and what bounds it is the number of drones a CU has in flight: 5 workgroups
// x 128 drones against the 28 waves x 64 drones of the single-type kernels.  A third of the form's waves are the
// SPARE waves, which exist so that every type can start at a wave boundary and which, in a 64 / 64 tile, do nothing but
// hold a wave slot and its registers for the workgroup's lifetime.  Here a workgroup is the two natural waves only; the
// slot groups (whole waves of one type, as before) are dealt to them round-robin, so a tile that needs a third group
// (65 + 63, or three types) costs one of its waves a second pass instead of costing EVERY tile a third wave.  With
// the unpadded LDS image (18.4 KB) a CU holds 8 workgroups = 1 024 drones.
// (Measured and rejected, round 2: NO staging — the slot permutation applied to the lane offset of the single-type
// kernels' addressing, so that no LDS image bounds the drones in flight and no barrier sits in a workgroup's lifetime.
// A slot group's lanes then use every other dword of four 128-byte lines per instruction, and every line is requested
// by both waves of the tile: 421 us with streaming accesses (partial-line writes), 278 us with the default policy,
// against 199 us for this form — the staging buys whole-line traffic, which is worth more than the occupancy.  Its
// other lesson is kept: two inlined laws behind one branch need ~113 VGPRs where each alone needs 72-75, and a loop
// around them makes the compiler hoist all 36 field addresses into SGPR pairs until the scalar file spills.)
struct Stage64u { float st[DSIM_NF_HEXA][64]; float tg[DSIM_NT][64]; };      // 26 + 10 rows, no padding: 9 KB
template <int AUX>
__device__ __forceinline__ void dma_block64u(const float* state_block, const float* target_block, Stage64u& dst, unsigned lane) {
  const float* sp = state_block + 4 * lane;            // 16 bytes per lane: four rows per DMA
  const float* tp = target_block + 4 * lane;
  float* ls = &dst.st[0][0];
  float* lt = &dst.tg[0][0];
#pragma unroll
  for (int q = 0; q < 6; ++q) __builtin_amdgcn_global_load_lds(sp + 256 * q, ls + 256 * q, 16, 0, AUX);
  __builtin_amdgcn_global_load_lds(state_block + 24 * 64 + lane, ls + 24 * 64, 4, 0, AUX);          // rows 24, 25: one row each
  __builtin_amdgcn_global_load_lds(state_block + 25 * 64 + lane, ls + 25 * 64, 4, 0, AUX);
  __builtin_amdgcn_global_load_lds(tp, lt, 16, 0, AUX);
  __builtin_amdgcn_global_load_lds(tp + 256, lt + 256, 16, 0, AUX);
  __builtin_amdgcn_global_load_lds(target_block + 8 * 64 + lane, lt + 8 * 64, 4, 0, AUX);           // rows 8, 9
  __builtin_amdgcn_global_load_lds(target_block + 9 * 64 + lane, lt + 9 * 64, 4, 0, AUX);
}
template <bool HEXA, bool NOISE, bool S1, bool BIN, class DT>
__device__ __forceinline__ void staged_body4(DT& T, const StepK& a, long long i, Stage64u* tile, unsigned d,
                                             bool active) {
  constexpr int NA = HEXA ? 6 : 4;
  float (*st)[64] = tile[d >> 6].st;
  float (*tt)[64] = tile[d >> 6].tg;
  unsigned c = d & 63u;
  Rigid s;
  CtrlMem<NA> m;
  Target tg;
  s.pos = v3(st[0][c], st[1][c], st[2][c]);
  s.q = Q4{st[3][c], st[4][c], st[5][c], st[6][c]};
  s.vel = v3(st[7][c], st[8][c], st[9][c]);
  s.w = v3(st[10][c], st[11][c], st[12][c]);
#pragma unroll
  for (int j = 0; j < NA; ++j) m.cmd[j] = st[20 + j][c];
  V3 ext = v3(0, 0, 0);
  if (a.ext_force) ext = v3(a.ext_force[i], a.ext_force[a.n_pad + i], a.ext_force[2 * a.n_pad + i]);
  V3 pos_e;
  float yaw_e;
  if constexpr (HEXA) hexa_substeps<NOISE, false, S1, false, !S1>(T, a, i, s, m.cmd, a.step_index, ext);
  else quad_substeps<NOISE ? 1 : 0, 4, false, S1 ? 1 : 0, false, -1, !S1>(T, a, i, s, m.cmd, a.step_index, ext);
  // what only the law reads — the rest of the controller memory and the targets — comes out of LDS BEHIND the sub-steps
  // (tied to their result): read in front of them it is 17 registers held through the physics
  asm volatile("" : "+v"(c) : "v"(s.pos.z));
  int bcell = 0, bslot = -1;                                  // next step's neighbour grid: reserve the slot now
  const bool binning = BIN && active && i < a.n;              // (BIN instances are launched when a.bin.count is set)
  if (binning) bslot = bin_reserve(a.bin, s.pos.x, s.pos.y, bcell);
  m.last_vel = v3(st[13][c], st[14][c], st[15][c]);
  m.last_rates = v3(st[16][c], st[17][c], st[18][c]);
  m.last_thrust = st[19][c];
  tg.pos = v3(tt[0][c], tt[1][c], tt[2][c]);
  tg.vel = v3(tt[3][c], tt[4][c], tt[5][c]);
  tg.acc = v3(tt[6][c], tt[7][c], tt[8][c]);
  tg.yaw = tt[9][c];
  if constexpr (HEXA) indi_hexa<false>(T, a.dt_ctrl, s, tg, m, pos_e, yaw_e, a.fb, active ? i : -1LL);
  else indi_quad<false>(T, a.dt_ctrl, s, tg, m, pos_e, yaw_e);
  ground_watch(T, s, a.fb.counters, active && i < a.n);      // (behind the law: in front of it it costs registers)
  if (!active) return;
  st[0][c] = s.pos.x; st[1][c] = s.pos.y; st[2][c] = s.pos.z;
  st[3][c] = s.q.x; st[4][c] = s.q.y; st[5][c] = s.q.z; st[6][c] = s.q.w;
  st[7][c] = s.vel.x; st[8][c] = s.vel.y; st[9][c] = s.vel.z;
  st[10][c] = s.w.x; st[11][c] = s.w.y; st[12][c] = s.w.z;
  st[13][c] = m.last_vel.x; st[14][c] = m.last_vel.y; st[15][c] = m.last_vel.z;
  st[16][c] = m.last_rates.x; st[17][c] = m.last_rates.y; st[18][c] = m.last_rates.z;
  st[19][c] = m.last_thrust;
#pragma unroll
  for (int j = 0; j < NA; ++j) st[20 + j][c] = m.cmd[j];
  if (binning) bin_commit(a.bin, bcell, bslot, s.pos.x, s.pos.y, s.pos.z, a.bin.local_offset + i);
}
// wave-tiled layout only (state of 26 fields and per-drone targets, as for the ring); up to DSIM_MIXED2_TYPES types
// NTY = number of types in the table (2..4): the ballot loop and the group bookkeeping are sized for it
template <bool NOISE, bool NT, bool S1, int NTY, bool BIN>
__global__ __launch_bounds__(128, S1 ? 4 : 3) void k_step_mixed4(StepK a) {
  constexpr int TILE = 128;
  __shared__ __attribute__((aligned(16))) Stage64u tile[2];                 // [half]: 18.4 KB
  const unsigned t = threadIdx.x, w = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;     // (w: an SGPR, so is the group loop)
  const long long i0 = a.first + (long long)blockIdx.x * TILE;
  if (NOISE && a.step_index_dev) a.step_index += *a.step_index_dev;
  constexpr int AUX = NT ? 2 : 0;
  const long long ih = i0 + 64 * (long long)w;
  if (ih < a.n_pad)                                                         // each wave brings its own half in
    dma_block64u<AUX>(a.st.base + (ih >> 6) * a.st.block_stride, a.tg.base + (ih >> 6) * a.tg.block_stride, tile[w], lane);
  // ---- partition (overlaps the DMAs): both waves ballot both halves; masks and counts are wave-uniform (SGPRs)
  const int t0 = (i0 + lane < a.n_pad) ? min((int)a.type_id[i0 + lane], DSIM_MAX_TYPES - 1) : DSIM_MAX_TYPES;
  const int t1 = (i0 + 64 + lane < a.n_pad) ? min((int)a.type_id[i0 + 64 + lane], DSIM_MAX_TYPES - 1) : DSIM_MAX_TYPES;
  unsigned long long m0[NTY], m1[NTY];
  unsigned g0[NTY + 1];                                       // first slot group of each type
  unsigned long long hexa_mine = 0;                           // the hexas of this wave's own half (for the store phase)
  g0[0] = 0;
#pragma unroll
  for (int ty = 0; ty < NTY; ++ty) {
    m0[ty] = __ballot(t0 == ty); m1[ty] = __ballot(t1 == ty);
    g0[ty + 1] = g0[ty] + (((unsigned)__popcll(m0[ty]) + (unsigned)__popcll(m1[ty]) + 63u) >> 6);
    if ((a.hexa_types >> ty) & 1u) hexa_mine |= w ? m1[ty] : m0[ty];
  }
  __builtin_amdgcn_s_waitcnt(0x0f70);                               // vmcnt(0): this wave's DMAs have landed
  __syncthreads();
  // ---- the slot groups, dealt round-robin to the two waves: group g of type ty = its drones of rank 64 (g - g0[ty]) ...
  for (unsigned g = w; g < g0[NTY]; g += 2) {
    int ty = 0;
#pragma unroll
    for (int k = 1; k < NTY; ++k) ty += (g >= g0[k]) ? 1 : 0;
    ty = __builtin_amdgcn_readfirstlane(ty);
    unsigned long long ma = 0, mb = 0;
#pragma unroll
    for (int k = 0; k < NTY; ++k) if (k == ty) { ma = m0[k]; mb = m1[k]; }
    const unsigned c0 = (unsigned)__popcll(ma), tot = c0 + (unsigned)__popcll(mb);
    const unsigned r = (g - g0[ty]) * 64 + lane;
    const bool active = r < tot;
    const unsigned rr = active ? r : 0u;
    const unsigned d = rr < c0 ? nth_set_bit64(ma, rr) : 64u + nth_set_bit64(mb, rr - c0);
    const long long i = i0 + d;
    CDevType& T = dev_type(a.types, ty);         // (constant address space, dsim_device.h: 192 -> 182 us)
    if (T.kind == DSIM_DEV_KIND_HEXA) staged_body4<true, NOISE, S1, BIN>(T, a, i, tile, d, active);
    else staged_body4<false, NOISE, S1, BIN>(T, a, i, tile, d, active);
  }
  __syncthreads();
  if (i0 + t < a.n_pad) {
    const bool nat_hexa = (hexa_mine >> lane) & 1ull;           // (from the ballots: no second read of type_id in front of the stores)
    float* sp = a.st.base + ((i0 + t) >> 6) * a.st.block_stride + lane;
    float (*rows)[64] = tile[w].st;
#pragma unroll
    for (int f = 0; f < 24; ++f) stg<NT>(sp + f * 64, 0u, rows[f][lane]);
    if (nat_hexa) { stg<NT>(sp + 24 * 64, 0u, rows[24][lane]); stg<NT>(sp + 25 * 64, 0u, rows[25][lane]); }
  }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
// dsim_step's ragged tail, or the whole fleet when no single-type fast path applies (called by dsim_step, dsim_step.hip): the
// general kernels (k_step_gen / _lean / _plane) and the LDS-staged kernels of mixed fleets kept in the caller's own order.
// first: first drone of this part (a multiple of 256); fb_open: the WLS fallback queue is already prepared for this step.
int step_general(dsim_ctx* ctx, int64_t n, const dsim_view& state, const dsim_view& targets, const dsim_step_args* args, StepK& a,
                 long long first, bool fb_open, hipStream_t st_) {
  int rc = DSIM_OK;
  const bool noise = args->noise_seed != 0 || args->noise_replay != nullptr;
  const bool uni = args->type_id == nullptr;
  const bool six = ctx->max_act == 6;
  const bool fine = noise && !args->noise_replay && (a.options & DSIM_OPT_NOISE_FINE) != 0;
  const bool fine_slow = fine && a.substeps > 1;
  const bool phys_opts = (args->options & (DSIM_OPT_DRAG | DSIM_OPT_GROUND | DSIM_OPT_PLANE)) != 0 || fine_slow;
  const bool plane = (args->options & DSIM_OPT_PLANE) != 0;
  bool any_quadlaw6 = false;
  for (int t = 0; t < ctx->n_types; ++t) any_quadlaw6 |= ctx->h_types[t].kind == DSIM_KIND_HEXA_QUADLAW;
  a.first = first;
  const dim3 g(grid_for(a.n_pad - first));
  const bool lean = !args->action && !args->noise_replay && !a.wp_table && a.n_steps == 1 && !phys_opts;     // (fine_slow is a phys_opt)
  if (lean && !uni && a.tg.base && ctx->n_types <= 4 && ctx->max_act == 6 && !any_quadlaw6) {
    // a heterogeneous fleet kept in the CALLER's own order (CtrlAviary(storage="caller"); storage="auto" stores it
    // type-major and never comes here): the LDS-staged kernels, which partition every tile by type
    const bool nt = stream_policy(args, state.n_pad, 240.0);
    bool any_hexa = false;
    for (int t = 0; t < ctx->n_types; ++t) any_hexa |= ctx->h_types[t].kind == DSIM_KIND_HEXA6DOF;
    if (any_hexa && !fb_open) {
      rc = fb_prepare(ctx, a.n_pad, st_);
      if (rc) return rc;
      a.fb.entries = ctx->d_fb;
    }
    if (first == 0) bin_next_prepare(ctx, n, args, &a, st_);      // (the whole fleet goes through this kernel)
    // LDS-DMA of whole 1 KB row groups needs the wave-tiled layout [n/64][F][64] for the state (26 fields: a table with
    // a morphing hexa) and for per-drone targets
    const bool tiled = state.block == 64 && state.field_stride == 64 &&
                       !(args->options & DSIM_OPT_BCAST_TGT) && targets.block == 64 && targets.field_stride == 64;
    if (tiled) {
      // two waves per tile, slot groups dealt round-robin (wave-tiled layout)
      const dim3 gm((unsigned)((a.n_pad - first + 127) / 128)), bm(128);
#define DSIM_MIXED4_CASE3(S_, Y_, B_)                                                                             \
do { if (noise) { if (nt) hipLaunchKernelGGL((k_step_mixed4<true, true, S_, Y_, B_>), gm, bm, 0, st_, a);            \
                  else hipLaunchKernelGGL((k_step_mixed4<true, false, S_, Y_, B_>), gm, bm, 0, st_, a); }            \
     else { if (nt) hipLaunchKernelGGL((k_step_mixed4<false, true, S_, Y_, B_>), gm, bm, 0, st_, a);                 \
            else hipLaunchKernelGGL((k_step_mixed4<false, false, S_, Y_, B_>), gm, bm, 0, st_, a); } } while (0)
#define DSIM_MIXED4_CASE2(S_, Y_) do { if (a.bin.count) DSIM_MIXED4_CASE3(S_, Y_, true); else DSIM_MIXED4_CASE3(S_, Y_, false); } while (0)
      // (a table of three types runs the four-type instance: an empty type has no ballots set and no slot group — sixteen
      // instances less for a storage order the host avoids by default)
        // (round 6: and so does a table of two — 181.2 against 181.4 us for the two-type instance at 4 194 304 drones, same box:
        // another sixteen instances less.  The three-wave form below does NOT bear it: five waves for two types 300 against 207 us.)
#define DSIM_MIXED4_CASE(S_) DSIM_MIXED4_CASE2(S_, 4)
      if (a.substeps == 1) DSIM_MIXED4_CASE(true); else DSIM_MIXED4_CASE(false);
#undef DSIM_MIXED4_CASE
#undef DSIM_MIXED4_CASE2
#undef DSIM_MIXED4_CASE3
    } else {
      // any other layout: one tile per workgroup, row DMAs in natural order, ballot partition
      const dim3 gm((unsigned)((a.n_pad - first + 127) / 128));
#define DSIM_MIXED3_CASE2(W_, S_)                                                                                  \
do { const dim3 bm(64 * W_);                                                                                    \
     /* (default cache policy only: a fleet kept in the caller's order on a layout that is not wave-tiled is two steps off */ \
     /*  every default — twelve streaming instances less, round 6)                                                        */ \
     if (noise) hipLaunchKernelGGL((k_step_mixed3<true, false, W_, S_, false>), gm, bm, 0, st_, a);              \
     else hipLaunchKernelGGL((k_step_mixed3<false, false, W_, S_, false>), gm, bm, 0, st_, a); } while (0)
#define DSIM_MIXED3_CASE(W_) do { if (a.substeps == 1) DSIM_MIXED3_CASE2(W_, true); else DSIM_MIXED3_CASE2(W_, false); } while (0)
      if (ctx->n_types == 2) DSIM_MIXED3_CASE(3); else if (ctx->n_types == 3) DSIM_MIXED3_CASE(4); else DSIM_MIXED3_CASE(5);
#undef DSIM_MIXED3_CASE
#undef DSIM_MIXED3_CASE2
    }
    if (any_hexa) fb_finish(ctx, a, st_);
    bin_next_commit(ctx, n, args, a);
    return (int)hipGetLastError();
  } else if (!six) {
    if (lean) DSIM_LAUNCH_GEN(k_step_lean, noise, uni, false, g, a, st_);     // (fine_slow is a phys_opt: never lean)
    else if (plane) DSIM_LAUNCH_GEN_ANY(k_step_plane, noise, false, g, a, st_);
    else DSIM_LAUNCH_GEN_ANY(k_step_gen, noise, false, g, a, st_);
  } else {
    // hexa fleets: deferred WLS fallbacks must land before the next Env.step reads cmd, so several
    // steps per call become several launches (each followed by the tiny fallback kernel)
    const int steps = a.n_steps;
    a.n_steps = 1;
    for (int k = 0; k < steps; ++k) {
      if (!fb_open) {                 // (open already when the tiles went through k_step_hexa: one queue, one fallback pass)
        rc = fb_prepare(ctx, a.n_pad, st_);
        if (rc) return rc;
        a.fb.entries = ctx->d_fb;
      }
      fb_open = false;
      if (lean && !a.action) DSIM_LAUNCH_GEN(k_step_lean, noise, uni, true, g, a, st_);
      else if (plane) DSIM_LAUNCH_GEN_ANY(k_step_plane, noise, true, g, a, st_);
      else DSIM_LAUNCH_GEN_ANY(k_step_gen, noise, true, g, a, st_);
      fb_finish(ctx, a, st_);
      a.step_index += 1;
      a.action = nullptr;             // an explicit action applies to the first Env.step only
    }
  }
  return (int)hipGetLastError();
}
