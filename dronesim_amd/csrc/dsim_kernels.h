// dsim_kernels.h — what the translation units of libdronesim_amd.so share (gfx950 only): the context, the blocked-SoA
// addressing, the launch arguments (StepK), the global-access helpers, the physics sub-step loops of one Env.step, the
// type waterfall and the run table of type-major fleets, and the host-side helpers the entry points have in common.
//
//   dsim_api.hip        context, reset, observation, trajectory sampler, WLS fallback, noise draw; the shared host helpers
//   dsim_step.hip       dsim_step: the fused Env.step + computeControl kernels of single-type fleets and type-major runs
//   dsim_step_mixed.hip ... its general kernels and the LDS-staged kernels of mixed fleets in the caller's order
//   dsim_two_call.hip   dsim_physics / dsim_control / dsim_control2 / dsim_step_adaptor, Physics.DYN (k_physics_*, k_control_*,
//                       k_dyn, k_adaptor*)
//   dsim_downwash.hip   neighbour downwash, adjacency, halo exchange of a sharded fleet (k_dw_*, k_halo_pack, ...)
//
// Execution shape: one drone per lane, 64-drone waves, 256-thread workgroups.
// State is blocked SoA (include/dronesim_amd.h): consecutive lanes read
// consecutive floats of one field, so every global access of a wave is one
// fully-coalesced 256-byte segment.  The fused step kernel reads each state
// field once and writes it once per Env.step(): 232 B per drone-step for a quad
// with per-drone targets (192 B with a broadcast target); physics sub-steps and
// the whole INDI law stay in registers.  The bound is HBM bandwidth.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <new>

#include "../../include/dronesim_amd.h"
#include "dsim_device.h"


// minimum waves per SIMD the fused kernel is compiled for (2nd __launch_bounds__ argument):
// bounds the VGPR budget (512 / waves); tuned on MI355X, see DESIGN.md
// (round 3, A/B of 3 / 4 / 5 / 6 on the final build, profiles/r03_ab_waves.txt: indifferent for one sub-step per launch,
// 153.5-154.5 us whatever the bound; with the examples' five sub-steps the looped kernel fits 95 VGPRs at 5 and runs
// 241 us instead of 248 at 4 (110 VGPRs); no instance spills at 5)
#ifndef DSIM_STEP_WAVES
#define DSIM_STEP_WAVES 5
#endif
#ifndef DSIM_GEN_WAVES
#define DSIM_GEN_WAVES 2
#endif
#ifndef DSIM_IO_ROWS_NT
#define DSIM_IO_ROWS_NT 0      // rows scattered to the caller's numbering: streaming hint or not
#endif

struct dsim_ctx {
  int device;
  int n_types;
  int max_act;                            // 4: quads only; 6: the table holds a morphing hexa
  DevType* d_types;                       // device copy of the type table
  unsigned long long* d_counters;         // [0..1] diagnostics (dsim_query), [2] fallback queue length, [3] its ticket
  FbEntry* d_fb;                          // deferred WLS fallback queue, grown to the largest fleet seen
  long long fb_cap;
  const int32_t* dw_ws;                   // downwash grid: workspace / shape / count-buffer parity of the last call
  long long dw_cells;
  int dw_parity;
  int dw_mode;                            // 0: counting sort, 1: cell buckets (which layout the count buffers hold)
  int n_cu;                               // compute units of the device
  bool dw_prebin;                         // the count buffer dw_parity holds the local drones, binned by the last dsim_step
  bool dw_prebin_valid;                   // ... and no call has moved the positions since without re-binning them
  long long dw_prebin_n, dw_prebin_off;
  float dw_prebin_geo[3];
  int dw_prebin_nx, dw_prebin_ny;
  long long dw_local_m;                   // overflow capacity of the local grid in the workspace (layout of what follows it)
  int dw_prebin_kind;                     // what that dsim_step did: 0 binned the drones, 1 refreshed the kept lists' positions (BinK.pbuild)
  const int32_t* dw_keep_ws;              // kept candidate lists (dsim_downwash_args.keep): the buffer / grid / fleet of the last BUILD query, null: none
  long long dw_keep_cells, dw_keep_n;
  float dw_keep_geo[4];                   // xmin, ymin, cell, skin
  int dw_keep_nx, dw_keep_ny;
  long long dw_reuses;                    // DSIM_Q_DW_REUSES
  volatile int* h_keep_fb;                // host memory the device writes into (mapped): [0] overflow length the last finished REUSE query saw, [1] which query
  int* d_keep_fb;                         // ... as the device addresses it (null: no feedback)
  int dwh_parity;                         // halo grid (split-phase downwash): count-buffer parity
  const int32_t* dwh_ws;                  // ... and the workspace / shape it was zeroed for
  long long dwh_cells;
  unsigned* d_bounds;                     // dsim_fleet_bounds: 5 order-preserving keys + a ticket
  int* d_block_map;                       // RunTab.block_map of the last side-by-side launch (DSIM_OPT_CALLER_IO), and what it was made for
  int* h_block_map;
  int block_map_cap, block_map_blocks, block_map_runs;
  dsim_type_run block_map_key[DSIM_MAX_TYPES];
  dsim_type_params h_types[DSIM_MAX_TYPES];
};

// ---------------------------------------------------------------------------
// blocked-SoA addressing
// ---------------------------------------------------------------------------
struct KView {
  float* base;
  long long field_stride, block_stride;
  long long mask;   // block - 1 (block is a power of two) ; -1 for plain SoA
  int shift;        // log2(block) ; 63 for plain SoA
};
__device__ __forceinline__ long long kv_off(const KView& v, long long i) {
  return (i >> v.shift) * v.block_stride + (i & v.mask);
}
// Workgroups are 256 drones starting at a multiple of 256 and block sizes are powers of two,
// so kv_off(i0 + t) = kv_off(i0) + kv_lane(t): a wave-uniform 64-bit part (kept in SGPRs and
// folded into the scalar base of each access) plus a small per-lane 32-bit part (ONE VGPR shared by
// every field).  Without the split every field costs a 64-bit VGPR address pair.
__device__ __forceinline__ unsigned kv_lane(const KView& v, unsigned t) {
  return v.shift >= 8 ? (t & (unsigned)v.mask)   // plain SoA / blocks >= 256: mask keeps t; broadcast view: mask = 0
                      : (t >> v.shift) * (unsigned)v.block_stride + (t & (unsigned)v.mask);
}

// ---- neighbour grid, bucket form (downwash P8 / adjacency) ---------------------------------------------------------
// Uniform xy grid of cells of HALF the 10 m cut-off or more (a receiver scans the 5 x 5 cells around its own: 625 m^2
// for 5 m cells against the 900 m^2 of 3 x 3 cells of 10 m — 30 % fewer candidate pairs, and four times fewer drones
// per cell counter); every cell owns a bucket of DW_CAP entries (x, y, z, world index), entries that find their bucket
// full go to one shared overflow list that every receiver scans too, so results never depend on the capacity.  The
// step kernels can fill the grid for the NEXT Env.step themselves (BinK in StepK): the new position is in registers
// when the state is stored, which removes the binning launch from the step chain.
#define DW_CAP 64
#define DW_CUTOFF 10.0f
// ints behind the per-cell counts of a bucket grid's count buffer: [0] overflow length; [1..4] the cell range that holds
// entries, as maxima so that an all-zero buffer is the neutral element: nx-1-cx_min, cx_max, ny-1-cy_min, cy_max (kept by
// the halo binning only: the halo pass of the query leaves at once where no halo entry can be in reach); [5] kept lists: the
// drones a refresh found more than half the skin from where they were when the lists were made
#define DW_CNT_EXTRA 6
struct BinK {
  int* count;          // [ncells + DW_CNT_EXTRA]: entries per cell, then the extras above.  null = no binning
  float4* buckets;     // [ncells][DW_CAP]
  float4* overflow;    // [m]
  float xmin, ymin, inv_cell;
  int nx, ny;
  long long local_offset;   // world index of local drone 0
  // Kept candidate lists (dsim_downwash_args.keep, "kept candidate lists" below): the NEXT query re-uses the lists of an earlier
  // one, so this step does not bin.  It REFRESHES: the drone's new position goes where its entry was when the lists were made —
  // its slot of `buckets` (pbuild[i].w; -1: it had none) — and a drone that has moved further than the lists' skin from where it
  // was then (pbuild[i]) leaves them: z = -inf in its slot (as a candidate it is above nobody, as a receiver its cell passes it
  // over) and an entry in the overflow list, whose members are candidates wherever they are in reach and are served by the cell
  // they are in NOW.  null = the step bins.
  const float4* pbuild;     // [n_pad] (x, y, z when the lists were made, bucket slot as int bits)
  float skin2;              // skin^2
  long long* drift;         // the skin moves with the fleet: DW_DRIFT below
  int drift_r;              // which refresh since the lists were made this is (0: the first)
  int drift_mask;           // the sample: drones whose index has none of these bits (dw_drift_mask)
};
// ---- a skin that moves with the fleet ---------------------------------------------------------------------------------
// The lists are invariant under a common translation of the fleet: if every drone that stays in them is within the skin of
// (where it was) + u, ANY u, two of them are within twice the skin of their old relative position — which is all the proof of
// the lists' completeness uses.  A formation in flight therefore leaves no skin at all if u follows it.  u is the mean displacement
// of a sample of the fleet (eight to sixteen drones, those inside the skin: integer atomics in 2^-16 m, deterministic — and few,
// because atomics on one address queue up: every 64th drone of a 65 536-drone shard took the step kernel from 10 to 56 us), predicted one
// refresh ahead from the last two (2 M[r-1] - M[r-2], M[-1] = 0: nobody has moved when the lists are made) by the query in
// front of that refresh; the refresh that uses it writes down what it used, and the query that follows reads it there: movers are
// binned and tested for reach at (position - u), because the cells' lists speak of where their drones WERE.  long longs: a ring of
// four sums (x, y, z, count) | [16, 18) what the last refresh used (u as three floats, r) | [18, 22) the prediction for even / odd r.
#define DW_DRIFT_WORDS 22
#define DW_DRIFT_Q 65536.0f
struct Drift3 { float x, y, z; };
static inline int dw_drift_mask(long long n) {          // every 2^k-th drone, 8 .. 16 of them (at least every 64th)
  long long stride = 64;
  while (stride * 16 < n) stride <<= 1;
  return (int)(stride - 1);
}
__device__ __forceinline__ Drift3 drift_predict(const long long* __restrict__ ring, int r) {
  float m[2][3];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const long long* s = ring + 4 * ((r + 3 - k) & 3);          // M[r - 1 - k]
    const long long cnt = r - 1 - k >= 0 ? s[3] : 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) m[k][c] = cnt > 0 ? (float)((double)s[c] / ((double)DW_DRIFT_Q * (double)cnt)) : 0.0f;
  }
  return Drift3{2.0f * m[0][0] - m[1][0], 2.0f * m[0][1] - m[1][1], 2.0f * m[0][2] - m[1][2]};
}
// ---- kept candidate lists ----------------------------------------------------------------------------
// A fleet moves centimetres per Env.step: which candidates a cell's receivers have to look at, and in which height band each lies,
// changes slowly.  A BUILD query (the banded cell-centred query, with its reach and band tests widened by twice the skin) writes, per
// cell, what it worked out — for every slot of the cell's own bucket the receiver's place in height order and its type, for every
// slot of the 25 neighbouring buckets the candidate's place in the tile (band 0, below every receiver, included: a drone that
// arrives later may need it) — and the following REUSE queries (k_dw_query_kept) load the table and the buckets side by side and go
// straight to the pair loops.  The steps in between keep the buckets current (BinK.pbuild above).  Exact for ANY motion: every pair
// is still tested against the cut-off and the height order on current positions; the lists only have to be SUPERSETS, which the
// skin guarantees for drones that stayed within it and the overflow list for those that did not.  A cell's list (int32 words):
#define DW_LHDR 16                        // [0] receivers [1] candidates in the tile [2] groups G [3] flags (1: banded) [4..11] band 1..8 totals
#define DW_LRECV DW_LHDR                  // [DW_CAP] slot of the cell's own bucket -> the receiver's place in height order
#define DW_LRTY (DW_LHDR + DW_CAP)        // [DW_CAP] slot -> type
#define DW_LCAND (DW_LHDR + 2 * DW_CAP)   // banded: uint16 [25][DW_CAP] (neighbour, bucket slot) -> place in the tile (bands G .. 1, then 0), 0xFFFF: not in
                                          //   reach.  The positions themselves stay in the grid's BUCKETS, refreshed in place by the steps: what a
                                          //   REUSE query loads depends on nothing it has to wait for.  Unbanded (crowded neighbourhoods): [<= DW_LCAP]
#define DW_LCAP (25 * DW_CAP)             //   bucket slots of every entry of the 5 x 5 cells, any order
#define DW_LSTRIDE (DW_LCAND + DW_LCAP)
#define DW_MOV_TILE 32                    // movers (overflow entries) in reach of a cell that a REUSE workgroup keeps in its tile
struct KeepK {
  int* lists;               // [ncells][DW_LSTRIDE]
  float4* pbuild;           // [n_pad] positions when the lists were made (NaN: never in a list), .w = the drone's bucket slot
  float skin;
  long long* drift;               // [DW_DRIFT_WORDS] the moving skin's sums and what the last refresh used (BinK.drift)
  unsigned long long* counters;   // the ctx's diagnostics ([5]: DSIM_Q_DW_MOVERS)
  int* feedback;                  // nullable, host-mapped: [0] = the overflow length this REUSE query finds, [1] = seq (dsim_downwash_keep_stats)
  int seq;
};
__device__ __forceinline__ void bin_refresh(const BinK& b, float x, float y, float z, long long world_index) {
  const long long i = world_index - b.local_offset;
  const float4 pb = b.pbuild[i];
  Drift3 u = Drift3{0.0f, 0.0f, 0.0f};
  if (b.drift) {
    const float* __restrict__ const pu = reinterpret_cast<const float*>(b.drift + 18 + 2 * (b.drift_r & 1));
    u = Drift3{pu[0], pu[1], pu[2]};
  }
  const float ax = x - pb.x, ay = y - pb.y, az = z - pb.z;             // how far it has come since the lists were made
  const float dx = ax - u.x, dy = ay - u.y, dz = az - u.z;             // ... and how far from where the fleet's drift would have it
  const float d2 = dx * dx + dy * dy + dz * dz;
  const bool stay = d2 <= b.skin2;                                     // (NaN anywhere: a mover)
  if (b.drift) {
    if (stay && (i & b.drift_mask) == 0) {                             // the sample the NEXT refreshes' drift is made of
      long long* s = b.drift + 4 * (b.drift_r & 3);
      atomicAdd((unsigned long long*)&s[0], (unsigned long long)(long long)__float2ll_rn(ax * DW_DRIFT_Q));
      atomicAdd((unsigned long long*)&s[1], (unsigned long long)(long long)__float2ll_rn(ay * DW_DRIFT_Q));
      atomicAdd((unsigned long long*)&s[2], (unsigned long long)(long long)__float2ll_rn(az * DW_DRIFT_Q));
      atomicAdd((unsigned long long*)&s[3], 1ULL);
    }
    if ((i & b.drift_mask) == 0 || (blockIdx.x == 0 && threadIdx.x == 0)) {   // what this refresh used: the query that follows reads it here (a few threads, the same words)
      float* h = reinterpret_cast<float*>(b.drift + 16);
      h[0] = u.x; h[1] = u.y; h[2] = u.z;
      reinterpret_cast<int*>(b.drift + 16)[3] = b.drift_r;
    }
  }
  // how many are HALF WAY out: what a host that paces the BUILDs reads before a fleet on the march leaves the skin together
  // (dsim_downwash_keep_stats; one atomic per wave that has any)
  const unsigned long long half = __ballot(!(d2 <= 0.25f * b.skin2)), me = 1ULL << __lane_id();
  if ((half & me) && !(half & (me - 1ULL))) atomicAdd(&b.count[b.nx * b.ny + 5], (int)__popcll(half));
  const int home = __float_as_int(pb.w);
  if (home >= 0) b.buckets[home] = make_float4(x, y, stay ? z : -__builtin_inff(), __int_as_float((int)world_index));
  if (!stay) b.overflow[atomicAdd(&b.count[b.nx * b.ny], 1)] = make_float4(x, y, z, __int_as_float((int)world_index));
}
__device__ __forceinline__ int bin_cell(const BinK& b, float x, float y) {
  const int cx = min(max((int)floorf((x - b.xmin) * b.inv_cell), 0), b.nx - 1);
  const int cy = min(max((int)floorf((y - b.ymin) * b.inv_cell), 0), b.ny - 1);
  return cy * b.nx + cx;
}
// the two halves of bin_entry: the slot's reservation is an atomic round trip to another XCD's L2 (~2 us); issued as soon
// as the new position exists it is hidden behind the control law instead of standing at the end of the workgroup
__device__ __forceinline__ int bin_reserve(const BinK& b, float x, float y, int& cell) {
  cell = 0;
  if (b.pbuild) return 0;
  cell = bin_cell(b, x, y);
  return atomicAdd(&b.count[cell], 1);
}
__device__ __forceinline__ void bin_commit(const BinK& b, int cell, int slot, float x, float y, float z, long long world_index) {
  if (b.pbuild) { bin_refresh(b, x, y, z, world_index); return; }
  const float4 e = make_float4(x, y, z, __int_as_float((int)world_index));
  if (slot < DW_CAP) b.buckets[(long long)cell * DW_CAP + slot] = e;
  else b.overflow[atomicAdd(&b.count[b.nx * b.ny], 1)] = e;
}
// (REFRESH = false: an instance that cannot afford the second form — the host then never asks it to, bin_next_prepare)
template <bool REFRESH = true>
__device__ __forceinline__ void bin_entry(const BinK& b, float x, float y, float z, long long world_index) {
  if (REFRESH && b.pbuild) { bin_refresh(b, x, y, z, world_index); return; }
  const int c = bin_cell(b, x, y);
  const float4 e = make_float4(x, y, z, __int_as_float((int)world_index));
  const int slot = atomicAdd(&b.count[c], 1);
  if (slot < DW_CAP) b.buckets[(long long)c * DW_CAP + slot] = e;
  else b.overflow[atomicAdd(&b.count[b.nx * b.ny], 1)] = e;
}

// bucket form: grids of up to 65 536 cells with at most 5/8 DW_CAP = 40 entries per cell on average (BASELINE config 5:
// one drone per m^2 = 25 per 5 m cell); the buckets take ncells * DW_CAP * 16 bytes of the workspace (67 MB at most)
static inline bool dw_use_buckets(int64_t m, int64_t ncells) { return ncells <= 65536 && m <= ncells * (DW_CAP * 5 / 8); }
// where the bucket form keeps things inside the workspace (ints): count x2 | 16-byte aligned buckets | overflow
static inline void bucket_layout(int32_t* ws, long long ncells, int parity, BinK* b) {
  const long long cstride = ncells + DW_CNT_EXTRA;
  b->count = ws + (long long)parity * cstride;
  uintptr_t sp = (uintptr_t)(ws + 2 * cstride);
  b->buckets = (float4*)((sp + 15) & ~(uintptr_t)15);
  b->overflow = b->buckets + ncells * DW_CAP;
}

struct StepK {
  KView st, tg;
  const DevType* types;
  const uint8_t* type_id;
  const float* noise_replay;
  const float* action;        // SoA [n_act][n_pad] or null (= stored cmd)
  int action_rows;            // DSIM_OPT_ACTION_ROWS: action is row-major [n][4] (the one-launch quad kernels only)
  float* echo;                // physics kernel: clipped action out, or null
  float* pos_e_out;           // control kernel only
  float* yaw_e_out;
  float* cmd_out;             // control kernel only: SoA [n_act][n_pad] copy of the new command, or null
  float* obs_out;             // physics kernel: fused observation rows [n][obs_w], or null
  int obs_w;                  // 16 + the table's largest actuator count: width of an observation row / rows of echo, cmd_out
  long long n;                // drones (rows of obs_out)
  FbList fb;                  // deferred WLS fallbacks (hexa)
  long long n_pad;
  long long first;            // general step kernel: first drone of this launch
  const float* wp_table;      // waypoint mode (null = targets view)
  int* wp_counter;
  const float* wp_offset;
  const float* ext_force;     // SoA [3][n_pad] body-frame force at the COM, or null
  const unsigned long long* step_index_dev;   // added to step_index (graph replay), or null
  int n_wp, n_steps;
  unsigned long long seed, step_index;
  int substeps;
  float dt_phys, dt_ctrl;
  unsigned options;
  long long lo, last;         // run kernels: first drone of the run (the launch starts at the tile that holds it), one past its last
  int run_type;               // run kernels: the run's type
  const int* drone_id;        // the caller's index of storage slot i (keys the noise counter), or null = i
  const int* io_id;           // DSIM_OPT_CALLER_IO: = drone_id, the per-drone arrays beside the state are indexed by it; else null
  unsigned hexa_types;        // bit t set: type t of the table is a morphing hexa (26 state fields in use)
  BinK bin;                   // grid of the next Env.step's downwash (k_step_mixed / k_step_run), count = null: none
  float* dyn_rates;           // Physics.DYN: BaseAviary.rpy_rates, SoA [3][n_pad] in-out (k_dyn only)
};

// Global accesses.  NT = nontemporal (streaming) hint: each state field is read once and written
// once per step, so for fleets larger than the caches the lines should not linger in L2/MALL
// (measured on MI355X with this access shape: +12-15 % HBM rate, tools/membench.hip).  Small
// fleets that fit the Infinity Cache keep the default policy so consecutive steps hit on-die.
// (uniform base pointer, per-lane BYTE offset): the form that maps onto
// `global_load_dword v, v_off, s[base:base+1]` (scalar base + 32-bit VGPR offset).
template <bool NT> __device__ __forceinline__ float ldg(const float* ub, unsigned boff) {
  const float* p = reinterpret_cast<const float*>(reinterpret_cast<const char*>(ub) + boff);
  return NT ? __builtin_nontemporal_load(p) : *p;
}
template <bool NT> __device__ __forceinline__ void stg(float* ub, unsigned boff, float v) {
  float* p = reinterpret_cast<float*>(reinterpret_cast<char*>(ub) + boff);
  if (NT) __builtin_nontemporal_store(v, p); else *p = v;
}

// The stores sit in a later basic block than the loads; instruction selection works per block and would no
// longer see that the lane offset is a zero-extended 32-bit value, so every store would get a 64-bit VGPR
// address (one v_lshl_add_u64 + two VGPRs per field).  Re-materialising the offset in the store's block keeps
// the scalar-base + 32-bit-lane-offset form there too.
__device__ __forceinline__ unsigned pin_lane_offset(unsigned off) {
  asm volatile("" : "+v"(off));
  return off;
}

template <bool NT = false>
__device__ __forceinline__ void load_rigid(const float* ub, long long fs, unsigned lo /* bytes */, Rigid& s) {
  s.pos = v3(ldg<NT>(ub + 0 * fs, lo), ldg<NT>(ub + 1 * fs, lo), ldg<NT>(ub + 2 * fs, lo));
  s.q = Q4{ldg<NT>(ub + 3 * fs, lo), ldg<NT>(ub + 4 * fs, lo), ldg<NT>(ub + 5 * fs, lo), ldg<NT>(ub + 6 * fs, lo)};
  s.vel = v3(ldg<NT>(ub + 7 * fs, lo), ldg<NT>(ub + 8 * fs, lo), ldg<NT>(ub + 9 * fs, lo));
  s.w = v3(ldg<NT>(ub + 10 * fs, lo), ldg<NT>(ub + 11 * fs, lo), ldg<NT>(ub + 12 * fs, lo));
}
template <bool NT = false>
__device__ __forceinline__ void store_rigid(float* ub, long long fs, unsigned lo /* bytes */, const Rigid& s) {
  stg<NT>(ub + 0 * fs, lo, s.pos.x); stg<NT>(ub + 1 * fs, lo, s.pos.y); stg<NT>(ub + 2 * fs, lo, s.pos.z);
  stg<NT>(ub + 3 * fs, lo, s.q.x); stg<NT>(ub + 4 * fs, lo, s.q.y); stg<NT>(ub + 5 * fs, lo, s.q.z); stg<NT>(ub + 6 * fs, lo, s.q.w);
  stg<NT>(ub + 7 * fs, lo, s.vel.x); stg<NT>(ub + 8 * fs, lo, s.vel.y); stg<NT>(ub + 9 * fs, lo, s.vel.z);
  stg<NT>(ub + 10 * fs, lo, s.w.x); stg<NT>(ub + 11 * fs, lo, s.w.y); stg<NT>(ub + 12 * fs, lo, s.w.z);
}
// CH (chained): last_vel / last_rates are neither read nor written (DSIM_OPT_CHAINED)
template <int NACT, bool NT = false, bool CH = false>
__device__ __forceinline__ void load_mem(const float* ub, long long fs, unsigned lo /* bytes */, CtrlMem<NACT>& m) {
  if (!CH) {
    m.last_vel = v3(ldg<NT>(ub + 13 * fs, lo), ldg<NT>(ub + 14 * fs, lo), ldg<NT>(ub + 15 * fs, lo));
    m.last_rates = v3(ldg<NT>(ub + 16 * fs, lo), ldg<NT>(ub + 17 * fs, lo), ldg<NT>(ub + 18 * fs, lo));
  }
  m.last_thrust = ldg<NT>(ub + 19 * fs, lo);
#pragma unroll
  for (int j = 0; j < NACT; ++j) m.cmd[j] = ldg<NT>(ub + (20 + j) * fs, lo);
}
template <int NACT, bool NT = false, bool CH = false>
__device__ __forceinline__ void store_mem(float* ub, long long fs, unsigned lo /* bytes */, const CtrlMem<NACT>& m) {
  if (!CH) {
    stg<NT>(ub + 13 * fs, lo, m.last_vel.x); stg<NT>(ub + 14 * fs, lo, m.last_vel.y); stg<NT>(ub + 15 * fs, lo, m.last_vel.z);
    stg<NT>(ub + 16 * fs, lo, m.last_rates.x); stg<NT>(ub + 17 * fs, lo, m.last_rates.y); stg<NT>(ub + 18 * fs, lo, m.last_rates.z);
  }
  stg<NT>(ub + 19 * fs, lo, m.last_thrust);
#pragma unroll
  for (int j = 0; j < NACT; ++j) stg<NT>(ub + (20 + j) * fs, lo, m.cmd[j]);
}
// A broadcast target view has mask = 0, so kv_off() is 0 for every lane: all lanes read the same
// ten floats (one cache line per wave-instruction), no separate code path.
template <bool NT = false>
__device__ __forceinline__ void load_target(const float* ub, long long fs, unsigned lo /* bytes */, Target& t) {
  t.pos = v3(ldg<NT>(ub + 0 * fs, lo), ldg<NT>(ub + 1 * fs, lo), ldg<NT>(ub + 2 * fs, lo));
  t.vel = v3(ldg<NT>(ub + 3 * fs, lo), ldg<NT>(ub + 4 * fs, lo), ldg<NT>(ub + 5 * fs, lo));
  t.acc = v3(ldg<NT>(ub + 6 * fs, lo), ldg<NT>(ub + 7 * fs, lo), ldg<NT>(ub + 8 * fs, lo));
  t.yaw = ldg<NT>(ub + 9 * fs, lo);
}

// Waypoint-table targets (examples/fly_INDI_TrajectoryTrack.py:242-245): row wp of the table (+ the
// drone's own position offset).  The 48 KB table is gathered per lane and stays L1/L2-resident.
__device__ __forceinline__ void waypoint_target(const StepK& a, long long i, int wp, Target& t) {
  const float* r = a.wp_table + (long long)wp * 10;
  t.pos = v3(r[0], r[1], r[2]);
  if (a.wp_offset) t.pos = t.pos + v3(a.wp_offset[i], a.wp_offset[a.n_pad + i], a.wp_offset[2 * a.n_pad + i]);
  t.vel = v3(r[3], r[4], r[5]);
  t.acc = v3(r[6], r[7], r[8]);
  t.yaw = r[9];
}
// wp_counters[j] + 1 if < NUM_WP - 1 else 0   (fly_INDI_TrajectoryTrack.py:253-256)
__device__ __forceinline__ int waypoint_next(int wp, int n_wp) { return wp < n_wp - 1 ? wp + 1 : 0; }

// physics sub-steps of one Env.step for a quad (BaseAviary.py:510-545)
// NOISE: 0 = off, 1 = in-kernel counter-based noise, 2 = replay buffer if given else in-kernel.
// NROW = rows per sub-step of the replay buffer's force / moment halves (the kernel's NACT).
// OPTS: honour the drag / ground-effect option bits (general kernels only).  prev = the action of the
// previous Env.step (last_clipped_action) for the drag of sub-step 0, or null = this step's action.
// NSUB > 0: the number of sub-steps is a compile-time constant and the code is straight-line.  Used for 1 (BASELINE's
// metric definition): without the loop the compiler keeps the headline kernel in 91 instead of 110 VGPRs
// (5 waves/SIMD) and 836 instead of 887 vector instructions.  (Measured and rejected: 2 — no change, 198 us
// either way; 5 — the unrolled body spills, 533 vs 310 us.)
// FINE: whether this instance carries the 16 + 16-bit noise lattice (the FINE bit of StepK.options, resolved by the entry point:
// every launch of ONE sub-step, and DSIM_OPT_NOISE_FINE at any count — a wave-uniform run-time switch) beside the 8 + 8-bit one.
// -1 = the rule: every instance does, except the LOOPED ones — the fast paths over several sub-steps, bound by vector issue and
// tuned to their register budgets; the launchers hand a fine-lattice launch of several sub-steps to the general kernels.
// LOOPED: the launcher picked this instance because the launch has SEVERAL sub-steps (its single-sub-step twin takes the others):
// the loop carries the body-frame form of the step (dsim_device.h:bullet_step_body).
// TABS: whether `tab` points at the tables — 1 yes, 0 no, -1 ask the pointer (an LDS address is never known to be non-null at
// compile time: left to the pointer, BOTH forms of the Box-Muller pairs are compiled into every looped instance)
template <int NOISE, int NROW = 4, bool OPTS = false, int NSUB = 0, bool PLANE = false, int FINE = -1, bool LOOPED = false, int TABS = -1, class DT>
__device__ __forceinline__ void quad_substeps(DT& T, const StepK& a, long long i, Rigid& s,
                                              const float cmd[4], unsigned long long step_index,
                                              V3 ext = V3{-0.0f, -0.0f, -0.0f} /* x + -0 = x for EVERY x: a caller without a force pays no add */,
                                              const float* prev = nullptr, long long nid = -1,
                                              const NoiseTab* tab = nullptr /* LDS tables of the Box-Muller pairs, or none */) {
  // nid: the drone's index in the caller's numbering when the fleet is stored in another order (StepK.drone_id): the
  // key of its noise stream.  -1 (a constant at the call sites of the single-order kernels) = i.
  const uint64_t noise_key = (uint64_t)(nid >= 0 ? nid : i);
  V3 F, tau;
  if (NOISE == 0) quad_wrench(T, cmd, nullptr, F, tau);   // cmd is constant over the sub-steps
  // (several sub-steps with noise: the noise-free part of the map once, the normals' part per sub-step)
  constexpr bool SPLIT = NOISE != 0 && NSUB != 1 && !OPTS;
  QuadBase qb = QuadBase{0.0f, V3{0.0f, 0.0f, 0.0f}};
  if (SPLIT) qb = quad_wrench_base(T, cmd);
  const int n_sub = NSUB > 0 ? NSUB : a.substeps;
  uint32_t nb[4] = {0u, 0u, 0u, 0u};       // the Threefry block: ONE serves two consecutive sub-steps (dsim_device.h:noise_normals)
  // the looped fast instances carry the body-frame form of the step across the sub-steps (dsim_device.h:bullet_step_body)
  // (only where several sub-steps are certain: with one, w' = R' (R^T w + a_b dt) costs the stored angular velocity two more
  // matrix roundings than w + R a_b dt and saves nothing.  The neutral zero-sub-step pass of the placement trials runs on
  // k_physics_fast, which is not LOOPED: there the state goes back bit for bit.)
  constexpr bool BODY_OK = LOOPED && !OPTS && !PLANE && NSUB != 1;
  constexpr bool BODY = BODY_OK;
  RigidB sb = RigidB{};
  const bool body = BODY && n_sub > 0;     // (wave-uniform) a zero-sub-step pass hands the state back bit for bit (placement trials)
  if (body) sb = body_begin(s);
  for (int k = 0; k < n_sub; ++k) {
    if (NOISE != 0) {
      float nz[8];
      if (NOISE == 2 && a.noise_replay) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          nz[j] = a.noise_replay[((long long)k * 2 * NROW + j) * a.n_pad + i];
          nz[4 + j] = a.noise_replay[((long long)k * 2 * NROW + NROW + j) * a.n_pad + i];
        }
      } else if ((FINE >= 0 ? FINE != 0 : (NSUB == 1 || OPTS || !LOOPED)) && (a.options & DSIM_OPT_NOISE_FINE)) {   // (wave-uniform) the 16 + 16-bit lattice
        quad_normals_fine(a.seed, noise_key, step_index * (uint64_t)a.substeps + (uint64_t)k, nz);
      } else {
        const uint64_t sub = step_index * (uint64_t)a.substeps + (uint64_t)k;          // (wave-uniform)
        if (k == 0 || (sub & 1ull) == 0) noise_block(a.seed, noise_key, sub >> 1, nb);  // a new block every other sub-step
        if (TABS < 0 ? tab != nullptr : TABS != 0) quad_normals_from_block_tab(*tab, nb, (sub & 1ull) != 0, nz);          // (the same bits, from LDS)
        else quad_normals_from_block(nb, (sub & 1ull) != 0, nz);                        // N(0,.01) | N(0,.001)
      }
      if (SPLIT) quad_wrench_noise(T, qb, nz, F, tau); else quad_wrench(T, cmd, nz, F, tau);
    }
    if (OPTS && (PLANE || (a.options & (DSIM_OPT_DRAG | DSIM_OPT_GROUND)))) {
      V3 F2 = F + ext, tau2 = tau;
      if (a.options & DSIM_OPT_GROUND) ground_effect_quad(T, s, cmd, F2, tau2);              // BaseAviary.py:528-529
      if (a.options & DSIM_OPT_DRAG) {                                                        // :531-532
        float lc[4];     // rotor speeds of the PREVIOUS action on sub-step 0 (values selected, not pointers: a pointer
#pragma unroll           // select between two register arrays sends both to scratch)
        for (int j = 0; j < 4; ++j) lc[j] = (k == 0 && prev) ? prev[j] : cmd[j];
        F2 = F2 + drag_quad(T, s, lc);
      }
      bullet_step<PLANE>(T, a.dt_phys, s, F2, tau2);
      continue;
    }
    if constexpr (BODY_OK) bullet_step_body(T, a.dt_phys, sb, F + ext, tau);      // (the loop runs: BODY holds)
    else bullet_step(T, a.dt_phys, s, F + ext, tau);
  }
  if (body) body_end(sb, s);
}

// the same for the morphing hexa (BaseAviary.py:1389-1457); replay rows: f[6], m[6]
template <bool NOISE, bool REPLAY = true, bool ONE = false, bool PLANE = false, bool LOOPED = false, int TABS = -1, class DT>
__device__ __forceinline__ void hexa_substeps(DT& T, const StepK& a, long long i, Rigid& s,
                                              const float cmd[6], unsigned long long step_index,
                                              V3 ext = V3{-0.0f, -0.0f, -0.0f}, long long nid = -1,
                                              const NoiseTab* tab = nullptr /* LDS tables of the Box-Muller pairs, or none */) {
  const uint64_t noise_key = (uint64_t)(nid >= 0 ? nid : i);
  V3 F, tau;
  if (!NOISE) hexa_wrench(T, cmd, nullptr, F, tau);
  constexpr bool SPLIT = NOISE && !ONE;
  HexaBase hb = HexaBase{V3{0.0f, 0.0f, 0.0f}, V3{0.0f, 0.0f, 0.0f}};
  if (SPLIT) hb = hexa_wrench_base(T, cmd);
  // The state holds what PyBullet reports — the BASE link's centre of mass (dsim_type_params.base_offset); the composite
  // body is integrated about its own: p = p_b - R d, v = v_b - w x (R d) in front of the sub-steps, and back behind them.
  // The position never makes the round trip: p_b' = p_b + sum(dt v_com) + (R' d - R d) — the sub-steps move the stored
  // position by the composite's displacement and the CHANGE of the offset is added behind them (millimetres, where
  // subtracting and re-adding the offset itself costs two roundings at the magnitude of the position: 1.4 ulp32(x) at
  // x = 34 m was the worst margin of the hexa kernels, 0.70 of the step's bar).  With the plane the contact geometry
  // wants the composite's position itself.
  if (!ONE && a.substeps <= 0) return;     // (wave-uniform) a zero-sub-step pass hands the state back bit for bit (placement trials)
  const V3 o0 = mul(matrix_from_quat(s.q), v3(T.base_off[0], T.base_off[1], T.base_off[2]));
  if (PLANE) s.pos = s.pos - o0;
  s.vel = s.vel - cross(s.w, o0);
  const int n_sub = ONE ? 1 : a.substeps;
  constexpr bool BODY_OK = LOOPED && !ONE && !PLANE && !REPLAY;       // the looped fast instances (quad_substeps: LOOPED): dsim_device.h:bullet_step_body
  constexpr bool BODY = BODY_OK;
  RigidB sb = RigidB{};
  if (BODY) sb = body_begin(s);
  // the noise-free wrench of the default streams (the command is constant over the sub-steps): split off once per Env.step
  V3 F0 = V3{0.0f, 0.0f, 0.0f}, tau0 = V3{0.0f, 0.0f, 0.0f};
  if (NOISE) {
    if (SPLIT) { F0 = hb.F; tau0 = hb.tau; }
    else if (!(REPLAY && a.noise_replay)) hexa_wrench(T, cmd, nullptr, F0, tau0);
  }
  uint32_t nb[4] = {0u, 0u, 0u, 0u};       // the Threefry block: ONE serves two consecutive sub-steps (dsim_device.h:noise_normals)
  for (int k = 0; k < n_sub; ++k) {
    if (NOISE) {
      if (REPLAY && a.noise_replay) {          // per-rotor normals as an input: f[6], m[6] (BaseAviary.py:1429-1430)
        float nz[12];
#pragma unroll
        for (int j = 0; j < 12; ++j) nz[j] = a.noise_replay[((long long)k * 12 + j) * a.n_pad + i];
        if (SPLIT) hexa_wrench_noise(T, hb, nz, F, tau); else hexa_wrench(T, cmd, nz, F, tau);
      } else {                                 // the default streams: the six normals of the body wrench (dsim_device.h:noise_normals)
        float z[6];
        const uint64_t sub = step_index * (uint64_t)a.substeps + (uint64_t)k;          // (wave-uniform)
        if ((ONE || REPLAY || !LOOPED) && (a.options & DSIM_OPT_NOISE_FINE)) {           // (wave-uniform) the fine lattice: every instance but the looped fast ones
          hexa_z_fine(a.seed, noise_key, sub, z);
        } else {
          if (k == 0 || (sub & 1ull) == 0) noise_block(a.seed, noise_key, sub >> 1, nb);  // a new block every other sub-step
          if (TABS < 0 ? tab != nullptr : TABS != 0) hexa_z_from_block_tab(*tab, nb, (sub & 1ull) != 0, z);
          else hexa_z_from_block(nb, (sub & 1ull) != 0, z);
        }
        hexa_wrench_z(T, F0, tau0, z, F, tau);
      }
    }
    if constexpr (BODY_OK) bullet_step_body(T, a.dt_phys, sb, F + ext, tau);      // (the loop runs: BODY holds)
    else bullet_step<PLANE>(T, a.dt_phys, s, F + ext, tau);
  }
  if (BODY) body_end(sb, s);
  {
    const V3 o = mul(matrix_from_quat(s.q), v3(T.base_off[0], T.base_off[1], T.base_off[2]));
    s.pos = s.pos + (PLANE ? o : o - o0); s.vel = s.vel + cross(s.w, o);
  }
}

__device__ __forceinline__ long long noise_id(const StepK& a, long long i) { return a.drone_id ? (long long)a.drone_id[i] : -1LL; }

typedef float vf4 __attribute__((ext_vector_type(4)));        // (a native vector: what the nontemporal builtins take)

#ifndef DSIM_HEXA_WAVES
#define DSIM_HEXA_WAVES 3
#endif
#ifndef DSIM_LATE_STORE_BASE
#define DSIM_LATE_STORE_BASE 1
#endif
#ifndef DSIM_LATE_STORE_BASE_S1
#define DSIM_LATE_STORE_BASE_S1 0
#endif
// p, as a wave-uniform value the compiler knows nothing about, available only behind `after`: the offset 0 goes through an empty
// asm that also takes `after` in, and comes back through v_readfirstlane (which is what tells the compiler that it is uniform:
// an asm's own output counts as divergent, and the loads behind it as per-lane loads)
template <class P>
__device__ __forceinline__ const P* opaque_after(const P* p, float after) {
  int z = 0;
  asm("" : "+v"(z) : "v"(after));
  return reinterpret_cast<const P*>(reinterpret_cast<const char*>(p) + __builtin_amdgcn_readfirstlane(z));
}
// Mixed fleets: every lane carries a type id, but the per-type constants must stay wave-uniform
// (scalar loads into SGPRs: ~150 floats per type would otherwise sit in VGPRs per lane — 256 VGPRs
// plus spills).  Waterfall: the wave peels one type per iteration with the lanes of that type active.
#define DSIM_FOR_MY_TYPE(UNIFORM, a, i, BODY)                                   \
  do {                                                                          \
    if (UNIFORM) { const DevType& T = (a).types[0]; BODY; }                     \
    else {                                                                      \
      const int my_t_ = (a).type_id ? (int)(a).type_id[i] : 0;                  \
      for (;;) {                                                                \
        const int cur_t_ = __builtin_amdgcn_readfirstlane(my_t_);               \
        if (my_t_ == cur_t_) { const DevType& T = (a).types[cur_t_]; BODY; break; } \
      }                                                                         \
    }                                                                           \
  } while (0)

// Before the waterfall, a mixed tile is PARTITIONED by type: the 256 lanes of the workgroup re-assign the
// tile's 256 drones among themselves so that drones of one type sit in consecutive lanes (a stable counting
// sort on the type id: per-wave ballots + popcounts, per-wave/per-type counts and the slot -> drone table in
// LDS).  Waves become type-homogeneous except where one type's run ends inside a wave (at most n_types - 1
// waves per tile), so the waterfall runs once instead of once per type present — in config 5 (even index
// quad, odd index hexa) every wave would otherwise execute BOTH laws at half occupancy of its lanes.  Lanes
// then gather their drone's fields from within the same 256-drone tile (same cache lines, HBM traffic
// unchanged).  Everything keyed by the drone index (noise stream, per-drone buffers) is unaffected.
// slot (lane of the workgroup) that processes this lane's natural drone: a stable counting sort of the tile on
// the type id `my` (0..DSIM_MAX_TYPES, the last value = no drone, sorted last).  One barrier.
template <int WAVES>
__device__ __forceinline__ unsigned tile_dest(int my) {
  __shared__ unsigned short cnt[WAVES][DSIM_MAX_TYPES + 1];
  const unsigned t = threadIdx.x, w = t >> 6, lane = t & 63;
  const unsigned long long lt = (1ULL << lane) - 1ULL;
  unsigned rank = 0, c_mine = 0;
#pragma unroll
  for (int ty = 0; ty <= DSIM_MAX_TYPES; ++ty) {
    const unsigned long long mask = __ballot(my == ty);
    if (my == ty) rank = (unsigned)__popcll(mask & lt);
    if ((int)lane == ty) c_mine = (unsigned)__popcll(mask);
  }
  if (lane <= DSIM_MAX_TYPES) cnt[w][lane] = (unsigned short)c_mine;
  __syncthreads();
  unsigned dest = rank;
  for (int ty = 0; ty <= DSIM_MAX_TYPES; ++ty) {
#pragma unroll
    for (unsigned ww = 0; ww < WAVES; ++ww) {
      const unsigned c = cnt[ww][ty];
      dest += (ty < my || (ty == my && ww < w)) ? c : 0u;
    }
  }
  return dest;
}
__device__ __forceinline__ unsigned tile_partition(const uint8_t* type_id, long long i0, long long n_pad) {
  __shared__ unsigned char slot2drone[256];
  const unsigned t = threadIdx.x;
  const int my = (i0 + t < n_pad) ? min((int)type_id[i0 + t], DSIM_MAX_TYPES - 1) : DSIM_MAX_TYPES;   // out of range: sorted last
  slot2drone[tile_dest<4>(my)] = (unsigned char)t;
  __syncthreads();
  return slot2drone[t];
}
template <bool UNIFORM>
__device__ __forceinline__ unsigned tile_slot(const uint8_t* type_id, long long i0, long long n_pad) {
  if (UNIFORM || !type_id) return threadIdx.x;       // (a homogeneous fleet on an any-fleet instance: wave-uniform, no partition)
  return tile_partition(type_id, i0, n_pad);
}

// General form: per-drone type ids (mixed quad / hexa fleets, NACT = 6), explicit action
// override, noise replay, external force, ragged sizes.  a.first = first drone this launch covers
// (a multiple of 256, so the scalar-base + lane-offset addressing of the fast kernel applies).
struct Addr { float* sb; const float* tb; unsigned sl, tl; long long sfs, tfs; };
__device__ __forceinline__ Addr make_addr(const StepK& a, long long i0, unsigned p /* drone within the tile */) {
  Addr r;
  r.sb = a.st.base + kv_off(a.st, i0);
  r.tb = a.tg.base ? a.tg.base + kv_off(a.tg, i0) : nullptr;
  r.sl = 4u * kv_lane(a.st, p);
  r.tl = 4u * kv_lane(a.tg, p);
  r.sfs = a.st.field_stride; r.tfs = a.tg.field_stride;
  return r;
}
// Kernels that loop over several sub-steps take the Box-Muller pairs of the rotor noise from LDS tables (NoiseTab, dsim_device.h:
// bit-identical to direct evaluation): filled by the whole workgroup before any lane leaves.  `ntab` = the tables, or null.
#define DSIM_NOISE_TAB(ON, THREADS)                                                                  \
  __shared__ NoiseTab ntab_[1];                                                                      \
  const NoiseTab* const ntab = (ON) ? &ntab_[0] : nullptr;                                           \
  if (ON) {                                                                                          \
    for (unsigned e_ = threadIdx.x; e_ < 256u; e_ += (THREADS)) noise_tab_init(ntab_[0], e_);        \
    __syncthreads();                                                                                 \
  }
// All the runs of a type-major fleet in ONE launch: a workgroup finds its run by its index (constant-index walk over the
// table, everything wave-uniform) and runs that run's law.  A 65 536-drone shard of BASELINE config 5 is two runs of 128
// workgroups: 9.5 + 9.0 us as two dependent launches, 12.0 us as one; at 4 194 304 drones 160.1 against 165.2 us.  One
// launch per run (k_step_run) serves fleets with a single run (and more than DSIM_MAX_TYPES of them).
struct RunTab {
  int blk0[DSIM_MAX_TYPES + 1];            // first workgroup of run q (blk0[q] = the total for q >= n_runs)
  long long first[DSIM_MAX_TYPES], lo[DSIM_MAX_TYPES], last[DSIM_MAX_TYPES];
  int type[DSIM_MAX_TYPES];
  unsigned hexa_mask;        // bit q: run q flies morphing-hexa physics (six actuators: DSIM_KIND_HEXA6DOF and _HEXA_QUADLAW)
  unsigned quadlaw6_mask;    // bit q: ... with the quad law on its six actuators (DSIM_KIND_HEXA_QUADLAW: k_control_runs)
  // null: workgroup b serves the runs one after the other (blk0).  Else [blocks] device ints, (tile << 3) | run: the runs are
  // served SIDE BY SIDE, each at a rate proportional to its size.  For DSIM_OPT_CALLER_IO: a drone's outputs go to its caller
  // index, and the drones of every run are spread over the caller's whole range (even index quad, odd index hexa ...), so
  // one run alone fills every other 88-byte row, every other dword of the command arrays — partial memory bursts, which
  // cost a read-modify-write each (measured: Env.step of 4 194 304 interleaved drones 351 us run after run).  Side by
  // side, the runs' halves of a line arrive within microseconds of each other and meet in the memory-side cache.
  const int* block_map;
};
// the run a workgroup belongs to: constant-index walk over the table, everything wave-uniform (SGPRs).  A macro, not a
// function: a kernel argument handed on by reference is copied to scratch (264 bytes per lane) before the walk.
struct RunOf { long long i0, lo, last; int type; bool hexa, quadlaw6; };
#define DSIM_RUN_OF_BLOCK(rt, ro, BIDX)                                                                             \
  RunOf ro;                                                                                                         \
  {                                                                                                                 \
    const int bidx_ = __builtin_amdgcn_readfirstlane((int)(BIDX));                                                  \
    int r_ = 0, tile_ = -1;                                                                                         \
    if (rt.block_map) { const int e_ = rt.block_map[bidx_]; r_ = e_ & 7; tile_ = e_ >> 3; }                         \
    else { _Pragma("unroll") for (int q = 1; q < DSIM_MAX_TYPES; ++q) if (bidx_ >= rt.blk0[q]) r_ = q; }            \
    r_ = __builtin_amdgcn_readfirstlane(r_);                                                                        \
    long long first_ = rt.first[0];                                                                                 \
    int b0_ = rt.blk0[0];                                                                                           \
    ro.lo = rt.lo[0]; ro.last = rt.last[0]; ro.type = rt.type[0];                                                   \
    _Pragma("unroll") for (int q = 1; q < DSIM_MAX_TYPES; ++q)                                                      \
      if (q == r_) { first_ = rt.first[q]; ro.lo = rt.lo[q]; ro.last = rt.last[q]; ro.type = rt.type[q]; b0_ = rt.blk0[q]; } \
    if (!rt.block_map) tile_ = bidx_ - b0_;                                                                         \
    ro.i0 = first_ + (long long)__builtin_amdgcn_readfirstlane(tile_) * 256;                                        \
    ro.hexa = (rt.hexa_mask >> r_) & 1u;                                                                            \
    ro.quadlaw6 = (rt.quadlaw6_mask >> r_) & 1u;                                                                    \
    if (tile_ < 0) ro.last = ro.lo = 0;                 /* a padding entry of the map: nothing to serve */          \
  }

// ---------------------------------------------------------------------------
// host side: helpers shared by the entry points (defined in dsim_api.hip unless noted; not exported)
// ---------------------------------------------------------------------------
#pragma GCC visibility push(hidden)
int make_kview(const dsim_view& v, int need_fields, KView* k, bool bcast = false);
int fill_stepk(dsim_ctx* ctx, int64_t n, const dsim_view& state, const dsim_view* targets, const dsim_step_args* args, StepK* a);
int fb_prepare(dsim_ctx* ctx, long long n_pad, hipStream_t st);
void fb_finish(dsim_ctx* ctx, const StepK& a, hipStream_t st);
int make_runtab(const dsim_ctx* ctx, long long n_pad, const dsim_type_run* runs, int n_runs, RunTab* rt, bool* any_hexa);
int side_by_side_map(dsim_ctx* ctx, hipStream_t st, const dsim_type_run* runs, int n_runs, RunTab* rt);
int observe_impl(dsim_ctx* ctx, void* stream, int64_t n, dsim_view state, const float* last_action, float* obs_out, int32_t obs_width,
                 int soa);
void bin_next_prepare(dsim_ctx* ctx, int64_t n, const dsim_step_args* args, StepK* a, hipStream_t st);      // dsim_downwash.hip
void bin_next_commit(dsim_ctx* ctx, int64_t n, const dsim_step_args* args, const StepK& a);                 // dsim_downwash.hip
int dyn_check(const dsim_ctx* ctx, const dsim_step_args* args, const StepK& a);                             // dsim_two_call.hip
int dyn_launch(bool ctrl, const StepK& a, bool nt, hipStream_t st);                                         // dsim_two_call.hip
int step_general(dsim_ctx* ctx, int64_t n, const dsim_view& state, const dsim_view& targets, const dsim_step_args* args, StepK& a,
                 long long first, bool fb_open, hipStream_t st_);                                           // dsim_step_mixed.hip
#pragma GCC visibility pop


static inline unsigned grid_for(long long n) { return (unsigned)((n + 255) / 256); }
// Which lattice a launch's rotor noise is drawn on (include/dronesim_amd.h: DSIM_OPT_NOISE_FINE / _COARSE): resolved ONCE per
// entry point into the FINE bit of StepK.options — the kernels test that bit, the launchers route on it.
static inline uint32_t resolve_noise_lattice(uint32_t options, int substeps) {
  if (options & DSIM_OPT_NOISE_FINE) return options;
  if (options & DSIM_OPT_NOISE_COARSE) return options;
  return substeps == 1 ? (options | DSIM_OPT_NOISE_FINE) : options;
}

// Streaming (nontemporal) accesses once one step's traffic exceeds what the 256 MB Infinity Cache can keep between
// consecutive steps; DSIM_OPT_STREAM_ON / _OFF override (the library reads no environment variables).
static inline bool stream_policy(const dsim_step_args* a, long long n_pad, double bytes_per_drone) {
  if (a->options & DSIM_OPT_STREAM_ON) return true;
  if (a->options & DSIM_OPT_STREAM_OFF) return false;
  return (double)n_pad * bytes_per_drone > 192.0 * 1024 * 1024;
}

// ... of a general kernel that serves homogeneous and mixed fleets with ONE instance per (noise, actuator count): the type waterfall
// runs once for a homogeneous fleet (type_id null: type 0), its partition is skipped — these kernels are not on a measured path, and
// the UNIFORM specialisation doubled their instance count (round 6: 293 -> ... instances)
#define DSIM_LAUNCH_GEN_ANY(KERNEL, NOISE, SIX, g, a, stream)                                           \
  do {                                                                                                  \
    const dim3 b_(256);                                                                                 \
    switch (((NOISE) ? 2 : 0) | ((SIX) ? 1 : 0)) {                                                      \
      case 0: hipLaunchKernelGGL((KERNEL<false, false, 4>), g, b_, 0, stream, a); break;                \
      case 1: hipLaunchKernelGGL((KERNEL<false, false, 6>), g, b_, 0, stream, a); break;                \
      case 2: hipLaunchKernelGGL((KERNEL<true, false, 4>), g, b_, 0, stream, a); break;                 \
      default: hipLaunchKernelGGL((KERNEL<true, false, 6>), g, b_, 0, stream, a); break;                \
    }                                                                                                   \
  } while (0)

// (noise, uniform) x actuator count dispatch of a general kernel
#define DSIM_LAUNCH_GEN(KERNEL, NOISE, UNI, SIX, g, a, stream)                                          \
  do {                                                                                                  \
    const dim3 b_(256);                                                                                 \
    switch (((NOISE) ? 4 : 0) | ((UNI) ? 2 : 0) | ((SIX) ? 1 : 0)) {                                    \
      case 0: hipLaunchKernelGGL((KERNEL<false, false, 4>), g, b_, 0, stream, a); break;                \
      case 1: hipLaunchKernelGGL((KERNEL<false, false, 6>), g, b_, 0, stream, a); break;                \
      case 2: hipLaunchKernelGGL((KERNEL<false, true, 4>), g, b_, 0, stream, a); break;                 \
      case 3: hipLaunchKernelGGL((KERNEL<false, true, 6>), g, b_, 0, stream, a); break;                 \
      case 4: hipLaunchKernelGGL((KERNEL<true, false, 4>), g, b_, 0, stream, a); break;                 \
      case 5: hipLaunchKernelGGL((KERNEL<true, false, 6>), g, b_, 0, stream, a); break;                 \
      case 6: hipLaunchKernelGGL((KERNEL<true, true, 4>), g, b_, 0, stream, a); break;                  \
      default: hipLaunchKernelGGL((KERNEL<true, true, 6>), g, b_, 0, stream, a); break;                 \
    }                                                                                                   \
  } while (0)
