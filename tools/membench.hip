// membench.hip — dev micro-benchmark (not part of the product): what HBM rate can THIS traffic
// shape reach on MI355X?  Shape of k_step_fast: per drone read 34 floats (24 state + 10 target),
// write 24 floats, one drone per lane.  Variants differ only in layout / access width.
//   hipcc -O3 --offload-arch=gfx950 -o membench tools/membench.hip && ./membench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int FS = 24, FT = 10;
typedef float f4 __attribute__((ext_vector_type(4)));

// light dependent arithmetic so that nothing is optimised away
__device__ __forceinline__ void mix(const float* in, int nin, float* out, int nout) {
  float acc = 0.f;
  for (int k = 0; k < nin; ++k) acc = fmaf(in[k], 1.0001f, acc);
  for (int k = 0; k < nout; ++k) out[k] = in[k] + acc * 1e-9f;
}

// K0: plain SoA, one dword per lane per field
template <bool NT>
__global__ __launch_bounds__(256) void k_soa(float* st, const float* tg, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float v[FS + FT], o[FS];
#pragma unroll
  for (int f = 0; f < FS; ++f) v[f] = NT ? __builtin_nontemporal_load(st + f * n + i) : st[f * n + i];
#pragma unroll
  for (int f = 0; f < FT; ++f) v[FS + f] = NT ? __builtin_nontemporal_load(tg + f * n + i) : tg[f * n + i];
  mix(v, FS + FT, o, FS);
#pragma unroll
  for (int f = 0; f < FS; ++f) { if (NT) __builtin_nontemporal_store(o[f], st + f * n + i); else st[f * n + i] = o[f]; }
}

// K1: wave tiles [n/64][F][64], dword per lane
template <bool NT>
__global__ __launch_bounds__(256) void k_tile(float* st, const float* tg, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float* ps = st + (i >> 6) * (FS * 64) + (i & 63);
  const float* pt = tg + (i >> 6) * (FT * 64) + (i & 63);
  float v[FS + FT], o[FS];
#pragma unroll
  for (int f = 0; f < FS; ++f) v[f] = NT ? __builtin_nontemporal_load(ps + f * 64) : ps[f * 64];
#pragma unroll
  for (int f = 0; f < FT; ++f) v[FS + f] = NT ? __builtin_nontemporal_load(pt + f * 64) : pt[f * 64];
  mix(v, FS + FT, o, FS);
#pragma unroll
  for (int f = 0; f < FS; ++f) { if (NT) __builtin_nontemporal_store(o[f], ps + f * 64); else ps[f * 64] = o[f]; }
}

// K1b: generic blocked SoA [n/B][F][B], dword per lane, TPB threads per block
template <int B, int TPB, bool NT>
__global__ __launch_bounds__(TPB) void k_blk(float* st, const float* tg, long long n) {
  const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  float* ps = st + (i / B) * (long long)(FS * B) + (i % B);
  const float* pt = tg + (i / B) * (long long)(FT * B) + (i % B);
  float v[FS + FT], o[FS];
#pragma unroll
  for (int f = 0; f < FS; ++f) v[f] = NT ? __builtin_nontemporal_load(ps + f * B) : ps[f * B];
#pragma unroll
  for (int f = 0; f < FT; ++f) v[FS + f] = NT ? __builtin_nontemporal_load(pt + f * B) : pt[f * B];
  mix(v, FS + FT, o, FS);
#pragma unroll
  for (int f = 0; f < FS; ++f) { if (NT) __builtin_nontemporal_store(o[f], ps + f * B); else ps[f * B] = o[f]; }
}


// K1c: k_blk with ~WORK dependent-ish FMAs per lane between the loads and the stores (the real step kernels do ~700 vector
// instructions per drone): does a layout's advantage survive when the waves no longer move in lock step?
template <int B, int TPB, int WORK>
__global__ __launch_bounds__(TPB) void k_blk_work(float* st, const float* tg, long long n) {
  const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  float* ps = st + (i / B) * (long long)(FS * B) + (i % B);
  const float* pt = tg + (i / B) * (long long)(FT * B) + (i % B);
  float v[FS + FT], o[FS];
#pragma unroll
  for (int f = 0; f < FS; ++f) v[f] = __builtin_nontemporal_load(ps + f * B);
#pragma unroll
  for (int f = 0; f < FT; ++f) v[FS + f] = __builtin_nontemporal_load(pt + f * B);
  float a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3];
#pragma unroll 8
  for (int k = 0; k < WORK / 4; ++k) {
    a0 = fmaf(a0, 0.999f, v[(k) % (FS + FT)]); a1 = fmaf(a1, 0.998f, a0); a2 = fmaf(a2, 0.997f, a1); a3 = fmaf(a3, 0.996f, a2);
  }
  mix(v, FS + FT, o, FS);
  o[0] += (a0 + a1 + a2 + a3) * 1e-12f;
#pragma unroll
  for (int f = 0; f < FS; ++f) __builtin_nontemporal_store(o[f], ps + f * B);
}

// K1d: the headline shape with the new state written to ANOTHER array (ping-pong) instead of in place
__global__ __launch_bounds__(256) void k_tile_pp(const float* st, const float* tg, float* out, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float* ps = st + (i >> 6) * (FS * 64) + (i & 63);
  const float* pt = tg + (i >> 6) * (FT * 64) + (i & 63);
  float* po = out + (i >> 6) * (FS * 64) + (i & 63);
  float v[FS + FT], o[FS];
#pragma unroll
  for (int f = 0; f < FS; ++f) v[f] = __builtin_nontemporal_load(ps + f * 64);
#pragma unroll
  for (int f = 0; f < FT; ++f) v[FS + f] = __builtin_nontemporal_load(pt + f * 64);
  mix(v, FS + FT, o, FS);
#pragma unroll
  for (int f = 0; f < FS; ++f) __builtin_nontemporal_store(o[f], po + f * 64);
}

// K2: wave tiles, 16 B per lane global accesses, transposed through wave-private LDS
template <bool NT>
__global__ __launch_bounds__(256) void k_tile_lds(float* st, const float* tg, long long n) {
  __shared__ f4 lds4[4][(FS + FT) * 16];            // per wave: 34 rows x 64 floats
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long tile = (long long)blockIdx.x * 4 + wave;
  if (tile * 64 >= n) return;
  f4* gs = reinterpret_cast<f4*>(st + tile * (FS * 64));
  const f4* gt = reinterpret_cast<const f4*>(tg + tile * (FT * 64));
  f4* L = lds4[wave];
  f4 a[6], b[3];
#pragma unroll
  for (int j = 0; j < 6; ++j) a[j] = NT ? __builtin_nontemporal_load(gs + j * 64 + lane) : gs[j * 64 + lane];
#pragma unroll
  for (int j = 0; j < 3; ++j) if (j * 64 + lane < FT * 16) b[j] = NT ? __builtin_nontemporal_load(gt + j * 64 + lane) : gt[j * 64 + lane];
#pragma unroll
  for (int j = 0; j < 6; ++j) L[j * 64 + lane] = a[j];
#pragma unroll
  for (int j = 0; j < 3; ++j) if (j * 64 + lane < FT * 16) L[FS * 16 + j * 64 + lane] = b[j];
  __builtin_amdgcn_wave_barrier();
  const float* Lf = reinterpret_cast<const float*>(L);
  float v[FS + FT], o[FS];
#pragma unroll
  for (int f = 0; f < FS + FT; ++f) v[f] = Lf[f * 64 + lane];
  mix(v, FS + FT, o, FS);
  __builtin_amdgcn_wave_barrier();
  float* Lw = reinterpret_cast<float*>(L);
#pragma unroll
  for (int f = 0; f < FS; ++f) Lw[f * 64 + lane] = o[f];
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int j = 0; j < 6; ++j) { f4 w = L[j * 64 + lane]; if (NT) __builtin_nontemporal_store(w, gs + j * 64 + lane); else gs[j * 64 + lane] = w; }
}

// K3: pure copy yardstick, float4, same byte count split 34 read / 24 write is impossible for a copy;
// so: read R bytes, write W bytes with R:W = 34:24 by reading two arrays and writing one
__global__ __launch_bounds__(256) void k_copy4(const float4* a, float4* b, long long n4) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n4) b[i] = a[i];
}

// K4 / K5: one-directional yardsticks.  HBM3E on MI355X does not serve reads and writes at the same rate, so a kernel's
// floor depends on its read : write mix (bench.py prints both and the floor of the headline mix).
__global__ __launch_bounds__(256) void k_read4(const f4* a, float* sink, long long n4) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  f4 v = {0.f, 0.f, 0.f, 0.f};
  if (i < n4) v = __builtin_nontemporal_load(a + i);
  const float s = v.x + v.y + v.z + v.w;
  if (s == 12345.678f) sink[0] = s;            // never true: keeps the load alive without a store
}
__global__ __launch_bounds__(256) void k_write4(f4* b, long long n4, float x) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const f4 v = {x, x + 1.f, x + 2.f, x + 3.f};
  if (i < n4) __builtin_nontemporal_store(v, b + i);
}


// K6: the access shape of Env.step WITH its observation (k_physics_fast): per drone read 13 state floats (wave tiles) + 4
// action floats (SoA), write 13 state + 4 echoed action floats + one row-major [n][20] observation row — 68 B read,
// 148 B written.  ROWS16: the rows leave through a wave-private LDS block as five 16-byte stores per lane over the wave's
// contiguous 5 120 bytes (what the product does); otherwise 20 dword stores per lane, 80 bytes apart.
template <bool ROWS16>
__global__ __launch_bounds__(256) void k_env_shape(float* st, const float* act, float* echo, float* rows, long long n) {
  __shared__ f4 blk[4][20 * 16];
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* ps = st + (i >> 6) * (FS * 64) + (i & 63);
  float v[17], o[13];
#pragma unroll
  for (int f = 0; f < 13; ++f) v[f] = __builtin_nontemporal_load(ps + f * 64);
#pragma unroll
  for (int f = 0; f < 4; ++f) v[13 + f] = __builtin_nontemporal_load(act + f * n + i);
  mix(v, 17, o, 13);
#pragma unroll
  for (int f = 0; f < 13; ++f) __builtin_nontemporal_store(o[f], ps + f * 64);
#pragma unroll
  for (int f = 0; f < 4; ++f) __builtin_nontemporal_store(v[13 + f] + o[0] * 1e-9f, echo + f * n + i);
  if (ROWS16) {
    float* L = reinterpret_cast<float*>(blk[wave]);
#pragma unroll
    for (int f = 0; f < 20; ++f) L[lane * 20 + f] = f < 13 ? o[f] : v[f - 7] + o[1];
    __builtin_amdgcn_wave_barrier();
    f4* g = reinterpret_cast<f4*>(rows + (i - lane) * 20);
#pragma unroll
    for (int j = 0; j < 5; ++j) __builtin_nontemporal_store(blk[wave][j * 64 + lane], g + j * 64 + lane);
  } else {
#pragma unroll
    for (int f = 0; f < 20; ++f) __builtin_nontemporal_store(f < 13 ? o[f] : v[f - 7] + o[1], rows + i * 20 + f);
  }
}

// K6b: the same shape with the new state written to ANOTHER block (ping-pong) instead of in place: is it the read-modify-write
// of the state beside the rows' write stream that the memory system dislikes, or reading and writing the same region at all?
__global__ __launch_bounds__(256) void k_env_shape_pp(const float* st, float* st_out, const float* act, float* echo, float* rows, long long n) {
  __shared__ f4 blk[4][20 * 16];
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float* ps = st + (i >> 6) * (FS * 64) + (i & 63);
  float* po = st_out + (i >> 6) * (FS * 64) + (i & 63);
  float v[17], o[13];
#pragma unroll
  for (int f = 0; f < 13; ++f) v[f] = __builtin_nontemporal_load(ps + f * 64);
#pragma unroll
  for (int f = 0; f < 4; ++f) v[13 + f] = __builtin_nontemporal_load(act + f * n + i);
  mix(v, 17, o, 13);
#pragma unroll
  for (int f = 0; f < 13; ++f) __builtin_nontemporal_store(o[f], po + f * 64);
#pragma unroll
  for (int f = 0; f < 4; ++f) __builtin_nontemporal_store(v[13 + f] + o[0] * 1e-9f, echo + f * n + i);
  float* L = reinterpret_cast<float*>(blk[wave]);
#pragma unroll
  for (int f = 0; f < 20; ++f) L[lane * 20 + f] = f < 13 ? o[f] : v[f - 7] + o[1];
  __builtin_amdgcn_wave_barrier();
  f4* g = reinterpret_cast<f4*>(rows + (i - lane) * 20);
#pragma unroll
  for (int j = 0; j < 5; ++j) __builtin_nontemporal_store(blk[wave][j * 64 + lane], g + j * 64 + lane);
}

// K6c: the Env.step shape under other cache policies / store orders — does any of them not care where the rows lie?
//   MODE 0 all streaming (the product)   1 rows with the default policy   2 the state with the default policy (loads and stores)
//   3 everything default   4 all streaming, the rows stored BEFORE the state   5 state loads default, every store streaming
template <int MODE>
__global__ __launch_bounds__(256) void k_env_shape_v(float* st, const float* act, float* echo, float* rows, long long n) {
  __shared__ f4 blk[4][20 * 16];
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* ps = st + (i >> 6) * (FS * 64) + (i & 63);
  constexpr bool ST_LD_NT = MODE == 0 || MODE == 1 || MODE == 4, ST_ST_NT = MODE == 0 || MODE == 1 || MODE == 4 || MODE == 5;
  constexpr bool ROWS_NT = MODE == 0 || MODE == 2 || MODE == 4 || MODE == 5;
  float v[17], o[13];
#pragma unroll
  for (int f = 0; f < 13; ++f) v[f] = ST_LD_NT ? __builtin_nontemporal_load(ps + f * 64) : ps[f * 64];
#pragma unroll
  for (int f = 0; f < 4; ++f) v[13 + f] = __builtin_nontemporal_load(act + f * n + i);
  mix(v, 17, o, 13);
  float* L = reinterpret_cast<float*>(blk[wave]);
#pragma unroll
  for (int f = 0; f < 20; ++f) L[lane * 20 + f] = f < 13 ? o[f] : v[f - 7] + o[1];
  __builtin_amdgcn_wave_barrier();
  f4* g = reinterpret_cast<f4*>(rows + (i - lane) * 20);
  if (MODE == 4) {
#pragma unroll
    for (int j = 0; j < 5; ++j) __builtin_nontemporal_store(blk[wave][j * 64 + lane], g + j * 64 + lane);
  }
#pragma unroll
  for (int f = 0; f < 13; ++f) { if (ST_ST_NT) __builtin_nontemporal_store(o[f], ps + f * 64); else ps[f * 64] = o[f]; }
#pragma unroll
  for (int f = 0; f < 4; ++f) __builtin_nontemporal_store(v[13 + f] + o[0] * 1e-9f, echo + f * n + i);
  if (MODE != 4) {
#pragma unroll
    for (int j = 0; j < 5; ++j) { if (ROWS_NT) __builtin_nontemporal_store(blk[wave][j * 64 + lane], g + j * 64 + lane); else g[j * 64 + lane] = blk[wave][j * 64 + lane]; }
  }
}

// K7: two buffers streamed side by side, one float4 per lane from / to each: MODE 0 read a + read b, 1 read a + write b,
// 2 write a + write b, 3 read+write a + write b, 4 read+write a + read b
template <int MODE>
__global__ __launch_bounds__(256) void k_pair(f4* a, f4* b, float* sink, long long n4) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  f4 va = {1.f, 2.f, 3.f, 4.f}, vb = {0.f, 0.f, 0.f, 0.f};
  if (MODE == 0 || MODE == 1 || MODE == 3 || MODE == 4) va = __builtin_nontemporal_load(a + i);
  if (MODE == 0 || MODE == 4) vb = __builtin_nontemporal_load(b + i);
  if (MODE == 3 || MODE == 4 || MODE == 2) __builtin_nontemporal_store(va + vb * 1e-9f, a + i);
  if (MODE == 1 || MODE == 2 || MODE == 3) __builtin_nontemporal_store(va * 1.0001f, b + i);
  if (MODE == 0) { const float s_ = va.x + vb.y; if (s_ == 12345.678f) sink[0] = s_; }
}

template <typename F>
float time_it(F f, int iters) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) f();
  std::vector<float> ts;
  for (int r = 0; r < 5; ++r) {
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    ts.push_back(ms / iters);
  }
  std::sort(ts.begin(), ts.end());
  return ts[2];
}

int main(int argc, char** argv) {
  bool json = false;
  long long n = 1LL << 22;
  for (int k = 1; k < argc; ++k) { if (!strcmp(argv[k], "--json")) json = true; else if (argv[k][0] != '-') n = atoll(argv[k]); }
  bool placement = false;
  for (int k = 1; k < argc; ++k) if (!strcmp(argv[k], "--placement")) placement = true;
  if (placement) {
    // does WHERE the arrays lie matter?  The same two kernels on freshly allocated buffers, each time behind a dummy
    // allocation of a different size (which shifts every later base address)
    const dim3 g((unsigned)((n + 255) / 256)), b(256);
    const size_t pads[] = {0, 4096, 65536, 1 << 20, (1 << 21) + 12288, 3 << 20, (5 << 20) + 256, 1 << 24};
    for (size_t pad : pads) {
      void* dummy = nullptr;
      if (pad) CK(hipMalloc(&dummy, pad));
      float *st, *tg, *act, *echo, *rows;
      CK(hipMalloc(&st, sizeof(float) * FS * n)); CK(hipMalloc(&tg, sizeof(float) * FT * n));
      CK(hipMalloc(&act, sizeof(float) * 4 * n)); CK(hipMalloc(&echo, sizeof(float) * 4 * n)); CK(hipMalloc(&rows, sizeof(float) * 20 * n));
      CK(hipMemset(st, 0, sizeof(float) * FS * n)); CK(hipMemset(tg, 0, sizeof(float) * FT * n)); CK(hipMemset(act, 0, sizeof(float) * 4 * n));
      const float t1 = time_it([&] { hipLaunchKernelGGL(k_tile<true>, g, b, 0, 0, st, tg, n); }, 20);
      const float t2 = time_it([&] { hipLaunchKernelGGL((k_blk<1024, 256, true>), g, b, 0, 0, st, tg, n); }, 20);
      const float t3 = time_it([&] { hipLaunchKernelGGL(k_env_shape<true>, g, b, 0, 0, st, act, echo, rows, n); }, 20);
      const long long s4 = FS * n / 4, r4 = 20 * n / 4;
      const float t4 = time_it([&] { hipLaunchKernelGGL(k_read4, dim3((unsigned)((s4 + 255) / 256)), b, 0, 0, (const f4*)st, tg, s4); }, 20);
      const float t5 = time_it([&] { hipLaunchKernelGGL(k_write4, dim3((unsigned)((s4 + 255) / 256)), b, 0, 0, (f4*)st, s4, 1.0f); }, 20);
      const float t6 = time_it([&] { hipLaunchKernelGGL(k_write4, dim3((unsigned)((r4 + 255) / 256)), b, 0, 0, (f4*)rows, r4, 1.0f); }, 20);
      const float t7 = time_it([&] { hipLaunchKernelGGL(k_copy4, dim3((unsigned)((r4 + 255) / 256)), b, 0, 0, (const float4*)st, (float4*)rows, r4); }, 20);
      printf("pad %9zu  tile64 %.1f  blk1024 %.1f  Env.step shape %.1f us | GB/s: read st %.0f  write st %.0f  write rows %.0f  copy st->rows %.0f\n", pad,
             t1 * 1e3, t2 * 1e3, t3 * 1e3, 16.0 * s4 / t4 / 1e6, 16.0 * s4 / t5 / 1e6, 16.0 * r4 / t6 / 1e6, 32.0 * r4 / t7 / 1e6);
      CK(hipFree(st)); CK(hipFree(tg)); CK(hipFree(act)); CK(hipFree(echo)); CK(hipFree(rows));
      if (dummy) CK(hipFree(dummy));
    }
    return 0;
  }
  bool sets = false;
  for (int k = 1; k < argc; ++k) if (!strcmp(argv[k], "--sets")) sets = true;
  if (sets) {
    // four sets of buffers alive at once: is a set lucky or unlucky as a whole, and do buffers of different sets mix?
    const dim3 g((unsigned)((n + 255) / 256)), b(256);
    constexpr int S = 4;
    float *st[S], *tg[S], *act[S], *echo[S], *rows[S];
    for (int k = 0; k < S; ++k) {
      CK(hipMalloc(&st[k], sizeof(float) * FS * n)); CK(hipMalloc(&tg[k], sizeof(float) * FT * n));
      CK(hipMalloc(&act[k], sizeof(float) * 4 * n)); CK(hipMalloc(&echo[k], sizeof(float) * 4 * n)); CK(hipMalloc(&rows[k], sizeof(float) * 20 * n));
      CK(hipMemset(st[k], 0, sizeof(float) * FS * n)); CK(hipMemset(tg[k], 0, sizeof(float) * FT * n)); CK(hipMemset(act[k], 0, sizeof(float) * 4 * n));
      printf("set %d: st %p tg %p act %p echo %p rows %p\n", k, (void*)st[k], (void*)tg[k], (void*)act[k], (void*)echo[k], (void*)rows[k]);
    }
    for (int rep = 0; rep < 2; ++rep)
      for (int k = 0; k < S; ++k) {
        const float t1 = time_it([&] { hipLaunchKernelGGL(k_tile<true>, g, b, 0, 0, st[k], tg[k], n); }, 20);
        const float t3 = time_it([&] { hipLaunchKernelGGL(k_env_shape<true>, g, b, 0, 0, st[k], act[k], echo[k], rows[k], n); }, 20);
        printf("set %d: tile64 %.1f us  Env.step shape %.1f us\n", k, t1 * 1e3, t3 * 1e3);
      }
    for (int k = 0; k < S; ++k)
      for (int j = 0; j < S; ++j) {
        const float t1 = time_it([&] { hipLaunchKernelGGL(k_tile<true>, g, b, 0, 0, st[k], tg[j], n); }, 20);
        const float t3 = time_it([&] { hipLaunchKernelGGL(k_env_shape<true>, g, b, 0, 0, st[k], act[k], echo[k], rows[j], n); }, 20);
        printf("st of set %d with tg / rows of set %d: tile64 %.1f us  Env.step shape %.1f us\n", k, j, t1 * 1e3, t3 * 1e3);
      }
    return 0;
  }
  bool regions = false;
  for (int k = 1; k < argc; ++k) if (!strcmp(argv[k], "--regions")) regions = true;
  if (regions) {
    // 40 state-sized chunks allocated one after the other (15 GB): every one as the state block of the Env.step shape
    // against ONE fixed set of rows / action buffers, and as the state of the headline shape against one fixed target
    // block.  How does a chunk's class follow the order (= the physical region) it was allocated in?
    const dim3 g((unsigned)((n + 255) / 256)), b(256);
    float *tg, *act, *echo, *rows;
    CK(hipMalloc(&tg, sizeof(float) * FT * n)); CK(hipMalloc(&act, sizeof(float) * 4 * n));
    CK(hipMalloc(&echo, sizeof(float) * 4 * n)); CK(hipMalloc(&rows, sizeof(float) * 20 * n));
    CK(hipMemset(tg, 0, sizeof(float) * FT * n)); CK(hipMemset(act, 0, sizeof(float) * 4 * n));
    constexpr int C = 40;
    float* st[C];
    for (int k = 0; k < C; ++k) { CK(hipMalloc(&st[k], sizeof(float) * FS * n)); CK(hipMemset(st[k], 0, sizeof(float) * FS * n)); }
    for (int k = 0; k < C; ++k) {
      const float t1 = time_it([&] { hipLaunchKernelGGL(k_tile<true>, g, b, 0, 0, st[k], tg, n); }, 10);
      const float t3 = time_it([&] { hipLaunchKernelGGL(k_env_shape<true>, g, b, 0, 0, st[k], act, echo, rows, n); }, 10);
      printf("chunk %2d (after %5.0f MiB) va %p: tile64 %.1f us  Env.step shape %.1f us\n", k, 608.0 + 384.0 * k, (void*)st[k], t1 * 1e3, t3 * 1e3);
    }
    return 0;
  }
  bool pairs = false;
  for (int k = 1; k < argc; ++k) if (!strcmp(argv[k], "--pairs")) pairs = true;
  if (pairs) {
    // which kinds of side-by-side streams care about the class of their buffers?  24 chunks of 256 MiB; classes from the
    // Env.step shape against chunk 0, then every pair mode on (same class) and (different class) pairs
    const dim3 b(256);
    constexpr int C = 48;
    const size_t bytes = 256u << 20;
    const long long n4 = bytes / 16;
    const dim3 g4((unsigned)((n4 + 255) / 256));
    f4* ch[C]; float* sink; CK(hipMalloc(&sink, 256));
    for (int k = 0; k < C; ++k) { CK(hipMalloc(&ch[k], bytes)); CK(hipMemset(ch[k], 0, bytes)); }
    float t3[C];
    int other = -1;
    for (int k = 1; k < C; ++k) {
      t3[k] = time_it([&] { hipLaunchKernelGGL(k_pair<3>, g4, b, 0, 0, ch[0], ch[k], sink, n4); }, 10);
      printf("chunk %2d vs chunk 0, read+write a / write b: %.1f us\n", k, t3[k] * 1e3);
    }
    float lo = 1e9f, hi = 0.f;
    for (int k = 1; k < C; ++k) { lo = t3[k] < lo ? t3[k] : lo; hi = t3[k] > hi ? t3[k] : hi; }
    int same = -1;
    for (int k = 1; k < C; ++k) {
      if (same < 0 && t3[k] > 0.5f * (lo + hi)) same = k;
      if (other < 0 && t3[k] < 0.5f * (lo + hi)) other = k;
    }
    printf("spread %.1f .. %.1f us; taking chunk %d as the slow partner of chunk 0 and chunk %d as the fast one\n", lo * 1e3, hi * 1e3, same, other);
    {
      // the whole relation among 12 chunks spread over the allocation order
      int pick[12];
      for (int k = 0; k < 12; ++k) pick[k] = k * 4;
      for (int mode : {3, 4}) {
        printf("matrix, %s (rows: a, columns: b; us)\n      ", mode == 3 ? "read+write a + write b" : "read+write a + read b");
        for (int j = 0; j < 12; ++j) printf("%6d", pick[j]);
        printf("\n");
        for (int i = 0; i < 12; ++i) {
          printf("%4d: ", pick[i]);
          for (int j = 0; j < 12; ++j) {
            if (i == j) { printf("     -"); continue; }
            f4 *pa = ch[pick[i]], *pb = ch[pick[j]];
            const float t = mode == 3 ? time_it([&] { hipLaunchKernelGGL(k_pair<3>, g4, b, 0, 0, pa, pb, sink, n4); }, 6)
                                      : time_it([&] { hipLaunchKernelGGL(k_pair<4>, g4, b, 0, 0, pa, pb, sink, n4); }, 6);
            printf("%6.1f", t * 1e3);
          }
          printf("\n");
        }
      }
    }
    if (same < 0 || other < 0) return 0;
    const char* names[5] = {"read a + read b", "read a + write b", "write a + write b", "read+write a + write b", "read+write a + read b"};
    for (int rep = 0; rep < 2; ++rep) {
      float ts[5][2];
      for (int j = 0; j < 2; ++j) {
        f4* pb = ch[j == 0 ? same : other];
        ts[0][j] = time_it([&] { hipLaunchKernelGGL(k_pair<0>, g4, b, 0, 0, ch[0], pb, sink, n4); }, 10);
        ts[1][j] = time_it([&] { hipLaunchKernelGGL(k_pair<1>, g4, b, 0, 0, ch[0], pb, sink, n4); }, 10);
        ts[2][j] = time_it([&] { hipLaunchKernelGGL(k_pair<2>, g4, b, 0, 0, ch[0], pb, sink, n4); }, 10);
        ts[3][j] = time_it([&] { hipLaunchKernelGGL(k_pair<3>, g4, b, 0, 0, ch[0], pb, sink, n4); }, 10);
        ts[4][j] = time_it([&] { hipLaunchKernelGGL(k_pair<4>, g4, b, 0, 0, ch[0], pb, sink, n4); }, 10);
      }
      for (int m = 0; m < 5; ++m) printf("%-24s  with the slow partner %.1f us   with the fast partner %.1f us\n", names[m], ts[m][0] * 1e3, ts[m][1] * 1e3);
    }
    return 0;
  }
  bool ballast = false;
  for (int k = 1; k < argc; ++k) if (!strcmp(argv[k], "--ballast")) ballast = true;
  if (ballast) {
    // what tools/placement_probe.py does with the product, with the access-shape kernel instead of k_physics_fast: one
    // fixed state / action / echo set, candidate row arrays with 2 GiB of (untouched) ballast between them
    const dim3 g((unsigned)((n + 255) / 256)), b(256);
    float *st, *act, *echo, *sink;
    CK(hipMalloc(&st, sizeof(float) * FS * n)); CK(hipMalloc(&act, sizeof(float) * 4 * n)); CK(hipMalloc(&echo, sizeof(float) * 4 * n));
    CK(hipMalloc(&sink, 256));
    CK(hipMemset(st, 0, sizeof(float) * FS * n)); CK(hipMemset(act, 0, sizeof(float) * 4 * n));
    const long long n4 = 20 * n / 4;
    const dim3 g4((unsigned)((n4 + 255) / 256));
    for (int k = 0; k < 10; ++k) {
      float* rows; CK(hipMalloc(&rows, sizeof(float) * 20 * n)); CK(hipMemset(rows, 0, sizeof(float) * 20 * n));
      const float t3 = time_it([&] { hipLaunchKernelGGL(k_env_shape<true>, g, b, 0, 0, st, act, echo, rows, n); }, 6);
      const float tp = time_it([&] { hipLaunchKernelGGL(k_pair<3>, g4, b, 0, 0, (f4*)st, (f4*)rows, sink, n4); }, 6);
      printf("candidate %d  rows %p (st %p)  Env.step shape %.1f us   pair rw st + w rows %.1f us\n", k, (void*)rows, (void*)st, t3 * 1e3, tp * 1e3);
      void* bal; CK(hipMalloc(&bal, 2ull << 30));
    }
    return 0;
  }
  bool bigsweep = false;
  for (int k = 1; k < argc; ++k) if (!strcmp(argv[k], "--bigsweep")) bigsweep = true;
  if (bigsweep) {
    // ONE 24 GB allocation; the state block at its start, the rows D bytes further on, D in steps of 128 MiB: is the
    // behaviour a function of the distance inside one allocation?
    const dim3 g((unsigned)((n + 255) / 256)), b(256);
    // (DSIM_SWEEP_GIB: a larger allocation and coarser steps, to look for windows wider than 16 GiB)
    const char* sg = getenv("DSIM_SWEEP_GIB");
    const size_t total = (sg ? (size_t)atoi(sg) : 24ull) << 30, sz_st = sizeof(float) * FS * n, sz_a = sizeof(float) * 4 * n, sz_r = sizeof(float) * 20 * n;
    char* arena; CK(hipMalloc(&arena, total)); CK(hipMemset(arena, 0, total));
    float* st = (float*)arena; float* act = (float*)(arena + sz_st); float* echo = (float*)(arena + sz_st + sz_a);
    float* sink; CK(hipMalloc(&sink, 256));
    const long long n4 = (256u << 20) / 16;
    const dim3 g4((unsigned)((n4 + 255) / 256));
    for (size_t D = sz_st + 2 * sz_a; D + sz_r <= total; D += (sg ? 4096ull : 512ull) << 20) {
      float* rows = (float*)(arena + D);
      const float t3 = time_it([&] { hipLaunchKernelGGL(k_env_shape<true>, g, b, 0, 0, st, act, echo, rows, n); }, 6);
      const float tp = time_it([&] { hipLaunchKernelGGL(k_pair<3>, g4, b, 0, 0, (f4*)arena, (f4*)(arena + D), sink, n4); }, 6);
      const float tq = time_it([&] { hipLaunchKernelGGL(k_pair<4>, g4, b, 0, 0, (f4*)arena, (f4*)(arena + D), sink, n4); }, 6);
      float tpp = 0.f, tin = 0.f;
      if (D + sz_st <= total) {
        const float* tgp = (const float*)(arena + sz_st);            // targets right behind the state (inside act / echo's place)
        tpp = time_it([&] { hipLaunchKernelGGL(k_tile_pp, g, b, 0, 0, (const float*)st, tgp, (float*)(arena + D), n); }, 6);
        tin = time_it([&] { hipLaunchKernelGGL(k_tile<true>, g, b, 0, 0, st, tgp, n); }, 6);
      }
      printf("D = %6zu MiB  Env.step shape %.1f us   pair rw a + w b %.1f us   pair rw a + r b %.1f us   headline shape in place %.1f us, written to D %.1f us\n", D >> 20, t3 * 1e3, tp * 1e3, tq * 1e3, tin * 1e3, tpp * 1e3);
    }
    return 0;
  }
  bool envvar = false;
  for (int k = 1; k < argc; ++k) if (!strcmp(argv[k], "--envvar")) envvar = true;
  if (envvar) {
    // ONE allocation (DSIM_SWEEP_GIB, default 40): the state block at its start, the rows D bytes on; K6c's modes
    const dim3 g((unsigned)((n + 255) / 256)), b(256);
    const char* sg = getenv("DSIM_SWEEP_GIB");
    const size_t total = (sg ? (size_t)atoi(sg) : 40ull) << 30, sz_st = sizeof(float) * FS * n, sz_a = sizeof(float) * 4 * n, sz_r = sizeof(float) * 20 * n;
    char* arena; CK(hipMalloc(&arena, total)); CK(hipMemset(arena, 0, total));
    float* st = (float*)arena; float* act = (float*)(arena + sz_st); float* echo = (float*)(arena + sz_st + sz_a);
    const size_t GiB = 1ull << 30;
    for (size_t D : {2 * GiB, 8 * GiB, 14 * GiB, 18 * GiB, 26 * GiB, 34 * GiB}) {
      if (D + sz_r > total) continue;
      float* rows = (float*)(arena + D);
      const float t0 = time_it([&] { hipLaunchKernelGGL(k_env_shape_v<0>, g, b, 0, 0, st, act, echo, rows, n); }, 6);
      const float t1 = time_it([&] { hipLaunchKernelGGL(k_env_shape_v<1>, g, b, 0, 0, st, act, echo, rows, n); }, 6);
      const float t2 = time_it([&] { hipLaunchKernelGGL(k_env_shape_v<2>, g, b, 0, 0, st, act, echo, rows, n); }, 6);
      const float t3 = time_it([&] { hipLaunchKernelGGL(k_env_shape_v<3>, g, b, 0, 0, st, act, echo, rows, n); }, 6);
      const float t4 = time_it([&] { hipLaunchKernelGGL(k_env_shape_v<4>, g, b, 0, 0, st, act, echo, rows, n); }, 6);
      const float t5 = time_it([&] { hipLaunchKernelGGL(k_env_shape_v<5>, g, b, 0, 0, st, act, echo, rows, n); }, 6);
      printf("rows at %2zu GiB:  all streaming %.1f   rows default %.1f   state default %.1f   all default %.1f   rows first %.1f   state loads default %.1f us\n",
             D >> 30, t0 * 1e3, t1 * 1e3, t2 * 1e3, t3 * 1e3, t4 * 1e3, t5 * 1e3);
    }
    return 0;
  }
  bool envpp = false;
  for (int k = 1; k < argc; ++k) if (!strcmp(argv[k], "--envpp")) envpp = true;
  if (envpp) {
    // ONE allocation (DSIM_SWEEP_GIB, default 40): the state block at its start; Env.step's shape in place and ping-pong
    // (new state written E bytes on), rows D bytes on
    const dim3 g((unsigned)((n + 255) / 256)), b(256);
    const char* sg = getenv("DSIM_SWEEP_GIB");
    const size_t total = (sg ? (size_t)atoi(sg) : 40ull) << 30, sz_st = sizeof(float) * FS * n, sz_a = sizeof(float) * 4 * n, sz_r = sizeof(float) * 20 * n;
    char* arena; CK(hipMalloc(&arena, total)); CK(hipMemset(arena, 0, total));
    float* st = (float*)arena; float* act = (float*)(arena + sz_st); float* echo = (float*)(arena + sz_st + sz_a);
    const size_t GiB = 1ull << 30;
    for (size_t E : {2 * GiB, 20 * GiB, 36 * GiB}) {
      for (size_t D : {4 * GiB, 12 * GiB, 18 * GiB, 24 * GiB, 34 * GiB, 38 * GiB}) {
        if (D + sz_r > total || E + sz_st > total) continue;
        float* rows = (float*)(arena + D);
        float* st2 = (float*)(arena + E);
        const float t_in = time_it([&] { hipLaunchKernelGGL(k_env_shape<true>, g, b, 0, 0, st, act, echo, rows, n); }, 6);
        const float t_pp = time_it([&] { hipLaunchKernelGGL(k_env_shape_pp, g, b, 0, 0, (const float*)st, st2, act, echo, rows, n); }, 6);
        printf("new state at %2zu GiB, rows at %2zu GiB:  in place %.1f us   ping-pong %.1f us\n", E >> 30, D >> 30, t_in * 1e3, t_pp * 1e3);
      }
    }
    return 0;
  }
  bool sweep = false;
  for (int k = 1; k < argc; ++k) if (!strcmp(argv[k], "--sweep")) sweep = true;
  if (sweep) {
    // one arena; the state block at its start, the target block `off` bytes behind the state's end: which relative
    // placements of two concurrently streamed arrays does the memory system like?
    const dim3 g((unsigned)((n + 255) / 256)), b(256);
    const size_t sz_st = sizeof(float) * FS * n, sz_tg = sizeof(float) * FT * n, slack = 64u << 20;
    char* arena; CK(hipMalloc(&arena, sz_st + sz_tg + slack)); CK(hipMemset(arena, 0, sz_st + sz_tg + slack));
    const size_t offs[] = {0, 256, 1024, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 524288, 1 << 20, 2 << 20, 3 << 20, 4 << 20,
                           6 << 20, 8 << 20, 12 << 20, 16 << 20, 24 << 20, 32 << 20, 48 << 20};
    for (size_t off : offs) {
      float* st = (float*)arena; float* tg = (float*)(arena + sz_st + off);
      const float t1 = time_it([&] { hipLaunchKernelGGL(k_tile<true>, g, b, 0, 0, st, tg, n); }, 20);
      printf("tg - st_end = %9zu  (tg - st) %% 16MB = %9zu  tile64 %.1f us\n", off, (size_t)((char*)tg - (char*)st) % (16u << 20), t1 * 1e3);
    }
    CK(hipFree(arena));
    // the Env.step shape: state | act | echo | rows in one arena, the rows `off` bytes behind the echo's end
    const size_t sz_a = sizeof(float) * 4 * n, sz_r = sizeof(float) * 20 * n;
    CK(hipMalloc(&arena, sz_st + 2 * sz_a + sz_r + 2 * slack)); CK(hipMemset(arena, 0, sz_st + 2 * sz_a + sz_r + 2 * slack));
    for (size_t off2 : {(size_t)0, (size_t)(1 << 20) + 4096}) for (size_t off : offs) {
      float* st = (float*)arena; float* act = (float*)(arena + sz_st + off2); float* echo = (float*)(arena + sz_st + off2 + sz_a);
      float* rows = (float*)(arena + sz_st + off2 + 2 * sz_a + off);
      const float t3 = time_it([&] { hipLaunchKernelGGL(k_env_shape<true>, g, b, 0, 0, st, act, echo, rows, n); }, 20);
      printf("act - st_end = %8zu  rows - echo_end = %9zu  Env.step shape %.1f us\n", off2, off, t3 * 1e3);
    }
    return 0;
  }
  if (json) {
    // the two yardsticks bench.py prints beside the headline: the float4 copy rate of this device, and the floor of the
    // headline kernel's ACCESS SHAPE (58 dword accesses per lane on the wave-tiled layout, streaming, no arithmetic)
    float *st, *tg;
    CK(hipMalloc(&st, sizeof(float) * FS * n));
    CK(hipMalloc(&tg, sizeof(float) * FT * n));
    CK(hipMemset(st, 0, sizeof(float) * FS * n));
    CK(hipMemset(tg, 0, sizeof(float) * FT * n));
    const dim3 g((unsigned)((n + 255) / 256)), b(256);
    const float shape_ms = time_it([&] { hipLaunchKernelGGL(k_tile<true>, g, b, 0, 0, st, tg, n); }, 20);
    const long long n4 = FS * n / 4;
    float4* dst; CK(hipMalloc(&dst, sizeof(float4) * n4));
    const float copy_ms = time_it([&] { hipLaunchKernelGGL(k_copy4, dim3((unsigned)((n4 + 255) / 256)), b, 0, 0, (const float4*)st, dst, n4); }, 20);
    const float read_ms = time_it([&] { hipLaunchKernelGGL(k_read4, dim3((unsigned)((n4 + 255) / 256)), b, 0, 0, (const f4*)st, tg, n4); }, 20);
    const float write_ms = time_it([&] { hipLaunchKernelGGL(k_write4, dim3((unsigned)((n4 + 255) / 256)), b, 0, 0, (f4*)dst, n4, 1.0f); }, 20);
    printf("{\"drones\": %lld, \"access_shape_floor_us\": %.2f, \"access_shape_GBps\": %.1f, \"float4_copy_GBps\": %.1f, "
           "\"float4_read_GBps\": %.1f, \"float4_write_GBps\": %.1f}\n", n,
           shape_ms * 1e3, 232.0 * n / (shape_ms * 1e-3) / 1e9, 2.0 * 16 * n4 / (copy_ms * 1e-3) / 1e9,
           16.0 * n4 / (read_ms * 1e-3) / 1e9, 16.0 * n4 / (write_ms * 1e-3) / 1e9);
    return 0;
  }
  float *st, *tg;
  CK(hipMalloc(&st, sizeof(float) * FS * n));
  CK(hipMalloc(&tg, sizeof(float) * FT * n));
  CK(hipMemset(st, 0, sizeof(float) * FS * n));
  CK(hipMemset(tg, 0, sizeof(float) * FT * n));
  const double bytes = 232.0 * n;
  const dim3 g((unsigned)((n + 255) / 256)), b(256);
  auto rep = [&](const char* name, float ms) { printf("%-28s %8.1f us  %7.1f GB/s (algorithmic 232 B/drone)\n", name, ms * 1e3, bytes / (ms * 1e-3) / 1e9); };
  rep("soa dword", time_it([&] { hipLaunchKernelGGL(k_soa<false>, g, b, 0, 0, st, tg, n); }, 20));
  rep("soa dword nt", time_it([&] { hipLaunchKernelGGL(k_soa<true>, g, b, 0, 0, st, tg, n); }, 20));
  rep("tile64 dword", time_it([&] { hipLaunchKernelGGL(k_tile<false>, g, b, 0, 0, st, tg, n); }, 20));
  rep("tile64 dword nt", time_it([&] { hipLaunchKernelGGL(k_tile<true>, g, b, 0, 0, st, tg, n); }, 20));
  rep("tile64 x4 via LDS", time_it([&] { hipLaunchKernelGGL(k_tile_lds<false>, g, b, 0, 0, st, tg, n); }, 20));
  rep("tile64 x4 via LDS nt", time_it([&] { hipLaunchKernelGGL(k_tile_lds<true>, g, b, 0, 0, st, tg, n); }, 20));
  rep("blk256 dword nt", time_it([&] { hipLaunchKernelGGL((k_blk<256, 256, true>), g, b, 0, 0, st, tg, n); }, 20));
  rep("blk1024 dword nt", time_it([&] { hipLaunchKernelGGL((k_blk<1024, 256, true>), g, b, 0, 0, st, tg, n); }, 20));
  rep("blk4096 dword nt", time_it([&] { hipLaunchKernelGGL((k_blk<4096, 256, true>), g, b, 0, 0, st, tg, n); }, 20));
  rep("tile64 nt tpb64", time_it([&] { hipLaunchKernelGGL((k_blk<64, 64, true>), dim3((unsigned)(n / 64)), dim3(64), 0, 0, st, tg, n); }, 20));
  rep("tile64 nt tpb128", time_it([&] { hipLaunchKernelGGL((k_blk<64, 128, true>), dim3((unsigned)(n / 128)), dim3(128), 0, 0, st, tg, n); }, 20));
  rep("tile64 nt tpb512", time_it([&] { hipLaunchKernelGGL((k_blk<64, 512, true>), dim3((unsigned)(n / 512)), dim3(512), 0, 0, st, tg, n); }, 20));
  rep("tile64 nt tpb1024", time_it([&] { hipLaunchKernelGGL((k_blk<64, 1024, true>), dim3((unsigned)(n / 1024)), dim3(1024), 0, 0, st, tg, n); }, 20));
  rep("tile64 nt + 600 FMAs", time_it([&] { hipLaunchKernelGGL((k_blk_work<64, 256, 600>), g, b, 0, 0, st, tg, n); }, 20));
  rep("blk1024 nt + 600 FMAs", time_it([&] { hipLaunchKernelGGL((k_blk_work<1024, 256, 600>), g, b, 0, 0, st, tg, n); }, 20));
  rep("blk4096 nt + 600 FMAs", time_it([&] { hipLaunchKernelGGL((k_blk_work<4096, 256, 600>), g, b, 0, 0, st, tg, n); }, 20));
  rep("tile64 nt + 1200 FMAs", time_it([&] { hipLaunchKernelGGL((k_blk_work<64, 256, 1200>), g, b, 0, 0, st, tg, n); }, 20));
  rep("blk1024 nt + 1200 FMAs", time_it([&] { hipLaunchKernelGGL((k_blk_work<1024, 256, 1200>), g, b, 0, 0, st, tg, n); }, 20));
  rep("blk4096 nt + 1200 FMAs", time_it([&] { hipLaunchKernelGGL((k_blk_work<4096, 256, 1200>), g, b, 0, 0, st, tg, n); }, 20));
  // copy yardstick: 2 x 4 B x n4 bytes moved
  const long long n4 = FS * n / 4;
  float4* dst; CK(hipMalloc(&dst, sizeof(float4) * n4));
  float ms = time_it([&] { hipLaunchKernelGGL(k_copy4, dim3((unsigned)((n4 + 255) / 256)), b, 0, 0, (const float4*)st, dst, n4); }, 20);
  printf("%-28s %8.1f us  %7.1f GB/s (read+write bytes)\n", "float4 copy", ms * 1e3, 2.0 * 16 * n4 / (ms * 1e-3) / 1e9);
  // the Env.step shape (216 B per drone: 68 read, 148 written)
  float *act, *echo, *rows;
  CK(hipMalloc(&act, sizeof(float) * 4 * n)); CK(hipMalloc(&echo, sizeof(float) * 4 * n)); CK(hipMalloc(&rows, sizeof(float) * 20 * n));
  CK(hipMemset(act, 0, sizeof(float) * 4 * n));
  ms = time_it([&] { hipLaunchKernelGGL(k_env_shape<true>, g, b, 0, 0, st, act, echo, rows, n); }, 20);
  printf("%-28s %8.1f us  %7.1f GB/s (algorithmic 216 B/drone)\n", "Env.step shape, rows x4 LDS", ms * 1e3, 216.0 * n / (ms * 1e-3) / 1e9);
  ms = time_it([&] { hipLaunchKernelGGL(k_env_shape<false>, g, b, 0, 0, st, act, echo, rows, n); }, 20);
  printf("%-28s %8.1f us  %7.1f GB/s (algorithmic 216 B/drone)\n", "Env.step shape, rows dword", ms * 1e3, 216.0 * n / (ms * 1e-3) / 1e9);
  return 0;
}
