// membench.hip — dev micro-benchmark (not part of the product): what HBM rate can THIS traffic
// shape reach on MI355X?  Shape of k_step_fast: per drone read 34 floats (24 state + 10 target),
// write 24 floats, one drone per lane.  Variants differ only in layout / access width.
//   hipcc -O3 --offload-arch=gfx950 -o membench tools/membench.hip && ./membench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
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
  for (int k = 1; k < argc; ++k) { if (!strcmp(argv[k], "--json")) json = true; else n = atoll(argv[k]); }
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
  // copy yardstick: 2 x 4 B x n4 bytes moved
  const long long n4 = FS * n / 4;
  float4* dst; CK(hipMalloc(&dst, sizeof(float4) * n4));
  float ms = time_it([&] { hipLaunchKernelGGL(k_copy4, dim3((unsigned)((n4 + 255) / 256)), b, 0, 0, (const float4*)st, dst, n4); }, 20);
  printf("%-28s %8.1f us  %7.1f GB/s (read+write bytes)\n", "float4 copy", ms * 1e3, 2.0 * 16 * n4 / (ms * 1e-3) / 1e9);
  return 0;
}
