// Probe: what an "empty" kernel costs behind a real kernel on the same stream, as a function of its static LDS size
// and grid (the deferred-WLS-fallback kernel with an empty queue).  hipcc --offload-arch=gfx950 -O3 launch_lds.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int LDS_FLOATS>
__global__ __launch_bounds__(64) void k_probe(const int* q, float* out) {
  __shared__ float w[LDS_FLOATS];
  if (*q == 0) return;
  w[threadIdx.x] = 1.0f;
  __syncthreads();
  out[blockIdx.x * 64 + threadIdx.x] = w[(threadIdx.x + 1) & 63];
}
__global__ __launch_bounds__(256) void k_busy(float* x, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) { float v = x[i]; for (int k = 0; k < 64; ++k) v = v * 1.0001f + 0.5f; x[i] = v; }
}
template <int L>
static float run(int grid, hipStream_t st, const int* q, float* out, float* x, long long n, bool with_probe) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 400;
  for (int w = 0; w < 2; ++w) {
    hipEventRecord(e0, st);
    for (int r = 0; r < reps; ++r) {
      hipLaunchKernelGGL(k_busy, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, n);
      if (with_probe) hipLaunchKernelGGL((k_probe<L>), dim3(grid), dim3(64), 0, st, q, out);
    }
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1000.0f / reps;
}
__global__ __launch_bounds__(256) void k_probe256(const int* q, float* out) {
  if (*q == 0) return;
  out[blockIdx.x * 256 + threadIdx.x] = 1.0f;
}
static float run256(int grid, hipStream_t st, const int* q, float* out, float* x, long long n) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 400;
  for (int w = 0; w < 2; ++w) {
    hipEventRecord(e0, st);
    for (int r = 0; r < reps; ++r) {
      hipLaunchKernelGGL(k_busy, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, n);
      hipLaunchKernelGGL(k_probe256, dim3(grid), dim3(256), 0, st, q, out);
    }
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1000.0f / reps;
}
int main() {
  hipStream_t st; hipStreamCreate(&st);
  int* q; float *out, *x; const long long n = 65536 * 16;
  hipMalloc(&q, 4); hipMemset(q, 0, 4); hipMalloc(&out, 4 * 256 * 65536); hipMalloc(&x, 4 * n); hipMemset(x, 0, 4 * n);
  const float base = run<64>(256, st, q, out, x, n, false);
  printf("busy kernel alone: %.2f us per iteration\n", base);
  printf("+ probe LDS 256 B,  grid 256: +%.2f us\n", run<64>(256, st, q, out, x, n, true) - base);
  printf("+ probe LDS 256 B,  grid   1: +%.2f us\n", run<64>(1, st, q, out, x, n, true) - base);
  printf("+ probe LDS 144 KB, grid 256: +%.2f us\n", run<36864>(256, st, q, out, x, n, true) - base);
  printf("+ probe LDS 144 KB, grid  32: +%.2f us\n", run<36864>(32, st, q, out, x, n, true) - base);
  printf("+ probe LDS 144 KB, grid   1: +%.2f us\n", run<36864>(1, st, q, out, x, n, true) - base);
  printf("+ probe LDS 64 KB,  grid 256: +%.2f us\n", run<16384>(256, st, q, out, x, n, true) - base);
  // dispatch rate of single-wave workgroups that leave at once (4 KB of LDS each)
  for (int g : {1024, 2048, 4096, 5372, 8192, 16384, 65536})
    printf("+ probe LDS 4 KB,   grid %5d: +%.2f us\n", g, run<1024>(g, st, q, out, x, n, true) - base);
  // ... and of 256-thread workgroups (the step kernels' shape), no LDS
  for (int g : {256, 1024, 4096, 16384})
    printf("+ probe 256 threads, grid %5d: +%.2f us\n", g, run256(g, st, q, out, x, n) - base);
  return 0;
}
