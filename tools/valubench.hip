// valubench.hip — the vector pipe's issue rate on this device (dev yardstick; not part of the product).
//   hipcc -O3 --offload-arch=gfx950 -o tools/valubench tools/valubench.hip && tools/valubench
// Every wave runs a long stream of INDEPENDENT v_fma_f32 (eight accumulators), or of v_exp_f32 / v_rcp_f32, with 1, 2, 4 and
// 8 waves per SIMD: cycles per wave64 instruction per SIMD = what bounds a kernel that is limited by vector issue
// (bench.py's "valu" roofline of the neighbour query).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const float m = 0.999f, c = 1e-3f;
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) {
      a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c); a2 = __builtin_fmaf(a2, m, c); a3 = __builtin_fmaf(a3, m, c);
      a4 = __builtin_fmaf(a4, m, c); a5 = __builtin_fmaf(a5, m, c); a6 = __builtin_fmaf(a6, m, c); a7 = __builtin_fmaf(a7, m, c);
    } else if (KIND == 1) {
      a0 = __builtin_amdgcn_exp2f(a0); a1 = __builtin_amdgcn_exp2f(a1); a2 = __builtin_amdgcn_exp2f(a2); a3 = __builtin_amdgcn_exp2f(a3);
      a4 = __builtin_amdgcn_exp2f(a4); a5 = __builtin_amdgcn_exp2f(a5); a6 = __builtin_amdgcn_exp2f(a6); a7 = __builtin_amdgcn_exp2f(a7);
    } else if (KIND == 3) {          // v_pk_fma_f32: two fp32 FMAs per lane and instruction (counted as ONE instruction below)
      typedef float f2 __attribute__((ext_vector_type(2)));
      f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
      const f2 mm = {m, m}, cc = {c, c};
      for (int r = 0; r < 2; ++r) {
        p0 = __builtin_elementwise_fma(p0, mm, cc); p1 = __builtin_elementwise_fma(p1, mm, cc);
        p2 = __builtin_elementwise_fma(p2, mm, cc); p3 = __builtin_elementwise_fma(p3, mm, cc);
      }
      a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
    } else {
      a0 = __builtin_amdgcn_rcpf(a0); a1 = __builtin_amdgcn_rcpf(a1); a2 = __builtin_amdgcn_rcpf(a2); a3 = __builtin_amdgcn_rcpf(a3);
      a4 = __builtin_amdgcn_rcpf(a4); a5 = __builtin_amdgcn_rcpf(a5); a6 = __builtin_amdgcn_rcpf(a6); a7 = __builtin_amdgcn_rcpf(a7);
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  const double clk = p.clockRate * 1e3;     // Hz
  float* out;
  hipMalloc(&out, sizeof(float) * 256 * cus * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  printf("{\"cus\": %d, \"clock_hz\": %.0f", cus, clk);
  const char* names[4] = {"v_fma_f32", "v_exp_f32", "v_rcp_f32", "v_pk_fma_f32"};
  for (int kind = 0; kind < 4; ++kind)
    for (int wps = 1; wps <= 8; wps *= 2) {          // workgroups of 4 waves: one wave per SIMD each; wps of them per CU
      const dim3 g(cus * wps), b(256);
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (kind == 0) hipLaunchKernelGGL(k<0>, g, b, 0, 0, out, iters, 1.0f);
        else if (kind == 1) hipLaunchKernelGGL(k<1>, g, b, 0, 0, out, iters, 1.0f);
        else if (kind == 3) hipLaunchKernelGGL(k<3>, g, b, 0, 0, out, iters, 1.0f);
        else hipLaunchKernelGGL(k<2>, g, b, 0, 0, out, iters, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
      }
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double inst_per_simd = (double)iters * 8 * wps;          // wave-instructions one SIMD issued
      printf(", \"%s_cycles_per_wave_instr_at_%d_waves_per_simd\": %.2f", names[kind], wps, ms * 1e-3 * clk / inst_per_simd);
    }
  printf("}\n");
  return 0;
}
