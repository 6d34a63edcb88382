/* A C caller of the C-ABI with no Python and no torch in the process: create a ctx, reset a
 * fleet, take fused steps, read the state back.  Built and run by tests/test_gpu_parity.py. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../include/dronesim_amd.h"

#define CK(x) do { int r_ = (x); if (r_) { printf("FAIL %s -> %d (%s)\n", #x, r_, dsim_strerror(r_)); return 1; } } while (0)

int main(void) {
  dsim_type_params t;
  memset(&t, 0, sizeof(t));
  t.kind = DSIM_KIND_QUAD; t.n_act = 4; t.mass = 0.75; t.kf = 2.0e-8; t.km = 2.74e-10;
  t.inertia[0] = t.inertia[1] = 6.2e-4; t.inertia[2] = 1.1e-3;
  const double rp[4][3] = {{.11, .11, 0}, {-.11, .11, 0}, {-.11, -.11, .03}, {.11, -.11, .04}};
  const double G1[4][4] = {{50, 50, -50, -50}, {-50, 50, 50, -50}, {-7, 7, -7, 7}, {1.7, 1.7, 1.7, 1.7}};
  /* pinv(G1/0.05) of this X-quad: columns of G1 are orthogonal, so pinv = 0.05 * G1^T / row_norm^2 */
  for (int j = 0; j < 4; ++j) {
    t.pwm2rpm_scale[j] = 20000; t.pwm_max[j] = 1; t.rotor_axis[j][2] = 1; t.rotor_spin[j] = (j & 1) ? 1 : -1;
    for (int k = 0; k < 3; ++k) t.rotor_pos[j][k] = rp[j][k];
    for (int i = 0; i < 4; ++i) {
      double nn = 0;
      for (int c = 0; c < 4; ++c) nn += G1[i][c] * G1[i][c];
      t.G1[i][j] = G1[i][j];
      t.alloc[j][i] = 0.05 * G1[i][j] / nn;
    }
  }
  t.kp_pos = 1.0; t.kd_pos = 2.2;
  t.att_gain[0] = t.att_gain[1] = 7; t.att_gain[2] = 5; t.rate_gain[0] = t.rate_gain[1] = 18; t.rate_gain[2] = 10;
  t.gravity = 9.8; t.lin_damping = t.ang_damping = 0.04f; t.max_coord_vel = 100; t.max_speed_kmh = 30;

  dsim_ctx* ctx = NULL;
  CK(dsim_create(&ctx, 0, &t, 1));
  const int64_t n = 1000, n_pad = 1024;
  float *state, *tgt, *pos, *rpy;
  if (hipMalloc((void**)&state, sizeof(float) * 24 * n_pad) || hipMalloc((void**)&tgt, sizeof(float) * 10 * n_pad) ||
      hipMalloc((void**)&pos, sizeof(float) * 3 * n_pad) || hipMalloc((void**)&rpy, sizeof(float) * 3 * n_pad)) return 2;
  float* h = (float*)calloc(24 * n_pad, sizeof(float));
  for (int64_t i = 0; i < n_pad; ++i) { h[i] = (float)(i % 32); h[n_pad + i] = (float)(i / 32); h[2 * n_pad + i] = 0.5f; }
  hipMemcpy(pos, h, sizeof(float) * 3 * n_pad, hipMemcpyHostToDevice);
  hipMemset(rpy, 0, sizeof(float) * 3 * n_pad);
  /* targets: hover 0.3 m above the start */
  for (int64_t i = 0; i < n_pad; ++i) h[2 * n_pad + i] = 0.8f;
  hipMemset(tgt, 0, sizeof(float) * 10 * n_pad);
  hipMemcpy(tgt, h, sizeof(float) * 3 * n_pad, hipMemcpyHostToDevice);
  dsim_view sv = {state, n_pad, n_pad, n_pad, n_pad * 24, 24, 0};
  dsim_view tv = {tgt, n_pad, n_pad, n_pad, n_pad * 10, 10, 0};
  CK(dsim_reset(ctx, NULL, n, sv, pos, rpy, NULL, NULL, NULL));
  dsim_step_args a;
  memset(&a, 0, sizeof(a));
  a.phys_substeps = 5; a.dt_phys = 1.0f / 240; a.dt_ctrl = 5.0f / 240;
  for (int k = 0; k < 240; ++k) { a.step_index = k; CK(dsim_step(ctx, NULL, n, sv, tv, &a)); }
  if (hipDeviceSynchronize()) return 3;
  hipMemcpy(h, state, sizeof(float) * 24 * n_pad, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int64_t i = 0; i < n; ++i) { const double e = fabs(h[2 * n_pad + i] - 0.8); if (e > worst) worst = e; }
  printf("c_abi_smoke: 1000 drones, 240 fused steps (5 s): max |z - 0.8| = %.4f m\n", worst);
  CK(dsim_destroy(ctx));
  return worst < 0.05 ? 0 : 4;
}
