/* dsim_oracle.c — CPU restatement (fp64, scalar) of the reference's per-drone
 * dynamics + INDI control step.
 *
 * TEST INFRASTRUCTURE ONLY.  Imported/linked by tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg, as the checker / the reported CPU baseline.
 * The product (dronesim_amd/) never calls into this file and has no CPU path.
 *
 * Every function cites the reference code it restates (paths relative to the
 * reference repo enac-drones/dronesim @ 2024_08_07).
 *
 * PARITY STATUS
 *   control half (C1..C8): PINNED by golden vectors produced by running the
 *     reference's own INDIControl / INDIControl_6DOF / wls_alloc / utils.math
 *     code (tests/golden/make_goldens.py; the three pybullet closed-form math
 *     helpers are a stand-in there, see that file's header).
 *   env-side half (P0..P3 force map, P6..P8 aero terms, action adaptors): PINNED by
 *     recordings of what the reference's own BaseAviary._physics/_drag/_groundEffect/
 *     _downwash/_preprocessAction hand to the engine (tests/golden/env_side.npz,
 *     tests/test_oracle_env_side.py).
 *   integrator (P4): "PARITY UNPINNED".  The reference delegates rigid-body
 *     integration to the third-party engine PyBullet (`pybullet`, version
 *     unpinned in setup.py:14, not vendored, not installed, no network).  P4 below
 *     restates Bullet 3.x's published btMultiBody floating-base step
 *     (btMultiBody::computeAccelerationsArticulatedBodyAlgorithmMultiDof,
 *     applyDeltaVeeMultiDof, stepPositionsMultiDof) from knowledge of that
 *     engine; it is anchored on the reference's call sites (BaseAviary.py:542-543,
 *     673-675, 681-694, 1443-1457, 1529-1543) and on analytic known-answer tests
 *     (tests/test_oracle_physics.py), not on a run of the reference.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "../include/dronesim_amd.h"

#define ORC_PI 3.14159265358979323846
#define ORC_FLT_EPSILON 1.1920929e-07  /* wls_alloc.py FLT_EPSILON (Paparazzi float) */

/* ======================================================================= */
/* C7: dronesim/utils/math.py helpers                                       */
/* ======================================================================= */

/* utils/math.py:23-31  quat_inv_comp (xyzw, w index 3) */
void orc_quat_inv_comp(const double q1[4], const double q2[4], double qerr[4]) {
  const int i = 3, x = 0, y = 1, z = 2;
  qerr[i] = q1[i] * q2[i] + q1[x] * q2[x] + q1[y] * q2[y] + q1[z] * q2[z];
  qerr[x] = q1[i] * q2[x] - q1[x] * q2[i] - q1[y] * q2[z] + q1[z] * q2[y];
  qerr[y] = q1[i] * q2[y] + q1[x] * q2[z] - q1[y] * q2[i] - q1[z] * q2[x];
  qerr[z] = q1[i] * q2[z] - q1[x] * q2[y] + q1[y] * q2[x] - q1[z] * q2[i];
}

/* utils/math.py:4-20  quat_comp */
void orc_quat_comp(const double a2b[4], const double b2c[4], double a2c[4]) {
  const int qi = 3, qx = 0, qy = 1, qz = 2;
  a2c[qi] = a2b[qi] * b2c[qi] - a2b[qx] * b2c[qx] - a2b[qy] * b2c[qy] - a2b[qz] * b2c[qz];
  a2c[qx] = a2b[qi] * b2c[qx] + a2b[qx] * b2c[qi] + a2b[qy] * b2c[qz] - a2b[qz] * b2c[qy];
  a2c[qy] = a2b[qi] * b2c[qy] - a2b[qx] * b2c[qz] + a2b[qy] * b2c[qi] + a2b[qz] * b2c[qx];
  a2c[qz] = a2b[qi] * b2c[qz] + a2b[qx] * b2c[qy] - a2b[qy] * b2c[qx] + a2b[qz] * b2c[qi];
}

/* utils/math.py:46-51  quat_wrap_shortest (in place) */
void orc_quat_wrap_shortest(double q[4]) {
  if (q[3] < 0) for (int i = 0; i < 4; ++i) q[i] = -q[i];
}

/* utils/math.py:75-80  norm_ang (while loops: +pi stays +pi, -pi stays -pi) */
double orc_norm_ang(double x) {
  while (x > ORC_PI) x -= 2 * ORC_PI;
  while (x < -ORC_PI) x += 2 * ORC_PI;
  return x;
}

/* ======================================================================= */
/* C8: the three PyBullet math helpers the controller calls                  */
/* [BULLET-INTERNAL] restated from Bullet 3.x pybullet.c / btMatrix3x3.h     */
/* call sites: INDIControl.py:225,301,388,428; BaseAviary.py:729             */
/* ======================================================================= */

/* p.getEulerFromQuaternion: ZYX, gimbal clamp at |sarg| >= 0.99999 */
void orc_euler_from_quat(const double q[4], double rpy[3]) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double sqx = x * x, sqy = y * y, sqz = z * z, squ = w * w;
  const double sarg = -2.0 * (x * z - w * y);
  if (sarg <= -0.99999) {
    rpy[0] = 0; rpy[1] = -0.5 * ORC_PI; rpy[2] = 2 * atan2(x, -y);
  } else if (sarg >= 0.99999) {
    rpy[0] = 0; rpy[1] = 0.5 * ORC_PI; rpy[2] = 2 * atan2(-x, y);
  } else {
    rpy[0] = atan2(2 * (y * z + w * x), squ - sqx - sqy + sqz);
    rpy[1] = asin(sarg);
    rpy[2] = atan2(2 * (x * y + w * z), squ + sqx - sqy - sqz);
  }
}

/* p.getQuaternionFromEuler: half-angle product, then normalise */
void orc_quat_from_euler(const double rpy[3], double q[4]) {
  const double phi = rpy[0] / 2.0, the = rpy[1] / 2.0, psi = rpy[2] / 2.0;
  q[0] = sin(phi) * cos(the) * cos(psi) - cos(phi) * sin(the) * sin(psi);
  q[1] = cos(phi) * sin(the) * cos(psi) + sin(phi) * cos(the) * sin(psi);
  q[2] = cos(phi) * cos(the) * sin(psi) - sin(phi) * sin(the) * cos(psi);
  q[3] = cos(phi) * cos(the) * cos(psi) + sin(phi) * sin(the) * sin(psi);
  const double len = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int i = 0; i < 4; ++i) q[i] /= len;
}

/* p.getMatrixFromQuaternion: btMatrix3x3::setRotation, row-major, s = 2/|q|^2 */
void orc_matrix_from_quat(const double q[4], double R[9]) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double d = x * x + y * y + z * z + w * w;
  const double s = 2.0 / d;
  const double xs = x * s, ys = y * s, zs = z * s;
  const double wx = w * xs, wy = w * ys, wz = w * zs;
  const double xx = x * xs, xy = x * ys, xz = x * zs;
  const double yy = y * ys, yz = y * zs, zz = z * zs;
  R[0] = 1.0 - (yy + zz); R[1] = xy - wz;         R[2] = xz + wy;
  R[3] = xy + wz;         R[4] = 1.0 - (xx + zz); R[5] = yz - wx;
  R[6] = xz - wy;         R[7] = yz + wx;         R[8] = 1.0 - (xx + yy);
}

/* ======================================================================= */
/* numpy.linalg.pinv / lstsq equivalents (SVD, singular-value cut-off)        */
/* INDIControl.py:336,459; wls_alloc.py:252                                  */
/* ======================================================================= */

/* One-sided Jacobi SVD of A (m x n, m >= n, row-major): A = U diag(s) V^T.
 * U overwrites a copy of A (m x n), V is n x n. */
static void jacobi_svd(const double* A, int m, int n, double* U, double* s, double* V) {
  memcpy(U, A, sizeof(double) * m * n);
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) V[i * n + j] = (i == j);
  for (int sweep = 0; sweep < 60; ++sweep) {
    int rotated = 0;
    for (int p = 0; p < n - 1; ++p) for (int q = p + 1; q < n; ++q) {
      double alpha = 0, beta = 0, gamma = 0;
      for (int i = 0; i < m; ++i) {
        alpha += U[i * n + p] * U[i * n + p];
        beta += U[i * n + q] * U[i * n + q];
        gamma += U[i * n + p] * U[i * n + q];
      }
      if (fabs(gamma) <= 1e-300 || fabs(gamma) <= 2.3e-16 * sqrt(alpha * beta)) continue;
      rotated = 1;
      const double zeta = (beta - alpha) / (2.0 * gamma);
      const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
      const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
      for (int i = 0; i < m; ++i) {
        const double up = U[i * n + p], uq = U[i * n + q];
        U[i * n + p] = c * up - sn * uq;
        U[i * n + q] = sn * up + c * uq;
      }
      for (int i = 0; i < n; ++i) {
        const double vp = V[i * n + p], vq = V[i * n + q];
        V[i * n + p] = c * vp - sn * vq;
        V[i * n + q] = sn * vp + c * vq;
      }
    }
    if (!rotated) break;
  }
  for (int j = 0; j < n; ++j) {
    double nn = 0;
    for (int i = 0; i < m; ++i) nn += U[i * n + j] * U[i * n + j];
    s[j] = sqrt(nn);
    if (s[j] > 0) for (int i = 0; i < m; ++i) U[i * n + j] /= s[j];
  }
}

/* Moore-Penrose pseudo-inverse, numpy semantics: singular values
 * <= rcond * s_max are treated as zero (np.linalg.pinv default rcond=1e-15). */
void orc_pinv(const double* A, int m, int n, double rcond, double* Ainv /* n x m */) {
  double At[144] = {0}, U[144], V[144], s[12];
  if (m >= n) {
    jacobi_svd(A, m, n, U, s, V);
    double smax = 0;
    for (int j = 0; j < n; ++j) if (s[j] > smax) smax = s[j];
    for (int i = 0; i < n; ++i) for (int k = 0; k < m; ++k) {
      double acc = 0;
      for (int j = 0; j < n; ++j) if (s[j] > rcond * smax) acc += V[i * n + j] * U[k * n + j] / s[j];
      Ainv[i * m + k] = acc;
    }
  } else { /* pinv(A) = pinv(A^T)^T */
    double Bt[144];
    for (int i = 0; i < m; ++i) for (int j = 0; j < n; ++j) At[j * m + i] = A[i * n + j];
    orc_pinv(At, n, m, rcond, Bt); /* m x n */
    for (int i = 0; i < m; ++i) for (int j = 0; j < n; ++j) Ainv[j * m + i] = Bt[i * n + j];
  }
}

/* ======================================================================= */
/* controller memory (what the reference keeps on each INDIControl instance)  */
/* INDIControl.py:109-146 ; 6DOF: INDIControl_6DOF.py:214-252                */
/* ======================================================================= */
typedef struct orc_ctrl_mem {
  double last_vel[3];
  double last_rates[3];
  double last_thrust;
  double cmd[DSIM_MAX_ACT];
} orc_ctrl_mem;

void orc_ctrl_reset(const dsim_type_params* P, orc_ctrl_mem* m) {
  memset(m, 0, sizeof(*m));
  if (P->kind == DSIM_KIND_HEXA6DOF) { /* INDIControl_6DOF.py:232-234 */
    m->last_thrust = 0.3;
    for (int i = 0; i < P->n_act; ++i) m->cmd[i] = 0.5;
  } /* quad: last_thrust = 0, cmd = 0 (INDIControl.py:127-129) */
}

static void build_G(const double rpy[3], double G[9]) {
  /* INDIControl.py:301-333 (same matrix at INDIControl_6DOF.py:423-457) */
  const double phi = rpy[0], theta = rpy[1], psi = rpy[2];
  const double sph = sin(phi), sth = sin(theta), sps = sin(psi);
  const double cph = cos(phi), cth = cos(theta), cps = cos(psi);
  const double T = 9.81;
  G[0] = (cph * sps - sph * cps * sth) * T; G[1] = (cph * cps * cth) * T; G[2] = sph * sps + cph * cps * sth;
  G[3] = (-sph * sps * sth - cps * cph) * T; G[4] = (cph * sps * cth) * T; G[5] = cph * sps * sth - cps * sph;
  G[6] = -cth * sph * T; G[7] = -sth * cph * T; G[8] = cph * cth;
}

/* C2: INDIControl._INDIPositionControl, INDIControl.py:232-351 */
void orc_indi_position(const dsim_type_params* P, double dt, const double pos[3], const double quat[4],
                       const double vel[3], const double tpos[3], const double trpy[3],
                       const double tvel[3], const double tacc[3], orc_ctrl_mem* mem,
                       double* thrust, double target_euler[3], double pos_e[3]) {
  double accel_e[3], rpy[3], G[9], Ginv[9], inc[3];
  for (int k = 0; k < 3; ++k) {
    pos_e[k] = tpos[k] - pos[k];                          /* :278 */
    const double speed_sp = pos_e[k] * P->kp_pos;         /* :281 */
    const double vel_e = speed_sp + tvel[k] - vel[k];     /* :283 */
    const double accel_sp = vel_e * P->kd_pos;            /* :286 */
    const double cur_accel = (vel[k] - mem->last_vel[k]) / dt; /* :289 */
    mem->last_vel[k] = vel[k];                            /* :291 */
    double e = accel_sp + tacc[k] - cur_accel;            /* :293 */
    accel_e[k] = e < -6.0 ? -6.0 : (e > 6.0 ? 6.0 : e);   /* :296 */
  }
  orc_euler_from_quat(quat, rpy);                         /* :301 */
  build_G(rpy, G);                                        /* :304-333 */
  orc_pinv(G, 3, 3, 1e-15, Ginv);                         /* :336 */
  for (int i = 0; i < 3; ++i)
    inc[i] = Ginv[i * 3] * accel_e[0] + Ginv[i * 3 + 1] * accel_e[1] + Ginv[i * 3 + 2] * accel_e[2]; /* :339 */
  const double yaw_inc = orc_norm_ang(trpy[2] - rpy[2]);  /* :341 */
  target_euler[0] = rpy[0] + inc[0];                      /* :344-346 */
  target_euler[1] = rpy[1] + inc[1];
  target_euler[2] = rpy[2] + yaw_inc;
  *thrust = mem->last_thrust + inc[2];                    /* :347 */
}

/* C4: INDIControl._INDIRateControl, INDIControl.py:413-490 */
void orc_indi_rate(const dsim_type_params* P, double dt, double thrust, const double quat[4],
                   const double ang_vel_world[3], const double rate_sp[3], orc_ctrl_mem* mem) {
  double R[9], wb[3], v[4], Gs[4 * DSIM_MAX_ACT], Gp[4 * DSIM_MAX_ACT];
  const int na = P->n_act; /* 4; 6 for hexa_6DOF_simple (self.indi_actuator_nr, INDIControl.py:70, 128) */
  orc_matrix_from_quat(quat, R);                          /* :428 */
  for (int k = 0; k < 3; ++k)                             /* :430  R.T.dot(w) */
    wb[k] = R[0 * 3 + k] * ang_vel_world[0] + R[1 * 3 + k] * ang_vel_world[1] + R[2 * 3 + k] * ang_vel_world[2];
  const double* Kr = P->rate_gain;
  for (int k = 0; k < 3; ++k) {
    const double angular_accel = (wb[k] - mem->last_rates[k]) / (1.0 * dt); /* :433 */
    mem->last_rates[k] = wb[k];                           /* :442 */
    const double ref = (rate_sp[k] - wb[k]) * Kr[k];      /* :446-448 */
    v[k] = ref - angular_accel;                           /* :451-453 */
  }
  v[3] = thrust - mem->last_thrust;                       /* :454 */
  mem->last_thrust = thrust;                              /* :455 */
  for (int i = 0; i < 4; ++i) for (int j = 0; j < na; ++j) Gs[i * na + j] = P->G1[i][j] / 0.05;
  orc_pinv(Gs, 4, na, 1e-15, Gp);                         /* :459  pinv(G1/0.05) : na x 4 */
  for (int j = 0; j < na; ++j) {
    double du = 0;
    for (int i = 0; i < 4; ++i) du += Gp[j * 4 + i] * v[i];
    double c = mem->cmd[j] + du;                          /* :486 */
    c = c < P->pwm_min[j] ? P->pwm_min[j] : (c > P->pwm_max[j] ? P->pwm_max[j] : c); /* :487 */
    mem->cmd[j] = c;
  }
}

/* C3: INDIControl._INDIAttitudeControl, INDIControl.py:355-411 */
void orc_indi_attitude(const dsim_type_params* P, double dt, double thrust, const double quat[4],
                       const double ang_vel_world[3], const double target_euler[3], orc_ctrl_mem* mem) {
  double tq[4], qerr[4], rate_sp[3];
  orc_quat_from_euler(target_euler, tq);                  /* :388 */
  orc_quat_inv_comp(quat, tq, qerr);                      /* :390 */
  orc_quat_wrap_shortest(qerr);                           /* :393 (mutates quat_err) */
  for (int k = 0; k < 3; ++k) rate_sp[k] = P->att_gain[k] * qerr[k]; /* :395-402 */
  orc_indi_rate(P, dt, thrust, quat, ang_vel_world, rate_sp, mem);   /* :404-410 */
}

/* ======================================================================= */
/* C6: wls_alloc, dronesim/control/wls_alloc.py:125-350                      */
/* Returns 0 and u_out on success, -1 on "solution failed" (:350), -2 where   */
/* the reference would raise (use of `alpha` before assignment, :304-325).    */
/* ======================================================================= */
#define WLS_MAX_U 6
#define WLS_MAX_V 6
#define WLS_MAX_C (WLS_MAX_U + WLS_MAX_V)
int orc_wls_alloc(const double* v, const double* umin, const double* umax, const double* B /* n_v x n_u */,
                  int n_u, int n_v, const double* u_guess, const double* W_init, const double* Wv,
                  const double* Wu, const double* up, double gamma_sq, int imax, double* u_out, int* iters) {
  const int n_c = n_u + n_v;
  double A[WLS_MAX_C][WLS_MAX_U] = {{0}}, A_free[WLS_MAX_C][WLS_MAX_U] = {{0}};
  double b[WLS_MAX_C] = {0}, d[WLS_MAX_C] = {0};
  int free_index[WLS_MAX_U] = {0}, free_index_lookup[WLS_MAX_U];
  int n_free = 0, free_chk = -1, iter = 0;
  double p_free[WLS_MAX_U] = {0}, p[WLS_MAX_U], u[WLS_MAX_U], u_opt[WLS_MAX_U], W[WLS_MAX_U], Lambda[WLS_MAX_U];
  int n_p_free = n_u; /* len(p_free) in the reference; starts as zeros(n_u) (:150) */
  double alpha = 0; int alpha_set = 0; int id_alpha = 0;

  for (int i = 0; i < n_u; ++i) u[i] = u_guess ? u_guess[i] : (umax[i] + umin[i]) * 0.5; /* :164-170 */
  for (int i = 0; i < n_u; ++i) W[i] = W_init ? W_init[i] : 0.0;                         /* :171-174 */
  for (int i = 0; i < n_u; ++i) free_index_lookup[i] = -1;
  for (int i = 0; i < n_u; ++i) if (W[i] == 0) { free_index_lookup[i] = n_free; free_index[n_free++] = i; } /* :177-182 */
  for (int i = 0; i < n_v; ++i) {                                                         /* :184-198 */
    b[i] = Wv ? gamma_sq * Wv[i] * v[i] : gamma_sq * v[i];
    d[i] = b[i];
    for (int j = 0; j < n_u; ++j) {
      A[i][j] = Wv ? gamma_sq * Wv[i] * B[i * n_u + j] : gamma_sq * B[i * n_u + j];
      d[i] -= A[i][j] * u[j];
    }
  }
  for (int i = n_v; i < n_c; ++i) {                                                       /* :199-213 */
    for (int j = 0; j < n_u; ++j) A[i][j] = 0;
    A[i][i - n_v] = Wu ? Wu[i - n_v] : 1.0;
    b[i] = up ? (Wu ? Wu[i - n_v] * up[i - n_v] : up[i - n_v]) : 0;
    d[i] = b[i] - A[i][i - n_v] * u[i - n_v];
  }
  while (iter < imax) {                                                                   /* :215 */
    iter += 1;
    for (int i = 0; i < n_u; ++i) { p[i] = 0; u_opt[i] = u[i]; }
    if (free_chk != n_free) {                                                             /* :225-231 */
      for (int i = 0; i < n_c; ++i) for (int j = 0; j < n_free; ++j) A_free[i][j] = A[i][free_index[j]];
      free_chk = n_free;
    }
    if (n_free) {                                                                         /* :235-247 lstsq */
      double Af[WLS_MAX_C * WLS_MAX_U], Ap[WLS_MAX_U * WLS_MAX_C];
      for (int i = 0; i < n_c; ++i) for (int j = 0; j < n_free; ++j) Af[i * n_free + j] = A_free[i][j];
      /* np.linalg.lstsq(rcond=None): cut-off eps*max(M,N) */
      orc_pinv(Af, n_c, n_free, 2.220446049250313e-16 * (n_c > n_free ? n_c : n_free), Ap);
      for (int j = 0; j < n_free; ++j) {
        double acc = 0;
        for (int i = 0; i < n_c; ++i) acc += Ap[j * n_c + i] * d[i];
        p_free[j] = acc;
      }
      n_p_free = n_free;
    }
    for (int i = 0; i < n_free; ++i) { p[free_index[i]] = p_free[i]; u_opt[free_index[i]] += p_free[i]; } /* :251-253 */
    int n_infeasible = 0;                                                                 /* :255-259, +-1.0 slack */
    for (int i = 0; i < n_u; ++i) if (u_opt[i] >= (umax[i] + 1.0) || u_opt[i] <= (umin[i] - 1.0)) n_infeasible++;
    if (n_infeasible == 0) {                                                              /* :261-289 */
      for (int i = 0; i < n_u; ++i) { u[i] = u_opt[i]; Lambda[i] = 0; }
      for (int i = 0; i < n_c; ++i) {
        for (int k = 0; k < n_free; ++k) d[i] -= A_free[i][k] * p_free[k];
        for (int k = 0; k < n_u; ++k) Lambda[k] += A[i][k] * d[i];
      }
      int break_flag = 1;
      for (int i = 0; i < n_u; ++i) {
        Lambda[i] *= W[i];
        if (Lambda[i] < -ORC_FLT_EPSILON) {
          break_flag = 0;
          W[i] = 0;
          if (free_index_lookup[i] < 0) { free_index_lookup[i] = n_free; free_index[n_free++] = i; }
        }
      }
      if (break_flag) {
        for (int i = 0; i < n_u; ++i) u_out[i] = u[i];
        *iters = iter;
        return 0;
      }
    } else {                                                                              /* :290-293 */
      alpha = INFINITY; alpha_set = 1; id_alpha = 0;
    }
    /* :295-346 runs in BOTH branches in the reference (the for loop is dedented) */
    if (!alpha_set) { *iters = iter; return -2; } /* reference: UnboundLocalError on `alpha` */
    for (int i = 0; i < n_free; ++i) {
      const int id = free_index[i];
      double alpha_tmp;
      if (fabs(p[id]) > ORC_FLT_EPSILON) alpha_tmp = p[id] < 0 ? (umin[id] - u[id]) / p[id] : (umax[id] - u[id]) / p[id];
      else alpha_tmp = INFINITY;
      if (alpha_tmp < alpha) { alpha = alpha_tmp; id_alpha = id; }
    }
    for (int i = 0; i < n_u; ++i) u[i] += alpha * p[i];                                   /* :310-311 */
    for (int i = 0; i < n_c; ++i) {                                                       /* :313-323 */
      const int k_len = n_free < n_p_free ? n_free : n_p_free;
      for (int k = 0; k < k_len; ++k) d[i] -= A_free[i][k] * alpha * p_free[k];
    }
    W[id_alpha] = p[id_alpha] > 0 ? 1.0 : -1.0;                                           /* :325-328 */
    n_free -= 1;                                                                          /* :332-339 */
    if (n_free < 0 || free_index_lookup[id_alpha] < 0) { *iters = iter; return -2; } /* reference would index [-1] */
    free_index[free_index_lookup[id_alpha]] = free_index[n_free];
    free_index_lookup[free_index[free_index_lookup[id_alpha]]] = free_index_lookup[id_alpha];
    free_index_lookup[id_alpha] = -1;
  }
  *iters = iter;
  return -1;                                                                              /* :350 */
}

/* ======================================================================= */
/* C5: INDIControl_6DOF (dronesim/control/INDIControl_6DOF.py:259-634)        */
/* ======================================================================= */
int orc_indi6_compute_control(const dsim_type_params* P, double dt, const double pos[3], const double quat[4],
                              const double vel[3], const double ang_vel_world[3], const double tpos[3],
                              const double tvel[3], const double trpy[3], orc_ctrl_mem* mem,
                              double pos_e[3], double* yaw_e, int* wls_iters) {
  double accel_e[3], rpy[3], G[9], Ginv[9], inc[3];
  for (int k = 0; k < 3; ++k) {                              /* :397-413 (no target_acc) */
    pos_e[k] = tpos[k] - pos[k];
    const double vel_e = pos_e[k] * P->kp_pos + tvel[k] - vel[k];
    const double accel_sp = vel_e * P->kd_pos;
    const double cur_accel = (vel[k] - mem->last_vel[k]) / dt;
    mem->last_vel[k] = vel[k];
    const double e = accel_sp - cur_accel;
    accel_e[k] = e < -6.0 ? -6.0 : (e > 6.0 ? 6.0 : e);
  }
  orc_euler_from_quat(quat, rpy);                            /* :418 */
  build_G(rpy, G);
  orc_pinv(G, 3, 3, 1e-15, Ginv);                            /* :464 */
  for (int i = 0; i < 3; ++i)
    inc[i] = Ginv[i * 3] * accel_e[0] + Ginv[i * 3 + 1] * accel_e[1] + Ginv[i * 3 + 2] * accel_e[2];
  /* :480-482 roll/pitch increments rotated by R(psi) -- they only feed target_euler,
     which :495 overwrites with zeros; thrust uses inc[2] (:492) */
  const double thrust = mem->last_thrust + inc[2];
  const double target_euler[3] = {0, 0, 0};                  /* :495 */
  /* _INDIAttitudeControl :499-634 */
  double tq[4], qerr[4], att_err[3], R[9], wb[3], v[6];
  orc_quat_from_euler(target_euler, tq);                     /* :538 */
  orc_quat_inv_comp(quat, tq, qerr);                         /* :540  (no shortest-wrap, :543-545) */
  {
    const double psi = rpy[2];                               /* :549-557: att_err.xy rotated by inv(R_psi) */
    const double c = cos(psi), s = sin(psi);
    /* inv([[c,-s],[s,c]]) = [[c,s],[-s,c]] / (c^2+s^2) ; numpy inv of a rotation */
    const double det = c * c + s * s;
    att_err[0] = (c * qerr[0] + s * qerr[1]) / det;
    att_err[1] = (-s * qerr[0] + c * qerr[1]) / det;
    att_err[2] = qerr[2];
  }
  orc_matrix_from_quat(quat, R);                             /* :566 */
  for (int k = 0; k < 3; ++k)
    wb[k] = R[0 * 3 + k] * ang_vel_world[0] + R[1 * 3 + k] * ang_vel_world[1] + R[2 * 3 + k] * ang_vel_world[2];
  for (int k = 0; k < 3; ++k) {
    const double rate_sp = P->att_gain[k] * att_err[k];      /* :560-562 */
    const double angular_accel = (wb[k] - mem->last_rates[k]) / (1.0 * dt); /* :571 */
    mem->last_rates[k] = wb[k];                              /* :580 */
    v[k] = (rate_sp - wb[k]) * P->rate_gain[k] - angular_accel; /* :583-592 */
  }
  for (int k = 0; k < 3; ++k)                                /* :589  R.T.dot(accel_error) */
    v[3 + k] = R[0 * 3 + k] * accel_e[0] + R[1 * 3 + k] * accel_e[1] + R[2 * 3 + k] * accel_e[2];
  mem->last_thrust = thrust;                                 /* :598 */
  const int na = P->n_act;
  double umin[6], umax[6], Bs[36], du[6];
  for (int i = 0; i < na; ++i) { umin[i] = P->pwm_min[i] - mem->cmd[i]; umax[i] = P->pwm_max[i] - mem->cmd[i]; } /* :607-612 */
  for (int i = 0; i < 6; ++i) for (int j = 0; j < na; ++j) Bs[i * na + j] = P->G1[i][j] / 0.05;
  const double Wv[6] = {1000, 1000, 0.1, 10, 10, 100};       /* :614 */
  double Wu[6] = {1, 1, 1, 1, 1, 1};                         /* :615 */
  int rc = orc_wls_alloc(v, umin, umax, Bs, na, 6, NULL, NULL, Wv, Wu, NULL, 100000, 100, du, wls_iters); /* :626-628 */
  if (rc != 0) return rc;  /* reference: `self.cmd += None` -> TypeError */
  for (int j = 0; j < na; ++j) {                             /* :630-631 */
    double c = mem->cmd[j] + du[j];
    mem->cmd[j] = c < P->pwm_min[j] ? P->pwm_min[j] : (c > P->pwm_max[j] ? P->pwm_max[j] : c);
  }
  *yaw_e = target_euler[2] - rpy[2];                         /* :336 */
  return 0;
}

/* C1: INDIControl.computeControl, INDIControl.py:154-227 (quad) */
int orc_indi_compute_control(const dsim_type_params* P, double dt, const double pos[3], const double quat[4],
                             const double vel[3], const double ang_vel_world[3], const double tpos[3],
                             const double tvel[3], const double tacc[3], const double trpy[3],
                             orc_ctrl_mem* mem, double pos_e[3], double* yaw_e) {
  if (P->kind == DSIM_KIND_HEXA6DOF) {
    int it;
    return orc_indi6_compute_control(P, dt, pos, quat, vel, ang_vel_world, tpos, tvel, trpy, mem, pos_e, yaw_e, &it);
  }
  double thrust, target_euler[3], rpy[3];
  orc_indi_position(P, dt, pos, quat, vel, tpos, trpy, tvel, tacc, mem, &thrust, target_euler, pos_e); /* :204-213 */
  orc_indi_attitude(P, dt, thrust, quat, ang_vel_world, target_euler, mem);                            /* :215-223 */
  orc_euler_from_quat(quat, rpy);                                                                      /* :225 */
  *yaw_e = target_euler[2] - rpy[2];                                                                   /* :227 */
  return 0;
}

/* ======================================================================= */
/* P1: CtrlAviary._preprocessAction, CtrlAviary.py:258-263                    */
/* ======================================================================= */
void orc_preprocess_action(const dsim_type_params* P, const double* action, double* clipped) {
  for (int j = 0; j < P->n_act; ++j)
    clipped[j] = action[j] < P->pwm_min[j] ? P->pwm_min[j] : (action[j] > P->pwm_max[j] ? P->pwm_max[j] : action[j]);
}

/* ======================================================================= */
/* P2: BaseAviary._quad_copter_physics standard branch, BaseAviary.py:1487-1490,
 * 1514-1543.  Returns the body-frame wrench about the COM that the five
 * p.applyExternalForce/Torque calls add up to.  f_noise/m_noise: 4 normals each
 * (N(0,.01), N(0,.001)), NULL = zero.                                       */
/* ======================================================================= */
static void cross3(const double a[3], const double b[3], double c[3]) {
  c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
}

void orc_quad_wrench(const dsim_type_params* P, const double cmd[4], const double* f_noise,
                     const double* m_noise, double F_b[3], double tau_b[3], double rpm_out[4]) {
  double forces[4], torques[4];
  for (int i = 0; i < 4; ++i) {
    const double rpm = P->pwm2rpm_scale[i] * cmd[i] + P->pwm2rpm_const[i]; /* :1487-1490 */
    if (rpm_out) rpm_out[i] = rpm;
    forces[i] = rpm * rpm * P->kf + (f_noise ? f_noise[i] : 0.0);   /* :1515,1524 */
    torques[i] = rpm * rpm * P->km + (m_noise ? m_noise[i] : 0.0);  /* :1516,1525 */
  }
  const double z_torque = -torques[0] + torques[1] - torques[2] + torques[3]; /* :1527 */
  F_b[0] = F_b[1] = F_b[2] = 0; tau_b[0] = tau_b[1] = tau_b[2] = 0;
  for (int i = 0; i < 4; ++i) {                                     /* :1528-1536 */
    const double f[3] = {f_noise ? f_noise[0] : 0.0, f_noise ? f_noise[1] : 0.0, forces[i]};
    double t[3];
    cross3(P->rotor_pos[i], f, t); /* LINK_FRAME force at the link's inertial origin */
    for (int k = 0; k < 3; ++k) { F_b[k] += f[k]; tau_b[k] += t[k]; }
  }
  tau_b[0] += m_noise ? m_noise[0] : 0.0;                           /* :1537-1543 */
  tau_b[1] += m_noise ? m_noise[1] : 0.0;
  tau_b[2] += z_torque;
}

/* P3: BaseAviary._morphing_hexa_physics, BaseAviary.py:1398-1403, 1429-1457.
 * Force [0,0,F_j] and torque [0,0,tau_j] in the (tilted) prop link frames;
 * rigid composite body (arm joints treated as locked).        */
void orc_hexa_wrench(const dsim_type_params* P, const double cmd[6], const double* f_noise,
                     const double* m_noise, double F_b[3], double tau_b[3], double rpm_out[6]) {
  F_b[0] = F_b[1] = F_b[2] = 0; tau_b[0] = tau_b[1] = tau_b[2] = 0;
  for (int j = 0; j < 6; ++j) {
    const double rpm = P->pwm2rpm_scale[j] * cmd[j] + P->pwm2rpm_const[j];
    if (rpm_out) rpm_out[j] = rpm;
    const double force = rpm * rpm * P->kf + (f_noise ? f_noise[j] : 0.0);      /* :1401,1431 */
    double torque = rpm * rpm * P->km + (m_noise ? m_noise[j] : 0.0);           /* :1402,1432 */
    torque *= P->rotor_spin[j];                                                  /* :1439-1440 (-1 for 0,2,4) */
    double f[3], t[3];
    for (int k = 0; k < 3; ++k) f[k] = P->rotor_axis[j][k] * force;
    cross3(P->rotor_pos[j], f, t);
    for (int k = 0; k < 3; ++k) { F_b[k] += f[k]; tau_b[k] += t[k] + P->rotor_axis[j][k] * torque; }
  }
}

/* P6: BaseAviary._drag, BaseAviary.py:1705-1732 (formula only; dead code in the fork).
 * drag = R^T-frame: base_rot = R; drag_factors = -coeff * sum(2*pi*rpm/60);
 * drag = dot(base_rot, drag_factors * vel) applied at the COM in LINK_FRAME.
 * NOTE: the reference multiplies R (not R^T) by the WORLD velocity and applies the
 * result in the LINK frame; restated literally.                              */
void orc_drag(const dsim_type_params* P, const double quat[4], const double vel[3], const double* rpm, double F_link[3]) {
  double R[9], s = 0, tmp[3];
  orc_matrix_from_quat(quat, R);
  for (int i = 0; i < P->n_act; ++i) s += 2 * ORC_PI * rpm[i] / 60.0;
  for (int k = 0; k < 3; ++k) tmp[k] = -1.0 * P->drag_coeff[k] * s * vel[k];
  for (int k = 0; k < 3; ++k) F_link[k] = R[k * 3] * tmp[0] + R[k * 3 + 1] * tmp[1] + R[k * 3 + 2] * tmp[2];
}

/* P7: BaseAviary._groundEffect, BaseAviary.py:1648-1699 (formula only).
 * Per rotor dF = kf*rpm^2*GND_EFF_COEFF*(PROP_RADIUS/(4 h))^2, h = rotor height
 * clipped to [gnd_eff_h_clip, inf); only if |roll|,|pitch| < pi/2.            */
void orc_ground_effect(const dsim_type_params* P, const double pos[3], const double quat[4],
                       const double* rpm, double dF[DSIM_MAX_ACT]) {
  double R[9], rpy[3];
  orc_matrix_from_quat(quat, R);
  orc_euler_from_quat(quat, rpy);
  for (int i = 0; i < P->n_act; ++i) {
    const double* r = P->rotor_pos[i];
    double h = pos[2] + R[6] * r[0] + R[7] * r[1] + R[8] * r[2];
    if (h < P->gnd_eff_h_clip) h = P->gnd_eff_h_clip;
    const double ratio = P->prop_radius / (4 * h);
    dF[i] = (fabs(rpy[0]) < ORC_PI / 2 && fabs(rpy[1]) < ORC_PI / 2)
                ? rpm[i] * rpm[i] * P->kf * P->gnd_eff_coeff * ratio * ratio : 0.0;
  }
}

/* ======================================================================= */
/* P4: p.stepSimulation for ONE floating-base body.  [BULLET-INTERNAL, PARITY
 * UNPINNED]  Restates Bullet 3.x btMultiBodyDynamicsWorld single step for a
 * base with only zero-mass fixed children (reference call sites
 * BaseAviary.py:542-543; gravity/timestep :673-675; URDF_USE_INERTIA_FROM_FILE :689):
 *  (1) gravity force m*g added in world frame (btMultiBodyDynamicsWorld::solveExternalForces)
 *  (2) btMultiBody::computeAccelerationsArticulatedBodyAlgorithmMultiDof, base part:
 *        bias = -(tau_b,F_b) + (J w (k1+k2|w|), m v (k1+k2|v|)) + (w x J w, m w x v)   [body frame]
 *        k1 = k2 = linear/angularDamping = 0.04f ; gyro term on
 *        world-frame output: wdot = R alpha_b ; vdot = R (a_b + w_b x v_b)
 *  (3) applyDeltaVeeMultiDof: vel += acc*dt, each coordinate clamped to +-maxCoordinateVelocity (100)
 *  (4) stepPositionsMultiDof: pos += v_new*dt (semi-implicit); orientation
 *        q <- dq(w_new, dt) * q with the exponential map of btTransformUtil:
 *        angle clamp ANGULAR_MOTION_THRESHOLD = pi/4, Taylor branch for |w|<0.001; normalise
 *  External forces are cleared after the step (BaseAviary re-applies them each sub-step).
 *  Ground contact: orc_plane_contact below (optional, DSIM_OPT_PLANE), between (3) and (4).  */
/* ======================================================================= */

/* Plane contact (DSIM_OPT_PLANE) — a PRODUCT-DEFINED model, "parity unpinned" like P4 and more so: the reference loads
 * pybullet_data's plane.urdf with collisions on (BaseAviary.py:680, 710-712) and leaves contact to Bullet's
 * btMultiBodyConstraintSolver (GJK/EPA manifold of the base link's collision cylinder, robobee.urdf:72-77, against the
 * plane's box; sequential impulses, 50 iterations, contact ERP 0.2, restitution 0, friction = product of the two
 * lateral-friction coefficients, 1.0 x 0.5).  Neither the engine nor any recording of it is available, so this is NOT a
 * restatement of Bullet's manifold generation; it keeps the documented ingredients of that solver on a fixed manifold:
 *   - the vehicle's collision shape is its bounding cylinder (radius collision_radius, half-height collision_below,
 *     centred on the COM, axis = body z); the manifold is ORC_PLANE_POINTS = 8 BODY-FIXED points on the rim of its lower
 *     face, at 45 degree steps from body x.  (A manifold that follows the rim's lowest point was tried first and
 *     dropped: the azimuth of a nearly level vehicle's tilt is noise, so the support polygon turned with the rounding
 *     of the state and a vehicle rocking through level under rotor torque came out 5 % different in fp32 and fp64.
 *     With fixed points an edge landing is seen at most r sin(tilt) (1 - cos 22.5 deg) = 0.076 r sin(tilt) late.)
 *     The lower face is the one the body z axis points away from; a vehicle lying exactly on its side switches faces.
 *   - a point closer than the contact breaking threshold (0.02 m) is a constraint: penetrating, its normal velocity
 *     is driven to erp depth / dt (erp = 0.2, restitution 0); separated, it may close no faster than gap / dt;
 *   - Coulomb friction mu = contact_friction on the two world tangents, each clamped to +-mu lambda_n (pyramid);
 *   - projected Gauss-Seidel, ORC_PLANE_ITERS sweeps, impulses applied to (v, w) through 1/m and the world inertia;
 * solved between the velocity update and the position update of the step, as a velocity-level solver does.
 * The plane is infinite (pybullet's is a 30 m x 30 m box).  Pinned only by analytic tests (rest, drop, slide, tip). */
#define ORC_PLANE_ITERS 24
#define ORC_PLANE_POINTS 8
#define ORC_PLANE_ERP 0.2
#define ORC_PLANE_MARGIN 0.02
static void inv_inertia_world(const dsim_type_params* P, const double R[9], const double x[3], double y[3]) {
  double b[3];
  for (int k = 0; k < 3; ++k) b[k] = (R[0 * 3 + k] * x[0] + R[1 * 3 + k] * x[1] + R[2 * 3 + k] * x[2]) / P->inertia[k];
  for (int k = 0; k < 3; ++k) y[k] = R[k * 3] * b[0] + R[k * 3 + 1] * b[1] + R[k * 3 + 2] * b[2];
}
void orc_plane_contact(const dsim_type_params* P, double dt, const double pos[3], const double q[4], double v[3], double w[3]) {
  if (!(P->collision_radius > 0)) return;
  double R[9];
  orc_matrix_from_quat(q, R);
  const double a[3] = {R[2], R[5], R[8]};                 /* body z axis in the world */
  const double ex[3] = {R[0], R[3], R[6]}, ey[3] = {R[1], R[4], R[7]};      /* body x, y */
  const double sgn = a[2] >= 0 ? 1.0 : -1.0;              /* which face is the lower one */
  double c[3];
  for (int k = 0; k < 3; ++k) c[k] = -sgn * P->collision_below * a[k];      /* lower face centre, relative to the COM */
  double r[ORC_PLANE_POINTS][3], gap[ORC_PLANE_POINTS], lam[ORC_PLANE_POINTS][3], K[ORC_PLANE_POINTS][3];
  int active[ORC_PLANE_POINTS], any = 0;
  for (int j = 0; j < ORC_PLANE_POINTS; ++j) {
    static const double cb[8] = {1, 0.70710678118654752440, 0, -0.70710678118654752440, -1, -0.70710678118654752440, 0, 0.70710678118654752440};
    static const double sb[8] = {0, 0.70710678118654752440, 1, 0.70710678118654752440, 0, -0.70710678118654752440, -1, -0.70710678118654752440};
    for (int k = 0; k < 3; ++k) r[j][k] = c[k] + P->collision_radius * (cb[j] * ex[k] + sb[j] * ey[k]);
    gap[j] = pos[2] + r[j][2];
    active[j] = gap[j] < ORC_PLANE_MARGIN;
    any |= active[j];
    for (int k = 0; k < 3; ++k) {
      lam[j][k] = 0.0;
      double ax[3] = {k == 0, k == 1, k == 2}, rxn[3], t[3], u[3];
      cross3(r[j], ax, rxn);
      inv_inertia_world(P, R, rxn, t);
      cross3(t, r[j], u);
      K[j][k] = 1.0 / P->mass + u[k];
    }
  }
  if (!any) return;
  const double mu = P->contact_friction;
  for (int it = 0; it < ORC_PLANE_ITERS; ++it)
    for (int j = 0; j < ORC_PLANE_POINTS; ++j) {
      if (!active[j]) continue;
      for (int pass = 0; pass < 3; ++pass) {
        const int k = pass == 0 ? 2 : pass - 1;            /* normal (z) first, then the tangents x, y */
        double wr[3];
        cross3(w, r[j], wr);
        const double u = v[k] + wr[k];
        double dl;
        if (k == 2) {
          const double target = gap[j] < 0 ? ORC_PLANE_ERP * (-gap[j]) / dt : -gap[j] / dt;
          const double nl = fmax(0.0, lam[j][2] + (target - u) / K[j][2]);
          dl = nl - lam[j][2]; lam[j][2] = nl;
        } else {
          const double lim = mu * lam[j][2];
          double nl = lam[j][k] - u / K[j][k];
          nl = nl < -lim ? -lim : (nl > lim ? lim : nl);
          dl = nl - lam[j][k]; lam[j][k] = nl;
        }
        double imp[3] = {0, 0, 0}, rxi[3], dw[3];
        imp[k] = dl;
        v[k] += dl / P->mass;
        cross3(r[j], imp, rxi);
        inv_inertia_world(P, R, rxi, dw);
        for (int m = 0; m < 3; ++m) w[m] += dw[m];
      }
    }
}

void orc_bullet_step_ex(const dsim_type_params* P, double dt, double pos[3], double q[4], double v[3],
                        double w[3], const double F_body[3], const double tau_body[3], int plane);
void orc_bullet_step(const dsim_type_params* P, double dt, double pos[3], double q[4], double v[3],
                     double w[3], const double F_body[3], const double tau_body[3]) {
  orc_bullet_step_ex(P, dt, pos, q, v, w, F_body, tau_body, 0);
}
void orc_bullet_step_ex(const dsim_type_params* P, double dt, double pos[3], double q[4], double v[3],
                        double w[3], const double F_body[3], const double tau_body[3], int plane) {
  double R[9], wb[3], vb[3], Jw[3], gyro[3], alpha_b[3], a_b[3];
  orc_matrix_from_quat(q, R); /* body -> world */
  for (int k = 0; k < 3; ++k) {
    wb[k] = R[0 * 3 + k] * w[0] + R[1 * 3 + k] * w[1] + R[2 * 3 + k] * w[2];
    vb[k] = R[0 * 3 + k] * v[0] + R[1 * 3 + k] * v[1] + R[2 * 3 + k] * v[2];
  }
  /* gravity in body frame: R^T (0,0,-m g) */
  double Fb[3];
  for (int k = 0; k < 3; ++k) Fb[k] = F_body[k] + R[2 * 3 + k] * (-P->gravity * P->mass);
  const double wn = sqrt(wb[0] * wb[0] + wb[1] * wb[1] + wb[2] * wb[2]);
  const double vn = sqrt(vb[0] * vb[0] + vb[1] * vb[1] + vb[2] * vb[2]);
  for (int k = 0; k < 3; ++k) Jw[k] = P->inertia[k] * wb[k];
  cross3(wb, Jw, gyro);
  for (int k = 0; k < 3; ++k) {
    const double damp_a = Jw[k] * (P->ang_damping + P->ang_damping * wn);
    alpha_b[k] = (tau_body[k] - damp_a - gyro[k]) / P->inertia[k];
    const double damp_l = P->mass * vb[k] * (P->lin_damping + P->lin_damping * vn);
    a_b[k] = (Fb[k] - damp_l) / P->mass; /* the m w x v bias cancels against +w_b x v_b on output */
  }
  for (int k = 0; k < 3; ++k) {
    const double wdot = R[k * 3] * alpha_b[0] + R[k * 3 + 1] * alpha_b[1] + R[k * 3 + 2] * alpha_b[2];
    const double vdot = R[k * 3] * a_b[0] + R[k * 3 + 1] * a_b[1] + R[k * 3 + 2] * a_b[2];
    w[k] += wdot * dt;
    v[k] += vdot * dt;
    const double mx = P->max_coord_vel;
    w[k] = w[k] < -mx ? -mx : (w[k] > mx ? mx : w[k]);
    v[k] = v[k] < -mx ? -mx : (v[k] > mx ? mx : v[k]);
  }
  if (plane) orc_plane_contact(P, dt, pos, q, v, w);       /* velocity-level contact solve, then the positions */
  for (int k = 0; k < 3; ++k) pos[k] += dt * v[k];
  double fAngle = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  if (fAngle * dt > 0.5 * (ORC_PI * 0.5)) fAngle = 0.5 * (ORC_PI * 0.5) / dt;
  double axis[3], scale;
  if (fAngle < 0.001) scale = 0.5 * dt - (dt * dt * dt) * 0.020833333333 * fAngle * fAngle;
  else scale = sin(0.5 * fAngle * dt) / fAngle;
  for (int k = 0; k < 3; ++k) axis[k] = w[k] * scale;
  const double dq[4] = {axis[0], axis[1], axis[2], cos(fAngle * dt * 0.5)};
  double qn[4]; /* qn = dq * q (Hamilton) */
  qn[3] = dq[3] * q[3] - dq[0] * q[0] - dq[1] * q[1] - dq[2] * q[2];
  qn[0] = dq[3] * q[0] + dq[0] * q[3] + dq[1] * q[2] - dq[2] * q[1];
  qn[1] = dq[3] * q[1] + dq[1] * q[3] + dq[2] * q[0] - dq[0] * q[2];
  qn[2] = dq[3] * q[2] + dq[2] * q[3] + dq[0] * q[1] - dq[1] * q[0];
  const double len = sqrt(qn[0] * qn[0] + qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3]);
  for (int k = 0; k < 4; ++k) q[k] = qn[k] / len;
}

/* One physics sub-step of BaseAviary.step's inner loop for one drone
 * (BaseAviary.py:510-545): wrench from the clipped action, optional add-ons,
 * then the engine step.  noise: [2*n_act] = f_noise then m_noise, or NULL.   */
void orc_physics_substep(const dsim_type_params* P, double dt, double rigid[13], const double* cmd,
                         const double* last_cmd, const double* noise, uint32_t options,
                         const double* ext_force_body /* nullable [3]: extra LINK_FRAME force at the COM */) {
  double F[3], tau[3], rpm[DSIM_MAX_ACT];
  const double* fn = noise;
  const double* mn = noise ? noise + P->n_act : NULL;
  double* pos = rigid; double* q = rigid + 3; double* v = rigid + 7; double* w = rigid + 10;
  /* The state holds what p.getBasePositionAndOrientation / getBaseVelocity report (BaseAviary.py:726-732): the BASE link's
   * centre of mass.  A vehicle flown as the rigid composite of several links (the morphing hexa) is integrated about the
   * composite's centre of mass, base_offset away: p = p_b - R d, v = v_b - w x (R d) in front of the step, and back. */
  const int shifted = P->base_offset[0] != 0.0 || P->base_offset[1] != 0.0 || P->base_offset[2] != 0.0;
  if (shifted) {
    double R[9], o[3], wo[3];
    orc_matrix_from_quat(q, R);
    for (int k = 0; k < 3; ++k) o[k] = R[3 * k] * P->base_offset[0] + R[3 * k + 1] * P->base_offset[1] + R[3 * k + 2] * P->base_offset[2];
    cross3(w, o, wo);
    for (int k = 0; k < 3; ++k) { pos[k] -= o[k]; v[k] -= wo[k]; }
  }
  /* BaseAviary._physics dispatches on the URDF's configuration type (BaseAviary.py:925-969): both hexa kinds are
     "morphing_hexa" (hexa_6DOF.urdf:25, hexa_6DOF_simple.urdf:25) */
  if (P->kind != DSIM_KIND_QUAD) orc_hexa_wrench(P, cmd, fn, mn, F, tau, rpm);
  else orc_quad_wrench(P, cmd, fn, mn, F, tau, rpm);
  if (options & DSIM_OPT_GROUND) { /* BaseAviary.py:528-529: extra thrust per rotor link */
    double dF[DSIM_MAX_ACT];
    orc_ground_effect(P, pos, q, rpm, dF);
    for (int i = 0; i < P->n_act; ++i) {
      double f[3], t[3];
      for (int k = 0; k < 3; ++k) f[k] = P->rotor_axis[i][k] * dF[i];
      cross3(P->rotor_pos[i], f, t);
      for (int k = 0; k < 3; ++k) { F[k] += f[k]; tau[k] += t[k]; }
    }
  }
  if (options & DSIM_OPT_DRAG) { /* BaseAviary.py:531-532: uses the LAST step's rpm */
    double lrpm[DSIM_MAX_ACT], D[3];
    for (int i = 0; i < P->n_act; ++i) lrpm[i] = P->pwm2rpm_scale[i] * last_cmd[i] + P->pwm2rpm_const[i];
    orc_drag(P, q, v, lrpm, D);
    for (int k = 0; k < 3; ++k) F[k] += D[k];
  }
  if (ext_force_body) for (int k = 0; k < 3; ++k) F[k] += ext_force_body[k];
  orc_bullet_step_ex(P, dt, pos, q, v, w, F, tau, (options & DSIM_OPT_PLANE) != 0);
  if (shifted) {
    double R[9], o[3], wo[3];
    orc_matrix_from_quat(q, R);
    for (int k = 0; k < 3; ++k) o[k] = R[3 * k] * P->base_offset[0] + R[3 * k + 1] * P->base_offset[1] + R[3 * k + 2] * P->base_offset[2];
    cross3(w, o, wo);
    for (int k = 0; k < 3; ++k) { pos[k] += o[k]; v[k] += wo[k]; }
  }
}

/* ======================================================================= */
/* Product noise definition (NOT a reference restatement: the reference draws
 * from the unseeded global numpy RNG, BaseAviary.py:1518-1525, which cannot be
 * reproduced).  Threefry4x32-12 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3",
 * SC'11 — the Random123 generator; add/rotate/xor only) keyed by the seed, counter = (drone, block index),
 * Box-Muller on 8+8-bit halves -> unit-variance normals on a 256 x 256 grid of cell centres, |n| <= 3.535 (orc_noise_normals).
 * Mirrors dsim_device.h so that tests can feed the oracle the very normals the kernel draws.
 * orc_threefry4x32 takes the round count so that the published known-answer vectors (13 and 20 rounds,
 * Random123 kat_vectors) pin the round function, rotation constants and key schedule
 * (tests/test_oracle_physics.py).                                                                               */
/* ======================================================================= */
static inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
void orc_threefry4x32(uint32_t x[4], const uint32_t key[4], int rounds) {
  static const int R0[8] = {10, 11, 13, 23, 6, 17, 25, 18}, R1[8] = {26, 21, 27, 5, 20, 11, 10, 20};
  const uint32_t ks[5] = {key[0], key[1], key[2], key[3], 0x1BD11BDAu ^ key[0] ^ key[1] ^ key[2] ^ key[3]};
  for (int i = 0; i < 4; ++i) x[i] += ks[i];
  for (int r = 0; r < rounds; ++r) {
    if ((r & 1) == 0) {
      x[0] += x[1]; x[1] = rotl32(x[1], R0[r & 7]) ^ x[0];
      x[2] += x[3]; x[3] = rotl32(x[3], R1[r & 7]) ^ x[2];
    } else {
      x[0] += x[3]; x[3] = rotl32(x[3], R0[r & 7]) ^ x[0];
      x[2] += x[1]; x[1] = rotl32(x[1], R1[r & 7]) ^ x[2];
    }
    if ((r & 3) == 3) {
      const int s = r / 4 + 1;
      x[0] += ks[s % 5]; x[1] += ks[(s + 1) % 5]; x[2] += ks[(s + 2) % 5]; x[3] += ks[(s + 3) % 5] + (uint32_t)s;
    }
  }
}
/* UNIT normals for (drone, sub-step counter): out[2 n_act] = f_noise[n_act] then m_noise[n_act] (the caller scales by 0.01 /
 * 0.001).  16 bits make a Box-Muller pair: radius from the high byte (u1 = (h + 1/2) / 256), angle from the low byte
 * (u2 = (l + 1/2) / 256 revolutions); the radius is scaled by ORC_BM8_CORR = 2 / mean(-2 ln u1 over the 256 values) so that the
 * variance is exactly 1 (|n| <= 3.535, kurtosis 2.977, no atoms: dsim_device.h:box_muller8).  One block = 8 pairs:
 *   block index = sub_counter >> 1; the even sub-step takes words 0 and 1, the odd one words 2 and 3;
 *   quad: force normals from the first word, moment normals from the second;
 *   six-actuator kinds: SIX normals (first word, low half of the second) — those of the body wrench, see below.
 *   Within a word the low 16 bits come first. */
#define ORC_BM8_CORR 1.0013550008475642
static void orc_bm8(uint32_t w, int half, double* n0, double* n1) {
  const uint32_t v = half ? (w >> 16) : (w & 0xFFFFu);
  const double u1 = ((double)(v >> 8) + 0.5) * (1.0 / 256.0);
  const double u2 = ((double)(v & 0xFFu) + 0.5) * (1.0 / 256.0);
  const double r = sqrt(-2.0 * ORC_BM8_CORR * log(u1));
  *n0 = r * cos(2 * ORC_PI * u2);
  *n1 = r * sin(2 * ORC_PI * u2);
}
void orc_noise_normals(uint64_t seed, uint64_t drone, uint64_t sub_counter, int n_act, double* out) {
  const uint32_t key[4] = {(uint32_t)seed, (uint32_t)(seed >> 32), 0u, 0u};
  const uint64_t blk = sub_counter >> 1;
  uint32_t c[4] = {(uint32_t)drone, (uint32_t)(drone >> 32), (uint32_t)blk, (uint32_t)(blk >> 32)};
  orc_threefry4x32(c, key, 12);
  const int odd = (int)(sub_counter & 1u);
  const uint32_t wa = odd ? c[2] : c[0], wb = odd ? c[3] : c[1];
  if (n_act == 4) {
    orc_bm8(wa, 0, out + 0, out + 1); orc_bm8(wa, 1, out + 2, out + 3);
    orc_bm8(wb, 0, out + 4, out + 5); orc_bm8(wb, 1, out + 6, out + 7);
  } else {
    /* six-actuator kinds: the six unit normals z of the BODY WRENCH (dsim_device.h:noise_normals: W = L z, L the Cholesky factor
     * of the covariance of what the twelve per-rotor normals of BaseAviary.py:1429-1430 add up to); out[6 .. 12) = 0 */
    orc_bm8(wa, 0, out + 0, out + 1); orc_bm8(wa, 1, out + 2, out + 3); orc_bm8(wb, 0, out + 4, out + 5);
    for (int j = 6; j < 12; ++j) out[j] = 0.0;
  }
}

/* DSIM_OPT_NOISE_FINE (dsim_device.h:box_muller16, quad_normals_fine, hexa_normals_fine): 16 + 16 bits per pair — radius from
 * the high half of a word (u1 = (h + 1/2) / 65536), direction from the low half; blocks in a domain of their own (counter word
 * 3's top bit), block `sub`; quad: words 0, 1 force, 2, 3 moment; six-actuator kinds: words 0, 1, 2.  Lattice points at the centres
 * of the cells (u1 = (h + 1/2) / 65536, u2 = (l + 1/2) / 65536): no draw is exactly 0. */
#define ORC_BM16_CORR 1.0000052883115735
static void orc_bm16(uint32_t w, double* n0, double* n1) {
  const double u1 = ((double)(w >> 16) + 0.5) * (1.0 / 65536.0);
  const double u2 = ((double)(w & 0xFFFFu) + 0.5) * (1.0 / 65536.0);
  const double r = sqrt(-2.0 * ORC_BM16_CORR * log(u1));
  *n0 = r * cos(2 * ORC_PI * u2);
  *n1 = r * sin(2 * ORC_PI * u2);
}
static void fine_block(uint64_t seed, uint64_t drone, uint64_t blk, uint32_t c[4]) {
  const uint32_t key[4] = {(uint32_t)seed, (uint32_t)(seed >> 32), 0u, 0u};
  c[0] = (uint32_t)drone; c[1] = (uint32_t)(drone >> 32); c[2] = (uint32_t)blk; c[3] = (uint32_t)(blk >> 32) | 0x80000000u;
  orc_threefry4x32(c, key, 12);
}
void orc_noise_normals_fine(uint64_t seed, uint64_t drone, uint64_t sub_counter, int n_act, double* out) {
  uint32_t c[4];
  fine_block(seed, drone, sub_counter, c);
  if (n_act == 4) {
    orc_bm16(c[0], out + 0, out + 1); orc_bm16(c[1], out + 2, out + 3); orc_bm16(c[2], out + 4, out + 5); orc_bm16(c[3], out + 6, out + 7);
  } else {                       /* the six normals of the body wrench (orc_noise_normals), words 0, 1, 2; out[6 .. 12) = 0 */
    orc_bm16(c[0], out + 0, out + 1); orc_bm16(c[1], out + 2, out + 3); orc_bm16(c[2], out + 4, out + 5);
    for (int j = 6; j < 12; ++j) out[j] = 0.0;
  }
}
/* many draws at once (distribution tests): out [n_drones][n_sub][2 n_act] */
void orc_noise_normals_batch(uint64_t seed, uint64_t drone0, int64_t n_drones, uint64_t sub0, int n_sub, int n_act, int fine,
                             double* out, int nthreads) {
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n_drones; ++i)
    for (int s = 0; s < n_sub; ++s) {
      double* o = out + ((size_t)i * n_sub + s) * 2 * n_act;
      if (fine) orc_noise_normals_fine(seed, drone0 + (uint64_t)i, sub0 + (uint64_t)s, n_act, o);
      else orc_noise_normals(seed, drone0 + (uint64_t)i, sub0 + (uint64_t)s, n_act, o);
    }
}

/* ======================================================================= */
/* batch drivers (AoS fp64, OpenMP over drones) — used by tests and the
 * cpu_baseline leg of bench.py                                               */
/* rigid [n][13] = pos3 quat4 vel3 angvel3 ; mem [n][13] = last_vel3 last_rates3
 * last_thrust cmd6 ; tgt [n][10] = pos3 vel3 acc3 yaw (or one row if bcast)   */
/* ======================================================================= */
static void mem_load(const double* m, orc_ctrl_mem* c) { memcpy(c, m, sizeof(double) * 13); }
static void mem_store(const orc_ctrl_mem* c, double* m) { memcpy(m, c, sizeof(double) * 13); }

int orc_control_batch(const dsim_type_params* types, const uint8_t* type_id, int64_t n, double dt,
                      const double* rigid, double* mem, const double* tgt, int bcast_tgt,
                      double* pos_e_out, double* yaw_e_out, int nthreads) {
  int fail = 0;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(static) reduction(| : fail)
  for (int64_t i = 0; i < n; ++i) {
    const dsim_type_params* P = &types[type_id ? type_id[i] : 0];
    const double* r = rigid + i * 13;
    const double* t = tgt + (bcast_tgt ? 0 : i * 10);
    orc_ctrl_mem c;
    mem_load(mem + i * 13, &c);
    const double trpy[3] = {0, 0, t[9]};
    double pe[3], ye;
    int rc = orc_indi_compute_control(P, dt, r, r + 3, r + 7, r + 10, t, t + 3, t + 6, trpy, &c, pe, &ye);
    if (rc) fail |= 1;
    mem_store(&c, mem + i * 13);
    if (pos_e_out) memcpy(pos_e_out + i * 3, pe, sizeof(pe));
    if (yaw_e_out) yaw_e_out[i] = ye;
  }
  return fail ? -1 : 0;
}

/* Env.step physics part: substeps with action [n][6] (NULL = the controller's stored
 * cmd); the clipped action goes to last_action_out [n][6] (nullable): the env's
 * last_clipped_action, BaseAviary.py:545.  The controller memory is not modified
 * (env and controller are separate objects in the reference).
 * noise [n][substeps][12] (f_noise6, m_noise6; quad uses [0:4] of each) or NULL. */
int orc_physics_batch(const dsim_type_params* types, const uint8_t* type_id, int64_t n, int substeps,
                      double dt, double* rigid, const double* action, const double* mem, const double* noise,
                      uint32_t options, double* last_action_out, const double* ext_force_body /* [n][3] nullable */,
                      int nthreads) {
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    const dsim_type_params* P = &types[type_id ? type_id[i] : 0];
    const double* m = mem + i * 13;
    double clipped[DSIM_MAX_ACT] = {0}, last[DSIM_MAX_ACT];
    orc_preprocess_action(P, action ? action + i * 6 : m + 7, clipped);
    /* drag uses the PREVIOUS step's action on sub-step 0 (BaseAviary.py:532, 545); without an env-side
       last_clipped_action buffer (fused stepping) the current action stands in */
    memcpy(last, last_action_out ? last_action_out + i * 6 : clipped, sizeof(last));
    for (int s = 0; s < substeps; ++s) {
      double nz[12];
      const double* np_ = NULL;
      if (noise) {
        const double* src = noise + (i * substeps + s) * 12;
        for (int j = 0; j < P->n_act; ++j) { nz[j] = src[j]; nz[P->n_act + j] = src[6 + j]; }
        np_ = nz;
      }
      orc_physics_substep(P, dt, rigid + i * 13, clipped, s == 0 ? last : clipped, np_, options,
                          ext_force_body ? ext_force_body + i * 3 : NULL);
    }
    if (last_action_out) memcpy(last_action_out + i * 6, clipped, sizeof(clipped));
  }
  return 0;
}

/* fused Env.step + computeControl, the example loop body (fly_INDI.py:223-239) */
int orc_step_batch(const dsim_type_params* types, const uint8_t* type_id, int64_t n, int substeps,
                   double dt_phys, double dt_ctrl, double* rigid, double* mem, const double* tgt,
                   int bcast_tgt, const double* noise, uint32_t options, const double* action,
                   const double* ext_force_body, int nthreads) {
  orc_physics_batch(types, type_id, n, substeps, dt_phys, rigid, action, mem, noise, options, NULL, ext_force_body, nthreads);
  return orc_control_batch(types, type_id, n, dt_ctrl, rigid, mem, tgt, bcast_tgt, NULL, NULL, nthreads);
}

/* P8: BaseAviary._downwash, BaseAviary.py:1736-1763 (formula only; dead code in the fork).
 * For drone i and every drone j of the world above it (dz > 0) within dxy < 10 m:
 *   alpha = DW1 (PROP_RADIUS / (4 dz))^2 ; beta = DW2 dz + DW3 ; Fz -= alpha exp(-0.5 (dxy/beta)^2)
 * with the RECEIVING drone's coefficients; applied along the body z axis at the COM (LINK_FRAME,
 * link 4).  Brute force O(n m).  pos_all [m][3]: every drone of the world; rigid [n][13]: the
 * receivers (their own entry in pos_all has dz = 0 and drops out).  fz_out [n]. */
void orc_downwash(const dsim_type_params* types, const uint8_t* type_id, int64_t n, const double* rigid,
                  const double* pos_all, int64_t m, double* fz_out, int nthreads) {
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    const dsim_type_params* P = &types[type_id ? type_id[i] : 0];
    const double* p = rigid + i * 13;
    double fz = 0;
    for (int64_t j = 0; j < m; ++j) {
      const double dz = pos_all[j * 3 + 2] - p[2];
      const double dx = pos_all[j * 3] - p[0], dy = pos_all[j * 3 + 1] - p[1];
      const double dxy = sqrt(dx * dx + dy * dy);
      if (dz > 0 && dxy < 10) {
        const double r = P->prop_radius / (4 * dz);
        const double alpha = P->dw_coeff[0] * r * r;
        const double beta = P->dw_coeff[1] * dz + P->dw_coeff[2];
        const double q = dxy / beta;
        fz += -alpha * exp(-0.5 * q * q);
      }
    }
    fz_out[i] = fz;
  }
}

/* Env.step with the neighbour-downwash term (Physics.PYB_DW), as BaseAviary.step loops it (BaseAviary.py:510-536): with
 * AGGR_PHY_STEPS > 1 the kinematic information is refreshed at the top of EVERY sub-step (:513-520) and _downwash(i) reads
 * those refreshed positions (:534-536, 1747-1751) — the term is evaluated per sub-step, from the positions at the start of
 * that sub-step (with one sub-step: from the positions the previous Env.step left, :547).  The world is the n drones of
 * `rigid`.  action / mem / noise [n][substeps][12] / last_action_out as orc_physics_batch. */
int orc_physics_downwash_batch(const dsim_type_params* types, const uint8_t* type_id, int64_t n, int substeps, double dt,
                               double* rigid, const double* action, const double* mem, const double* noise,
                               uint32_t options, double* last_action_out, int nthreads) {
  double* fz = (double*)malloc(sizeof(double) * (size_t)n);
  double* pos = (double*)malloc(sizeof(double) * 3 * (size_t)n);
  double* ext = (double*)calloc((size_t)n * 3, sizeof(double));
  double* nz = noise ? (double*)malloc(sizeof(double) * 12 * (size_t)n) : NULL;
  if (!fz || !pos || !ext || (noise && !nz)) { free(fz); free(pos); free(ext); free(nz); return -1; }
  for (int s = 0; s < substeps; ++s) {
    for (int64_t i = 0; i < n; ++i) memcpy(pos + 3 * i, rigid + 13 * i, sizeof(double) * 3);
    orc_downwash(types, type_id, n, rigid, pos, n, fz, nthreads);
    for (int64_t i = 0; i < n; ++i) ext[3 * i + 2] = fz[i];
    if (noise) for (int64_t i = 0; i < n; ++i) memcpy(nz + 12 * i, noise + (i * substeps + s) * 12, sizeof(double) * 12);
    orc_physics_batch(types, type_id, n, 1, dt, rigid, action, mem, nz, options, last_action_out, ext, nthreads);
  }
  free(fz); free(pos); free(ext); free(nz);
  return 0;
}

/* ======================================================================= */
/* D1: BaseAviary._dynamics, BaseAviary.py:1767-1828 — the reference's OWN explicit rigid-body model (Physics.DYN).
 * PINNED: tests/golden/dynamics.npz holds what the reference's function hands p.resetBasePositionAndOrientation /
 * p.resetBaseVelocity and stores in self.rpy_rates, for recorded inputs (tests/golden/make_goldens.py:capture_dynamics);
 * tests/test_oracle_dynamics.py compares at 1e-12.  The branch is dead code in the fork for plumbing reasons only (it reads
 * self.KF, self.M, self.J, self.J_INV, self.L, self.GRAVITY, self.DRONE_MODEL, which the fork moved into self.drones[i] /
 * dropped, and indexes the action dict as an array, :527): the arithmetic below is the function's own, line by line.
 * pos, vel, rpy_rates in-out; quat / rpy in (self.quat, self.rpy as _updateAndStoreKinematicInformation left them, :726-729);
 * rpy_out = the summed angles handed to getQuaternionFromEuler (:1812, 1817); quat_out = that quaternion.             */
/* ======================================================================= */
void orc_dynamics(const dsim_type_params* P, double dt, const double rpm[4], double pos[3], const double quat[4],
                  const double rpy[3], double vel[3], double rpy_rates[3], double rpy_out[3], double quat_out[4]) {
  double R[9], forces[4], zt[4];
  orc_matrix_from_quat(quat, R);                                           /* :1786 */
  double thrust = 0;
  for (int i = 0; i < 4; ++i) { forces[i] = rpm[i] * rpm[i] * P->kf; thrust += forces[i]; }    /* :1788-1789 */
  const double weight = P->gravity * P->mass;                              /* self.GRAVITY = self.G*self.M, :226 */
  const double fw[3] = {R[2] * thrust, R[5] * thrust, R[8] * thrust - weight};                  /* :1790-1791 */
  for (int i = 0; i < 4; ++i) zt[i] = rpm[i] * rpm[i] * P->km;             /* :1792 */
  const double z_torque = -zt[0] + zt[1] - zt[2] + zt[3];                  /* :1793 */
  double x_torque, y_torque;
  if (P->dyn_mixer == DSIM_DYN_MIXER_PLUS) {                               /* DroneModel.CF2P / HB, :1801-1803 */
    x_torque = (forces[1] - forces[3]) * P->arm;
    y_torque = (-forces[0] + forces[2]) * P->arm;
  } else {                                                                 /* DroneModel.CF2X, :1794-1800 */
    x_torque = (forces[0] + forces[1] - forces[2] - forces[3]) * (P->arm / sqrt(2.0));
    y_torque = (-forces[0] + forces[1] + forces[2] - forces[3]) * (P->arm / sqrt(2.0));
  }
  double tq[3] = {x_torque, y_torque, z_torque}, Jw[3], gy[3];
  for (int k = 0; k < 3; ++k) Jw[k] = P->inertia[k] * rpy_rates[k];        /* np.dot(self.J, rpy_rates), J = diag (:2066) */
  cross3(rpy_rates, Jw, gy);
  for (int k = 0; k < 3; ++k) tq[k] -= gy[k];                              /* :1805 */
  for (int k = 0; k < 3; ++k) {
    const double deriv = (1.0 / P->inertia[k]) * tq[k];                    /* np.dot(self.J_INV, torques), :1806, 2067 */
    vel[k] = vel[k] + dt * (fw[k] / P->mass);                              /* :1807, 1809 */
    rpy_rates[k] = rpy_rates[k] + dt * deriv;                              /* :1810 */
    pos[k] = pos[k] + dt * vel[k];                                         /* :1811 */
    rpy_out[k] = rpy[k] + dt * rpy_rates[k];                               /* :1812 */
  }
  orc_quat_from_euler(rpy_out, quat_out);                                  /* :1817 */
}

/* BaseAviary.step with PHYSICS == Physics.DYN (BaseAviary.py:510-545) for a fleet: per sub-step the kinematic information
 * is what the engine reports (refreshed at the top of the sub-step when AGGR_PHY_STEPS > 1, :513-520, and behind the
 * last one, :547: self.rpy = getEulerFromQuaternion(quat), :729), then _dynamics(clipped_action, i) (:525-527), no
 * p.stepSimulation (:541-543).  The argument is documented as RPMs (:1770-1775) and the fork's Env hands PWM commands
 * (CtrlAviary.py:258-263): rpm = PWM2RPM_SCALE * pwm + PWM2RPM_CONST as the fork's own force map does (:1487-1490).
 * rigid [n][13]: ang_v is what p.getBaseVelocity reports afterwards — the placeholder (-1, -1, -1) (:1821-1826), or with
 * DSIM_OPT_DYN_BODY_RATES (a product-defined deviation, include/dronesim_amd.h) R(quat) rpy_rates.  rates [n][3] =
 * self.rpy_rates.  action [n][6] (NULL = the stored cmd of mem), last_action_out [n][6] nullable. */
int orc_dyn_physics_batch(const dsim_type_params* types, const uint8_t* type_id, int64_t n, int substeps, double dt,
                          double* rigid, double* rates, const double* action, const double* mem, uint32_t options,
                          double* last_action_out, int nthreads) {
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
  int bad = 0;
#pragma omp parallel for schedule(static) reduction(| : bad)
  for (int64_t i = 0; i < n; ++i) {
    const dsim_type_params* P = &types[type_id ? type_id[i] : 0];
    if (P->n_act != 4) { bad |= 1; continue; }                             /* both mixers read forces[0..3] */
    double clipped[DSIM_MAX_ACT] = {0}, rpm[4];
    orc_preprocess_action(P, action ? action + i * 6 : mem + i * 13 + 7, clipped);
    for (int j = 0; j < 4; ++j) rpm[j] = P->pwm2rpm_scale[j] * clipped[j] + P->pwm2rpm_const[j];
    double* r = rigid + i * 13;
    for (int s = 0; s < substeps; ++s) {
      double rpy[3], rpy_new[3], q_new[4];
      orc_euler_from_quat(r + 3, rpy);                                     /* :729 */
      orc_dynamics(P, dt, rpm, r, r + 3, rpy, r + 7, rates + i * 3, rpy_new, q_new);
      memcpy(r + 3, q_new, sizeof(q_new));
    }
    if (substeps > 0) {
      if (options & DSIM_OPT_DYN_BODY_RATES) {
        double R[9];
        orc_matrix_from_quat(r + 3, R);
        const double* w = rates + i * 3;
        for (int k = 0; k < 3; ++k) r[10 + k] = R[3 * k] * w[0] + R[3 * k + 1] * w[1] + R[3 * k + 2] * w[2];
      } else {
        r[10] = r[11] = r[12] = -1.0;                                      /* :1824 */
      }
    }
    if (last_action_out) memcpy(last_action_out + i * 6, clipped, sizeof(clipped));
  }
  return bad ? -1 : 0;
}

/* the example loop body on Physics.DYN: Env.step, then computeControl on the state it reports (fly_INDI.py:223-239) */
int orc_dyn_step_batch(const dsim_type_params* types, const uint8_t* type_id, int64_t n, int substeps, double dt_phys,
                       double dt_ctrl, double* rigid, double* rates, double* mem, const double* tgt, int bcast_tgt,
                       uint32_t options, const double* action, int nthreads) {
  int rc = orc_dyn_physics_batch(types, type_id, n, substeps, dt_phys, rigid, rates, action, mem, options, NULL, nthreads);
  if (rc) return rc;
  return orc_control_batch(types, type_id, n, dt_ctrl, rigid, mem, tgt, bcast_tgt, NULL, NULL, nthreads);
}

/* trajGenerator.get_des_state + get_yaw, dronesim/utils/trajGen.py:108-143 (polyder: trajutils.py:13-21).
 * coeffs [n_seg*10][3] (row-major, as trajGenerator.coeffs), TS [n_seg+1].  yaw_state = (yaw, heading_x,
 * heading_y) is the sampler's memory (trajGen.py:128-143: yaw integrates the signed angle between
 * consecutive velocity headings), updated in place.  out10 = pos3 vel3 acc3 yaw. */
void orc_traj_sample(const double* coeffs, const double* TS, int n_seg, double t, double yaw_state[3], double out10[10]) {
  if (t > TS[n_seg]) t = TS[n_seg] - 0.001;                       /* :110-111 */
  int seg = 0;
  for (int k = 0; k <= n_seg; ++k) if (t >= TS[k]) seg = k;       /* :113 np.where(t >= TS)[0][-1] */
  if (seg >= n_seg) seg = n_seg - 1;                              /* t == TS[-1] cannot happen after the clamp above */
  t -= TS[seg];
  double pw[10];
  pw[0] = 1.0;
  for (int j = 1; j < 10; ++j) pw[j] = pw[j - 1] * t;
  for (int d = 0; d < 3; ++d) {
    double p = 0, v = 0, a = 0;
    for (int j = 0; j < 10; ++j) {
      const double c = coeffs[(seg * 10 + j) * 3 + d];
      p += c * pw[j];
      if (j >= 1) v += c * j * pw[j - 1];
      if (j >= 2) a += c * j * (j - 1) * pw[j - 2];
    }
    out10[d] = p; out10[3 + d] = v; out10[6 + d] = a;
  }
  /* get_yaw(vel[:2]) :128-143 */
  const double nv = sqrt(out10[3] * out10[3] + out10[4] * out10[4]);
  const double cx = out10[3] / nv, cy = out10[4] / nv;
  double cosine = yaw_state[1] * cx + yaw_state[2] * cy;
  cosine = fmax(-1.0, fmin(cosine, 1.0));
  const double dyaw = acos(cosine);
  const double cr = yaw_state[1] * cy - yaw_state[2] * cx;        /* np.cross of 2-vectors */
  yaw_state[0] += (cr > 0 ? 1.0 : (cr < 0 ? -1.0 : 0.0)) * dyaw;
  if (yaw_state[0] > ORC_PI) yaw_state[0] -= 2 * ORC_PI;
  if (yaw_state[0] < -ORC_PI) yaw_state[0] += 2 * ORC_PI;
  yaw_state[1] = cx; yaw_state[2] = cy;
  out10[9] = yaw_state[0];
}

/* Env.step of the alternate action adaptors: _preprocessAction runs (part of) the INDI law on the
 * CURRENT state, then BaseAviary.step's physics loop runs with that command.
 *   mode 0: VelocityAviary._preprocessAction, VelocityAviary.py:221-264
 *   mode 1: RPYTAviary._preprocessAction, RPYTAviary.py:181-193
 * action [n][4]. */
int orc_adaptor_step_batch(const dsim_type_params* types, const uint8_t* type_id, int64_t n, int mode,
                           int substeps, double dt_phys, double dt_ctrl, double* rigid, double* mem,
                           const double* action, int nthreads) {
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    const dsim_type_params* P = &types[type_id ? type_id[i] : 0];
    double* r = rigid + i * 13;
    const double* v = action + i * 4;
    orc_ctrl_mem c;
    mem_load(mem + i * 13, &c);
    if (mode == 0) {
      double rpy[3], tvel[3] = {0, 0, 0}, pe[3], ye;
      orc_euler_from_quat(r + 3, rpy);                                   /* state[9] */
      const double nrm = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);  /* :243-246 */
      const double speed_limit = P->max_speed_kmh * (1000.0 / 3600.0);   /* :92-94 */
      if (nrm != 0) for (int k = 0; k < 3; ++k) tvel[k] = speed_limit * fabs(v[3]) * (v[k] / nrm);   /* :256-258 */
      const double tacc[3] = {0, 0, 0}, trpy[3] = {0, 0, rpy[2]};        /* :255 */
      orc_indi_compute_control(P, dt_ctrl, r, r + 3, r + 7, r + 10, r /* target_pos = cur pos */, tvel, tacc,
                               trpy, &c, pe, &ye);
    } else {
      orc_indi_rate(P, dt_ctrl, v[3], r + 3, r + 10, v, &c);             /* RPYTAviary.py:185-190 */
    }
    mem_store(&c, mem + i * 13);
    for (int s = 0; s < substeps; ++s) orc_physics_substep(P, dt_phys, r, c.cmd, c.cmd, NULL, 0, NULL);
  }
  return 0;
}

/* P5: _getDroneStateVector, BaseAviary.py:780-790: [pos quat rpy vel ang_v last_action] */
void orc_state_vector(const dsim_type_params* P, const double rigid[13], const double* last_action, double* out) {
  memcpy(out, rigid, sizeof(double) * 7);
  orc_euler_from_quat(rigid + 3, out + 7);
  memcpy(out + 10, rigid + 7, sizeof(double) * 6);
  memcpy(out + 16, last_action, sizeof(double) * P->n_act);
}

/* the same for a fleet: rigid [n][13], last_action [n][6], out [n][width] (width = 16 + max n_act of the table;
 * rows of types with fewer actuators are zero-filled behind their own n_act, as the device rows are) */
void orc_state_vector_batch(const dsim_type_params* types, const uint8_t* type_id, int64_t n, const double* rigid,
                            const double* last_action, int width, double* out) {
  for (int64_t i = 0; i < n; ++i) {
    const dsim_type_params* P = &types[type_id ? type_id[i] : 0];
    double* o = out + i * width;
    for (int j = 16; j < width; ++j) o[j] = 0.0;
    orc_state_vector(P, rigid + i * 13, last_action + i * 6, o);
  }
}

int orc_sizeof_params(void) { return (int)sizeof(dsim_type_params); }
int orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
