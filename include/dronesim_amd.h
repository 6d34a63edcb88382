/* dronesim_amd.h — C-ABI of the MI355X-native fleet dynamics + INDI control step.
 *
 * The reference (enac-drones/dronesim, pure Python) has no FFI: its boundary on
 * this path is two Python call surfaces,
 *     gym.Env      reset()/step(action)      dronesim/envs/BaseAviary.py:406-555
 *     BaseControl  computeControl(...)       dronesim/control/BaseControl.py:107-149
 *                                            dronesim/control/INDIControl.py:154-227
 * so the entry points below are what a ctypes binding on the reference side
 * would bind to replace the per-drone Python loops behind those two surfaces
 * (INTEGRATION.md shows that binding).  Each entry point cites the reference
 * code it replaces.
 *
 * Conventions (same as the reference): quaternions xyzw (w at index 3,
 * dronesim/utils/math.py:6,25,47); world frame z-up; angular velocity stored in
 * WORLD frame (BaseAviary.py:730); PWM commands in [pwm_min, pwm_max].
 *
 * Memory: every state / target / action buffer is CALLER-OWNED DEVICE memory
 * (e.g. a torch-ROCm tensor's data_ptr()).  The library allocates nothing per
 * call, never frees caller memory and never synchronises the device; work is
 * stream-ordered on the caller's hipStream_t (passed as void*).  A dsim_ctx owns
 * only the per-type constant table (device copy), a few diagnostic counters and, for
 * fleets with the morphing hexa, the queue of deferred WLS fallbacks (grown when a
 * larger fleet is first seen).  One ctx per device; a ctx may be used from one host
 * thread at a time; the calling thread's current HIP device must be the ctx's device
 * (dsim_create makes it so) and the stream must belong to it.
 *
 * Errors: int return, 0 = OK, negative = library error (DSIM_E_*), positive =
 * hipError_t.  No exceptions or aborts cross the ABI.  There is NO CPU fallback:
 * without a HIP device dsim_create fails.
 */
#ifndef DRONESIM_AMD_H
#define DRONESIM_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSIM_ABI_VERSION 9
#define DSIM_MAX_ACT 6     /* actuators per vehicle (quad 4, morphing hexa 6) */
#define DSIM_MAX_TYPES 8

/* ---- state layout ---------------------------------------------------------
 * Fleet state is ONE fp32 device array in "blocked SoA":
 *     addr(field f, drone i) = base + (i / block) * block_stride
 *                                   + f * field_stride + (i % block)
 * Plain SoA [F][n_pad] is block = n_pad, field_stride = n_pad; the wave-tiled
 * form [n_pad/64][F][64] is block = 64, field_stride = 64, block_stride = F*64.
 * n_pad must be a multiple of 64 (padding lanes are computed and ignored).
 *
 * Field order (DSIM_F_*): the 13 rigid-body floats the reference reads back
 * from Bullet (BaseAviary.py:718-732) followed by the controller memory the
 * reference keeps on each INDIControl instance (INDIControl.py:109-146).      */
enum {
  DSIM_F_POS = 0,        /* 3  world position                                   */
  DSIM_F_QUAT = 3,       /* 4  xyzw                                            */
  DSIM_F_VEL = 7,        /* 3  world linear velocity                           */
  DSIM_F_ANGVEL = 10,    /* 3  world angular velocity                          */
  DSIM_F_LAST_VEL = 13,  /* 3  INDIControl.last_vel   (INDIControl.py:130,291) */
  DSIM_F_LAST_RATES = 16,/* 3  INDIControl.last_rates (INDIControl.py:125,442) */
  DSIM_F_LAST_THRUST = 19,/*1  INDIControl.last_thrust(INDIControl.py:127,455) */
  DSIM_F_CMD = 20,       /* n_act  INDIControl.cmd == the action fed to step() */
  DSIM_NF_QUAD = 24,     /* fields for a 4-actuator fleet                      */
  DSIM_NF_HEXA = 26      /* fields for a 6-actuator (or mixed) fleet           */
};

/* Per-step targets, same blocked-SoA addressing, 10 fields:
 * target_pos3, target_vel3, target_acc3, target_yaw (only target_rpy[2] is read
 * by the reference, INDIControl.py:341; target_rpy_rates is ignored, :404-410) */
enum { DSIM_T_POS = 0, DSIM_T_VEL = 3, DSIM_T_ACC = 6, DSIM_T_YAW = 9, DSIM_NT = 10 };

typedef struct dsim_view {
  float*  base;          /* device pointer                                      */
  int64_t n_pad;         /* padded drone count, multiple of 64                  */
  int64_t block;         /* drones per block (n_pad for plain SoA, or 64)       */
  int64_t field_stride;  /* floats between consecutive fields inside a block    */
  int64_t block_stride;  /* floats between consecutive blocks                   */
  int32_t n_fields;      /* DSIM_NF_QUAD / DSIM_NF_HEXA / DSIM_NT               */
  int32_t _pad;
} dsim_view;

/* ---- per-type constants (host side, fp64; converted to fp32 on upload) -----
 * Filled by the host from the vehicle URDF exactly as the reference does
 * (BaseAviary._parseURDFParameters, BaseAviary.py:2041-2140;
 *  INDIControl._parseURDFControlParameters, INDIControl.py:55-106).            */
enum {
  DSIM_KIND_QUAD = 0,          /* quad physics (BaseAviary.py:1477-1543) + the quad INDI law (INDIControl.py)                    */
  DSIM_KIND_HEXA6DOF = 1,      /* morphing-hexa physics (BaseAviary.py:1389-1457) + the 6-DOF INDI law with WLS allocation
                                  (INDIControl_6DOF.py): hexa_6DOF.urdf                                                        */
  DSIM_KIND_HEXA_QUADLAW = 2   /* morphing-hexa physics + the QUAD law on six actuators — G1 is 4 x 6, alloc = pinv(G1/0.05)
                                  is 6 x 4, every one of the six commands is incremented and clipped (INDIControl.py:457-487
                                  with actuator_nr = 6): hexa_6DOF_simple.urdf:25-34, examples/fly_hexa_6DOF_simple.py:18   */
};

typedef struct dsim_type_params {
  int32_t kind;                       /* DSIM_KIND_*                                        */
  int32_t n_act;                      /* 4 | 6   (indi actuator_nr); 6 for both hexa kinds   */
  double  mass;                       /* total rigid-body mass used by the integrator       */
  double  inertia[3];                 /* principal moments (URDF ixx,iyy,izz)               */
  double  kf, km;                     /* thrust / drag-torque coefficients                  */
  double  pwm2rpm_scale[DSIM_MAX_ACT];
  double  pwm2rpm_const[DSIM_MAX_ACT];
  double  pwm_min[DSIM_MAX_ACT], pwm_max[DSIM_MAX_ACT];
  double  rotor_pos[DSIM_MAX_ACT][3]; /* point of force application, body frame (link inertial origin) */
  double  rotor_axis[DSIM_MAX_ACT][3];/* thrust direction, body frame ((0,0,1) for quads)   */
  double  rotor_spin[DSIM_MAX_ACT];   /* sign of km*rpm^2 in the yaw torque (BaseAviary.py:1527: -,+,-,+) */
  double  G1[DSIM_MAX_ACT][DSIM_MAX_ACT];    /* control effectiveness, [n_out][n_act]       */
  double  alloc[DSIM_MAX_ACT][DSIM_MAX_ACT]; /* quad: pinv(G1/0.05) [n_act][n_out] (INDIControl.py:459);
                                                hexa: M1 of the WLS first iteration, u_opt = M1 v + M4 u0 */
  double  alloc2[DSIM_MAX_ACT][DSIM_MAX_ACT];/* hexa: M4 (see DESIGN.md "WLS allocation"); quad: unused  */
  double  kp_pos, kd_pos;             /* indi_guidance_gains                                */
  double  att_gain[3], rate_gain[3];  /* indi_att_gains att / rate                          */
  double  gravity;                    /* 9.8 (BaseAviary.py:182,673)                        */
  double  lin_damping, ang_damping;   /* Bullet multibody defaults 0.04f                    */
  double  max_coord_vel;              /* Bullet multibody default 100                       */
  double  drag_coeff[3];              /* BaseAviary._drag coefficients (formula P6)         */
  double  gnd_eff_coeff, prop_radius, gnd_eff_h_clip;   /* formula P7                       */
  double  dw_coeff[3];                /* formula P8                                         */
  double  max_speed_kmh;              /* URDF max_speed_kmh (VelocityAviary speed limit)    */
  /* bounding cylinder of the vehicle's collision shapes about its body z axis (URDF <collision>; robobee: the
   * 0.15 m x 0.1 m cylinder, robobee.urdf:72-77): radius, and extent below the centre of mass.  The reference loads
   * plane.urdf with collisions on (BaseAviary.py:680).  DSIM_OPT_PLANE enforces that plane on this cylinder (general
   * kernels); without the option the flight kernels do not model it, and every drone-step that ends with this
   * cylinder reaching z <= 0 is counted instead (DSIM_Q_GROUND_CONTACTS), so that a caller knows when a flight has
   * left the domain in which results are comparable.  0 = no watch and no contact for this type.  */
  double  collision_radius, collision_below;
  double  contact_friction;           /* DSIM_OPT_PLANE: Coulomb coefficient against the plane (PyBullet combines by product:
                                         plane.urdf's lateral_friction 1.0 x the vehicle's default 0.5)                 */
  /* Body-frame vector from the centre of mass the physics integrates to the point whose position and velocity the state
   * block holds — what p.getBasePositionAndOrientation / getBaseVelocity report (BaseAviary.py:726-732): the BASE
   * link's centre of mass.  Zero for the single-body quads; the morphing hexa is flown as the rigid composite of its 19
   * links, whose centre of mass lies 11 mm below the base link's (hexa_6DOF.urdf).  mass, inertia, rotor_pos and the
   * collision cylinder are about the integrated centre of mass. */
  double  base_offset[3];
  /* Physics.DYN (BaseAviary._dynamics, BaseAviary.py:1767-1828: the reference's own explicit rigid-body model, four-rotor
   * types only): the URDF's `arm` attribute L (BaseAviary.py:2058; robobee / tello 0.0635 — NOT the rotor links' lever arms
   * the PYB force map uses) and which of the two mixers of :1794-1803 turns the rotor forces into roll / pitch torques. */
  double  arm;
  int32_t dyn_mixer;                  /* DSIM_DYN_MIXER_X: (f0+f1-f2-f3, -f0+f1+f2-f3) L/sqrt 2 (DroneModel.CF2X, :1794-1800);
                                         DSIM_DYN_MIXER_PLUS: (f1-f3, -f0+f2) L (CF2P / HB, :1801-1803)                    */
  int32_t _pad_dyn;
} dsim_type_params;
enum { DSIM_DYN_MIXER_X = 0, DSIM_DYN_MIXER_PLUS = 1 };

/* ---- step options ---------------------------------------------------------- */
enum {
  DSIM_OPT_DRAG        = 1u << 0,   /* add formula P6 (BaseAviary.py:1705-1732)            */
  DSIM_OPT_GROUND      = 1u << 1,   /* add formula P7 (BaseAviary.py:1648-1699)            */
  DSIM_OPT_BCAST_TGT   = 1u << 2,   /* targets view holds ONE drone's targets, broadcast   */
  DSIM_OPT_CHAINED     = 1u << 3,   /* dsim_step only.  The caller asserts that the stored controller memory is
                                       the one the previous dsim_step / dsim_control left on the SAME rigid state,
                                       i.e. last_vel == vel and last_rates == R(quat)^T ang_vel.  Both are then
                                       recomputed from the rigid state instead of being read, and are NOT written
                                       back (6 of the 24 fields: 184 instead of 232 bytes of traffic per
                                       drone-step).  The six fields are stale until dsim_materialize is called. */
  /* -- tuning knobs (results do not depend on them; the library reads no environment variables) --------------- */
  DSIM_OPT_STREAM_ON   = 1u << 4,   /* force nontemporal (streaming) loads/stores of the state; default: on when one
                                       step's traffic exceeds what the 256 MB Infinity Cache keeps between steps   */
  DSIM_OPT_STREAM_OFF  = 1u << 5,   /* force the default cache policy                                              */
  /* (bits 6-9, 12, 13: A/B knobs of measured-and-rejected kernel forms; honoured only by a library built with
   * tools/variants/ in the git history up to round 5 — the library ignores them)                              */
  /* -- physics (changes results) ------------------------------------------------------------------------------------ */
  DSIM_OPT_PLANE       = 1u << 10,  /* ground plane z = 0 with contact and friction, as the reference's world has
                                       (BaseAviary.py:680 loads plane.urdf, collisions on).  A PRODUCT-DEFINED contact
                                       model (eight body-fixed rim points of the vehicle's collision cylinder, 24
                                       sequential-impulse sweeps, ERP 0.2, restitution 0, Coulomb friction): Bullet's own
                                       contact pipeline cannot be restated or pinned here.  Served
                                       by the k_step_plane / k_physics_plane / k_adaptor kernels; every airframe kind;
                                       not combined with DSIM_OPT_CHAINED.                                            */
  /* -- numbering of the per-drone arrays beside the state (dsim_physics, dsim_control2) -------------------------------------
   * A host class may STORE a heterogeneous fleet type-major (runs) behind its caller's numbering (dsim_step_args.drone_id).
   * With this bit the per-drone arrays those two entry points take and return besides the state and target views — the
   * action, obs_out, cmd_out, pos_e_out, yaw_e_out — are indexed by the CALLER's drone number drone_id[i] instead of the
   * storage slot i: Env.step() returns its observation rows and computeControl() its triple in the caller's order with no
   * second pass over them (BaseAviary.py:547-555, INDIControl.py:227), and the command goes from the one to the other as
   * it is.  last_action_out stays in storage order (it is the env's own memory, read back by dsim_observe).  Needs
   * drone_id and a fleet that the run kernels serve (runs given or one type; no noise replay, no drag / ground / plane
   * option): DSIM_E_UNSUPPORTED otherwise.  The first call with a NEW set of runs builds a small table the ctx keeps: it
   * waits for the device and may allocate, so it must not sit inside a stream capture (every later call with the same
   * runs may).                                                                                                        */
  DSIM_OPT_CALLER_IO   = 1u << 14,
  /* dsim_physics / dsim_step_adaptor: `action` is row-major [n][4] — one 4-vector per drone, as Env.step is handed it
   * (CtrlAviary.py:258-263, VelocityAviary.py:221-264, RPYTAviary.py:181-193) — instead of field-major [4][n_pad]; 16-byte
   * aligned.  Served by the one-launch forms (homogeneous quad fleet in whole tiles, no physics option, no downwash inputs):
   * DSIM_E_UNSUPPORTED otherwise.                                                                                        */
  DSIM_OPT_ACTION_ROWS = 1u << 15,
  /* -- Physics.DYN (changes results) -----------------------------------------------------------------------------------
   * dsim_physics / dsim_step integrate with the reference's OWN explicit model instead of the restated Bullet step:
   * BaseAviary._dynamics (BaseAviary.py:1767-1828; dispatch :525-527, no p.stepSimulation :541-543), per physics sub-step
   *     rpm = pwm2rpm_scale * clipped action + pwm2rpm_const   (the fork's PWM -> RPM map, BaseAviary.py:1487-1490; the
   *                                                              function's argument is documented as RPMs, :1770-1775)
   *     rpy = getEulerFromQuaternion(quat)                      (:513-520 / :547 refresh self.rpy from the engine, :729)
   *     thrust = sum kf rpm^2 along body z (R from quat, :1786-1790);  force_world = R thrust - (0, 0, G M)
   *     torques = mixer(forces; L) , z = -t0 + t1 - t2 + t3 (t = km rpm^2);  torques -= rpy_rates x (J rpy_rates)
   *     vel += dt force_world / M;  rpy_rates += dt J^-1 torques;  pos += dt vel;  rpy += dt rpy_rates
   *     quat = getQuaternionFromEuler(rpy)                      (:1814-1819)
   * There is no rotor noise in this model (noise_seed / noise_replay are ignored) and no damping.  Four-rotor types only
   * (both mixers read forces[0..3]): DSIM_E_UNSUPPORTED for a table with a six-actuator type, and with the drag / ground /
   * plane options, ext_force, waypoint tables, n_steps > 1, DSIM_OPT_CHAINED, _CALLER_IO, _ACTION_ROWS, bin_next.
   * The model's own state `rpy_rates` (BaseAviary.py:670-671, 1828: an env attribute beside pos / quat / vel) lives in
   * dsim_step_args.dyn_rpy_rates, REQUIRED with this bit.  The angular-velocity fields of the state block hold what
   * p.getBaseVelocity reports after the step: the reference stores the placeholder (-1, -1, -1) there ("ang_vel not computed
   * by DYN", :1821-1826), so that is what observations show and what a controller reads — bit for bit the reference's
   * behaviour, and the reason its INDI controller cannot close a loop on Physics.DYN.                                      */
  DSIM_OPT_DYN         = 1u << 16,
  /* With DSIM_OPT_DYN: a PRODUCT-DEFINED deviation that makes the mode flyable.  The angular-velocity fields receive
   * R(quat_new) rpy_rates_new instead of the placeholder — the world-frame image of the rates, which the model's own torque
   * equation treats as body rates (rpy_rates x J rpy_rates, :1805).  Off = the reference's (-1, -1, -1).                  */
  DSIM_OPT_DYN_BODY_RATES = 1u << 17,
  /* -- rotor noise (changes results) -------------------------------------------------------------------------------------
   * WHICH of the two lattices the rotor-noise normals are drawn on (see dsim_step_args.noise_seed).  A function of the launch
   * arguments alone, never of the kernel that happens to serve the launch:
   *   neither bit   phys_substeps == 1: the FINE lattice; phys_substeps > 1: the COARSE one
   *   _NOISE_FINE   the fine lattice at any sub-step count: 16 + 16 bits per Box-Muller pair, one Threefry block per quad sub-step
   *                 (two per hexa sub-step), radius and direction evaluated.  Free on launches of one sub-step (bound by HBM);
   *                 a launch of several sub-steps on it leaves the looped fast kernels for the general ones
   *   _NOISE_COARSE the coarse lattice at any count: 8 + 8 bits per pair, one block per TWO quad sub-steps (one per hexa
   *                 sub-step), radius and direction from 256-entry tables in the kernels that loop over sub-steps (bound by
   *                 vector issue).  A caller that splits one Env.step of several sub-steps into single-sub-step launches
   *                 (the per-sub-step neighbour downwash) passes the bit of the lattice its unsplit step would draw
   * Both bits: DSIM_E_ARG.                                                                                                */
  DSIM_OPT_NOISE_FINE  = 1u << 18,
  DSIM_OPT_NOISE_COARSE = 1u << 19,
  /* -- scheduling (results do not depend on it) ---------------------------------------------------------------------- */
  DSIM_OPT_DEFER_FALLBACK = 1u << 11 /* dsim_step / dsim_control2 of a table with a morphing hexa do NOT launch the deferred
                                       WLS fallback pass behind the step; the caller launches dsim_wls_fallback itself —
                                       on any stream ordered behind this call — before the commands are next read
                                       (INDIControl_6DOF.py:600-631 finishes the allocation inside computeControl; here its
                                       rare active-set tail may overlap the next Env.step's neighbour query).            */
};

/* A run of consecutive drones of one type (type-major storage of a heterogeneous fleet). */
typedef struct dsim_type_run {
  int64_t first;            /* first drone of the run (any index: the launch starts at the 256-drone tile that
                               holds it and the lanes in front of `first` retire)                        */
  int64_t count;            /* drones in the run                                                        */
  int32_t type;             /* index into the ctx type table                                            */
  int32_t _pad;
} dsim_type_run;

typedef struct dsim_step_args {
  int32_t  phys_substeps;   /* AGGR_PHY_STEPS: Bullet sub-steps per Env.step (BaseAviary.py:510) */
  float    dt_phys;         /* 1/SIM_FREQ  (BaseAviary.py:675)                                   */
  float    dt_ctrl;         /* control_timestep handed to computeControl (fly_INDI.py:231)       */
  uint32_t options;         /* DSIM_OPT_*                                                        */
  uint64_t noise_seed;      /* 0 = rotor noise off; else counter-based noise, N(0, .01) on the rotor forces and N(0, .001) on
                               the rotor moments per sub-step (BaseAviary.py:1518-1525, 1429-1432).  The reference draws
                               np.random.normal from the unseeded global generator; here the stream is PRODUCT-DEFINED:
                               Threefry4x32-12 keyed by the seed, counter = (drone, block), Box-Muller pairs on a LATTICE of
                               cell centres (u = (k + 1/2) / K: no draw is exactly 0) —
                                 fine      16 + 16 bits per pair: 2^32 distinct pairs, |n| <= 4.855 sigma (mass 1.2e-6 beyond),
                                           kurtosis 2.9998, Kolmogorov distance from the normal distribution below what 1e7 draws
                                           resolve.  The lattice of every launch of ONE sub-step (BASELINE's metric) and of
                                           DSIM_OPT_NOISE_FINE;
                                 coarse    8 + 8 bits per pair: 256 radii x 256 directions = 65 536 distinct pairs, |n| <= 3.535
                                           sigma (the reference's tails are unbounded: mass 4.1e-4 beyond 3.535 sigma; 2.75e-3 instead of 2.70e-3 beyond 3), variance
                                           exactly sigma^2 (the radius is rescaled), kurtosis 2.977 instead of 3, no atoms,
                                           Kolmogorov distance 1.4e-3 from the normal distribution.  The lattice of launches of
                                           SEVERAL sub-steps (the examples' five: kernels bound by vector issue) and of
                                           DSIM_OPT_NOISE_COARSE.
                               WHAT is drawn: a quad's eight normals per sub-step, one per rotor force and rotor moment as the
                               reference draws them (BaseAviary.py:1518-1521).  The six-actuator kinds draw SIX: the reference's twelve
                               per-rotor normals (:1429-1430) enter the rigid composite only through the body wrench they add up
                               to, a Gaussian 6-vector; the stream draws that vector (W = L z, L the Cholesky factor of its
                               covariance, computed from the type's rotor geometry by dsim_create) — the same distribution of the
                               wrench, hence of the flight, from half the bits (tests/test_noise_distribution.py:
                               test_hexa_wrench_noise_has_the_covariance_of_the_per_rotor_noise).  Per-rotor normals remain an
                               input: noise_replay.
                               tests/test_noise_distribution.py measures both lattices against N(0, 1) (Kolmogorov distance, moments,
                               tail mass, the absence of exact zeros) over 1e7 draws; dsim_noise_draw hands out the normals.   */
  uint64_t step_index;      /* env-step counter, mixed into the noise counter                    */
  const float* noise_replay;/* nullable; [phys_substeps][2*n_act][n_pad] recorded normals (tests)*/
  const uint8_t* type_id;   /* nullable; per-drone index into the ctx type table (mixed fleets)  */
  const float* action;      /* nullable; SoA [n_act][n_pad].  dsim_step: action for the physics part
                               instead of the stored cmd (first iteration of the example loop: the
                               initial action 0.4, fly_INDI.py:214, while INDIControl.cmd starts at 0).
                               dsim_physics: the action of Env.step(action); NULL = stored cmd.      */
  /* -- waypoint-table targets (examples/fly_INDI_TrajectoryTrack.py:178-189, 242-256) ------------
   * wp_table non-NULL: the targets view is ignored; drone i tracks row wp_counter[i] of the table
   * (row-major [n_wp][10] = pos3 vel3 acc3 yaw, device memory, read-only; 1200 rows = 48 KB for the
   * example) plus its own position offset, and after every control evaluation the counter advances
   * by one and wraps to 0 after n_wp-1, as the example does.                                        */
  const float* wp_table;
  int32_t*     wp_counter;  /* [n_pad] device, in-out                                              */
  const float* wp_offset;   /* nullable; SoA [3][n_pad] added to the table's target position       */
  int32_t      n_wp;
  /* -- multi-step launches -------------------------------------------------------------------------
   * n_steps > 1: dsim_step runs that many consecutive [Env.step + computeControl] iterations in ONE
   * launch with the state held in registers (step_index, step_index+1, ...); targets stay fixed
   * unless they come from the waypoint table.  0 is treated as 1.                                   */
  int32_t      n_steps;
  /* -- external body-frame force --------------------------------------------------------------------
   * nullable; SoA [3][n_pad]: extra force applied at the COM in the LINK frame every sub-step, e.g.
   * the neighbour downwash of dsim_downwash (BaseAviary.py:1755-1762 applies it to link 4 = COM).   */
  const float* ext_force;
  /* -- graph replay -----------------------------------------------------------------------------------
   * nullable device pointer; the effective env-step counter is step_index + *step_index_dev.  Lets a
   * captured hipGraph of [dsim_step, dsim_counter_add] pairs be replayed without repeating the noise
   * stream (kernel arguments are frozen at capture time, device memory is not).                        */
  const uint64_t* step_index_dev;
  /* -- type-major storage ------------------------------------------------------------------------------
   * nullable HOST array: the fleet is stored as n_runs runs of one type each (type_id[], still required by
   * the other entry points, must agree).  dsim_step then launches the single-type kernel of each run's kind
   * over that run — the mixed-fleet kernel, which has to hold every law and re-sort each tile by type, is
   * not needed.  Drones outside every run are not stepped.                                              */
  const dsim_type_run* runs;
  int32_t n_runs;
  /* -- fused observation (dsim_physics only) -----------------------------------------------------------------------
   * obs_out nullable: the rows of dsim_observe (BaseAviary._getDroneStateVector, BaseAviary.py:764-790) of the state
   * AFTER the step with the applied (clipped) action echoed, written by the same launch: what Env.step() returns
   * (BaseAviary.py:547-555).  Row-major [n][obs_width], obs_width = 16 + n_act of the ctx.                           */
  int32_t obs_width;
  float*  obs_out;
  /* -- neighbour grid for the NEXT Env.step (dsim_step only) --------------------------------------------------------
   * nullable: the grid description of the next dsim_downwash call.  The step kernel then appends the NEW position of
   * every local drone to its cell's bucket while the position is still in registers (or, when that call will re-use kept
   * candidate lists — its keep field says DSIM_DW_KEEP_REUSE —, refreshes the drone's bucket entry in place), and that
   * dsim_downwash call (same workspace and shape, args->prebinned = 1) skips the binning launch for the local drones.  Honoured on the
   * bucket form of the grid only (dsim_downwash_prebin_ok() != 0); otherwise ignored.                                */
  const struct dsim_downwash_args* bin_next;
  /* -- storage order --------------------------------------------------------------------------------------------------
   * nullable device array [n_pad]: the index of storage slot i in the CALLER's numbering.  Drones are independent, so a
   * host class may store a heterogeneous fleet type-major (runs) whatever order its caller uses; the rotor-noise
   * counter is then keyed by drone_id[i] instead of i, so that a drone draws the same noise wherever it is stored.   */
  const int32_t* drone_id;
  /* -- Physics.DYN ------------------------------------------------------------------------------------------------------
   * device SoA [3][n_pad], in-out; required with DSIM_OPT_DYN, ignored otherwise: BaseAviary.rpy_rates (BaseAviary.py:670-671
   * zeroed by _housekeeping, :1785 read, :1828 written by _dynamics).  Caller-owned like every other per-drone array.  */
  float* dyn_rpy_rates;
} dsim_step_args;

typedef struct dsim_ctx dsim_ctx;

/* library / build info */
int         dsim_abi_version(void);
const char* dsim_strerror(int code);

/* ctx: uploads the type table to `device`.  Replaces the per-instance parsing in
 * BaseAviary.__init__ (BaseAviary.py:200-235) + INDIControl.__init__ (:34-52).   */
int dsim_create(dsim_ctx** out, int device, const dsim_type_params* types, int n_types);
int dsim_destroy(dsim_ctx* ctx);

/* Plain device allocations straight from the driver (hipMalloc / hipFree on the ctx's device), outside any caching
 * allocator of the host framework.  (No counterpart in the reference.)  For host classes that choose WHERE a fleet-sized
 * array lies by trial (dronesim_amd/placement.py): candidates that are not kept go back to the driver at once, and no
 * framework-wide cache has to be emptied to walk through device memory.  The caller owns what it allocates; dsim_dev_free
 * takes ctx == NULL too (a block may outlive the ctx it was allocated through: dsim_destroy frees nothing of the caller's). */
int dsim_dev_alloc(dsim_ctx* ctx, int64_t bytes, void** out);
int dsim_dev_free(dsim_ctx* ctx, void* ptr);

/* reset(): BaseAviary._housekeeping (BaseAviary.py:640-714) + INDIControl.reset
 * (INDIControl.py:109-146).  init_* are device SoA [3][n_pad]; init_vel and
 * init_cmd nullable (zeros / controller reset value).  Writes every field.       */
int dsim_reset(dsim_ctx* ctx, void* stream, int64_t n, dsim_view state,
               const float* init_pos, const float* init_rpy, const float* init_vel,
               const float* init_cmd, const uint8_t* type_id);

/* One fused Env.step() + computeControl() for every drone:
 *   physics x phys_substeps with the stored cmd  (BaseAviary.py:510-545, 1477-1543 | 1389-1457,
 *                                                  p.stepSimulation :542)
 *   then the INDI law on the fresh state          (INDIControl.py:154-227)
 * i.e. the body of the example loop (examples/fly_INDI.py:223-239).             */
int dsim_step(dsim_ctx* ctx, void* stream, int64_t n, dsim_view state, dsim_view targets,
              const dsim_step_args* args);

/* Pre-allocates what the ctx owns for fleets of up to n_pad drones (today: the deferred-WLS-fallback queue of
 * tables that hold a morphing hexa; a no-op otherwise), so that no later call allocates or synchronises — which
 * hipGraph capture of dsim_step requires. */
int dsim_reserve(dsim_ctx* ctx, void* stream, int64_t n_pad);

/* *counter += inc on the stream (a one-thread kernel; the companion of step_index_dev). */
int dsim_counter_add(dsim_ctx* ctx, void* stream, uint64_t* counter, uint64_t inc);

/* Rewrites last_vel / last_rates from the rigid state (ends a DSIM_OPT_CHAINED sequence). */
int dsim_materialize(dsim_ctx* ctx, void* stream, int64_t n, dsim_view state);

/* Env.step() only: physics sub-steps with args->action (clipped in-kernel as
 * CtrlAviary._preprocessAction does, CtrlAviary.py:258-263; NULL = stored cmd).
 * The clipped action is written to last_action_out (SoA [n_act][n_pad], nullable):
 * the env's last_clipped_action (BaseAviary.py:545).  The controller memory
 * (state cmd fields) is NOT touched: env and controller are separate objects in
 * the reference.                                                              */
int dsim_physics(dsim_ctx* ctx, void* stream, int64_t n, dsim_view state,
                 float* last_action_out, const dsim_step_args* args);

/* Env.step() of the reference's two alternate action adaptors, which run (part of) the INDI law
 * INSIDE _preprocessAction and then the physics:  control on the current state first, then
 * phys_substeps with the new command.  action: SoA [4][n_pad].
 *   DSIM_ADAPT_VELOCITY  VelocityAviary._preprocessAction (VelocityAviary.py:221-264): action =
 *       (vx, vy, vz, speed fraction); full INDI with target_pos = current position, target yaw =
 *       current yaw, target_vel = SPEED_LIMIT |a3| unit(a0..2)
 *   DSIM_ADAPT_RPYT      RPYTAviary._preprocessAction (RPYTAviary.py:181-193): action = (p, q, r
 *       body-rate set-points, thrust) fed to _INDIRateControl only
 * dt_ctrl of args is the control_timestep (AGGR_PHY_STEPS * TIMESTEP).  Quad types only.
 * last_action_out (nullable, SoA [4][n_pad]) receives the applied command (last_clipped_action).
 * args->obs_out (nullable, [n][20], obs_width = 20): Env.step's return value, the rows of _computeObs for the NEW state
 * (BaseAviary.py:547-555, 780-790) — written by the same launch for a homogeneous fleet in whole tiles (which also takes the
 * action row-major, DSIM_OPT_ACTION_ROWS), by the observation kernel behind the step otherwise. */
enum { DSIM_ADAPT_VELOCITY = 0, DSIM_ADAPT_RPYT = 1 };
int dsim_step_adaptor(dsim_ctx* ctx, void* stream, int64_t n, dsim_view state, const float* action,
                      int32_t mode, float* last_action_out, const dsim_step_args* args);

/* computeControl() only (INDIControl.py:154-227 / INDIControl_6DOF.py:259-336):
 * reads the rigid fields, updates the controller-memory fields and cmd.
 * pos_e_out [3][n_pad] and yaw_e_out [n_pad] nullable.                           */
int dsim_control(dsim_ctx* ctx, void* stream, int64_t n, dsim_view state, dsim_view targets,
                 const dsim_step_args* args, float* pos_e_out, float* yaw_e_out);
/* The same, and the new command is also written to cmd_out (nullable, SoA [n_act][n_pad]): the first return value of
 * computeControl as a plain array, in the layout Env.step's action (dsim_step_args.action) takes — the command goes
 * from the controller to the next Env.step without a copy. */
int dsim_control2(dsim_ctx* ctx, void* stream, int64_t n, dsim_view state, dsim_view targets,
                  const dsim_step_args* args, float* pos_e_out, float* yaw_e_out, float* cmd_out);

/* _getDroneStateVector (BaseAviary.py:764-790): writes the reference's 20/22-wide
 * observation rows [pos3 quat4 rpy3 vel3 ang_v3 last_action] as row-major
 * [n][obs_width] fp32, obs_width = 16 + n_act.  last_action: SoA [n_act][n_pad]
 * (the env's last_clipped_action); NULL = the stored cmd.                        */
int dsim_observe(dsim_ctx* ctx, void* stream, int64_t n, dsim_view state, const float* last_action,
                 float* obs_out, int32_t obs_width);

/* Trajectory sampler on device: trajGenerator.get_des_state(t) (dronesim/utils/trajGen.py:108-126,
 * polyder trajutils.py:13-21) per drone, including the stateful yaw-from-velocity rule
 * (trajGen.py:128-143), written into a targets view that dsim_step / dsim_control then consume.
 * coeffs: device, row-major [n_seg*10][3] fp64 (trajGenerator.coeffs); ts: device [n_seg+1] fp64.
 * t: device [n_pad] fp64 in-out, each drone's own trajectory time; advanced by dt_advance after
 * sampling (the example samples at 1/control_freq).  yaw_state: device SoA [3][n_pad] fp64 in-out
 * (yaw, heading_x, heading_y), zero-initialised like trajGenerator; fp64 because the rule integrates
 * acos() of nearly parallel unit headings, whose error is sqrt(eps) per step.  offset: nullable fp32
 * SoA [3][n_pad] added to the sampled position.  Polynomials (degree 9, t up to ~7 s) in fp64. */
int dsim_traj_sample(dsim_ctx* ctx, void* stream, int64_t n, const double* coeffs, const double* ts,
                     int32_t n_seg, double* t, double dt_advance, double* yaw_state, const float* offset,
                     dsim_view targets_out);

/* The same rows as dsim_observe, written field-major: SoA [obs_width][n_pad] (coalesced), the slab a
 * device-side Logger appends per step (dronesim/utils/Logger.py:117-139 stores exactly this vector). */
int dsim_observe_soa(dsim_ctx* ctx, void* stream, int64_t n, dsim_view state, const float* last_action,
                     float* obs_out, int32_t obs_width);

/* Neighbour downwash, formula P8 (BaseAviary._downwash, BaseAviary.py:1736-1763; dead code in the
 * fork, intended semantics): for every local drone i and every drone j of the WORLD above it
 * (dz > 0) within dxy < 10 m,  Fz -= DW1 (PROP_RADIUS/(4 dz))^2 exp(-0.5 (dxy / (DW2 dz + DW3))^2)
 * with the receiving drone's coefficients, along its body z axis.
 * pos_all: SoA [3][m_pad] positions of ALL drones of the world (on several GPUs: the all-gather of
 * every rank's positions); the library bins them into a uniform xy grid (cell >= 10 m; drones outside
 * the box are clamped to the border cells, which keeps every pair within 10 m in adjacent cells) and
 * each local drone scans its 3x3 cells: O(n * neighbours) instead of the reference's O(n m).  Receivers
 * are processed in grid order (a wave's lanes sit in the same or neighbouring cells, so their scans
 * read the same sorted entries).
 * force_out: SoA [3][n_pad], x and y written as 0 (feed it to dsim_step_args.ext_force).
 * workspace: caller-owned device int32 buffer of at least dsim_downwash_workspace(m, nx, ny) entries. */
struct dsim_halo_plan;
typedef struct dsim_downwash_args {
  const float* pos_all;     /* SoA [3][m_pad] positions of every drone of the world, or NULL: the world is this
                               fleet alone (m = n, local_offset = 0) and positions are read from the state block */
  int64_t  m, m_pad;
  float    xmin, ymin, cell;
  int32_t  nx, ny;
  int32_t* workspace;
  int64_t  workspace_len;
  const uint8_t* type_id;   /* nullable; types of the LOCAL drones */
  int64_t  local_offset;    /* index inside pos_all of local drone 0 (the rank's shard begin); the local
                               drones' entries of pos_all must equal their state positions               */
  int32_t  prebinned;       /* 1: the previous dsim_step was given this grid as bin_next and has already binned the
                               local drones (see dsim_step_args.bin_next); only the other entries of pos_all are
                               binned by this call.  The library falls back to a full binning pass when its own
                               record of the last prebinning does not match (another grid, or the positions were
                               moved since by a call that did not re-bin them).                              */
  int32_t  phase;           /* DSIM_DW_*: with a halo plan the call can be split so that the exchange overlaps the local
                               part of the query (see dsim_halo_plan)                                        */
  const struct dsim_halo_plan* halo;   /* nullable: the rest of the world is what the neighbouring ranks sent (pos_all must
                               be NULL, local_offset 0, m = n + the sum of the plan's recv_cap); bucket form only */
  uint64_t* pairs_evaluated; /* nullable device counter, diagnostics only (results do not depend on it): += the number of
                               (receiver, candidate) pairs whose term the query's loops evaluate in this call — the unit of
                               the query's vector-pipe roofline (bench.py, config 5).  Bucket form only; one atomic per
                               receiver group when given, one scalar test when not                                        */
  /* -- kept candidate lists (no counterpart in the reference, whose loop is O(N^2) per drone) ---------------------------------
   * A fleet moves centimetres per Env.step, so WHICH candidates the receivers of a cell have to look at changes slowly.
   * DSIM_DW_KEEP_BUILD: the query also writes, per cell, the list it worked out (receivers in height order, candidates in
   * height-band order), with its reach and band tests widened by 2 keep_skin.  DSIM_DW_KEEP_REUSE: the query reads the lists of
   * the last BUILD call on this grid and the CURRENT positions and goes straight to the pair loops; nothing is binned — a
   * dsim_step that is given this block as bin_next refreshes every drone's position in the grid's buckets IN PLACE instead (no
   * atomic round trip for a bucket slot); a call that finds no such step in front of it refreshes them itself.  EXACT for any motion: every pair is still tested against the cut-off and the height order on current positions;
   * a drone that has moved further than keep_skin from where it was at the BUILD leaves the lists for the overflow list, which
   * every receiver scans, and is served where it is now.  How often to BUILD is the caller's choice (performance only: the
   * more drones have left the skin, the longer the overflow scan).  Honoured where dsim_downwash_keep_ok() != 0 (the world is
   * this fleet alone: pos_all = NULL, no halo plan; bucket form at a density that takes the banded query; cells of
   * 5 m + keep_skin or more — two rings of cells cover the widened reach — and below 10 m), otherwise ignored; a REUSE without usable
   * lists (none made yet, another grid or fleet size) is answered as a BUILD.  keep_ws: caller-owned,
   * dsim_downwash_keep_workspace(n_pad, nx, ny) int32 entries, owned by the library between a BUILD and the last REUSE. */
  int32_t  keep;            /* DSIM_DW_KEEP_* */
  int32_t  keep_age;        /* REUSE: how many REUSE calls the lists have served, this one included (1, 2, ...; the dsim_step that is
                               given this block as bin_next refreshes for that call).  The skin moves with the fleet — displacements are
                               measured against the mean drift of a sample of it, predicted from the two refreshes before —, and the
                               age says which of the ring of sums a refresh fills.  Performance only: a wrong age makes the prediction
                               worse, never the result (the query reads the drift its refresh used from device memory).             */
  float    keep_skin;       /* metres, > 0 */
  int32_t* keep_ws;
  int64_t  keep_ws_len;
} dsim_downwash_args;
enum { DSIM_DW_KEEP_OFF = 0, DSIM_DW_KEEP_BUILD = 1, DSIM_DW_KEEP_REUSE = 2 };
int64_t dsim_downwash_keep_workspace(int64_t n_pad, int32_t nx, int32_t ny);
/* How the lists are holding up, WITHOUT synchronising anything: *queries = the REUSE calls enqueued so far; *outside_skin = the
 * length of the overflow list that REUSE call number *of_query (1-based; 0: none has finished) found — the drones that had left the
 * skin, which every later call until the next BUILD will find too, and more; *half_way = the drones the refresh in front of that
 * call found further than HALF the skin from where they were when the lists were made (the warning: a fleet in coordinated motion
 * leaves any skin together, and a REUSE call whose overflow list holds the fleet costs a hundred plain ones).  The device writes
 * the three into host memory when that query starts; a caller that paces its BUILDs by them (dronesim_amd/downwash.py) reads values
 * a query or two old, which is what such pacing needs.  -1: no feedback memory. */
int     dsim_downwash_keep_stats(dsim_ctx* ctx, int64_t* outside_skin, int64_t* half_way, int64_t* of_query, int64_t* queries);
int     dsim_downwash_keep_ok(int64_t m, int32_t nx, int32_t ny, float cell, float keep_skin);
int64_t dsim_downwash_workspace(int64_t m, int32_t nx, int32_t ny);

/* ---- halo exchange of a spatially sharded fleet (BASELINE config 5: slabs along x, one rank per GPU) -----------------
 * The reference's downwash loop runs over the whole world (BaseAviary.py:1736-1763); a sharded fleet needs, per rank,
 * the positions of the other ranks' drones that can be within the 10 m cut-off of one of its own.  The library does the
 * device side of that exchange — selecting and packing the boundary drones, binning what arrived — and the caller moves
 * the packed buffers between ranks (RCCL send/recv; the library itself has no communicator).  Per Env.step, with no
 * host synchronisation anywhere:
 *     dsim_halo_pack -> [send/recv of the caller, on the collective library's stream] ............ wait for it
 *                    -> dsim_downwash(DSIM_DW_LOCAL) (beside the wire) -> dsim_downwash(DSIM_DW_HALO_BIN)
 *                    -> dsim_downwash(DSIM_DW_HALO_QUERY) -> dsim_step(bin_next)
 * or, with a transport that has no latency to hide, pack -> wire -> dsim_downwash(DSIM_DW_ALL) -> dsim_step.
 * A message is a HEADER (DSIM_HALO_HDR floats: the number of positions that follow, and the sender's xy bounding box at
 * the time of sending) followed by xyz triples.  dsim_halo_pack selects, for every peer, the local drones inside that
 * peer's box — read from the header of the LAST message that peer sent, in device memory — grown by reach[p]; with a
 * message every step the box is one Env.step old, so reach = cut-off + (max coordinate velocity) x dt_env is exact for ANY
 * motion the integrator allows (Bullet clamps every coordinate velocity to 100 m/s, P4).  Message sizes are fixed on
 * the host (send_cap / recv_cap, re-made rarely from counts read back); the count travels in the header, and a
 * selection that does not fit is dropped AND counted (DSIM_Q_HALO_OVERFLOW: 0 certifies that nothing was missed).      */
#define DSIM_MAX_PEERS 8
#define DSIM_HALO_HDR 8      /* floats: [0] count (int32 bits)  [1..4] xmin ymin xmax ymax of the sender  [5] max |coordinate
                                velocity| of the sender (diagnostic)  [6..7] reserved                                    */
enum {
  DSIM_DW_ALL = 0,          /* one grid, one query: local drones and whatever pos_all / the halo plan hold               */
  DSIM_DW_LOCAL = 1,        /* bin (unless pre-binned) and query the LOCAL drones only; force_out is written              */
  DSIM_DW_HALO_BIN = 2,     /* bin the plan's received positions into the halo grid (touches nothing else)                 */
  DSIM_DW_HALO_QUERY = 3    /* local receivers against the halo grid; force_out += (after LOCAL and HALO_BIN)             */
};
typedef struct dsim_halo_plan {
  int32_t world, rank;
  int64_t cap;              /* positions a peer's buffer can hold; buffer p begins at float (DSIM_HALO_HDR + 3 cap) p      */
  float*  send;             /* device [world][DSIM_HALO_HDR + 3 cap]: dsim_halo_pack writes header + triples for peer p    */
  const float* recv;        /* device, same shape: the last message of peer p (its header is read by the NEXT pack)       */
  int32_t* scratch;         /* device int32[32], zero-initialised by the caller once, owned by the plan's pack calls       */
  int32_t send_cap[DSIM_MAX_PEERS];   /* host: positions in the message to peer p (0: no message to p)                    */
  int32_t recv_cap[DSIM_MAX_PEERS];   /* host: positions in the message from peer p                                       */
  float   reach[DSIM_MAX_PEERS];      /* host: selection margin around peer p's last known box, >= the cut-off             */
} dsim_halo_plan;
/* out5 (device float[5]): xmin, ymin, xmax, ymax of the n local drones and their largest |coordinate velocity|. */
int dsim_fleet_bounds(dsim_ctx* ctx, void* stream, int64_t n, dsim_view state, float* out5);
/* One launch: for every peer p != rank, send[p] = header + the positions of the local drones inside p's last known box
 * grown by reach[p] (at most send_cap[p] of them; the header carries the number SELECTED, so the receiver sees an
 * overflow too), and this rank's own box in every header. */
int dsim_halo_pack(dsim_ctx* ctx, void* stream, int64_t n, dsim_view state, const dsim_halo_plan* plan);
/* workspace (int32 entries) of a split-phase downwash: the local grid of n drones + the halo grid of up to h drones */
int64_t dsim_downwash_workspace_halo(int64_t n, int64_t h, int32_t nx, int32_t ny);

/* The deferred WLS fallback pass of a table with a morphing hexa (wls_alloc.py:222-350 for the drones whose first
 * iteration left the box), as its own call: what dsim_step / dsim_control2 launch behind themselves unless
 * DSIM_OPT_DEFER_FALLBACK is set.  cmd_out nullable (SoA [n_act][n_pad], as dsim_control2). */
int dsim_wls_fallback(dsim_ctx* ctx, void* stream, int64_t n, dsim_view state, const uint8_t* type_id, float* cmd_out);
/* != 0 when a grid of this shape takes the bucket form (one binning pass, cell-centred LDS-tiled query), which is
 * the form dsim_step_args.bin_next can fill. */
int dsim_downwash_prebin_ok(int64_t m, int32_t nx, int32_t ny);
int dsim_downwash(dsim_ctx* ctx, void* stream, int64_t n, dsim_view state, const dsim_downwash_args* args,
                  float* force_out);
/* Forgets the ctx's bookkeeping of the neighbour grid (which of the two count buffers is current, whether a step has
 * filled one ahead): the next dsim_downwash clears both and bins everything itself.  Host-side only, nothing is launched.
 * For callers that replay a captured sequence of dsim_downwash / dsim_step calls (hipGraph): captured behind this call, the
 * sequence starts from no assumption about the buffers, and called again after a replay, so does whatever follows. */
int dsim_downwash_reset(dsim_ctx* ctx);

/* Neighbourhood adjacency at fleet scale (BaseAviary._getAdjacencyMatrix, BaseAviary.py:901-921: drones
 * i != j are neighbours when |pos_i - pos_j| < neighbourhood_radius).  The reference returns the dense
 * O(N^2) matrix row in every observation; here the same uniform grid as the downwash gives, per local
 * drone, the neighbour count and (optionally) up to max_k neighbour indices into pos_all in ascending
 * grid order (count_out [n_pad]; list_out [max_k][n_pad] nullable, unused slots -1).  args->cell must be
 * >= radius; pos_all / workspace / local_offset as for dsim_downwash (type_id is ignored). */
int dsim_adjacency(dsim_ctx* ctx, void* stream, int64_t n, dsim_view state, const dsim_downwash_args* args,
                   float radius, int32_t* count_out, int32_t* list_out, int32_t max_k);

/* The rotor-noise normals the step kernels draw (diagnostics / distribution studies; no counterpart in the reference, whose
 * draws come from numpy's global generator): for drones [0, n) and the physics sub-steps [0, substeps) of Env.step number
 * step_index, out [substeps][2 * n_act][n_pad] (device) receives the UNIT-variance normals — n_act = 4: rows 0 .. 3 the force noise,
 * 4 .. 7 the moment noise; n_act = 6: rows 0 .. 5 the six normals z of the body wrench (dsim_step_args.noise_seed), rows 6 .. 11 zero —
 * exactly as a launch with the same noise_seed / step_index / phys_substeps / options (DSIM_OPT_NOISE_FINE / _COARSE or neither)
 * draws them.  (dsim_step_args.noise_replay takes PER-ROTOR values, [substeps][2 * n_act][n_pad] times the
 * deviations.  n_act = 4 | 6.  drone_id nullable (the key of drone i's stream: dsim_step_args.drone_id).                 */
int dsim_noise_draw(dsim_ctx* ctx, void* stream, int64_t n, int64_t n_pad, int32_t n_act, uint64_t noise_seed, uint64_t step_index,
                    int32_t substeps, uint32_t options, const int32_t* drone_id, float* out);

/* Diagnostics counters kept by the ctx (device-side, cumulative; this call synchronises `stream`):
 *   DSIM_Q_WLS_FALLBACKS  drones x steps whose 6DOF allocation left the first-iteration fast path and
 *                         ran the full active-set loop of wls_alloc (wls_alloc.py:222-350)
 *   DSIM_Q_WLS_FAILURES   allocations on which the reference would have failed (wls_alloc returns None
 *                         -> `self.cmd += None` raises, INDIControl_6DOF.py:626-630); cmd is left unchanged
 *   DSIM_Q_GROUND_CONTACTS  drone x Env.steps that ended with the vehicle's collision cylinder at or below z = 0, where
 *                         PyBullet's ground plane would have acted (see dsim_type_params.collision_radius)
 *   DSIM_Q_HALO_OVERFLOW  positions dsim_halo_pack selected but could not ship (send_cap too small), or received
 *                         headers that announced more than recv_cap: the force of that step may have missed pairs
 *   DSIM_Q_DW_REUSES      dsim_downwash calls answered from kept candidate lists (dsim_downwash_args.keep)
 *   DSIM_Q_DW_MOVERS      overflow-list entries those calls found, summed: drones that had left the lists' skin (performance
 *                         only — results do not depend on it; / DSIM_Q_DW_REUSES = the mean length every receiver scanned)      */
enum { DSIM_Q_WLS_FALLBACKS = 0, DSIM_Q_WLS_FAILURES = 1, DSIM_Q_GROUND_CONTACTS = 2, DSIM_Q_HALO_OVERFLOW = 3,
       DSIM_Q_DW_REUSES = 4, DSIM_Q_DW_MOVERS = 5 };
int dsim_query(dsim_ctx* ctx, void* stream, int32_t what, int64_t* value_out);

/* error codes */
enum {
  DSIM_OK = 0,
  DSIM_E_ARG = -1,        /* null / inconsistent argument                       */
  DSIM_E_LAYOUT = -2,     /* view violates the layout contract                  */
  DSIM_E_NODEVICE = -3,   /* no HIP device / wrong arch                         */
  DSIM_E_TYPES = -4,      /* bad type table                                     */
  DSIM_E_UNSUPPORTED = -5
};

#ifdef __cplusplus
}
#endif
#endif /* DRONESIM_AMD_H */
