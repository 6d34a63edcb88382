"""Debug aid: the worst offenders of the plane-contact parity check, with their states (GPU)."""
import ctypes, sys
import numpy as np, torch
sys.path.insert(0, ".")
from dronesim_amd import params, _native as nat, fleet
from oracle import oracle as orc
from tests.util import K_ULP, f32, plane_terms, step_terms, increment_ratio, random_fleet
import tests.test_gpu_parity as T

DT = T.DT
model, sub = sys.argv[1], int(sys.argv[2])
t = params.builtin_type(model); na = t.n_act; n = 1800
ctx = fleet.Context([t]); st = fleet.FleetState(ctx, n, "soa", 256)
rigid, mem, tgt, kind = T._near_ground_fleet(t, n, 131 + sub, na)
st.load_aos(rigid, mem)
rng = np.random.default_rng(5)
act = f32(t.hover_pwm * rng.choice([0.0, 1.0, 1.6], (n, 1)) * np.ones((1, na)))
act_dev = torch.zeros((na, st.n_pad), device=ctx.device); act_dev[:, :n] = torch.from_numpy(act.T).float()
echo = torch.zeros((na, st.n_pad), device=ctx.device)
dtc = float(np.float32(sub / 240.0))
O = orc.Oracle([t])
a6 = np.zeros((n, 6)); a6[:, :na] = act
cur = rigid.copy()
for s_ in range(sub):
    a = T._args(nat, 1, DT, dtc, options=nat.OPT_PLANE, action=act_dev)
    nat.check(ctx.lib.dsim_physics(ctx.handle, ctx.stream_ptr(), n, st.view(), echo.data_ptr(), ctypes.byref(a)))
    got = st.rigid_aos()
    ref = cur.copy(); O.physics(ref, mem, 1, DT, action=a6, options=nat.OPT_PLANE)
    tr, _ = step_terms([t], None, cur, mem, tgt, DT, dtc, 1, False, act)
    ex = plane_terms([t], None, cur, dtc)
    rr = increment_ratio(got, ref, cur, tr + ex[0], K_ULP * 25)
    w = np.argsort(rr.max(1))[::-1][:4]
    print(f"sub-step {s_}: worst {rr.max():.3f}")
    for i in w:
        f = rr[i].argmax()
        r22 = 1 - 2 * (cur[i, 3] ** 2 + cur[i, 4] ** 2)
        print(f"  drone {i} kind {kind[i]} field {f} ratio {rr[i, f]:.2f} got {got[i, f]:.7g} ref {ref[i, f]:.7g} prev {cur[i, f]:.7g}"
              f" z {cur[i, 2]:.5f} tilt {np.arccos(min(1, abs(r22))):.5f} v {cur[i, 7:10]} w {cur[i, 10:13]}")
    cur = got

# the fused step from the device's state
tg = fleet.Targets(ctx, n, "soa", pad=256)
tg.set_fields(0, torch.from_numpy(np.ascontiguousarray(tgt.T)))
r0, m0 = st.rigid_aos(), st.mem_aos()
a2 = T._args(nat, sub, DT, dtc, options=nat.OPT_PLANE)
nat.check(ctx.lib.dsim_step(ctx.handle, ctx.stream_ptr(), n, st.view(), tg.view(), ctypes.byref(a2)))
r1, m1 = r0.copy(), m0.copy()
O.step(r1, m1, tgt, sub, DT, dtc, options=nat.OPT_PLANE)
from tests.util import tilt_gain
tr, tm = step_terms([t], None, r0, m0, tgt, DT, dtc, sub, True, None)
ex = plane_terms([t], None, r0, dtc)
k = K_ULP * sub * 25
gr, gm = st.rigid_aos(), st.mem_aos()
rr = increment_ratio(gr, r1, r0, tr + ex[0], k)
rm = increment_ratio(gm, m1, m0, tm + ex[1], k * tilt_gain([t], None, r1))
print(f"fused: worst rigid {rr.max():.3f} mem {rm.max():.3f}")
for i in np.argsort(rr.max(1))[::-1][:5]:
    f = rr[i].argmax()
    print(f"  rigid drone {i} kind {kind[i]} field {f} ratio {rr[i, f]:.2f} got {gr[i, f]:.7g} ref {r1[i, f]:.7g} prev {r0[i, f]:.7g} M {(tr + ex[0])[i, f]:.4g}"
          f" z {r0[i, 2]:.5f} q {r0[i, 3:7]} v {r0[i, 7:10]} w {r0[i, 10:13]} cmd {m0[i, 7:7 + na]}")
for i in np.argsort(rm.max(1))[::-1][:5]:
    f = rm[i].argmax()
    print(f"  mem drone {i} kind {kind[i]} field {f} ratio {rm[i, f]:.2f} got {gm[i, f]:.7g} ref {m1[i, f]:.7g} prev {m0[i, f]:.7g} M {(tm + ex[1])[i, f]:.4g}"
          f" z {r0[i, 2]:.5f} v {r0[i, 7:10]} w {r0[i, 10:13]}")
