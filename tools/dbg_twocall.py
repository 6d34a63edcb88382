"""Debug aid: does the reference-shaped loop's time depend on where its arrays land? (GPU)"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import bench
from dronesim_amd.control import INDIControl
from dronesim_amd.envs import CtrlAviary
xyz = bench.grid_fleet(4096, 1024)
keep = []
for trial in range(8):
    if trial:
        keep.append(torch.empty(((trial * 2 + 1) * 1024 * 1024 + 4096 * trial) // 4, dtype=torch.float32, device="cuda"))
    env = CtrlAviary(["robobee"], xyz.shape[0], initial_xyzs=xyz, aggregate_phy_steps=1, noise_seed=1, dict_io=False, layout="tile64")
    ctrl = INDIControl("robobee", env=env)
    tpos = torch.from_numpy(np.ascontiguousarray(xyz.T.astype(np.float32))).to(env.ctx.device)
    cmd = torch.full((xyz.shape[0], 4), 0.4, device=env.ctx.device)
    res = []
    for p in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(50):
            obs, _, _, _ = env.step(cmd)
            cmd, _, _ = ctrl.computeControlFromState(1 / 240, None, target_pos=tpos, target_rpy=np.array([0, 0, 0.4]))
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 50 * 1e3)
    ptrs = [env.state.data.data_ptr(), env._obs_buf.data_ptr() if getattr(env, "_obs_buf", None) is not None else 0, ctrl._cmd.data_ptr(), env._last_action.data_ptr()]
    print(f"trial {trial}: loop {res[1]:.1f} {res[2]:.1f} us   state/obs/cmd/echo at " + " ".join(hex(x) for x in ptrs))
    env.close(); del env, ctrl, tpos, cmd, obs
