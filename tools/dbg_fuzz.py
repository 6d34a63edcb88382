import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import test_gpu_fuzz as F
from tests.util import WHERE, WORST
L = F.Loop("quad", 1)
for j, op in enumerate(F.LONG):
    WHERE.clear(); WORST.clear()
    try:
        getattr(L, "op_" + op[0])(*op[1:]) if op[0] != "targets" else L.set_targets(op[1])
    except AssertionError as e:
        print(j, op, "FAIL", str(e)[:160])
        for k, v in WHERE.items():
            print("   ", k, v)
        print("   chain_live", L.env._chain_live, "chain_ok", L.env._chain_ok, "env_steps", L.env._env_steps, L.env_steps)
        break
    print(j, op, {k: round(v, 3) for k, v in WORST.items()})
