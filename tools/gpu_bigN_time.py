"""Time of one long candidate alone on the device: per-evaluation (eval hook, repeats) and whole solve."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import set_map, serpentine_path
from topay_amd import api
from harness import workload as wl
lib = os.environ.get("TOPAY_LIB")
w, lens, paths, scen = wl.cuboids_batch(3, 2)
for L in [float(a) for a in sys.argv[1:]] or [24.0, 40.0, 62.0]:
    p = serpentine_path(L)
    o = api.MomaTrajOptBatch(device=0, lib_path=lib)
    set_map(o, w)
    o.set_init_traj(np.array([len(p)], dtype=np.int32), p)
    N = o.n_pieces()[0]
    out = []
    for stage in (1, 2):
        o.eval_batch(stage, 3)
        o.eval_batch(stage, 50)
        ms, _ = o.last_kernel_ms()
        out.append(ms / 50 * 1e3)
    o.reset(); ok = o.optimize()
    ms, _ = o.last_kernel_ms()
    st = o.stats()[0]
    print(f"N {N}: eval stage1 {out[0]:.0f} us, stage2 {out[1]:.0f} us; solve {ms:.0f} ms, evals {st[2]}+{st[5]}, iters {st[1]}+{st[4]}, us/eval {ms*1e3/(st[2]+st[5]):.0f}", flush=True)
