"""Per-evaluation time of identical candidates as a function of how many are resident: 1 (alone on the device) ... 1024
(one per SIMD) ... 2048 (two rounds).  Separates the cost of the evaluation itself from what the loaded device adds
(memory-system contention, clocks under load).  usage: gpu_contention.py [path length in metres ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import set_map, serpentine_path
from topay_amd import api
from harness import workload as wl
w, lens, paths, scen = wl.cuboids_batch(3, 2)
for L in [float(a) for a in sys.argv[1:]] or [11.3, 62.0]:
    p = serpentine_path(L)
    for count in (1, 64, 256, 512, 768, 1024, 2048):
        o = api.MomaTrajOptBatch(device=0)
        set_map(o, w)
        o.set_init_traj(np.full(count, len(p), dtype=np.int32), np.concatenate([p] * count))
        N = o.n_pieces()[0]
        res = []
        for stage in (1, 2):
            o.eval_batch(stage, 3)
            o.eval_batch(stage, 40)
            ms, _ = o.last_kernel_ms()
            res.append(ms / 40 * 1e3 / max(1, (count + 1023) // 1024))
        print(f"N {N} x {count}: stage1 {res[0]:.0f} us, stage2 {res[1]:.0f} us per evaluation and round", flush=True)
        del o
