import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from topay_amd.harness import workload as wl
from topay_amd import api
w2, lens2, paths2, scen2 = wl.cuboids_batch(256, 8)
offs = np.concatenate([[0], np.cumsum(lens2)])
sel = [81, 427, 643]
lens = lens2[sel]; paths = np.concatenate([paths2[offs[i]:offs[i+1]] for i in sel])
tr = {}
for name, lib in (("w2", None), ("w3", "topay_amd/lib/libtopay_hip_w3.so"), ("emu", "tests/emu/libtopay_emu.so")):
    g = api.MomaTrajOptBatch(device=0, lib_path=lib)
    g.set_map(w2.origin, w2.res, w2.dims, w2.min_b, w2.max_b, w2.esdf2d, w2.esdf3d)
    g.set_init_traj(lens, paths)
    g.set_trace(400)
    ok = g.optimize()
    tr[name] = [g.get_trace(b) for b in range(3)]
    print(name, ok, g.stats()[:, [3, 4, 5]].tolist(), flush=True)
for b in range(3):
    for a, c in (("w2", "emu"), ("w3", "emu")):
        x, y = tr[a][b], tr[c][b]
        neq = np.nonzero(~((x == y) | (np.isnan(x) & np.isnan(y))))[0]
        k = neq[0] if len(neq) else -1
        print("traj", sel[b], a, "vs", c, "first diff at eval", k, (x[max(0,k-1):k+3], y[max(0,k-1):k+3]) if k >= 0 else "")
