"""How converged is "converged"?  The CPU oracle (the reference algorithm) solves the candidates of a scenario, then its
stage-2 loop is restarted at its own result (same x, same final lambda / rho, empty L-BFGS history).  A stationary
point would stop after `past` + 1 iterations with no change; the reference's stop test (|f_{k-3} - f_k| < 1e-4 |f_k|)
also fires on plateaus, so a fraction of the restarts keeps descending.  Test infrastructure.

    python tools/restart_experiment.py [n_scenarios] [candidates]
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc
from harness import workload as wl

S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
Cn = int(sys.argv[2]) if len(sys.argv) > 2 else 8
w, lens, paths, scen = wl.cuboids_batch(S, Cn)
m = orc.MapView(w.origin, w.res, w.dims, w.min_b, w.max_b, w.esdf2d, w.esdf3d)
offs = np.concatenate([[0], np.cumsum(lens)])
rows = []
for b in range(len(lens)):
    o = orc.Oracle(m)
    o.set_init_traj(paths[offs[b]:offs[b + 1]])
    if not o.optimize():
        continue
    x, alm, c = o.get_x(), o.alm_state(), o.traj_cost()
    _, _, kn = o.get_traj()
    o2 = orc.Oracle(m)
    o2.set_init_traj(paths[offs[b]:offs[b + 1]])
    o2.set_x(x)
    o2.set_alm(alm[:2], alm[2:])
    o2.set_param("alm_max_outer", 1)
    o2.optimize_warm()
    _, _, kn2 = o2.get_traj()
    rows.append((o2.stats()["stage2_iters"], abs(o2.traj_cost() - c) / abs(c), np.abs(kn2 - kn).max()))
r = np.array(rows)
print(f"{len(r)} converged candidates restarted at their own result:")
print(f"  stop within 5 iterations: {np.mean(r[:, 0] <= 5):.2f};  median iterations {np.median(r[:, 0]):.0f}, max {r[:, 0].max():.0f}")
for t in (1e-5, 1e-4, 1e-3, 1e-2):
    print(f"  relative cost change > {t:g}: {np.mean(r[:, 1] > t):.2f}")
print(f"  knot displacement: median {np.median(r[:, 2]):.2e} m, 90th percentile {np.percentile(r[:, 2], 90):.2e} m, max {r[:, 2].max():.2f} m")
