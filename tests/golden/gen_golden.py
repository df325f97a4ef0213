"""Generates tests/golden/*.npz from the CPU oracle.

The reference ships no golden vectors for this path and cannot be built here (Eigen/ROS/PCL/OMPL absent), so these
fixtures are oracle outputs on fixed seeded inputs: they pin the oracle against regressions (and document the exact
numbers the HIP path is compared with), they do not pin it against the reference binary.  Inputs are stored too:
the init path, decision vector and ALM state; the map is regenerated from its seed by the harness and its checksum
is part of the fixture.

    python tests/golden/gen_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402
from harness import workload as wl  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    w, lens, paths, scen = wl.cuboids_batch(3, 2)
    m = orc.MapView(w.origin, w.res, w.dims, w.min_b, w.max_b, w.esdf2d, w.esdf3d)
    offs = np.concatenate([[0], np.cumsum(lens)])
    rng = np.random.default_rng(2024)
    out = dict(map_seed=42, lens=lens, paths=paths, esdf2d_sum=w.esdf2d.sum(), esdf3d_sum=w.esdf3d.sum(),
               esdf3d_sq=(w.esdf3d ** 2).sum())
    for b in range(len(lens)):
        o = orc.Oracle(m)
        n = o.set_init_traj(paths[offs[b]:offs[b + 1]])
        x0 = o.get_x()
        x = x0 + 0.05 * rng.standard_normal(n)
        lam, rho = np.array([0.3, -0.2]), np.array([1e4, 2e4])
        o.set_alm(lam, rho)
        f1, g1 = o.eval(1, x)
        f2, g2 = o.eval(2, x)
        e2 = o.final_xy_error()
        out.update({f"N_{b}": o.N, f"x0_{b}": x0, f"x_{b}": x, f"lam_{b}": lam, f"rho_{b}": rho, f"f1_{b}": f1,
                    f"g1_{b}": g1, f"f2_{b}": f2, f"g2_{b}": g2, f"xyerr_{b}": e2})
        # capped solve: inside the horizon over which stage-2 iterates are reproducible to ~1e-9 (DESIGN.md "Parity")
        o2 = orc.Oracle(m)
        o2.set_param("s2_max_iterations", 12)
        o2.set_param("alm_max_outer", 1)
        o2.set_init_traj(paths[offs[b]:offs[b + 1]])
        o2.optimize()
        st = o2.stats()
        d, c, k = o2.get_traj()
        out.update({f"cap_stats_{b}": np.array([st[key] for key in st]), f"cap_cost_{b}": o2.traj_cost(),
                    f"cap_x_{b}": o2.get_x(), f"cap_dur_{b}": d, f"cap_knots_{b}": k})
    # one full solve (trajectory 0): status, cost, N, T, knots
    o3 = orc.Oracle(m)
    o3.set_init_traj(paths[offs[0]:offs[1]])
    ok = o3.optimize()
    st = o3.stats()
    d, c, k = o3.get_traj()
    feas, strict, rep = o3.check_feasible()      # printConstraintsSituations / checkFeasible on that trajectory
    out.update(full_ok=int(ok), full_stats=np.array([st[key] for key in st]), full_cost=o3.traj_cost(), full_dur=d,
               full_knots=k, full_coeffs=c, full_x=o3.get_x(), full_gate=np.array([int(feas), int(strict)]), full_gate_report=rep,
               full_car_seq=o3.car_seq(), full_state_mid=o3.traj_state(0.37 * d.sum()))
    np.savez_compressed(os.path.join(HERE, "cuboids_seed42.npz"), **out)
    print("wrote", os.path.join(HERE, "cuboids_seed42.npz"))


if __name__ == "__main__":
    main()
