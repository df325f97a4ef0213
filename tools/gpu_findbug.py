import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from topay_amd.harness import workload as wl
from topay_amd import api
from oracle import oracle as orc
w2, lens2, paths2, scen2 = wl.cuboids_batch(256, 8)
offs = np.concatenate([[0], np.cumsum(lens2)])
T = int(os.environ.get("TRAJ", "643"))
lens = lens2[[T]]; paths = paths2[offs[T]:offs[T+1]]
m = orc.MapView(w2.origin, w2.res, w2.dims, w2.min_b, w2.max_b, w2.esdf2d, w2.esdf3d)
o = orc.Oracle(m); o.set_init_traj(paths)
names = ["jerk", "time", "chassis_colli", "moment", "acc", "domega", "mani_colli", "self_colli", "mani_pos", "mani_vel", "mani_acc", "mean_time", "endp"]
for cap in [int(c) for c in os.environ.get("CAPS", "4,8,12,16,20,24").split(",")]:
    xs = {}
    ctxs = {}
    for name, lib in (("gpu", os.environ.get("TOPAY_LIB")), ("emu", "tests/emu/libtopay_emu.so")):
        p = api.default_params(api.load(lib))
        p.s2_lbfgs.max_iterations = cap
        p.alm_max_outer = 1
        g = api.MomaTrajOptBatch(params=p, device=0, lib_path=lib)
        g.set_map(w2.origin, w2.res, w2.dims, w2.min_b, w2.max_b, w2.esdf2d, w2.esdf3d)
        g.set_init_traj(lens, paths)
        g.optimize()
        xs[name] = g.get_x(0)
        ctxs[name] = g
    same = (xs["gpu"] == xs["emu"]).all()
    x = xs["emu"]
    lam, rho = [0.0, 0.0], [1e4, 1e4]
    o.set_alm(lam, rho)
    fo, go = o.eval(2, x)
    fg, gg, _ = ctxs["gpu"].eval(2, 0, x, lam, rho)
    fe, ge, _ = ctxs["emu"].eval(2, 0, x, lam, rho)
    terms = o.debug_terms()
    act = [k for k in names if terms[k] != 0]
    print("cap", cap, "x same", bool(same), "maxdx %.3e" % np.abs(xs["gpu"] - xs["emu"]).max(), "| f gpu-emu %.3e" % abs(fg - fe), "g gpu-emu %.3e" % np.abs(gg - ge).max(),
          "| g emu-oracle rel %.2e" % (np.abs(ge - go).max() / np.abs(go).max()), "g gpu-oracle rel %.2e" % (np.abs(gg - go).max() / np.abs(go).max()), "active:", act, flush=True)
    if np.abs(gg - ge).max() > 0:
        bad = np.nonzero(gg != ge)[0]
        N = (len(x) + 8) // 10
        print("    N", N, "bad idx", bad[:20], "layout: tau[0:%d] theta[%d:%d] arc[%d:%d] vq[%d:]" % (N, N, 2*N-1, 2*N-1, 3*N-1, 3*N-1))
        break
