"""First-contact GPU probe: math bit-exactness, eval parity, GPU-vs-emulator lockstep, batch timing."""
import sys, time, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from topay_amd.harness import workload as wl
from topay_amd import api
from oracle import oracle as orc

EMU = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "emu", "libtopay_emu.so")
t0 = time.time()
gpu = api.MomaTrajOptBatch(device=0, lib_path=os.environ.get('TOPAY_LIB'))
emu = api.MomaTrajOptBatch(lib_path=EMU)
print("ctx created", time.time() - t0, flush=True)

# 1. deterministic math: GPU vs emulator bitwise
rng = np.random.default_rng(0)
a = np.concatenate([rng.uniform(-10, 10, 20000), rng.uniform(-1e5, 1e5, 5000), rng.uniform(-1e-3, 1e-3, 1000)])
b = rng.uniform(-10, 10, len(a))
mg, me = gpu.test_math(a, b), emu.test_math(a, b)
print("math bitwise equal:", [bool((mg[:, k] == me[:, k]).all()) for k in range(4)], "max abs diff", np.abs(mg - me).max(), flush=True)

# 2. eval parity vs oracle + vs emu on a cuboids mini batch
w, lens, paths, scen = wl.cuboids_batch(3, 2)
m = orc.MapView(w.origin, w.res, w.dims, w.min_b, w.max_b, w.esdf2d, w.esdf3d)
for o_ in (gpu, emu):
    o_.set_map(w.origin, w.res, w.dims, w.min_b, w.max_b, w.esdf2d, w.esdf3d)
    o_.set_init_traj(lens, paths)
offs = np.concatenate([[0], np.cumsum(lens)])
o = orc.Oracle(m)
for bb in range(len(lens)):
    n = o.set_init_traj(paths[offs[bb]:offs[bb + 1]])
    x0 = o.get_x()
    print("traj", bb, "N", o.N, "x0: gpu-vs-oracle", np.abs(gpu.get_x(bb) - x0).max(), "gpu==emu", bool((gpu.get_x(bb) == emu.get_x(bb)).all()))
    for stage in (1, 2):
        x = x0 + 0.05 * rng.standard_normal(n)
        lam, rho = [0.3, -0.2], [1e4, 2e4]
        o.set_alm(lam, rho)
        f, g = o.eval(stage, x)
        fg, gg, eg = gpu.eval(stage, bb, x, lam, rho)
        fe, ge, ee = emu.eval(stage, bb, x, lam, rho)
        print("   stage", stage, "rel df(gpu,oracle) %.2e" % (abs(f - fg) / abs(f)), "max rel dg %.2e" % (np.abs(g - gg).max() / np.abs(g).max()),
              "| gpu==emu: f", fg == fe, "g", bool((gg == ge).all()), "maxdiff %.3e" % np.abs(gg - ge).max(), flush=True)

# 3. full solve lockstep GPU vs emulator (2 trajectories), with traces
for o_ in (gpu, emu):
    o_.set_init_traj(lens[:2], paths[:offs[2]])
    o_.set_trace(3000)
t = time.time(); sg = gpu.optimize(); tg = time.time() - t
t = time.time(); se = emu.optimize(); te = time.time() - t
print("solve gpu %.3fs emu %.1fs" % (tg, te), "succ", sg, se, "cost gpu", gpu.traj_cost, "emu", emu.traj_cost)
for bb in range(2):
    a_, b_ = gpu.get_trace(bb), emu.get_trace(bb)
    neq = np.nonzero(a_ != b_)[0]
    print("  traj", bb, "stats gpu", gpu.stats()[bb], "emu", emu.stats()[bb], "trace identical:", len(neq) == 0, "first diff", (neq[0] if len(neq) else -1),
          "x identical", bool((gpu.get_x(bb) == emu.get_x(bb)).all()), flush=True)

# 4. batch solves: timing + stats vs oracle
for S in ((32, 256) if not os.environ.get('PROBE_QUICK') else ()):
    w2, lens2, paths2, scen2 = wl.cuboids_batch(S, 8)
    gpu.set_map(w2.origin, w2.res, w2.dims, w2.min_b, w2.max_b, w2.esdf2d, w2.esdf3d)
    gpu.set_init_traj(lens2, paths2)
    t = time.time(); ok = gpu.optimize(); dt = time.time() - t
    ms, nl = gpu.last_kernel_ms()
    st = gpu.stats()
    print("batch", len(lens2), "wall %.3fs kernel %.1f ms" % (dt, ms), "traj/s %.0f" % (len(lens2) / (ms * 1e-3)), "success", ok.mean(),
          "mean s2 evals", st[:, 5].mean(), "max", st[:, 5].max(), "N hist", np.bincount(gpu.n_pieces()), flush=True)
    if S == 32:
        m2 = orc.MapView(w2.origin, w2.res, w2.dims, w2.min_b, w2.max_b, w2.esdf2d, w2.esdf3d)
        r = orc.optimize_batch(m2, lens2, paths2, nthreads=os.cpu_count())
        print("  oracle: %.2fs on %d threads -> %.1f traj/s, success %.3f, mean s2 evals %.1f" % (r['seconds'], os.cpu_count(), len(lens2) / r['seconds'], r['success'].mean(), r['stats'][:, 5].mean()))
        rel = np.abs(gpu.traj_cost - r['cost']) / np.abs(r['cost'])
        print("  rel cost diff gpu-vs-oracle: frac<1e-5 %.3f  frac<1e-2 %.3f  median %.2e" % ((rel < 1e-5).mean(), (rel < 1e-2).mean(), np.median(rel)))
        print("  mean cost gpu %.3f oracle %.3f" % (gpu.traj_cost[ok].mean(), r['cost'][r['success'] == 1].mean()))
print("done", time.time() - t0)
