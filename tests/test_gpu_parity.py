"""Parity tests proper: the HIP path on a real MI355X, through the C-ABI (topay_amd/lib/libtopay_hip.so).

Three layers, from tight to loose (DESIGN.md "Parity"):
  1. per evaluation, HIP vs the independent oracle: cost and gradient to ~1e-13 (asserted at 1e-11/1e-10);
  2. whole solves, HIP vs the CPU lane-emulator build of the same kernel sources: BIT-IDENTICAL iterates, evaluation
     traces and results (the reference's stage-2 L-BFGS is chaotically sensitive to rounding, so bit-exact arithmetic
     is the only way converged results can be reproduced at all);
  3. whole solves, HIP vs oracle: identical while rounding has not yet been amplified (capped iterations, 1e-7), and
     statistically equivalent at convergence (success rate, end-point error, cost distribution); plus size-independent
     properties at full batch size.
"""
import numpy as np
import pytest

from conftest import EMU_LIB, serpentine_path, set_map
from oracle import oracle as orc
from topay_amd import api
from harness import workload as wl

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu(cuboids_small):
    opt = api.MomaTrajOptBatch(device=0)
    set_map(opt, cuboids_small["world"])
    opt.set_init_traj(cuboids_small["lens"], cuboids_small["paths"])
    return opt


@pytest.fixture(scope="module")
def emu(cuboids_small):
    opt = api.MomaTrajOptBatch(lib_path=EMU_LIB)
    set_map(opt, cuboids_small["world"])
    opt.set_init_traj(cuboids_small["lens"], cuboids_small["paths"])
    return opt


def test_deterministic_math_is_bitwise_equal_to_cpu(gpu, emu):
    rng = np.random.default_rng(0)
    a = np.concatenate([rng.uniform(-10, 10, 20000), rng.uniform(-1e5, 1e5, 5000), rng.uniform(-1e-3, 1e-3, 1000),
                        [0.0, np.pi / 2, -np.pi, 1e6, 1e12]])
    b = rng.uniform(-10, 10, len(a))
    mg, me = gpu.test_math(a, b), emu.test_math(a, b)
    assert (mg == me).all()
    small = np.abs(a) <= 1e5  # accuracy range the solver needs; beyond it only bit-reproducibility matters
    assert np.abs(mg[small, 0] - np.sin(a[small])).max() < 2.5e-16 and np.abs(mg[small, 1] - np.cos(a[small])).max() < 2.5e-16
    assert np.abs(mg[:, 2] - np.arctan2(a, b)).max() < 1e-15


def test_init_traj_matches_oracle(gpu, cuboids_small):
    cs = cuboids_small
    o = orc.Oracle(cs["map"])
    Ns = gpu.n_pieces()
    for b in range(len(cs["lens"])):
        o.set_init_traj(cs["paths"][cs["offs"][b]:cs["offs"][b + 1]])
        assert Ns[b] == o.N
        assert np.allclose(gpu.get_x(b), o.get_x(), rtol=0, atol=1e-12)


@pytest.mark.parametrize("stage", [1, 2])
def test_eval_matches_oracle_and_emulator(gpu, emu, cuboids_small, stage):
    cs = cuboids_small
    rng = np.random.default_rng(10 + stage)
    o = orc.Oracle(cs["map"])
    for b in range(len(cs["lens"])):
        n = o.set_init_traj(cs["paths"][cs["offs"][b]:cs["offs"][b + 1]])
        N = o.N
        for trial in range(3):
            x = o.get_x().copy()
            if trial == 1:
                x += 0.05 * rng.standard_normal(n)
            if trial == 2:  # rare paths: joint velocity/acceleration limits, mean-time band, folded arm
                x[:N] -= 1.6
                x[N - 1] += 2.5
                x[3 * N - 1:] += np.tile([0.0, 1.5, 0.0, 2.4, 0.0, 1.9, 0.0], N - 1)
            lam, rho = [0.3, -0.2], [1e4, 2e4]
            o.set_alm(lam, rho)
            f, g = o.eval(stage, x)
            fg, gg, eg = gpu.eval(stage, b, x, lam, rho)
            fe, ge, ee = emu.eval(stage, b, x, lam, rho)
            # tolerance of the floating-point parity claim: 1e-11 relative on f, 1e-10 of max|g| on g (observed ~1e-14)
            assert abs(f - fg) <= 1e-11 * abs(f)
            assert np.abs(g - gg).max() <= 1e-10 * np.abs(g).max()
            assert fg == fe and (gg == ge).all() and (eg == ee).all()  # bit-identical to the CPU execution


def test_self_colliding_arm_poses_on_gpu(gpu, emu, cuboids_small):
    """The sphere-pair path of the manipulator block on the device (pair forces through the HBM block, round 4): strongly
    perturbed joints fold the arm onto itself in most samples; against the oracle at the parity tolerance and against the
    lane emulator bit for bit."""
    cs = cuboids_small
    o = orc.Oracle(cs["map"])
    rng = np.random.default_rng(11)
    hit = 0
    for b in range(min(4, len(cs["lens"]))):
        n = o.set_init_traj(cs["paths"][cs["offs"][b]:cs["offs"][b + 1]])
        N = o.N
        for trial in range(2):
            x = o.get_x().copy()
            x[3 * N - 1:] += 4.0 * rng.standard_normal(n - 3 * N + 1)
            o.set_alm([0.1, -0.2], [1e4, 2e4])
            f, g = o.eval(2, x)
            hit += o.debug_terms()["self_colli"] > 1e3
            fg, gg, eg = gpu.eval(2, b, x, [0.1, -0.2], [1e4, 2e4])
            fe, ge, ee = emu.eval(2, b, x, [0.1, -0.2], [1e4, 2e4])
            assert abs(f - fg) <= 1e-11 * abs(f)
            assert np.abs(g - gg).max() <= 1e-10 * np.abs(g).max()
            assert fg == fe and (gg == ge).all() and (eg == ee).all()
    assert hit >= 6


def test_cost_terms_match_the_oracle_breakdown(gpu, cuboids_small):
    """Per-term cost breakdown (DebugManager, moma_traj_opt.h:566-611) on the device through topay_set_params +
    topay_eval against the oracle's accumulators, at a point where the rare terms are active."""
    cs = cuboids_small
    o = orc.Oracle(cs["map"])
    for b in (0, 3):
        n = o.set_init_traj(cs["paths"][cs["offs"][b]:cs["offs"][b + 1]])
        N = o.N
        x = o.get_x().copy()
        x[:N] -= 1.6
        x[N - 1] += 2.5
        x[3 * N - 1:] += np.tile([0.0, 1.5, 0.0, 2.4, 0.0, 1.9, 0.0], N - 1)
        o.set_alm([0.3, -0.2], [1e4, 2e4])
        f, _ = o.eval(2, x)
        t = o.debug_terms()
        d = gpu.cost_terms(b, x, [0.3, -0.2], [1e4, 2e4])
        assert t["mani_vel"] > 0 and t["mean_time"] > 0
        assert all(abs(d[k] - t[k]) <= 1e-11 * max(abs(t[k]), 1e-6 * abs(f)) for k in t), (d, t)
        assert abs(sum(d.values()) - f) <= 1e-11 * abs(f)


def test_golden_fixture_evaluations(cuboids_small):
    import os

    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cuboids_seed42.npz"))
    gpu = api.MomaTrajOptBatch(device=0)          # the fixture's own inputs (init paths as stored), the map from its seed
    set_map(gpu, cuboids_small["world"])
    gpu.set_init_traj(gold["lens"], gold["paths"])
    for b in range(len(gold["lens"])):
        for stage in (1, 2):
            f, g, _ = gpu.eval(stage, b, gold[f"x_{b}"], gold[f"lam_{b}"], gold[f"rho_{b}"])
            assert abs(f - gold[f"f{stage}_{b}"]) <= 1e-11 * abs(f)
            assert np.abs(g - gold[f"g{stage}_{b}"]).max() <= 1e-10 * np.abs(g).max()


def test_full_solve_is_bit_identical_to_emulator(cuboids_small):
    """Complete optimizeTraj (stage 1 + ALM loop) of two candidates: every evaluated cost, the final iterate, the
    counters and the returned trajectory are identical bit for bit on the GPU and on the CPU."""
    cs = cuboids_small
    lens = cs["lens"][:2]
    paths = cs["paths"][:cs["offs"][2]]
    res = {}
    for name, lib in (("gpu", None), ("emu", EMU_LIB)):
        opt = api.MomaTrajOptBatch(device=0, lib_path=lib)
        set_map(opt, cs["world"])
        opt.set_init_traj(lens, paths)
        opt.set_trace(3000)
        ok = opt.optimize()
        res[name] = dict(ok=ok, cost=opt.traj_cost.copy(), stats=opt.stats(), trace=[opt.get_trace(b) for b in range(2)],
                         x=[opt.get_x(b) for b in range(2)], traj=[opt.getTraj(b) for b in range(2)])
    g, e = res["gpu"], res["emu"]
    assert (g["ok"] == e["ok"]).all() and (g["cost"] == e["cost"]).all() and (g["stats"] == e["stats"]).all()
    for b in range(2):
        assert (g["trace"][b] == e["trace"][b]).all() and (g["x"][b] == e["x"][b]).all()
        assert (g["traj"][b]["coeffs"] == e["traj"][b]["coeffs"]).all()
        assert (g["traj"][b]["knots_xy"] == e["traj"][b]["knots_xy"]).all()
    assert g["stats"][:, 4].min() > 50  # these are real stage-2 runs, not early exits


def test_capped_solve_matches_oracle(cuboids_small):
    cs = cuboids_small
    p = api.default_params()
    p.s2_lbfgs.max_iterations = 12
    p.alm_max_outer = 1
    opt = api.MomaTrajOptBatch(params=p, device=0)
    set_map(opt, cs["world"])
    opt.optimizeTraj(cs["lens"], cs["paths"])
    st = opt.stats()
    for b in range(len(cs["lens"])):
        o = orc.Oracle(cs["map"])
        o.set_param("s2_max_iterations", 12)
        o.set_param("alm_max_outer", 1)
        o.set_init_traj(cs["paths"][cs["offs"][b]:cs["offs"][b + 1]])
        o.optimize()
        so = o.stats()
        assert list(st[b]) == [so[k] for k in api.STAT_KEYS]
        # 1e-7: twelve stage-2 iterations amplify the 1e-14 per-evaluation differences by a few orders of magnitude
        assert np.allclose(opt.get_x(b), o.get_x(), rtol=1e-7, atol=1e-8)
        assert abs(opt.traj_cost[b] - o.traj_cost()) <= 1e-8 * abs(o.traj_cost())
        tr = opt.getTraj(b)
        d, c, kn = o.get_traj()
        assert np.allclose(tr["durations"], d, rtol=1e-8) and np.allclose(tr["knots_xy"], kn, atol=1e-7)
        # getTraj coefficients (moma_traj_opt.h:943-946) of the two solves: same amplification as x
        assert np.abs(tr["coeffs"] - c).max() <= 1e-6 * np.abs(c).max()


def test_alm_rounds_match_oracle(cuboids_small):
    """The ALM multiplier / penalty update (moma_traj_opt.cpp:451-459) and the work-budget exit on the HIP path against
    the oracle: counters and rho exact, lambda and the iterate to 1e-7 (three rounds of five stage-2 iterations)."""
    from test_emu_parity import alm_rounds_case

    mk = lambda p: api.MomaTrajOptBatch(params=p, device=0)
    assert alm_rounds_case(mk, cuboids_small, [0, 1, 2, 3, 4, 5], 24000) == [3] * 6
    assert max(alm_rounds_case(mk, cuboids_small, [0, 1, 2, 3], 60)) < 3


def test_more_than_32_pieces(cuboids_small):
    """N = 33, 48, 64 (the classes of long candidates; the reference has no cap, moma_traj_opt.cpp:245, 300-321): packed
    initial guess, per-evaluation cost / gradient against the oracle at three kinds of points, a capped solve with
    identical counters, and the same capped solve bit-identical to the CPU lane emulator.  N = 171 is reported failed
    without a solve (N = 96, 128 and 170: tests/test_multiwave.py)."""
    cs = cuboids_small
    paths = [serpentine_path(L) for L in (34.0, 50.0, 66.0, 178.0)]
    lens = np.array([len(p) for p in paths], dtype=np.int32)
    opt = api.MomaTrajOptBatch(device=0)
    set_map(opt, cs["world"])
    opt.set_init_traj(lens, np.concatenate(paths))
    N = opt.n_pieces()
    assert list(N) == [33, 48, 64, 0]
    rng = np.random.default_rng(21)
    for k in range(3):
        o = orc.Oracle(cs["map"])
        n = o.set_init_traj(paths[k])
        Nk = o.N
        assert np.allclose(opt.get_x(k), o.get_x(), rtol=0, atol=1e-12)
        for trial in range(3):
            x = o.get_x().copy()
            if trial == 1:
                x += 0.03 * rng.standard_normal(n)
            if trial == 2:
                x[:Nk] -= 1.6
                x[Nk - 1] += 2.5
                x[3 * Nk - 1:] += np.tile([0.0, 1.5, 0.0, 2.4, 0.0, 1.9, 0.0], Nk - 1)
            for stage in (1, 2):
                lam, rho = [0.1, 0.2], [1e4, 3e4]
                o.set_alm(lam, rho)
                f, g = o.eval(stage, x)
                fg, gg, _ = opt.eval(stage, k, x, lam, rho)
                assert abs(f - fg) <= 1e-11 * abs(f), (Nk, trial, stage)
                assert np.abs(g - gg).max() <= 1e-10 * np.abs(g).max(), (Nk, trial, stage)
    # capped solves: stage 1 in full (not chaotic), eight stage-2 iterations
    p = api.default_params()
    p.s2_lbfgs.max_iterations = 8
    p.alm_max_outer = 1
    cap = api.MomaTrajOptBatch(params=p, device=0)
    set_map(cap, cs["world"])
    ok = cap.optimizeTraj(lens, np.concatenate(paths))
    st = cap.stats()
    assert not ok[3] and np.isnan(cap.traj_cost[3]) and (st[3] == 0).all()
    for k in range(3):
        o = orc.Oracle(cs["map"])
        o.set_param("s2_max_iterations", 8)
        o.set_param("alm_max_outer", 1)
        o.set_init_traj(paths[k])
        o.optimize()
        so = o.stats()
        assert list(st[k]) == [so[key] for key in api.STAT_KEYS], (k, list(st[k]), so)
        assert np.allclose(cap.get_x(k), o.get_x(), rtol=1e-7, atol=1e-8)
        tr = cap.getTraj(k)
        d, c, kn = o.get_traj()
        assert np.allclose(tr["durations"], d, rtol=1e-8) and np.allclose(tr["knots_xy"], kn, atol=1e-7)
        assert np.abs(tr["coeffs"] - c).max() <= 1e-6 * np.abs(c).max()
    # N = 33 against the lane emulator, every bit (few iterations: the emulator is slow)
    p.s1_lbfgs.max_iterations = 5
    p.s2_lbfgs.max_iterations = 3
    res = []
    for lib in (None, EMU_LIB):
        o2 = api.MomaTrajOptBatch(params=p, device=0, lib_path=lib)
        set_map(o2, cs["world"])
        o2.set_init_traj(lens[:1], paths[0])
        o2.set_trace(64)
        o2.optimize()
        res.append((o2.stats(), o2.get_trace(0), o2.get_x(0), o2.getTraj(0)["coeffs"]))
    g, e = res
    assert (g[0] == e[0]).all() and (g[1] == e[1]).all() and (g[2] == e[2]).all() and (g[3] == e[3]).all()


def test_config2_tables_64_candidates():
    """BASELINE config 2: one 'tables' scenario x 64 candidates.  Properties of the converged batch + statistical
    agreement with the CPU oracle (converged values themselves are not comparable one by one: chaotic iteration)."""
    world, start, goal, lens, paths = wl.tables_scenario(0, 64)
    opt = api.MomaTrajOptBatch(device=0)
    set_map(opt, world)
    ok = opt.optimizeTraj(lens, paths)
    cost = opt.traj_cost.copy()
    st = opt.stats()
    assert ok.mean() > 0.8
    for b in np.nonzero(ok)[0][:16]:
        tr = opt.getTraj(int(b))
        assert np.all(tr["durations"] > 0) and np.isfinite(tr["coeffs"]).all()
        assert np.linalg.norm(tr["knots_xy"][-1] - goal[:2]) < 0.01   # ALM tolerance reached (moma_traj_opt.cpp:451)
        assert np.allclose(tr["knots_xy"][0], start[:2])
    # same batch again: identical bits (no atomics, no timing dependence)
    ok2 = opt.optimizeTraj(lens, paths)
    assert (ok2 == ok).all() and (opt.traj_cost[ok] == cost[ok]).all() and (opt.stats() == st).all()
    m = orc.MapView(world.origin, world.res, world.dims, world.min_b, world.max_b, world.esdf2d, world.esdf3d)
    r = orc.optimize_batch(m, lens, paths, nthreads=16)
    # stage 1 is short and not chaotic: identical counters
    assert (st[:, :3] == r["stats"][:, :3]).mean() > 0.95
    assert abs(ok.mean() - r["success"].mean()) <= 0.1
    both = ok & (r["success"] == 1)
    # Cost distribution of the converged candidates.  It is multi-modal (the 64 candidates of one scenario end in a few
    # local minima: around 358, 590, 700, 760 here) and a candidate near a watershed may end in another minimum than
    # the oracle's run of it, so single quantiles next to a gap between modes are ill-conditioned.  Asserted instead:
    # (a) most candidates end in the oracle's minimum of the same candidate (costs within 5 %), (b) the two empirical
    # distributions agree at every decile within 5 % in value and 0.15 in rank (a Kolmogorov-Smirnov band; the 5 %
    # critical distance for two samples of 60 is 0.25).
    dev, ref = cost[both], r["cost"][both]
    same = np.abs(dev / ref - 1.0) < 0.05
    print(f"config2: {both.sum()} converged in both, {same.mean():.2f} in the same minimum; deciles dev/ref: "
          f"{[round(float(np.percentile(dev, q) / np.percentile(ref, q)), 3) for q in range(10, 100, 10)]}")
    assert same.mean() > 0.7
    assert abs(np.median(dev) / np.median(ref) - 1.0) < 0.05
    for q in range(10, 100, 10):
        v = np.percentile(ref, q)
        assert np.mean(dev <= 1.05 * v) >= q / 100 - 0.15 and np.mean(dev < 0.95 * v) <= q / 100 + 0.15, q
    # Converged values cannot be compared one by one (chaotic iteration), and the reference's stop
    # test (three past costs within 1e-4) also fires on plateaus: restarted at its own result, the reference algorithm
    # itself moves on by more than 1e-3 in a quarter of the cases (tools/restart_experiment.py).  What can be asserted
    # is that the device's results are converged IN THE REFERENCE ALGORITHM'S OWN SENSE as often as the reference's:
    # continue every successful candidate with the oracle's stage-2 loop from the returned x and final (lambda, rho),
    # and count how often it stops within five iterations having lowered the cost by less than 1e-3 relative.
    offs = np.concatenate([[0], np.cumsum(lens)])
    alm = opt.alm_state()

    def settles(path, x, lam_rho, c0):
        o = orc.Oracle(m)
        o.set_init_traj(path)
        o.set_x(x)
        o.set_alm(lam_rho[:2], lam_rho[2:])
        o.set_param("alm_max_outer", 1)
        o.optimize_warm()
        return o.stats()["stage2_iters"] <= 5 and abs(o.traj_cost() - c0) <= 1e-3 * abs(c0)

    dev, ref = [], []
    for b in np.nonzero(both)[0]:
        path = paths[offs[b]:offs[b + 1]]
        dev.append(settles(path, opt.get_x(int(b)), alm[b], cost[b]))
        o = orc.Oracle(m)
        o.set_init_traj(path)
        o.optimize()
        ref.append(settles(path, o.get_x(), o.alm_state(), o.traj_cost()))
    assert np.mean(dev) >= np.mean(ref) - 0.15 and np.mean(dev) > 0.4, (np.mean(dev), np.mean(ref))
    from conftest import track_agreement
    track_agreement("config2_64_candidates", dict(success_dev=ok.mean(), success_ref=r["success"].mean(), same_minimum=same.mean(),
                                                  settles_dev=np.mean(dev), settles_ref=np.mean(ref)), band=0.08)


def test_converged_solves_equal_oracle_solver_in_device_order():
    """BASELINE configs[1] (one tables scenario x 64 candidates), converged parity against independent solver code: the
    oracle's own L-BFGS / line search / ALM logic, its vector arithmetic carried out in the device's summation order,
    fed with the device's cost and gradient through topay_eval, must reproduce every device solve bit for bit --
    iterates, costs, multipliers and counters of all 64 candidates, whatever their status.  (The evaluation layer itself
    is compared per call with the oracle's at 1e-11 in the tests above; see tests/test_emu_parity.py for the reasoning.)"""
    world, start, goal, lens, paths = wl.tables_scenario(0, 64)
    opt = api.MomaTrajOptBatch(device=0)
    set_map(opt, world)
    ok = opt.optimizeTraj(lens, paths)
    st, cost, alm = opt.stats(), opt.traj_cost.copy(), opt.alm_state()
    xs = [opt.get_x(b) for b in range(len(lens))]
    ev = api.MomaTrajOptBatch(device=0)
    set_map(ev, world)
    ev.set_init_traj(lens, paths)
    m = orc.MapView(world.origin, world.res, world.dims, world.min_b, world.max_b, world.esdf2d, world.esdf3d)
    offs = np.concatenate([[0], np.cumsum(lens)])
    iters = 0
    for b in range(len(lens)):
        o = orc.Oracle(m)
        o.set_init_traj(paths[offs[b]:offs[b + 1]])
        okh = o.optimize_device_order(lambda stage, xx, lam, rho: ev.eval(stage, b, xx, lam, rho), epl=opt.class_of(o.N)[1], nw=opt.class_of(o.N)[0])
        so = o.stats()
        assert okh == bool(ok[b]), b
        assert [so["stage1_ret"], so["stage1_iters"], so["stage1_evals"], so["stage2_last_ret"], so["stage2_iters"], so["stage2_evals"],
                so["alm_outer"], so["sum_bound"]] == list(st[b]), b
        assert (o.get_x() == xs[b]).all() and (o.alm_state() == alm[b]).all(), b
        if ok[b]:
            assert o.traj_cost() == cost[b], b
        iters += int(st[b][4])
    print(f"64 candidates, {iters} stage-2 iterations in all: device solves == oracle solver logic in device order, bit for bit")


def test_converged_solves_of_a_benchmark_slice_equal_oracle_solver_in_device_order():
    """The same on a slice of the headline batch (BASELINE configs[2]): the first 32 scenarios x 8 candidates of
    bench.py's seed-42 batch, one map each, all launch classes that occur -- 256 device solves against the oracle's solver
    logic in device order, bit for bit."""
    tb = wl.TablesBatch(32, 8, base_seed=42, nthreads=8)
    opt = api.MomaTrajOptBatch(device=0)
    ev = api.MomaTrajOptBatch(device=0)
    for k, s_ in enumerate(tb.scenarios):
        set_map(opt, tb.world(s_), map_id=k)
        set_map(ev, tb.world(s_), map_id=k)
    slot = {s_: k for k, s_ in enumerate(tb.scenarios)}
    map_ids = np.array([slot[s_] for s_ in tb.scen], dtype=np.int32)
    ok = opt.optimizeTraj(tb.lens, tb.paths, map_ids=map_ids)
    st, cost, alm, Np = opt.stats(), opt.traj_cost.copy(), opt.alm_state(), opt.n_pieces()
    ev.set_init_traj(tb.lens, tb.paths, map_ids=map_ids)
    offs = np.concatenate([[0], np.cumsum(tb.lens)])
    views = {}
    iters, same = 0, 0
    for b in range(len(tb.lens)):
        sc = tb.scen[b]
        if sc not in views:
            w = tb.world(sc)
            views[sc] = orc.MapView(w.origin, w.res, w.dims, w.min_b, w.max_b, w.esdf2d, w.esdf3d)
        o = orc.Oracle(views[sc])
        o.set_init_traj(tb.paths[offs[b]:offs[b + 1]])
        okh = o.optimize_device_order(lambda stage, xx, lam, rho: ev.eval(stage, b, xx, lam, rho), epl=opt.class_of(o.N)[1], nw=opt.class_of(o.N)[0])
        so = o.stats()
        assert okh == bool(ok[b]), b
        assert [so["stage1_ret"], so["stage1_iters"], so["stage1_evals"], so["stage2_last_ret"], so["stage2_iters"], so["stage2_evals"],
                so["alm_outer"], so["sum_bound"]] == list(st[b]), b
        assert (o.get_x() == opt.get_x(b)).all() and (o.alm_state() == alm[b]).all(), b
        iters += int(st[b][4])
        same += 1
    print(f"{same} candidates (pieces {Np.min()}..{Np.max()}), {iters} stage-2 iterations: device == oracle solver logic in device order, bit for bit")
    tb.close()


def test_large_batch_properties_and_multi_map():
    """Size-independent properties on a larger multi-map batch (32 tables scenarios x 8 candidates, one map each)."""
    tb = wl.TablesBatch(32, 8, base_seed=1000, nthreads=8)
    opt = api.MomaTrajOptBatch(device=0)
    for k, s in enumerate(tb.scenarios):
        set_map(opt, tb.world(s), map_id=k)
    slot = {s: k for k, s in enumerate(tb.scenarios)}
    map_ids = np.array([slot[s] for s in tb.scen], dtype=np.int32)
    ok = opt.optimizeTraj(tb.lens, tb.paths, map_ids=map_ids)
    st = opt.stats()
    assert len(ok) == 256 and ok.mean() > 0.7
    assert np.isfinite(opt.traj_cost[ok]).all()
    assert (st[:, 0] == 1).mean() > 0.95          # stage 1 stops on the past/delta test
    assert (st[ok, 6] >= 1).all() and (st[:, 5] >= st[:, 4]).all()   # at least one ALM round; evals >= iterations
    # candidates of scenario s evaluated against a different map must differ: spot-check via the eval hook
    b = int(np.nonzero(map_ids == 1)[0][0])
    x = opt.get_x(b)
    world1 = tb.world(tb.scenarios[1])
    m1 = orc.MapView(world1.origin, world1.res, world1.dims, world1.min_b, world1.max_b, world1.esdf2d, world1.esdf3d)
    o = orc.Oracle(m1)
    off = int(np.concatenate([[0], np.cumsum(tb.lens)])[b])
    o.set_init_traj(tb.paths[off:off + tb.lens[b]])
    o.set_alm([0, 0], [1e4, 1e4])
    f, g = o.eval(2, x)
    fg, gg, _ = opt.eval(2, b, x, [0, 0], [1e4, 1e4])
    # at a converged iterate the gradient is a small difference of ~1e6-sized terms: compare on that scale
    assert abs(f - fg) <= 1e-11 * abs(f) and np.abs(g - gg).max() <= 1e-9 * max(np.abs(g).max(), abs(f))
    tb.close()


def test_config5_high_resolution_esdf():
    """BASELINE config 5 at reduced extent: 0.02 m voxels, 20 x 20 x 1.6 m => 1000 x 1000 x 80 cells, a 640 MB 3-D ESDF
    (larger than L2 + Infinity Cache, so the sphere gathers really go to HBM).  The full 50 x 50 m map (4 GB) differs only
    in extent; bench.py --workload hires builds it.  Per-evaluation parity against the oracle on that map, then a solve."""
    w = wl.World(wl.CUBOIDS, seed=7, size_xy=20.0, size_z=1.6, res=0.02, cloud_res=0.02, nthreads=0)
    assert tuple(w.dims) == (1000, 1000, 80)
    lens_l, paths_l = [], []
    sid = 0
    while len(lens_l) < 4:
        ok, s, g = w.sample_scenario(500 + sid)
        sid += 1
        if not ok:
            continue
        lens, paths = w.init_paths(s, g, 4, 999 + sid)
        if len(lens) == 4:
            lens_l.append(lens)
            paths_l.append(paths)
    lens, paths = np.concatenate(lens_l), np.concatenate(paths_l)
    offs = np.concatenate([[0], np.cumsum(lens)])
    opt = api.MomaTrajOptBatch(device=0)
    set_map(opt, w)
    opt.set_init_traj(lens, paths)
    m = orc.MapView(w.origin, w.res, w.dims, w.min_b, w.max_b, w.esdf2d, w.esdf3d)
    o = orc.Oracle(m)
    rng = np.random.default_rng(3)
    for b in range(0, len(lens), 3):
        n = o.set_init_traj(paths[offs[b]:offs[b + 1]])
        x = o.get_x() + 0.03 * rng.standard_normal(n)
        o.set_alm([0.1, -0.1], [1e4, 1e4])
        f, g = o.eval(2, x)
        fg, gg, _ = opt.eval(2, b, x, [0.1, -0.1], [1e4, 1e4])
        assert abs(f - fg) <= 1e-11 * abs(f) and np.abs(g - gg).max() <= 1e-10 * np.abs(g).max()
    ok = opt.optimize()
    assert ok.mean() >= 0.5 and np.isfinite(opt.traj_cost[ok]).all()
    # converged parity on the map no cache holds: the oracle's solver logic in device order, fed with the device's
    # evaluations, reproduces the device's solves bit for bit (see test_converged_solves_equal_oracle_solver_in_device_order)
    st, alm = opt.stats(), opt.alm_state()
    xs = [opt.get_x(b) for b in range(len(lens))]
    ev = api.MomaTrajOptBatch(device=0)
    set_map(ev, w)
    ev.set_init_traj(lens, paths)
    for b in range(0, len(lens), 4):
        oh = orc.Oracle(m)
        oh.set_init_traj(paths[offs[b]:offs[b + 1]])
        okh = oh.optimize_device_order(lambda stage, xx, lam, rho: ev.eval(stage, b, xx, lam, rho), epl=opt.class_of(oh.N)[1], nw=opt.class_of(oh.N)[0])
        so = oh.stats()
        assert okh == bool(ok[b]) and (oh.get_x() == xs[b]).all() and (oh.alm_state() == alm[b]).all(), b
        assert [so["stage1_iters"], so["stage1_evals"], so["stage2_iters"], so["stage2_evals"], so["alm_outer"]] == \
            [st[b][1], st[b][2], st[b][4], st[b][5], st[b][6]], b
    w.close()


def test_config3_full_size_properties():
    """BASELINE config 3 at full size: 1024 start/goal scenarios x 8 candidates on one cuboids map (8192 trajectories).
    Size-independent properties: every success meets the ALM end-point tolerance, durations are positive, start knots
    are the start positions, the per-candidate counters are consistent, a second run is bit-identical, and a sample of
    converged candidates re-evaluated by the oracle at the returned x gives the same cost to 1e-11."""
    w, lens, paths, scen = wl.cuboids_batch(1024, 8)
    offs = np.concatenate([[0], np.cumsum(lens)])
    opt = api.MomaTrajOptBatch(device=0)
    set_map(opt, w)
    ok = opt.optimizeTraj(lens, paths)
    cost = opt.traj_cost.copy()
    st = opt.stats()
    assert len(ok) == 8192 and ok.mean() > 0.85
    assert (st[:, 5] >= st[:, 4]).all() and (st[ok, 6] >= 1).all()
    assert (st[:, 6] <= api.default_params().alm_max_outer).all()
    for b in np.nonzero(ok)[0][::517]:
        tr = opt.getTraj(int(b))
        goal = paths[offs[b + 1] - 1][:2]
        start = paths[offs[b]][:2]
        assert np.all(tr["durations"] > 0) and np.isfinite(tr["coeffs"]).all()
        assert np.linalg.norm(tr["knots_xy"][-1] - goal) < 0.01 and np.allclose(tr["knots_xy"][0], start)
    # the reported cost is the stage-2 cost at the returned x with the final ALM state: check with the oracle
    m = orc.MapView(w.origin, w.res, w.dims, w.min_b, w.max_b, w.esdf2d, w.esdf3d)
    o = orc.Oracle(m)
    alm = opt.alm_state()
    for b in np.nonzero(ok)[0][::1031]:
        o.set_init_traj(paths[offs[b]:offs[b + 1]])
        o.set_alm(alm[b, :2], alm[b, 2:])
        f, _ = o.eval(2, opt.get_x(int(b)))
        assert abs(f - cost[b]) <= 1e-9 * abs(f)
    ok2 = opt.optimizeTraj(lens, paths)
    assert (ok2 == ok).all() and (opt.traj_cost[ok] == cost[ok]).all() and (opt.stats() == st).all()
    w.close()


def test_eval_parity_every_bucket_multi_map():
    """Per-evaluation parity HIP vs oracle over every N bucket (rows per lane 1, 2, 3 and 6), three kinds of points
    (initial guess, random perturbation, rare-path trigger), both stages, every candidate against its own map."""
    tb = wl.TablesBatch(128, 8, base_seed=7000, nthreads=0)
    opt = api.MomaTrajOptBatch(device=0)
    views = {}
    slot = {}
    for k, s in enumerate(tb.scenarios):
        w = tb.world(s)
        set_map(opt, w, map_id=k)
        slot[s] = k
        views[s] = orc.MapView(w.origin, w.res, w.dims, w.min_b, w.max_b, w.esdf2d, w.esdf3d)
    map_ids = np.array([slot[s] for s in tb.scen], dtype=np.int32)
    opt.set_init_traj(tb.lens, tb.paths, map_ids=map_ids)
    N = opt.n_pieces()
    offs = np.concatenate([[0], np.cumsum(tb.lens)])
    rng = np.random.default_rng(1)
    buckets_seen = 0
    for lo, hi in ((3, 7), (8, 10), (11, 13), (14, 16), (17, 21), (22, 26), (27, 32), (33, 64)):
        idx = np.nonzero((N >= lo) & (N <= hi))[0]
        if len(idx) == 0:
            continue
        buckets_seen += 1
        for b in rng.choice(idx, size=min(6, len(idx)), replace=False):
            b = int(b)
            o = orc.Oracle(views[int(tb.scen[b])])
            n = o.set_init_traj(tb.paths[offs[b]:offs[b + 1]])
            Nb = o.N
            for trial in range(3):
                x = o.get_x().copy()
                if trial == 1:
                    x += 0.08 * rng.standard_normal(n)
                if trial == 2:
                    x[:Nb] -= 1.5
                    x[Nb - 1] += 2.2
                    x[3 * Nb - 1:] += np.tile([0.0, 1.5, 0.0, 2.4, 0.0, 1.9, 0.0], Nb - 1)
                lam, rho = rng.uniform(-1, 1, 2), np.array([1e4, 3e5])
                o.set_alm(lam, rho)
                for stage in (1, 2):
                    f, g = o.eval(stage, x)
                    if not np.isfinite(f):
                        continue
                    fg, gg, _ = opt.eval(stage, b, x, lam, rho)
                    assert abs(f - fg) <= 1e-11 * abs(f)
                    assert np.abs(g - gg).max() <= 1e-10 * np.abs(g).max()
    assert buckets_seen >= 6
    tb.close()


def test_capped_solves_of_long_trajectories_are_bit_identical_to_emulator():
    """Rows-per-lane 2 and 3 code paths (N in 14..21 and 22..32): stage 1 plus ten stage-2 iterations, GPU vs the CPU
    lane emulator, every bit (a full solve of such a trajectory takes minutes in the emulator)."""
    tb = wl.TablesBatch(128, 8, base_seed=7000, nthreads=0)
    probe = api.MomaTrajOptBatch(device=0)
    w0 = tb.world(tb.scenarios[0])
    set_map(probe, w0)
    probe.set_init_traj(tb.lens, tb.paths)       # piece counts only depend on the paths
    N = probe.n_pieces()
    offs = np.concatenate([[0], np.cumsum(tb.lens)])
    picks = [int(np.nonzero((N >= lo) & (N <= hi))[0][0]) for lo, hi in ((14, 21), (22, 32))]
    p = api.default_params()
    p.s2_lbfgs.max_iterations = 10
    p.alm_max_outer = 1
    for b in picks:
        w = tb.world(int(tb.scen[b]))
        path = tb.paths[offs[b]:offs[b + 1]]
        res = []
        for lib in (None, EMU_LIB):
            opt = api.MomaTrajOptBatch(params=p, device=0, lib_path=lib)
            set_map(opt, w)
            opt.set_init_traj(np.array([len(path)], dtype=np.int32), path)
            opt.set_trace(200)
            ok = opt.optimize()
            res.append((ok.copy(), opt.traj_cost.copy(), opt.stats(), opt.get_trace(0), opt.get_x(0)))
        g, e = res
        assert (g[0] == e[0]).all() and (g[2] == e[2]).all() and (g[3] == e[3]).all() and (g[4] == e[4]).all()
        assert (g[1] == e[1]).all() or (np.isnan(g[1]).all() and np.isnan(e[1]).all())
        assert g[2][0, 5] >= 10
    tb.close()


def test_two_batches_in_flight_give_the_serial_results():
    """topay_optimize_async / topay_synchronize: two contexts issued back to back on one GPU return exactly what a
    serial solve returns (no cross-talk through the dispatch gate or the shared parameter block)."""
    tb = wl.TablesBatch(16, 8, base_seed=4242, nthreads=8)
    half = len(tb.lens) // 2
    offs = np.concatenate([[0], np.cumsum(tb.lens)])
    parts = [(tb.lens[:half], tb.paths[:offs[half]], tb.scen[:half]), (tb.lens[half:], tb.paths[offs[half]:], tb.scen[half:])]

    def make(part):
        lens, paths, scen = part
        o = api.MomaTrajOptBatch(device=0)
        slot = {}
        for s in sorted(set(scen.tolist())):
            slot[s] = len(slot)
            set_map(o, tb.world(s), map_id=slot[s])
        o.set_init_traj(lens, paths, map_ids=np.array([slot[s] for s in scen], dtype=np.int32))
        return o

    serial = []
    for part in parts:
        o = make(part)
        ok = o.optimize()
        serial.append((ok.copy(), o.traj_cost.copy(), o.stats()))
    a, b = make(parts[0]), make(parts[1])
    a.optimize_async()
    b.optimize_async()
    res = [(a.finish(), a.traj_cost.copy(), a.stats()), (b.finish(), b.traj_cost.copy(), b.stats())]
    for (ok_s, c_s, st_s), (ok_p, c_p, st_p) in zip(serial, res):
        assert (ok_s == ok_p).all() and (st_s == st_p).all()
        assert ((c_s == c_p) | (np.isnan(c_s) & np.isnan(c_p))).all()
    tb.close()


@pytest.mark.gpu
def test_persistent_queues_do_not_change_results(monkeypatch):
    """The persistent launch (resident workgroups taking candidates from a queue; which wave solves which candidate is
    timing-dependent) returns bit for bit what the one-workgroup-per-candidate launch returns.  1536 candidates on 1024
    SIMD slots, so every queue is really drained by looping workgroups."""
    tb = wl.TablesBatch(192, 8, base_seed=777, nthreads=8)
    slot = {s: k for k, s in enumerate(tb.scenarios)}
    map_ids = np.array([slot[s] for s in tb.scen], dtype=np.int32)

    def solve():
        o = api.MomaTrajOptBatch(device=0)
        for s in tb.scenarios:
            set_map(o, tb.world(s), map_id=slot[s])
        o.set_init_traj(tb.lens, tb.paths, map_ids=map_ids)
        ok = o.optimize()
        out = (ok.copy(), o.traj_cost.copy(), o.stats(), o.total_durations(), [o.get_x(b) for b in (0, 500, 1535)])
        o.close()
        return out

    monkeypatch.setenv("TOPAY_PERSISTENT", "0")
    ref = solve()
    monkeypatch.setenv("TOPAY_PERSISTENT", "1")
    per = solve()
    per2 = solve()
    for got in (per, per2):
        assert (ref[0] == got[0]).all() and (ref[2] == got[2]).all()
        assert ((ref[1] == got[1]) | (np.isnan(ref[1]) & np.isnan(got[1]))).all()
        assert ((ref[3] == got[3]) | (np.isnan(ref[3]) & np.isnan(got[3]))).all()
        for xa, xb in zip(ref[4], got[4]):
            assert (xa == xb).all()
    assert ref[0].mean() > 0.8
    tb.close()


@pytest.mark.gpu
def test_contexts_with_different_parameters_in_flight():
    """The kernels read the parameters from one __constant__ block: two contexts with different parameters must not
    overlap on the device (the library makes the second wait), and each must return its own serial result."""
    tb = wl.TablesBatch(8, 8, base_seed=999, nthreads=8)
    slot = {s: k for k, s in enumerate(tb.scenarios)}
    map_ids = np.array([slot[s] for s in tb.scen], dtype=np.int32)
    lib = api.load()
    pa = api.default_params(lib)
    pb = api.default_params(lib)
    pb.s2_lbfgs.max_iterations = 15
    pb.alm_max_outer = 1
    pb.s2_time_weight = 80.0            # a weight the kernels read from the constant block throughout the solve

    def make(p):
        o = api.MomaTrajOptBatch(params=p, device=0)
        for s in tb.scenarios:
            set_map(o, tb.world(s), map_id=slot[s])
        o.set_init_traj(tb.lens, tb.paths, map_ids=map_ids)
        return o

    serial = []
    for p in (pa, pb):
        o = make(p)
        ok = o.optimize()
        serial.append((ok.copy(), o.traj_cost.copy(), o.stats()))
        o.close()
    assert (serial[0][2] != serial[1][2]).any()          # the two parameter sets really behave differently
    a, b = make(pa), make(pb)
    a.optimize_async()
    b.optimize_async()
    got = [(a.finish(), a.traj_cost.copy(), a.stats()), (b.finish(), b.traj_cost.copy(), b.stats())]
    for (ok_s, c_s, st_s), (ok_p, c_p, st_p) in zip(serial, got):
        assert (ok_s == ok_p).all() and (st_s == st_p).all()
        assert ((c_s == c_p) | (np.isnan(c_s) & np.isnan(c_p))).all()
    tb.close()


@pytest.mark.gpu
def test_rccl_record_gather_on_one_gpu():
    """The multi-GPU code path of bench.py (RCCL process group, per-scenario record all-gather overlapped with the next
    batch's persistent solve, max-over-ranks timing) on the one GPU there is: one rank under torch.distributed.run with
    TOPAY_FORCE_DIST=1.  The launcher is started before anything touches the GPU in that process tree; this test only
    reads the JSON line.  What came back over RCCL must be the rank's own records, bit for bit."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TOPAY_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--scenarios", "64",
           "--no-cpu-baseline", "--no-config1", "--no-serial"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    g = out["config"]["record_gather"]
    assert g is not None and g["gathers"] >= 4 and g["own_rows_match"] is True and g["rows"] == 64 == g["rows_expected"]
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["config"]["n_not_launched"] == 0


@pytest.mark.gpu
def test_config5_full_size_field_evaluation_parity():
    """BASELINE configs[4] at FULL size: 50 x 50 x 1.6 m at 0.02 m = 2500 x 2500 x 80 cells, a 4.0 GB 3-D field whose
    byte offsets pass 2^31 and 2^32.  The occupancy comes from the harness, the fields are built on the device (the
    serial-envelope passes: lines of 2500 cells) and copied back for the oracle, which then evaluates the same x on the
    same buffers: per-evaluation parity at the far corner of the map, where the offsets are largest."""
    w = wl.World(wl.CUBOIDS, seed=42, size_xy=50.0, size_z=1.6, res=0.02, cloud_res=0.02, nthreads=-1)
    assert tuple(w.dims) == (2500, 2500, 80)
    opt = api.MomaTrajOptBatch(device=0)
    opt.build_esdf(w.origin, w.res, w.dims, w.min_b, w.max_b, w.occ2d, w.occ3d)
    e2, e3, ms = opt.get_map()
    assert e3.nbytes == 4_000_000_000 and np.isfinite(e3[::100003]).all() and (e3[::100003] < 50.0).all()
    m = orc.MapView(w.origin, w.res, w.dims, w.min_b, w.max_b, e2, e3)
    rng = np.random.default_rng(8)
    paths = []
    for corner in ((21.0, 21.0), (-22.0, 20.0), (18.0, -23.0)):        # x = 21 m: cell 2300 of 2500, offset ~3.7e9 B
        a = np.array(corner)
        ang = rng.uniform(-np.pi, np.pi)
        g = np.clip(a + 5.0 * np.array([np.cos(ang), np.sin(ang)]), -23.5, 23.5)
        t = np.linspace(0, 1, 9)[:, None]
        q0, q1 = rng.uniform(-1, 1, 7), rng.uniform(-1, 1, 7)
        paths.append(np.concatenate([a + t * (g - a), np.full((9, 1), np.arctan2(*(g - a)[::-1])), q0 + t * (q1 - q0)], axis=1))
    lens = np.array([len(p) for p in paths], dtype=np.int32)
    opt.set_init_traj(lens, np.concatenate(paths))
    o = orc.Oracle(m)
    for b, p in enumerate(paths):
        n = o.set_init_traj(p)
        assert np.allclose(opt.get_x(b), o.get_x(), rtol=0, atol=1e-12)
        for trial in range(2):
            x = o.get_x() + (0.05 * rng.standard_normal(n) if trial else 0.0)
            o.set_alm([0.2, -0.1], [1e4, 2e4])
            f, g = o.eval(2, x)
            fg, gg, _ = opt.eval(2, b, x, [0.2, -0.1], [1e4, 2e4])
            assert abs(f - fg) <= 1e-11 * abs(f) and np.abs(g - gg).max() <= 1e-10 * np.abs(g).max()
    ok = opt.optimize()
    assert np.isfinite(opt.traj_cost[ok]).all()
    print(f"4 GB field built on the device in {ms:.0f} ms")
    w.close()
