"""Kernel logic vs the oracle on the CPU: the HIP kernel sources are compiled unmodified with g++ against the
lane emulator in tests/emu (test infrastructure, never a product fallback) and driven through the same C-ABI.
Per-evaluation parity is tight (~1e-13); whole solves are compared only while the reference's chaotic stage-2
iteration has not yet amplified rounding differences (capped iterations), see DESIGN.md "Parity"."""
import numpy as np
import pytest

from conftest import EMU_LIB, serpentine_path, set_map
from oracle import oracle as orc
from topay_amd import api
from harness import workload as wl


@pytest.fixture(scope="module")
def emu(cuboids_small):
    opt = api.MomaTrajOptBatch(lib_path=EMU_LIB)
    set_map(opt, cuboids_small["world"])
    opt.set_init_traj(cuboids_small["lens"], cuboids_small["paths"])
    return opt


def test_init_traj_matches_oracle(emu, cuboids_small):
    cs = cuboids_small
    o = orc.Oracle(cs["map"])
    Ns = emu.n_pieces()
    for b in range(len(cs["lens"])):
        n = o.set_init_traj(cs["paths"][cs["offs"][b]:cs["offs"][b + 1]])
        assert Ns[b] == o.N and n == 10 * o.N - 8
        assert np.allclose(emu.get_x(b), o.get_x(), rtol=0, atol=1e-12)


def test_shared_map_slots_give_the_owner_results(emu, cuboids_small):
    """topay_share_maps: a second context that borrows the first one's resident map evaluates to the same bits (the
    reference's optimisers all hold the planner's one GridMap::Ptr, planner.cpp:59-75)."""
    cs = cuboids_small
    other = api.MomaTrajOptBatch(lib_path=EMU_LIB)
    with pytest.raises(api.TopayError):
        other.set_init_traj(cs["lens"], cs["paths"])          # no map yet
    other.share_maps(emu, 0, 1)
    other.set_init_traj(cs["lens"], cs["paths"])
    for b in (0, 3):
        x = emu.get_x(b)
        f0, g0, _ = emu.eval(2, b, x, [0.1, 0.2], [1e4, 1e4])
        f1, g1, _ = other.eval(2, b, x, [0.1, 0.2], [1e4, 1e4])
        assert f0 == f1 and (g0 == g1).all()
    other.close()


@pytest.mark.parametrize("stage", [1, 2])
def test_eval_matches_oracle(emu, cuboids_small, stage):
    cs = cuboids_small
    rng = np.random.default_rng(stage)
    o = orc.Oracle(cs["map"])
    for b in range(len(cs["lens"])):  # N = 4..11: both row classes (one / two system rows per lane)
        n = o.set_init_traj(cs["paths"][cs["offs"][b]:cs["offs"][b + 1]])
        for trial in range(2):
            x = o.get_x() + (0.05 * rng.standard_normal(n) if trial else 0.0)
            lam, rho = [0.3, -0.2], [1e4, 2e4]
            o.set_alm(lam, rho)
            f, g = o.eval(stage, x)
            fe, ge, e = emu.eval(stage, b, x, lam, rho)
            assert abs(f - fe) <= 1e-12 * abs(f)
            assert np.abs(g - ge).max() <= 1e-11 * np.abs(g).max()
            if stage == 2:
                assert np.allclose(e, o.final_xy_error(), atol=1e-12)


def test_eval_rare_paths_match_oracle(emu, cuboids_small):
    """Much shorter pieces switch on the joint velocity / acceleration limits (gradBeta rows 1 and 2 of the joints,
    moma_traj_opt.cpp:1674-1710) and the mean-time band; folded joints switch on self collision terms."""
    cs = cuboids_small
    o = orc.Oracle(cs["map"])
    for b in range(len(cs["lens"])):
        n = o.set_init_traj(cs["paths"][cs["offs"][b]:cs["offs"][b + 1]])
        N = o.N
        x = o.get_x().copy()
        x[:N] -= 1.6
        x[N - 1] += 2.5  # one long piece: mean-time band
        x[3 * N - 1:] += np.tile([0.0, 1.5, 0.0, 2.4, 0.0, 1.9, 0.0], N - 1)  # fold the arm
        o.set_alm([0, 0], [1e4, 1e4])
        f, g = o.eval(2, x)
        t = o.debug_terms()
        assert t["mani_vel"] > 0 and t["mani_acc"] > 0 and t["mean_time"] > 0
        fe, ge, _ = emu.eval(2, b, x, [0, 0], [1e4, 1e4])
        assert abs(f - fe) <= 1e-12 * abs(f)
        assert np.abs(g - ge).max() <= 1e-11 * np.abs(g).max()
        if b < 2:
            # the per-term breakdown (DebugManager, moma_traj_opt.h:566-611) through topay_set_params + topay_eval
            d = emu.cost_terms(b, x, [0.3, -0.2], [1e4, 2e4])
            o.set_alm([0.3, -0.2], [1e4, 2e4])
            f2, _ = o.eval(2, x)
            t2 = o.debug_terms()
            assert all(abs(d[k] - t2[k]) <= 1e-11 * max(abs(t2[k]), 1e-6 * abs(f2)) for k in t2), (d, t2)
            assert abs(sum(d.values()) - f2) <= 1e-11 * abs(f2)
            assert emu.eval(2, b, x, [0, 0], [1e4, 1e4])[0] == fe        # the context's own parameters are back
    assert t["self_colli"] >= 0


def test_capped_solve_matches_oracle(cuboids_small):
    """Stage 1 in full and the first stage-2 iterations: same iteration/evaluation counts, same iterate."""
    cs = cuboids_small
    p = api.default_params(api.load(EMU_LIB))
    p.s2_lbfgs.max_iterations = 12
    p.alm_max_outer = 1
    opt = api.MomaTrajOptBatch(params=p, lib_path=EMU_LIB)
    set_map(opt, cs["world"])
    sel = [0, 1, 5]
    lens = cs["lens"][sel]
    paths = np.concatenate([cs["paths"][cs["offs"][i]:cs["offs"][i + 1]] for i in sel])
    opt.optimizeTraj(lens, paths)
    st = opt.stats()
    for k, b in enumerate(sel):
        o = orc.Oracle(cs["map"])
        o.set_param("s2_max_iterations", 12)
        o.set_param("alm_max_outer", 1)
        o.set_init_traj(cs["paths"][cs["offs"][b]:cs["offs"][b + 1]])
        o.optimize()
        so = o.stats()
        assert list(st[k][:7]) == [so[key] for key in api.STAT_KEYS[:7]]
        assert st[k][7] == so["sum_bound"]
        assert np.allclose(opt.get_x(k), o.get_x(), rtol=1e-7, atol=1e-8)
        assert abs(opt.traj_cost[k] - o.traj_cost()) <= 1e-8 * abs(o.traj_cost())
        tr = opt.getTraj(k)
        d, c, kn = o.get_traj()
        assert np.allclose(tr["durations"], d, rtol=1e-8) and np.allclose(tr["coeffs"], c, rtol=1e-6, atol=1e-7)
        assert np.allclose(tr["knots_xy"], kn, atol=1e-7)
    # the batched getter (the planner's winners in one call) and the PolyTraj.msg layout return the same numbers
    pick = [2, 0]
    pk = opt.getTrajs(pick)
    for j, k in enumerate(pick):
        tr = opt.getTraj(k)
        a, b = pk["piece_off"][j], pk["piece_off"][j + 1]
        assert (pk["durations"][a:b] == tr["durations"]).all() and (pk["coeffs"][a:b] == tr["coeffs"]).all()
        assert (pk["knots_xy"][a + j:b + j + 1] == tr["knots_xy"]).all()
        order, cf, du, dirs = opt.polytraj_msg(k)
        assert order == 5 and (cf == tr["coeffs"].astype(np.float32)).all() and (du == tr["durations"].astype(np.float32)).all()
        assert set(dirs.tolist()) <= {-1, 1} and len(dirs) == len(du)


def alm_rounds_case(opt_factory, cs, sel, budget):
    """Shared by the emulator and the GPU test: three ALM rounds of five stage-2 iterations each (every round ends with
    LBFGSERR_MAXIMUMITERATION, so the multiplier / penalty update of moma_traj_opt.cpp:451-459 runs between rounds), or,
    with a small work budget, an exit through the deterministic stand-in for the reference's 1 s clock (403-407).
    Counters and final (lambda, rho) must equal the oracle's exactly, the iterate to 1e-7."""
    lib = opt_factory(None).L
    p = api.default_params(lib)
    p.s2_lbfgs.max_iterations = 5
    p.alm_max_outer = 3
    p.alm_work_budget = budget
    opt = opt_factory(p)
    set_map(opt, cs["world"])
    lens = cs["lens"][sel]
    paths = np.concatenate([cs["paths"][cs["offs"][i]:cs["offs"][i + 1]] for i in sel])
    opt.optimizeTraj(lens, paths)
    st = opt.stats()
    alm = opt.alm_state()
    rounds = []
    for k, b in enumerate(sel):
        o = orc.Oracle(cs["map"])
        o.set_param("s2_max_iterations", 5)
        o.set_param("alm_max_outer", 3)
        o.set_param("alm_work_budget", budget)
        o.set_init_traj(cs["paths"][cs["offs"][b]:cs["offs"][b + 1]])
        ok = o.optimize()
        so = o.stats()
        assert list(st[k]) == [so[key] for key in api.STAT_KEYS], (k, list(st[k]), so)
        assert np.allclose(opt.get_x(k), o.get_x(), rtol=1e-7, atol=1e-8)
        a = o.alm_state()
        # rho is a product of exact constants; lambda accumulates rho * final_xy_error of each round (1e-7 like the iterate)
        assert (alm[k, 2:] == a[2:]).all(), (alm[k], a)
        assert np.allclose(alm[k, :2], a[:2], rtol=1e-6, atol=1e-9), (alm[k], a)
        assert abs(opt.traj_cost[k] - o.traj_cost()) <= 1e-7 * abs(o.traj_cost())
        rounds.append(so["alm_outer"])
    return rounds


def test_alm_rounds_match_oracle(cuboids_small):
    mk = lambda p: api.MomaTrajOptBatch(params=p, lib_path=EMU_LIB)
    rounds = alm_rounds_case(mk, cuboids_small, [0, 1, 5], 24000)
    assert rounds == [3, 3, 3]                      # lambda / rho were updated twice and a third round ran with them
    rounds = alm_rounds_case(mk, cuboids_small, [0, 1], 60)
    assert max(rounds) < 3                          # the work budget ended the loop, in both implementations alike


def _zigzag_path(n_legs, leg=1.3):
    """Synthetic long init path (inside the 20 x 20 m map) whose time allocation needs many pieces."""
    pts = [np.array([-8.5, -8.5])]
    for k in range(n_legs):
        step = np.array([leg, 0.0]) if k % 2 == 0 else np.array([0.0, leg * (1 if (k // 2) % 2 == 0 else -1)])
        nxt = pts[-1] + step
        if abs(nxt[0]) > 8.8:
            break
        pts.append(nxt)
    states = []
    for k, p in enumerate(pts):
        th = 0.0 if k == 0 else np.arctan2(*(pts[k] - pts[k - 1])[::-1])
        q = np.linspace(0.2, -0.3, 7) * (k / max(1, len(pts) - 1))
        states.append(np.concatenate([p, [th], q]))
    return np.array(states)


def test_three_rows_per_lane_class(cuboids_small):
    """N in 22..32 uses three system rows per lane."""
    cs = cuboids_small
    mid = _zigzag_path(17)
    opt = api.MomaTrajOptBatch(lib_path=EMU_LIB)
    set_map(opt, cs["world"])
    short = cs["paths"][cs["offs"][0]:cs["offs"][1]]
    lens = np.array([len(mid), len(short)], dtype=np.int32)
    opt.set_init_traj(lens, np.concatenate([mid, short]))
    N = opt.n_pieces()
    o = orc.Oracle(cs["map"])
    o.set_init_traj(mid)
    assert N[0] == o.N and 22 <= N[0] <= 32, N
    assert np.allclose(opt.get_x(0), o.get_x(), atol=1e-12)
    x = o.get_x() + 0.03 * np.random.default_rng(5).standard_normal(o.n)
    for stage in (1, 2):
        o.set_alm([0.1, 0.2], [1e4, 1e4])
        f, g = o.eval(stage, x)
        fe, ge, _ = opt.eval(stage, 0, x, [0.1, 0.2], [1e4, 1e4])
        assert abs(f - fe) <= 1e-12 * abs(f) and np.abs(g - ge).max() <= 1e-11 * np.abs(g).max()


def test_long_classes_and_too_long_paths(cuboids_small):
    """N in 33..64 (the reference has no cap, moma_traj_opt.cpp:245, 300-321) runs in the classes of long candidates
    (several waves per trajectory by default, tests/test_multiwave.py): packed initial guess and per-evaluation parity at
    N = 33, 48 and 64, three kinds of points.  Beyond 170 pieces the candidate is reported as failed without a solve."""
    cs = cuboids_small
    paths = [serpentine_path(L) for L in (34.0, 50.0, 66.0, 178.0)]
    opt = api.MomaTrajOptBatch(lib_path=EMU_LIB)
    set_map(opt, cs["world"])
    opt.set_init_traj(np.array([len(p) for p in paths], dtype=np.int32), np.concatenate(paths))
    N = opt.n_pieces()
    assert list(N) == [33, 48, 64, 0]
    rng = np.random.default_rng(11)
    for k in range(3):
        o = orc.Oracle(cs["map"])
        n = o.set_init_traj(paths[k])
        Nk = o.N
        assert Nk == N[k] and np.allclose(opt.get_x(k), o.get_x(), rtol=0, atol=1e-12)
        for trial in range(3):
            x = o.get_x().copy()
            if trial == 1:
                x += 0.03 * rng.standard_normal(n)
            if trial == 2:  # rare paths: joint velocity/acceleration limits, mean-time band, folded arm
                x[:Nk] -= 1.6
                x[Nk - 1] += 2.5
                x[3 * Nk - 1:] += np.tile([0.0, 1.5, 0.0, 2.4, 0.0, 1.9, 0.0], Nk - 1)
            for stage in (1, 2):
                o.set_alm([0.1, 0.2], [1e4, 3e4])
                f, g = o.eval(stage, x)
                fe, ge, _ = opt.eval(stage, k, x, [0.1, 0.2], [1e4, 3e4])
                assert abs(f - fe) <= 1e-12 * abs(f) and np.abs(g - ge).max() <= 1e-11 * np.abs(g).max(), (Nk, trial, stage)
    o_long = orc.Oracle(cs["map"])
    o_long.set_init_traj(paths[3])
    assert o_long.N > 170
    with pytest.raises(api.TopayError):
        opt.get_x(3)
    r = opt.getTraj(3)
    assert r["success"] is False and len(r["durations"]) == 0


def test_edge_case_inputs_follow_the_oracle():
    """Degenerate inputs the reference accepts: a two-state hop, a rotation in place, identical start and goal (no inner
    point: one piece, stage 1 stops with an error code and the candidate is reported failed) and a 24 m path.  Piece
    counts, the packed initial guess, the stage-1 counters (not chaotic) and the success flags agree with the oracle."""
    w = wl.World(wl.CUBOIDS, seed=42)
    m = orc.MapView(w.origin, w.res, w.dims, w.min_b, w.max_b, w.esdf2d, w.esdf3d)
    ok, s, g = w.sample_scenario(42)
    assert ok
    hop = s.copy(); hop[0] += 0.6; hop[1] += 0.2
    turn = s.copy(); turn[2] += 1.5
    a, b = s.copy(), s.copy()
    a[0] = a[1] = -8.5
    b[0] = b[1] = 8.5
    far = np.stack([a + (b - a) * t for t in np.linspace(0, 1, 40)])
    far[:, 2] = np.arctan2(1.0, 1.0)
    cases = [np.stack([s, hop]), np.stack([s, turn]), np.stack([s, s.copy()])]
    lens = np.array([len(c) for c in cases], dtype=np.int32)
    paths = np.concatenate(cases)
    emu = api.MomaTrajOptBatch(lib_path=EMU_LIB)
    set_map(emu, w)
    okv = emu.optimizeTraj(lens, paths)
    st = emu.stats()
    Ns = emu.n_pieces()
    for k, c in enumerate(cases):
        o = orc.Oracle(m)
        o.set_init_traj(c)
        r = o.optimize()
        so = o.stats()
        assert Ns[k] == o.N
        assert list(st[k, :3]) == [so["stage1_ret"], so["stage1_iters"], so["stage1_evals"]]
        assert bool(okv[k]) == bool(r)
    assert okv[0] and okv[1] and not okv[2] and Ns[2] == 1 and np.isnan(emu.traj_cost[2])
    # the long path: initialisation only (a 23-piece solve takes minutes in the lane emulator; the GPU tests solve such)
    emu.set_init_traj(np.array([len(far)], dtype=np.int32), far)
    o = orc.Oracle(m)
    o.set_init_traj(far)
    assert emu.n_pieces()[0] == o.N == 23
    assert np.allclose(emu.get_x(0), o.get_x(), rtol=0, atol=1e-12)
    w.close()


def test_optimizer_yaml_parameters_are_wired_not_hard_coded(cuboids_small):
    """Every weight of optimizer.yaml that the oracle lets a test change is changed (one at a time, by an odd factor) in
    both implementations: the kernel's cost and gradient must follow the oracle's, and must actually move."""
    cs = cuboids_small
    b = 3
    path = cs["paths"][cs["offs"][b]:cs["offs"][b + 1]]
    names = ["relu_mu", "s1_time_weight", "s1_moment_weight", "s1_acc_weight", "s1_domega_weight", "s1_path_pos_weight",
             "s2_time_weight", "s2_moment_weight", "s2_acc_weight", "s2_domega_weight", "s2_collision_weight",
             "s2_mani_colli_weight", "s2_self_colli_weight", "s2_mani_pos_weight", "s2_mani_vel_weight", "s2_mani_acc_weight",
             "s2_mean_time_weight"]
    base = api.default_params()
    o0 = orc.Oracle(cs["map"])
    n = o0.set_init_traj(path)
    N = o0.N
    x = o0.get_x().copy()
    x[:N] -= 1.6                       # short durations: velocity / acceleration / joint-rate penalties are active
    x[N - 1] += 2.5                    # and one long piece: the mean-time band
    x[3 * N - 1:] += np.tile([0.0, 1.5, 0.0, 2.4, 0.0, 1.9, 0.0], N - 1)   # folded arm: joint limits, self collision
    lam, rho = [0.3, -0.2], [1e4, 2e4]
    o0.set_alm(lam, rho)
    f_base = {s: o0.eval(s, x)[0] for s in (1, 2)}
    moved = 0
    for name in names:
        p = api.default_params()
        setattr(p, name, getattr(base, name) * 1.37)
        emu = api.MomaTrajOptBatch(params=p, lib_path=EMU_LIB)
        set_map(emu, cs["world"])
        emu.set_init_traj(cs["lens"][b:b + 1], path)
        o = orc.Oracle(cs["map"])
        o.set_param(name, getattr(base, name) * 1.37)
        o.set_init_traj(path)
        o.set_alm(lam, rho)
        stage = 1 if name.startswith("s1_") else 2
        f, g = o.eval(stage, x)
        fe, ge, _ = emu.eval(stage, 0, x, lam, rho)
        assert abs(f - fe) <= 1e-11 * abs(f) and np.abs(g - ge).max() <= 1e-10 * np.abs(g).max(), name
        if abs(f - f_base[stage]) > 1e-9 * abs(f):
            moved += 1
    assert moved >= len(names) - 2     # at this point nearly every term is active


def test_persistent_queue_loop_in_the_emulator(cuboids_small, monkeypatch):
    """The emulated device has 8 SIMD slots (tests/emu): 18 candidates of mixed classes are drained by workgroups that
    loop over their class's queue; the capped solves must equal the one-workgroup-per-candidate launch bit for bit."""
    cs = cuboids_small
    sel = [0, 1, 2, 3, 4, 5] * 3
    lens = cs["lens"][sel]
    paths = np.concatenate([cs["paths"][cs["offs"][i]:cs["offs"][i + 1]] for i in sel])

    def solve():
        p = api.default_params(api.load(EMU_LIB))
        p.s1_lbfgs.max_iterations = 6
        p.s2_lbfgs.max_iterations = 4
        p.alm_max_outer = 1
        opt = api.MomaTrajOptBatch(params=p, lib_path=EMU_LIB)
        set_map(opt, cs["world"])
        opt.optimizeTraj(lens, paths)
        return opt.stats().copy(), opt.traj_cost.copy(), [opt.get_x(k) for k in range(len(sel))]

    monkeypatch.setenv("TOPAY_PERSISTENT", "0")
    st0, c0, x0 = solve()
    monkeypatch.setenv("TOPAY_PERSISTENT", "1")
    st1, c1, x1 = solve()
    assert (st0 == st1).all() and (c0 == c1).all()
    for a, b in zip(x0, x1):
        assert (a == b).all()
    # the three copies of every candidate are identical too (a workgroup's LDS state does not leak into its next candidate)
    for k in range(6):
        assert (x1[k] == x1[k + 6]).all() and (x1[k] == x1[k + 12]).all() and c1[k] == c1[k + 6] == c1[k + 12]


def test_converged_solve_equals_oracle_solver_in_device_order(cuboids_small):
    """Converged parity against independent solver code.  Converged values of two implementations cannot be compared
    through the reference-order arithmetic (rounding differences are amplified).  But the solve is
    two layers: the evaluation (checked per call against the oracle, 1e-12) and the solver logic around it -- L-BFGS,
    Lewis-Overton line search, stop tests, ALM updates.  Here the ORACLE's solver logic (its restatement of lbfgs.hpp and
    of optimizeTraj:359-497, not the kernel sources) runs with its vector arithmetic in the device's summation order and
    takes cost / gradient from the evaluation hook of the kernel sources: the full solve to convergence must come out
    bit for bit -- iterate, cost, multipliers, every counter."""
    cs = cuboids_small
    lens, paths = cs["lens"][:1], cs["paths"][:cs["offs"][1]]
    emu = api.MomaTrajOptBatch(lib_path=EMU_LIB)
    set_map(emu, cs["world"])
    ok = emu.optimizeTraj(lens, paths)
    st, x, cost, alm = emu.stats()[0], emu.get_x(0), emu.traj_cost[0], emu.alm_state()[0]
    ev = api.MomaTrajOptBatch(lib_path=EMU_LIB)
    set_map(ev, cs["world"])
    ev.set_init_traj(lens, paths)
    o = orc.Oracle(cs["map"])
    o.set_init_traj(paths)
    okh = o.optimize_device_order(lambda stage, xx, lam, rho: ev.eval(stage, 0, xx, lam, rho))
    so = o.stats()
    assert okh == bool(ok[0]) and st[4] > 100          # a real solve: more than a hundred stage-2 iterations
    assert [so["stage1_ret"], so["stage1_iters"], so["stage1_evals"], so["stage2_last_ret"], so["stage2_iters"], so["stage2_evals"],
            so["alm_outer"], so["sum_bound"]] == list(st)
    assert (o.get_x() == x).all() and o.traj_cost() == cost and (o.alm_state() == alm).all()


def test_self_colliding_arm_poses_match_oracle(emu, cuboids_small):
    """The sphere-pair path of the manipulator block (moma_traj_opt.cpp:1566-1611).  Since round 4 the pair forces of a
    self-colliding sample do not live in registers beside the sphere centres: the lanes that have any accumulate them in
    an HBM block and every sphere's own terms start from there.  Strongly perturbed joints fold the arm onto itself in
    most samples (the self-collision term dominates the cost); cost and gradient against the oracle."""
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
            t = o.debug_terms()
            hit += t["self_colli"] > 1e3
            fe, ge, _ = emu.eval(2, b, x, [0.1, -0.2], [1e4, 2e4])
            assert abs(f - fe) <= 1e-12 * abs(f)
            assert np.abs(g - ge).max() <= 1e-11 * np.abs(g).max()
    assert hit >= 6
