"""Cancellation of a planning call's candidates (planner.cpp:829-952: 100 ms after the first candidate that succeeds and
passes the gate, threads.interrupt_all(); interruption points moma_traj_opt.cpp:402, 887) on the deterministic work
clock of piece-evaluations, and topay_cancel (interrupt everything now).

The rule itself is restated in the oracle (oracle/oracle_capi.cpp: orc_group_cancel).  What is asserted:
  * the device's interrupted set == the oracle's rule applied to the candidates' own clocks and verdicts (those of the
    same batch run without cancellation; on the GPU additionally those of the oracle's solver logic in device order, so
    that clocks, success flags and gate verdicts all come from oracle code);
  * a candidate that is not interrupted returns bit for bit what it returns without cancellation;
  * the outcome is the same on every run (it does not depend on which candidate the device happened to run first).
"""
import numpy as np
import pytest

from conftest import EMU_LIB, set_map
from oracle import oracle as orc
from topay_amd import api
from harness import workload as wl


LATENCY_MODE = 0   # (set by the helper-wave variant of the CPU test)


def _run(lib, world, lens, paths, params, groups=None, budget=0):
    opt = api.MomaTrajOptBatch(params=params, device=0, lib_path=lib)
    set_map(opt, world)
    opt.set_init_traj(lens, paths)
    opt.set_latency_mode(LATENCY_MODE)
    if groups is not None:
        opt.set_groups(groups, budget)
    ok = opt.optimize()
    gate = opt.check_feasible()
    out = dict(ok=ok.copy(), gate=gate.copy(), stats=opt.stats().copy(), cost=opt.traj_cost.copy(), N=opt.n_pieces().copy(),
               intr=opt.interrupted().copy(), x=[opt.get_x(b) for b in range(len(lens))])
    return opt, out


def _check_rule(lib, world, lens, paths, params, groups):
    """Run without cancellation, derive the clocks, pick a budget that cuts the group roughly in half, run with it."""
    _, base = _run(lib, world, lens, paths, params)
    clock = (base["stats"][:, 2] + base["stats"][:, 5]).astype(np.int64) * base["N"]
    acc = base["ok"] & base["gate"]
    assert acc.any(), "the case needs at least one accepted candidate"
    res = []
    for budget in (int(np.median(clock[groups >= 0]) - clock[acc].min()) // 2 + 1, 10 ** 8):
        budget = max(1, budget)
        want = orc.group_cancel(groups, clock, acc, budget)
        for rep in range(2):
            _, got = _run(lib, world, lens, paths, params, groups, budget)
            assert (got["intr"] == want).all(), (budget, got["intr"], want)
            assert not got["ok"][want].any() and not got["gate"][want].any()
            assert (got["stats"][want, 3] == -2000).all()
            keep = ~want
            assert (got["ok"][keep] == base["ok"][keep]).all() and (got["gate"][keep] == base["gate"][keep]).all()
            assert (got["stats"][keep] == base["stats"][keep]).all()
            assert ((got["cost"][keep] == base["cost"][keep]) | np.isnan(base["cost"][keep])).all()
            for b in np.nonzero(keep)[0]:
                assert (got["x"][b] == base["x"][b]).all()
        res.append((budget, int(want.sum())))
    assert res[0][1] > 0 and res[1][1] == 0, res       # the tight window interrupts somebody, the wide one nobody
    return base, clock, acc


def test_cancellation_rule_on_cpu(cuboids_small):
    """Kernel sources in the lane emulator, capped solves (the emulator is slow): two planning calls of three candidates."""
    cs = cuboids_small
    p = api.default_params(api.load(EMU_LIB))
    p.s2_lbfgs.max_iterations = 25
    p.alm_max_outer = 2
    p.alm_tolerance = 10.0            # every capped solve "succeeds": the rule needs accepted candidates
    groups = np.array([0, 0, 0, 1, 1, -1], dtype=np.int32)
    _check_rule(EMU_LIB, cs["world"], cs["lens"], cs["paths"], p, groups)


def test_cancellation_rule_with_helper_waves_on_cpu(cuboids_small, monkeypatch):
    """The same through the helper-wave kernels (topay_set_latency_mode 2): wave 0 takes the interruption and releases the others."""
    import sys
    monkeypatch.setattr(sys.modules[__name__], "LATENCY_MODE", 2)
    test_cancellation_rule_on_cpu(cuboids_small)


def test_cancellation_rule_with_a_two_wave_candidate_on_cpu(cuboids_small):
    """The same rule when one candidate of the call runs on two waves (33 pieces: the verdict of the poll travels to the
    second wave through LDS)."""
    from conftest import serpentine_path

    cs = cuboids_small
    p = api.default_params(api.load(EMU_LIB))
    p.s1_lbfgs.max_iterations = 40
    p.s2_lbfgs.max_iterations = 25
    p.alm_max_outer = 2
    p.alm_tolerance = 10.0
    long_ = serpentine_path(34.0)
    sel = [cs["paths"][cs["offs"][b]:cs["offs"][b + 1]] for b in (0, 3)] + [long_]
    lens = np.array([len(q) for q in sel], dtype=np.int32)
    paths = np.concatenate(sel)
    groups = np.array([0, 0, 0], dtype=np.int32)
    base, clock, acc = _check_rule(EMU_LIB, cs["world"], lens, paths, p, groups)
    assert base["N"][2] == 33


def test_oracle_rule_known_answers():
    g = np.array([0, 0, 0, 1, 1, -1, 2])
    clock = np.array([100, 350, 351, 50, 5000, 10 ** 9, 70])
    acc = np.array([1, 0, 1, 0, 0, 1, 0])
    # call 0: first accepted finish at 100 -> 350 stays, 351 is interrupted; call 1 and 2: nobody succeeds, nobody is interrupted
    assert list(orc.group_cancel(g, clock, acc, 250)) == [False, False, True, False, False, False, False]
    assert list(orc.group_cancel(g, clock, acc, 0)) == [False, True, True, False, False, False, False]


@pytest.mark.gpu
def test_cancellation_rule_on_gpu_matches_oracle():
    """One tables scenario x 16 candidates = one planning call, full solves.  The expected set comes from oracle code only:
    the oracle's solver logic in device order (bit-identical counters), its own gate on its own final trajectory."""
    world, start, goal, lens, paths = wl.tables_scenario(5, 16)
    offs = np.concatenate([[0], np.cumsum(lens)])
    groups = np.zeros(len(lens), dtype=np.int32)
    p = api.default_params()
    base, clock, acc = _check_rule(None, world, lens, paths, p, groups)
    m = orc.MapView(world.origin, world.res, world.dims, world.min_b, world.max_b, world.esdf2d, world.esdf3d)
    ev = api.MomaTrajOptBatch(device=0)
    set_map(ev, world)
    ev.set_init_traj(lens, paths)
    oclock, oacc = [], []
    for b in range(len(lens)):
        o = orc.Oracle(m)
        o.set_init_traj(paths[offs[b]:offs[b + 1]])
        okh = o.optimize_device_order(lambda stage, xx, lam, rho: ev.eval(stage, b, xx, lam, rho), epl=ev.class_of(o.N)[1], nw=ev.class_of(o.N)[0])
        so = o.stats()
        oclock.append((so["stage1_evals"] + so["stage2_evals"]) * o.N)
        # (in device-order mode the oracle's spline is not the one of its final iterate -- the evaluations came from the
        # device: load the final x with the final multipliers first, then gate it with the oracle's own playback)
        a_ = o.alm_state()
        o.set_alm(a_[:2], a_[2:])
        o.eval(2, o.get_x())
        oacc.append(bool(okh) and o.check_feasible()[0])
    assert list(oclock) == list(clock), (oclock, list(clock))
    assert list(oacc) == list(acc), (oacc, list(acc), list(base["ok"]), list(base["gate"]))
    budget = 2400                      # the reference's 100 ms
    _, got = _run(None, world, lens, paths, p, groups, budget)
    assert (got["intr"] == orc.group_cancel(groups, oclock, oacc, budget)).all()
    print(f"16 candidates of one call: {int(got['intr'].sum())} interrupted at the reference's window, "
          f"{int((got['ok'] & got['gate']).sum())} accepted, clocks {sorted(int(c) for c in clock)}")
    world.close()


@pytest.mark.gpu
def test_cancel_everything_in_flight():
    """topay_cancel == threads.interrupt_all(): the solve in flight ends early, every candidate that had not finished is
    interrupted (no trajectory, success 0), and the context solves normally afterwards."""
    import time

    tb = wl.TablesBatch(64, 8, base_seed=99, nthreads=8)
    slot = {s_: k for k, s_ in enumerate(tb.scenarios)}
    map_ids = np.array([slot[s_] for s_ in tb.scen], dtype=np.int32)
    opt = api.MomaTrajOptBatch(device=0)
    for s_ in tb.scenarios:
        set_map(opt, tb.world(s_), map_id=slot[s_])
    opt.set_init_traj(tb.lens, tb.paths, map_ids=map_ids)
    t0 = time.perf_counter()
    ok_full = opt.optimize()
    t_full = time.perf_counter() - t0
    st_full = opt.stats().copy()
    opt.reset()
    opt.optimize_async()
    time.sleep(0.02)
    opt.cancel()
    t0 = time.perf_counter()
    ok = opt.finish()
    t_cancel = time.perf_counter() - t0
    intr = opt.interrupted()
    assert intr.mean() > 0.5 and not ok[intr].any() and (opt.stats()[intr, 3] == -2000).all()
    assert t_cancel < 0.5 * t_full, (t_cancel, t_full)
    opt.reset()
    ok2 = opt.optimize()
    assert (ok2 == ok_full).all() and (opt.stats() == st_full).all() and not opt.interrupted().any()
    tb.close()


@pytest.mark.gpu
def test_wall_clock_budget_interrupts_what_has_not_finished():
    """topay_optimize_within: the wall-clock knob of a planning call (max_replan_time; the reference's 1.0 s cap of the ALM
    loop is of the same kind).  A budget far below the batch's solve time interrupts most candidates -- they return
    TOPAY_INTERRUPTED and no success -- and whoever finished in time keeps the result of the unbudgeted solve bit for bit; a
    generous budget changes nothing."""
    tb = wl.TablesBatch(64, 8, base_seed=42, nthreads=8)
    gpu = api.MomaTrajOptBatch(device=0)
    worlds = [tb.world(s_) for s_ in tb.scenarios]
    w0 = worlds[0]
    gpu.build_esdf_batch(w0.origin, w0.res, w0.dims, w0.min_b, w0.max_b, np.stack([w.occ2d for w in worlds]), np.stack([w.occ3d for w in worlds]))
    slot = {s_: k for k, s_ in enumerate(tb.scenarios)}
    map_ids = np.array([slot[s_] for s_ in tb.scen], dtype=np.int32)
    ok_full = gpu.optimizeTraj(tb.lens, tb.paths, map_ids=map_ids)
    cost_full = gpu.traj_cost.copy()
    ms_full = gpu.last_kernel_ms()[0]
    gpu.reset()
    ok_gen, timed_out = gpu.optimize_within(60000.0)
    assert not timed_out and (ok_gen == ok_full).all() and np.array_equal(np.nan_to_num(gpu.traj_cost), np.nan_to_num(cost_full))
    assert not gpu.interrupted().any()
    gpu.reset()
    # (a tenth of the full solve: the full time is the longest candidate's, and the faster the long candidates got relative to
    # the rest -- the two-loop recursion of round 5 -- the fewer are still running at a fixed fraction of it: 0.15 left 29.9 %)
    ok_short, timed_out = gpu.optimize_within(max(1.0, 0.10 * ms_full))
    intr = gpu.interrupted().astype(bool)
    print(f"full solve {ms_full:.0f} ms; budget {0.10 * ms_full:.0f} ms: {int(intr.sum())} of {len(intr)} interrupted, {int(ok_short.sum())} succeeded in time")
    assert timed_out and intr.sum() > 0.3 * len(intr)
    assert not (ok_short & intr).any()
    done = ok_short & ~intr
    assert (ok_full[done]).all() and np.array_equal(gpu.traj_cost[done], cost_full[done])
    st = gpu.stats()
    assert (st[intr, 3] == -2000).all()
    tb.close()


@pytest.mark.gpu
def test_wall_clock_budget_does_not_start_what_is_still_queued():
    """A batch several times larger than the resident grid: when the deadline of topay_optimize_within passes, the candidates
    still in the queues must not be begun at all (they used to be dequeued and run through the whole of stage 1
    one after the other, so the overshoot grew with the queue).  They come back interrupted with no evaluation counted, and
    the call returns within a small multiple of the budget."""
    import time

    tb = wl.TablesBatch(768, 8, base_seed=7, nthreads=8)
    gpu = api.MomaTrajOptBatch(device=0)
    worlds = [tb.world(s_) for s_ in tb.scenarios]
    w0 = worlds[0]
    gpu.build_esdf_batch(w0.origin, w0.res, w0.dims, w0.min_b, w0.max_b, np.stack([w.occ2d for w in worlds]), np.stack([w.occ3d for w in worlds]))
    slot = {s_: k for k, s_ in enumerate(tb.scenarios)}
    map_ids = np.array([slot[s_] for s_ in tb.scen], dtype=np.int32)
    ok_full = gpu.optimizeTraj(tb.lens, tb.paths, map_ids=map_ids)
    ms_full = gpu.last_kernel_ms()[0]
    assert len(ok_full) >= 3 * 1024                 # (more candidates than SIMD slots: most of them wait in the queues at first)
    gpu.reset()
    budget = max(2.0, 0.05 * ms_full)
    t0 = time.perf_counter()
    ok_short, timed_out = gpu.optimize_within(budget)
    wall_ms = (time.perf_counter() - t0) * 1e3
    intr = gpu.interrupted().astype(bool)
    st = gpu.stats()
    never = intr & (st[:, 2] == 0) & (st[:, 5] == 0)
    print(f"full solve {ms_full:.0f} ms; budget {budget:.0f} ms -> returned after {wall_ms:.0f} ms; {int(intr.sum())} of {len(intr)} interrupted, "
          f"{int(never.sum())} of them never begun")
    assert timed_out and never.sum() > 0.3 * len(intr)
    assert (st[intr, 3] == -2000).all() and not ok_short[intr].any()
    assert np.isnan(gpu.traj_cost[never]).all()
    # what was resident at the deadline finishes its stage 1 (no interruption point there) and one stage-2 iteration
    assert wall_ms < budget + 0.35 * ms_full, (wall_ms, budget, ms_full)
    gpu.reset()
    ok2 = gpu.optimize()
    assert (ok2 == ok_full).all() and not gpu.interrupted().any()
    tb.close()
