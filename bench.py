#!/usr/bin/env python3
"""Headline benchmark: candidate trajectories optimised to L-BFGS convergence per second (BASELINE.json metric).

Workload ("benchmark_tables batch"): S independent 'tables' scenarios per GPU, each with its own freshly generated map
(the reference's benchmark loop regenerates the map every episode, src/planner/src/planner.cpp:514-521) and 8 topological
candidate init paths (the reference's cap, planner.cpp:59,829).  One step = the whole hot path over the batch with the
maps resident in HBM: upload of the raw init paths + the init kernel (optimizeTraj lines 146-357) + the persistent solve
kernels (lines 359-497: stage-1 L-BFGS, stage-2 ALM loop; the solving wave also applies the feasibility gate of
planner.cpp:878-880 to its result) + download of
flags, costs and durations, the per-scenario winner selection (planner.cpp:999-1016) and download of the winners'
trajectories + the gather of per-scenario result records (RCCL all-gather when N > 1).  Three batches are kept in
flight by default (--inflight); the strictly serial figures are measured in the same run and reported beside the line.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \
        bench.py --gpus 8 --steps 3 --warmup 1

Scaling is weak: every rank owns S scenarios (different seeds); no collective inside the solve.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

# before anything initialises HIP (torch.cuda does): the solver's bucket streams need distinct hardware queues
# (three batches in flight use 18; with 16 in all the RCCL gather's stream shares a queue with a solve launch and the
# multi-GPU path runs at half speed -- tools/ab_dist.sh: 4.5k against 9.3k trajectories/s with 20 or 24)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
# A sample pass with n < 64 active lanes is cheaper than a full one -- its gathers and its penalty branches scale with the
# lanes that are there.  Measured on the evaluation kernel alone (tools/gpu_eval_by_n.py, round 5: N = 4..16, 2048 candidates
# each): cost(n) / cost(64) = 0.42 + 0.58 n / 64.  Feeds lane_utilisation_effective.
PASS_COST_FIXED = 0.42


def algorithmic_bytes(stats, n_pieces):
    """Algorithmic bytes of one solve of the batch (no cache credit), DESIGN.md "Roofline accounting":
       stage-2 evaluation of N pieces  : 13N x (4 + 12*8) ESDF doubles + coefficients/gradient in/out = 11 280 N B
       stage-1 evaluation              : coefficients + T in, gdC + gdT out                          =    880 N B
       accepted L-BFGS iteration       : two-loop reads 4 x bound x n x 8 B (bound summed by the solver) + history
                                         write 2n x 8 B + x, g, d, xp, gp traffic ~ 10n x 8 B
    """
    N = n_pieces.astype(np.float64)
    n = 10.0 * N - 8.0
    ev1, ev2 = stats[:, 2].astype(np.float64), stats[:, 5].astype(np.float64)
    it = (stats[:, 1] + stats[:, 4]).astype(np.float64)
    sumb = stats[:, 7].astype(np.float64)
    b = ev2 * 11280.0 * N + ev1 * 880.0 * N + sumb * 4.0 * n * 8.0 + it * 12.0 * n * 8.0
    return float(b.sum())


def host_cores():
    """CPUs this process may actually use: affinity mask capped by the cgroup CPU quota (the GPU boxes expose all 256
    hardware threads but grant a 16-CPU quota; oversubscribing it only adds throttling)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def main():
    # stdout carries the one JSON line and nothing else: whatever libraries print there meanwhile (RCCL's version banner
    # when the first communicator is created) goes to stderr
    sys.stdout.flush()
    _real_stdout = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scenarios", type=int, default=1024, help="scenarios per GPU (x 8 candidates each)")
    ap.add_argument("--candidates", type=int, default=8)
    ap.add_argument("--cpu-sample", type=int, default=1536, help="trajectories of the batch timed on the host cores")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-serial", action="store_true", help="skip the strictly serial steps measured beside a pipelined run")
    ap.add_argument("--own-maps", action="store_true", help="every context in flight keeps its own copy of the maps (default: shared, topay_share_maps)")
    ap.add_argument("--front-end", action="store_true",
                    help="take the init paths from the device's layered joint-space search (topay_mcrrt_plan == MCRRTs::plan along the "
                         "batch's chassis paths, untimed set-up) instead of the generator's straight joint interpolation; candidates whose "
                         "search finds no path keep the generator's.  The workload then differs from the default line's and says so")
    ap.add_argument("--no-planner", action="store_true",
                    help="skip the planner-semantics figure (the same steps with the reference's cancellation of a planning call's "
                         "remaining candidates 100 ms after its first accepted one, planner.cpp:943-952), measured beside the line")
    ap.add_argument("--no-config1", action="store_true", help="skip the configs[1] latency figure (profiling runs)")
    ap.add_argument("--workload", choices=["tables", "hires"], default="tables",
                    help="tables: the headline benchmark_tables batch; hires: BASELINE config 5, ONE cuboids map at 0.02 m "
                         "voxels (--hires-size metres square; 50 = the 4 GB 3-D ESDF) shared by all scenarios")
    ap.add_argument("--hires-size", type=float, default=50.0)
    ap.add_argument("--inflight", type=int, default=None,
                    help="batches (contexts) in flight per GPU (default 3): step i+1 is issued while step i is still solving and its waves "
                         "take the SIMDs step i's tail leaves idle; 1 = strictly serial steps (also measured and reported beside the line)")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    # TOPAY_FORCE_DIST=1: take the multi-rank code path (RCCL init, record gather, reductions) with a single rank too,
    # so that it can be validated on a one-GPU box
    distributed = world > 1 or (os.environ.get("TOPAY_FORCE_DIST") == "1" and "RANK" in os.environ)
    # (more ranks than devices: the ranks share the devices there are -- how the rank-indexed code is exercised with
    # WORLD_SIZE 2 on a one-GPU box, tests/test_sharding.py; on a full node local_rank < device_count and nothing changes)
    dev_index = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # TOPAY_DIST_BACKEND=gloo: the record exchange over gloo with host tensors (two ranks on ONE device cannot form an RCCL
    # communicator); default nccl == RCCL on ROCm
    backend = os.environ.get("TOPAY_DIST_BACKEND", "nccl")
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from topay_amd import api, dist as tdist
    from harness import workload as wl

    # ---- synthetic inputs (CPU harness, untimed): S scenarios of this rank, one map each
    S, Ccand = args.scenarios, args.candidates
    t0 = time.time()
    hires = args.workload == "hires"
    if hires:
        # BASELINE config 5: one high-resolution map, many scenarios on it (the 3-D ESDF no longer fits L2 + Infinity Cache)
        class _OneMap:  # same fields as TablesBatch
            pass
        tb = _OneMap()
        hw, tb.lens, tb.paths, tb.scen = wl.cuboids_batch(S, Ccand, map_seed=42 + rank, base_seed=42 + rank * 100000,
                                                          size_xy=args.hires_size, size_z=1.6, res=0.02, cloud_res=0.02)
        tb.scenarios = [0]
        tb.world = lambda s_: hw
        tb_map_of = np.zeros(len(tb.lens), dtype=np.int32)
    else:
        # host side of the generator: as many threads as this rank may use, and only the scenarios the CPU baseline
        # solves keep their CPU-built 3-D distance field (the device builds its own from the occupancy grids): 1 GB
        # instead of 5.5 GB of host memory per rank
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", "1")) if distributed else 1
        gen_threads = max(1, host_cores() // max(1, local_world))
        keep = None
        if distributed or args.no_cpu_baseline:
            keep = 0
        else:
            keep = min(S, (args.cpu_sample + Ccand - 1) // Ccand + 32)
        tb = wl.TablesBatch(S, Ccand, base_seed=42 + rank * 100000, nthreads=gen_threads, keep_esdf3d=keep)
    B = len(tb.lens)
    # `--inflight` contexts hold the same batch (in a sweep they would hold consecutive batches): step i runs on context
    # i mod inflight, so the tail of one step -- a few long candidates, most SIMDs idle -- overlaps the bulk of the next.
    if args.inflight is None:
        args.inflight = 3   # measured on one box (tools/ab_inflight.sh): 2 -> 7.9k, 3 -> 8.7-8.9k, 4 -> 7.9k (8.6k with 24 hardware queues) trajectories/s
    depth = max(1, args.inflight)
    opts = []
    map_ids_of = {}
    edt_ms = None
    for _ in range(depth):
        o_ = api.MomaTrajOptBatch(device=dev_index)
        slot = {s: k for k, s in enumerate(tb.scenarios)}
        if opts and not args.own_maps:
            # the batches in flight share one resident copy of the maps (the reference's optimisers share one GridMap::Ptr)
            o_.share_maps(opts[0], 0, 1 if hires else len(tb.scenarios))
        elif hires:
            w = tb.world(0)
            o_.set_map(w.origin, w.res, w.dims, w.min_b, w.max_b, w.esdf2d, w.esdf3d, map_id=0)
        else:
            # the maps' distance fields are built on the GPU from the occupancy grids (GridMap::updateESDF, the step
            # before optimizeTraj in the benchmark loop; bit-identical to the CPU construction, tests/test_edt.py)
            worlds = [tb.world(s) for s in tb.scenarios]
            w0 = worlds[0]
            o_.build_esdf_batch(w0.origin, w0.res, w0.dims, w0.min_b, w0.max_b, np.stack([w.occ2d for w in worlds]),
                                np.stack([w.occ3d for w in worlds]))
            edt_ms = o_.get_map(0)[2]
        map_ids = tb_map_of if hires else np.array([slot[s] for s in tb.scen], dtype=np.int32)
        o_.set_init_traj(tb.lens, tb.paths, map_ids=map_ids)   # (sizes the workspace; every timed step uploads the paths again)
        map_ids_of[id(o_)] = map_ids
        opts.append(o_)
    opt = opts[0]
    front_end = None
    if args.front_end and not hires:
        # MCRRTs::plan for every candidate's chassis path (the dense path's (x, y, theta, dt)) on the device, outside the timed
        # region like the rest of the front-end; the whole-body paths replace the generator's joints
        offs_ = np.concatenate([[0], np.cumsum(tb.lens)])
        car_ = np.c_[tb.paths[:, :3], tb.dts]
        prm_ = opt.mcrrt_params(seed=42 + rank)
        tf0 = time.perf_counter()
        wbs_, mst_, _ = opt.mcrrt_plan(tb.lens, car_, tb.paths[offs_[:-1]], tb.paths[offs_[1:] - 1], prm_, map_ids=map_ids_of[id(opt)],
                                       first_instance=rank * S * Ccand)
        tf1 = time.perf_counter() - tf0
        fe_paths = tb.paths.copy()
        for k_ in np.nonzero(mst_[:, 0] == 1)[0]:
            fe_paths[offs_[k_]:offs_[k_ + 1]] = wbs_[k_]
        tb.paths = fe_paths
        for o_ in opts:
            o_.set_init_traj(tb.lens, tb.paths, map_ids=map_ids_of[id(o_)])
        front_end = {"searches": int(B), "ms": tf1 * 1e3, "searches_per_s": B / tf1, "found_fraction": float((mst_[:, 0] == 1).mean()),
                     "mean_iterations": float(mst_[:, 2].mean()), "mean_nodes": float(mst_[:, 1].mean()), "max_nodes": int(mst_[:, 1].max()),
                     "whole_body_checks": int(mst_[:, 7].astype(np.int64).sum()),
                     "note": "topay_mcrrt_plan, one wave per chassis path, including the copies of its inputs and results; untimed set-up"}
    n_pieces = opt.n_pieces()
    n_not_launched = int((n_pieces <= 0).sum())
    setup_s = time.time() - t0
    scen_global = tb.scen.astype(np.int64) + rank * S
    scen_ids = np.array(sorted(set(scen_global.tolist())), dtype=np.int64)

    cancel_budget = [0]                  # > 0: the planner's cancellation window, in piece-evaluations (2400 = 100 ms)

    def issue(o_):
        # optimizeTraj:146-357: host-to-device copy of the raw init paths + the init kernel, then the persistent solve
        # kernels (optimizeTraj:359-497).  The maps stay resident (the reference builds its map before optimizeTraj).
        o_.set_init_traj(tb.lens, tb.paths, map_ids=map_ids_of[id(o_)])
        if cancel_budget[0] > 0:
            o_.set_groups(tb.scen, cancel_budget[0])
        o_.optimize_async()

    gathers = []
    trace = os.environ.get("TOPAY_BENCH_TRACE") == "1"
    tr0 = time.perf_counter()

    winners = {}
    sent = []                            # local records of the gathers in flight, oldest first
    gather_stats = {"gathers": 0, "rows": 0, "own_rows_match": True}

    def check_gather(rows):
        # every rank finds its own records, bit for bit, in what came back over RCCL (with one rank: the whole result)
        mine = sent.pop(0)
        got = rows[np.isin(rows[:, 0], mine[:, 0])]
        got = got[np.argsort(got[:, 0], kind="stable")]
        ref = mine[np.argsort(mine[:, 0], kind="stable")]
        same = got.shape == ref.shape and bool(np.all((got == ref) | (np.isnan(got) & np.isnan(ref))))
        gather_stats["gathers"] += 1
        gather_stats["rows"] = int(rows.shape[0])
        gather_stats["distinct_ids"] = int(len(np.unique(rows[:, 0])))
        gather_stats["own_rows_match"] = gather_stats["own_rows_match"] and same

    def finish(o_):
        ta = time.perf_counter()
        ok = o_.finish()                 # waits for this context's stream; success flags and costs come back
        ms, _ = o_.last_kernel_ms()
        # What the planner does with a solved scenario (planner.cpp:878-885, 999-1016): a candidate counts when
        # optimizeTraj succeeded AND printConstraintsSituations passes; of those the shortest one is kept, and that
        # trajectory is what leaves the device (coefficients, durations, knots of the per-scenario winners).
        feas = o_.check_feasible()
        dur = o_.total_durations()
        recs, best = tdist.scenario_records(scen_ids, scen_global, (ok & feas).astype(np.int32), o_.traj_cost, n_pieces, dur,
                                            return_winners=True)
        winners["last"] = o_.getTrajs(best, n_pieces)
        winners["n"] = len(best)
        tb_ = time.perf_counter()
        if distributed:                  # the one exchange of the path: per-scenario result records over RCCL
            tc = time.perf_counter()
            # the exchange of step i is started here and collected while step i+1's records are being prepared: the RCCL
            # kernel has to find a compute unit on a device whose SIMDs all hold resident solver waves of the next batch
            gathers.append(tdist.gather_records_begin(recs, max_rows=S, device=dev if backend == "nccl" else None))
            sent.append(recs)
            while len(gathers) > 1:
                check_gather(tdist.gather_records_end(gathers.pop(0)))
            if trace:
                print(f"[trace] finish at {1e3 * (ta - tr0):.0f} ms: wait {1e3 * (tb_ - ta):.1f}, records {1e3 * (tc - tb_):.1f}, "
                      f"gather {1e3 * (time.perf_counter() - tc):.1f} ms", file=sys.stderr)
        return ok, ms

    def run(nsteps, depth_=None):
        """nsteps steps, at most `depth_` in flight; every step is waited for and its records gathered."""
        depth_ = depth if depth_ is None else depth_
        out_ = []
        for i in range(nsteps):
            o_ = opts[i % depth_]
            if i >= depth_:
                out_.append(finish(o_))
            issue(o_)
        for i in range(max(0, nsteps - depth_), nsteps):
            out_.append(finish(opts[i % depth_]))
        while gathers:                   # every step's records are on every rank before the step counts as done
            check_gather(tdist.gather_records_end(gathers.pop(0)))
        return out_

    run(args.warmup)
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    t_start = time.perf_counter()
    res = run(args.steps)
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    ok = res[-1][0]
    kernel_ms = [r[1] for r in res]
    # per-candidate figures of the line (evaluation counts, device times, gate verdicts) come from the last TIMED step, read
    # here: the passes below (serial steps, planner semantics with cancellation) reuse the contexts
    o_full = opts[(args.steps - 1) % depth]
    stats = o_full.stats().copy()
    gate = o_full.check_feasible().copy()          # printConstraintsSituations over the batch (also part of every timed step)
    # waves a candidate's WORKGROUP holds: one for the classes up to 32 pieces, four for the long classes (class index >= 4; their
    # solver runs on the first wave -- topay_class_of reports the solver's division -- but all four hold their SIMD slots)
    waves_of = np.array([4.0 if o_full.class_of(max(3, n_))[2] >= 4 else 1.0 for n_ in range(0, 129)], dtype=np.float64)
    # slot utilisation: device time of the candidates of one step summed (each is measured on the device's constant
    # clock from its first to its last instruction, times the SIMD slots its workgroup holds) over the SIMD slots there are
    slot_seconds = float((o_full.elapsed_us() * waves_of[np.clip(o_full.n_pieces(), 0, 128)]).sum() * 1e-6)
    # the same steps strictly one after the other (outside the timed region): the per-launch figures of the roofline
    # block and of the rocprofv3 kernel trace are only well defined when launches of different steps do not overlap
    serial = None
    if depth > 1 and not args.no_serial:
        ns = max(1, min(3, args.steps))
        torch.cuda.synchronize()
        ts = time.perf_counter()
        rs = run(ns, 1)
        torch.cuda.synchronize()
        serial = {"steps": ns, "ms_per_step": (time.perf_counter() - ts) / ns * 1e3, "kernel_ms": float(np.mean([r[1] for r in rs]))}
    # The same steps as the planner runs them: the candidates of a planning call still running 100 ms (2400
    # piece-evaluations) after the call's first accepted candidate are interrupted and count as failed.  Outside the timed
    # region and beside the line: `value` is the full figure, every candidate solved to its own end.
    planner = None
    if not args.no_planner and not hires:
        np_ = max(depth, min(6, args.steps))
        cancel_budget[0] = 2400
        run(depth)                        # (fill the pipeline with cancelling batches first)
        torch.cuda.synchronize()
        tp = time.perf_counter()
        rp_ = run(np_)
        torch.cuda.synchronize()
        tpe = time.perf_counter() - tp
        cancel_budget[0] = 0
        o_last = opts[(np_ - 1) % depth]
        intr = o_last.interrupted()
        acc_ = rp_[-1][0] & o_last.check_feasible()
        solved_scen = len(set(tb.scen[acc_].tolist()))
        planner = {"steps": np_, "ms_per_step": tpe / np_ * 1e3,
                   "trajectories_per_s_per_gpu": (B - n_not_launched) / (tpe / np_),
                   "cancel_window": "2400 piece-evaluations after a scenario's first accepted candidate (= the reference's 100 ms at 42 us per piece-evaluation)",
                   "interrupted_fraction": float(intr.mean()), "scenarios_with_a_trajectory": int(solved_scen)}
    rank_report = None
    if distributed:
        rdev = dev if backend == "nccl" else None
        tt = torch.tensor([elapsed], dtype=torch.float64, device=rdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        tot = torch.tensor([float(B - n_not_launched)], dtype=torch.float64, device=rdev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_traj = float(tot.item())
        # what every rank worked on (rank-indexed quantities, for the two-ranks-on-one-device test): its scenario ids, a
        # checksum of its init paths (different seeds -> different inputs) and its share of the solved candidates
        import hashlib
        mine = {"rank": rank, "device": dev_index, "scenario_id_min": int(scen_ids.min()), "scenario_id_max": int(scen_ids.max()),
                "n_scenarios": int(len(scen_ids)), "input_sha": hashlib.sha256(tb.paths.tobytes()).hexdigest()[:16],
                "solved": int(B - n_not_launched), "success_fraction": float(ok.mean())}
        allr = [None] * world
        dist.all_gather_object(allr, mine)
        rank_report = allr
    else:
        total_traj = float(B - n_not_launched)   # only candidates the device actually solved count

    # BASELINE configs[1] (one tables scenario x 64 candidates) as a latency figure beside the throughput line: the batch
    # is far too small to fill the device (64 wavefronts), so its time is the longest candidate's.
    cfg1 = None
    if rank == 0 and not hires and not args.no_config1:
        w1, _, _, lens1, paths1 = wl.tables_scenario(0, 64)
        o1 = api.MomaTrajOptBatch(device=dev_index)
        o1.set_map(w1.origin, w1.res, w1.dims, w1.min_b, w1.max_b, w1.esdf2d, w1.esdf3d)
        o1.set_init_traj(lens1, paths1)
        o1.optimize()
        o1.reset()
        ok1 = o1.optimize()
        ms1, _ = o1.last_kernel_ms()
        cost1, stats1 = np.nan_to_num(o1.traj_cost.copy()), o1.stats().copy()
        # the same call on the helper-wave kernels (topay_set_latency_mode 1: a batch of at most one candidate per SIMD
        # runs on four-wave workgroups whose extra waves join the evaluations only): same bits, shorter solves
        o1.set_latency_mode(1)
        o1.reset()
        o1.optimize()
        o1.reset()
        ok1h = o1.optimize()
        ms1h, _ = o1.last_kernel_ms()
        # solve_ms = the default kernels in every round (comparable round over round; round 4's line had the helper-wave
        # time under this key and the default kernels' under solve_ms_default_kernels); the planning-call mode beside it
        cfg1 = {"workload": "BASELINE configs[1]: 1 tables scenario x 64 candidates", "solve_ms": float(ms1),
                "solve_ms_latency_mode": float(ms1h),
                "trajectories_per_s": float(len(lens1) / (ms1h * 1e-3)), "success_fraction": float(ok1h.mean()),
                "latency_mode_kernels": "helper waves (topay_set_latency_mode 1), %d of %d launches" % (o1.last_helper_launches(), o1.last_kernel_ms()[1]),
                "bit_identical_to_default_kernels": bool((ok1 == ok1h).all() and (cost1 == np.nan_to_num(o1.traj_cost)).all()
                                                         and (stats1 == o1.stats()).all())}
        o1.close()
        w1.close()
    simd_slots = 4 * torch.cuda.get_device_properties(dev_index).multi_processor_count
    ev_ = (stats[:, 2] + stats[:, 5]).astype(np.float64) * (n_pieces > 0)
    np_ = np.maximum(n_pieces, 1).astype(np.float64)
    lane_util = float((ev_ * 13.0 * np_).sum() / max(1.0, (ev_ * 64.0 * np.ceil(13.0 * np_ / 64.0)).sum()))
    gate_timeouts = int(sum(o_.gate_timeouts() for o_ in opts))
    abytes = algorithmic_bytes(stats, n_pieces)
    # Launch duration for the roofline: HIP events on the launch stream bracket each step's solve; with steps
    # overlapping, those spans overlap too, so the average wall time per step is used instead (never smaller than the
    # true per-launch cost).
    kms_wall = float(np.mean(kernel_ms)) if depth == 1 else elapsed / args.steps * 1e3
    sustained = abytes / (kms_wall * 1e-3) / 1e9
    if serial is not None:
        serial["trajectories_per_s_per_gpu"] = (B - n_not_launched) / (serial["ms_per_step"] * 1e-3)
        serial["achieved_GBps"] = abytes / (serial["kernel_ms"] * 1e-3) / 1e9
        serial["frac"] = serial["achieved_GBps"] / HBM_PEAK_GBS
    # The roofline figure proper is PER LAUNCH: the algorithmic bytes of one step over the HIP-event span of its launches when
    # steps do not overlap (the serial steps run right after the timed region; with --inflight 1 the timed steps themselves).
    # With batches in flight the launches of consecutive steps overlap and wall time per step is shorter than any launch:
    # that quotient is the sustained rate of the device, reported beside it, never as `achieved`.
    if depth == 1:
        kms, kdef = kms_wall, "mean HIP-event span of the batch's concurrent launches (timed steps, one batch in flight)"
    elif serial is not None:
        kms, kdef = serial["kernel_ms"], ("mean HIP-event span of the batch's concurrent launches over the strictly serial steps run after the timed "
                                          "region (serial_steps); the timed region overlaps three batches -- see `sustained`")
    else:
        kms, kdef = kms_wall, "NO serial steps in this run (--no-serial): wall time of the timed region / steps, i.e. the sustained figure"
    achieved = abytes / (kms * 1e-3) / 1e9
    # the same with the measured cost of a partly filled pass instead of a whole pass for every pass (PASS_COST_FIXED)
    tail_ = 13.0 * np_ - 64.0 * (np.ceil(13.0 * np_ / 64.0) - 1.0)
    eff_passes = (np.ceil(13.0 * np_ / 64.0) - 1.0) + (PASS_COST_FIXED + (1.0 - PASS_COST_FIXED) * tail_ / 64.0)
    lane_util_eff = float((ev_ * 13.0 * np_).sum() / max(1.0, (ev_ * 64.0 * eff_passes).sum()))
    # HBM-side bytes of one step from the committed PMC passes of this same command (tools/profile_round.sh; FETCH_SIZE
    # and WRITE_SIZE need separate rocprofv3 runs, so they cannot be collected live here).  Only quoted when the
    # workload is the one that was profiled.
    # The figure is tied to the kernel sources it was measured on (source_sha written by tools/summarize_profile.py): with
    # other sources in the tree it is not quoted (traffic null, the reason in traffic_source).
    traffic, traffic_src = None, None
    try:
        import glob
        import hashlib

        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic_hires.json" if hires else "r*_traffic.json")))
        if cands and S == 1024 and Ccand == 8:
            tj = json.load(open(cands[-1]))
            h_ = hashlib.sha256()
            csrc = os.path.join(ROOT, "topay_amd", "csrc")
            for fn in sorted(os.listdir(csrc)):
                h_.update(open(os.path.join(csrc, fn), "rb").read())
            if tj.get("source_sha") == h_.hexdigest()[:16]:
                traffic = float(tj["traffic_bytes"])
                traffic_src = os.path.relpath(cands[-1], ROOT) + " (PMC passes of this command on these kernel sources)"
            else:
                traffic_src = (os.path.relpath(cands[-1], ROOT) + " was measured on other kernel sources (source_sha differs): not quoted; "
                               "tools/profile_round.sh collects it again")
    except (OSError, ValueError, KeyError):
        pass
    out = {
        "metric": "trajectories/sec to L-BFGS convergence (benchmark_tables batch) at 1/2/4/8 GPU",
        "value": total_traj * args.steps / elapsed,
        "unit": "trajectories/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": (f"BASELINE configs[4] high-resolution ESDF: ONE cuboids map {args.hires_size:g}x{args.hires_size:g}x1.6 m "
                         f"@0.02 m ({tb.world(0).esdf3d.nbytes / 1e9:.2f} GB 3-D ESDF), {S} scenarios/GPU x {Ccand} candidates = {B} "
                         "trajectories/GPU, both stages + ALM to convergence") if hires else
                        (f"benchmark_tables batch (BASELINE configs[3] per-GPU share: 8192 scenarios / 8 GPUs): {S} "
                         f"scenarios/GPU x {Ccand} candidates = {B} trajectories/GPU, tables map 20x20x1.6 m @0.1 m "
                         "regenerated per scenario, both stages + ALM to convergence"
                         + ("; init paths from the device's MCRRTs::plan (--front-end)" if front_end else "")),
            "front_end": front_end,
            "scenarios_per_gpu": S, "candidates": Ccand, "trajectories_per_gpu": B,
            "parallelism": f"scenario-sharded x{world}, one workgroup (1 or 4 wavefronts) per trajectory, {depth} batches in flight per GPU",
            "mean_pieces": float(n_pieces.mean()), "max_pieces": int(n_pieces.max()),
            "pieces_over_32": int((n_pieces > 32).sum()),
            # candidates the device did not solve (more pieces than the build supports): none may hide in `value`
            "n_not_launched": n_not_launched, "success_fraction": float(ok.mean()),
            "gate_pass_fraction_of_successes": float(gate[ok].mean()) if ok.any() else 0.0,
            # what a planner could use of `value`: candidates that converged AND pass printConstraintsSituations
            # (planner.cpp:878-880), this rank's fraction applied to the whole-job rate
            "accepted_fraction": float((ok & gate).mean()),
            "accepted_trajectories_per_s": float((ok & gate).sum() / max(1, B - n_not_launched) * total_traj * args.steps / elapsed),
            # samples evaluated per sample slot of the 64-lane passes (13 N samples in ceil(13 N / 64) passes), weighted by
            # every candidate's evaluations: every multiple of five pieces pays a whole pass for N / 5 samples
            "lane_utilisation_sweeps": lane_util, "lane_utilisation_effective": lane_util_eff,
            "mean_evals_per_traj": float((stats[:, 2] + stats[:, 5]).mean()),
            "mean_iters_per_traj": float((stats[:, 1] + stats[:, 4]).mean()),
            "max_evals_per_traj": int((stats[:, 2] + stats[:, 5]).max()),
            "p99_evals_per_traj": float(np.percentile(stats[:, 2] + stats[:, 5], 99)),
            "timed_region": "per step: host-to-device copy of the raw init paths + init kernel + persistent solve kernels (the "
                            "solving wave also runs the feasibility gate on its result) + device-to-host copy of flags, costs, "
                            "durations and of the per-scenario winners' trajectories (+ the RCCL record gather when N > 1); maps resident",
            "h2d_bytes_per_step": int(tb.paths.nbytes + tb.lens.nbytes + 4 * B),
            "winners_per_step": int(winners.get("n", 0)),
            "d2h_winner_bytes_per_step": int(sum(v.nbytes for v in winners["last"].values())) if "last" in winners else 0,
            "record_gather": (dict(gather_stats, collective=("RCCL" if backend == "nccl" else backend) + " all-gather of 6 x f64 per scenario",
                                   rows_expected=int(S * world), world=int(world), scenario_ids_distinct=int(gather_stats.get("distinct_ids", 0)))
                              if distributed else None),
            "ranks": rank_report,
            "setup_seconds_untimed": setup_s,
            "esdf_build_ms_gpu_untimed": edt_ms,
            "config1_latency": cfg1,
            "planner_semantics": planner,
            "workspace_bytes_per_context": int(opt.workspace_bytes()), "contexts": depth,
            # the batches in flight against the device's memory (maps are shared between the contexts and not in the figure: 5.4 MB each)
            "workspace_fraction_of_hbm": float(opt.workspace_bytes()) * depth / float(torch.cuda.get_device_properties(dev_index).total_memory),
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_unit": "bytes per step (2 x FETCH_SIZE + WRITE_SIZE)", "traffic_source": traffic_src,
            "kernel": "k_solve1/2/3 + k_long5/14 (persistent solve: one workgroup of 1 or 4 waves per trajectory takes candidates from its class's queue; the "
                      "up to six class launches of a batch run concurrently; the long classes solve on one wave and evaluate on four)",
            "kernel_ms": kms,
            "kernel_ms_definition": kdef,
            "sustained": {"achieved": sustained, "frac": sustained / HBM_PEAK_GBS, "ms_per_step": kms_wall,
                          "definition": "algorithmic bytes per step over the wall time of the timed region per step (batches in flight overlap)"},
            "kernel_span_ms_each": [float(k) for k in kernel_ms], "steps_in_flight": depth,
            "serial_steps": serial,
            "algorithmic_bytes_per_step": abytes,
            # wave-seconds of one step (per-candidate device times x the waves of its workgroup) over the SIMDs there are.
            # Since round 4 the kernels fit two waves per SIMD (256 registers, LDS permitting), so the mean number of
            # resident waves per SIMD can pass 1 (at most 2); a wave that shares its SIMD runs slower, so wave-seconds
            # are not comparable with rounds 1-3's one-wave-per-SIMD slot-seconds.
            "slot_seconds_per_step": slot_seconds, "simd_slots": simd_slots,
            "work_ms_per_slot": slot_seconds / simd_slots * 1e3,
            "slot_utilisation": slot_seconds / simd_slots / (elapsed / args.steps),
            "resident_waves_per_simd_mean": slot_seconds / simd_slots / (elapsed / args.steps),
            "waves_per_simd_built_for": 2,
        },
    }
    if gate_timeouts:
        out["error"] = f"dispatch gate timed out {gate_timeouts} time(s): a batch waited 120 s for its predecessor to become resident"

    if rank == 0 and world == 1 and not distributed and not args.no_cpu_baseline and not hires:
        # CPU baseline: the oracle (a C++ port of the reference path; the reference itself needs Eigen/ROS/Boost and
        # cannot be built here) on the host cores of this box.  One pool of worker threads, one trajectory per task,
        # every trajectory against its own scenario's map; bounded sample = the first `cpu_sample` trajectories.
        from oracle import oracle as orc

        cores = host_cores()
        nsamp = min(args.cpu_sample, B)
        while nsamp < B and tb.scen[nsamp] == tb.scen[nsamp - 1]:
            nsamp += 1                    # whole scenarios only
        offs = np.concatenate([[0], np.cumsum(tb.lens)])
        used = sorted(set(tb.scen[:nsamp].tolist()))
        mslot = {s_: k for k, s_ in enumerate(used)}
        views = []
        for s_ in used:
            w = tb.world(s_)
            views.append(orc.MapView(w.origin, w.res, w.dims, w.min_b, w.max_b, w.esdf2d, w.esdf3d))
        mid = np.array([mslot[s_] for s_ in tb.scen[:nsamp]], dtype=np.int32)
        r = orc.optimize_batch_maps(views, mid, tb.lens[:nsamp], tb.paths[:offs[nsamp]], nthreads=cores)
        # BASELINE configs[0] (single start/goal, one candidate at a time, CPU L-BFGS): the same oracle on ONE thread over
        # the first candidates of the batch -- the per-candidate latency of the reference's own execution model
        n1 = min(64, nsamp)
        r1 = orc.optimize_batch_maps(views, mid[:n1], tb.lens[:n1], tb.paths[:offs[n1]], nthreads=1)
        out["cpu_baseline"] = {
            "value": nsamp / r["seconds"], "unit": "trajectories/s", "cores": cores, "kind": "port",
            "sample": f"first {nsamp} trajectories ({len(used)} scenarios) of the same batch, CPU oracle (C++ port of the "
                      f"reference path, oracle/), {cores} worker threads (= usable CPUs: affinity capped by the cgroup quota; the box "
                      f"has {os.cpu_count()} hardware threads), {r['seconds']:.1f} s wall, "
                      f"{r['seconds_each'].sum():.1f} thread-seconds, success {r['success'].mean():.3f}, "
                      f"{int((r['n_pieces'] > 32).sum())} of them with more than 32 pieces (all solved, as on the device)",
            "single_thread": {"value": n1 / r1["seconds"], "unit": "trajectories/s", "cores": 1,
                              "sample": f"BASELINE configs[0]: first {n1} trajectories one after the other on one thread, "
                                        f"{r1['seconds']:.1f} s, mean {1e3 * r1['seconds'] / n1:.0f} ms per candidate"},
        }
    sys.stdout.flush()
    os.dup2(_real_stdout, 1)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
