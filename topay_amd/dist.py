"""Multi-GPU sharding of independent scenarios (one process per GPU, torch.distributed: "nccl" is RCCL on ROCm).

Every candidate trajectory is independent (own decision vector, own L-BFGS state, read-only map), so there is no
collective inside the solve.  Scenarios are partitioned over ranks with all candidates of a scenario on one rank,
so the reference's pick-the-shortest-duration step (src/planner/src/planner.cpp:999-1010) stays local.  The only
exchange is one all-gather of fixed-size per-scenario result records at the end.
"""
import numpy as np

RECORD_FIELDS = ("scenario_id", "best_candidate", "success", "n_pieces", "cost", "duration")
RECORD_WIDTH = 6  # float64 columns


def shard_range(n_items, rank, world):
    """Contiguous block partition [lo, hi) of n_items over `world` ranks (sizes differ by at most one)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def scenario_records(scenario_ids, scen_of_traj, success, cost, n_pieces, durations, return_winners=False):
    """Local argmin-by-duration per scenario -> records [n_scen, 6] (float64).
    scen_of_traj[b] is the (global) scenario id of trajectory b; durations[b] its total duration; success[b] is what the
    planner requires of a candidate (planner.cpp:878-885: optimizeTraj's return value AND printConstraintsSituations).
    With return_winners also the trajectory indices of the per-scenario winners (scenarios without one are left out)."""
    scenario_ids = np.asarray(scenario_ids)
    scen_of_traj = np.asarray(scen_of_traj)
    recs = np.zeros((len(scenario_ids), RECORD_WIDTH))
    recs[:, 0] = scenario_ids
    recs[:, 1] = -1.0
    recs[:, 4:] = np.nan
    winners = []
    if len(scen_of_traj) == 0:
        return (recs, np.zeros(0, dtype=np.int32)) if return_winners else recs
    # first trajectory of every scenario (candidate index = position - first) and the best successful one
    order = np.argsort(scen_of_traj, kind="stable")
    s_sorted = scen_of_traj[order]
    first_pos = {}
    starts = np.nonzero(np.r_[True, s_sorted[1:] != s_sorted[:-1]])[0]
    for st_ in starts:
        first_pos[s_sorted[st_]] = order[st_]
    row_of = {sid: r for r, sid in enumerate(scenario_ids.tolist())}
    ok = np.nonzero(np.asarray(success) > 0)[0]
    if len(ok):
        d_ok = np.asarray(durations)[ok]
        # lexicographic: scenario, then duration, then trajectory index -> the first entry per scenario is the argmin
        o2 = np.lexsort((ok, d_ok, scen_of_traj[ok]))
        so = scen_of_traj[ok][o2]
        firsts = np.nonzero(np.r_[True, so[1:] != so[:-1]])[0]
        for f_ in firsts:
            best = ok[o2[f_]]
            sid = so[f_]
            r = row_of.get(int(sid))
            if r is None:
                continue
            recs[r, 1:] = (best - first_pos[sid], 1.0, n_pieces[best], cost[best], durations[best])
            winners.append(best)
    return (recs, np.array(winners, dtype=np.int32)) if return_winners else recs


def gather_records_begin(local_records, max_rows, device=None):
    """Starts the all-gather of the per-scenario records (padded to max_rows rows per rank) and returns a handle for
    gather_records_end.  Nothing here waits for the collective: with a persistent solve of the next batch resident on
    every SIMD, the RCCL kernel only gets a compute unit when workgroups of that solve exit, so the caller collects the
    result one step later.  Uses the default process group (nccl/RCCL on GPUs, gloo in CPU tests)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size()
    buf = torch.full((max_rows, RECORD_WIDTH + 1), float("nan"), dtype=torch.float64)
    n = local_records.shape[0]
    if n:
        buf[:n, :RECORD_WIDTH] = torch.from_numpy(np.ascontiguousarray(local_records))
        buf[:n, RECORD_WIDTH] = 1.0  # valid flag
    if device is not None:
        buf = buf.to(device)
    out = [torch.empty_like(buf) for _ in range(world)]
    work = dist.all_gather(out, buf, async_op=True)
    return work, out, buf


def gather_records_end(handle):
    """Waits for a gather started by gather_records_begin; returns the concatenated valid rows (on every rank)."""
    work, out, _ = handle
    work.wait()
    rows = np.concatenate([o.cpu().numpy() for o in out])   # per-rank copies: no concatenation kernel on a busy device
    valid = rows[:, RECORD_WIDTH] == 1.0
    return rows[valid, :RECORD_WIDTH]


def gather_records(local_records, max_rows, device=None):
    """All-gather of the per-scenario records, synchronous form."""
    return gather_records_end(gather_records_begin(local_records, max_rows, device))
