"""N > 1 path on CPU: scenario sharding + the single end-of-run all-gather, world_size 2 over gloo."""
import os
import subprocess
import sys

import numpy as np

from topay_amd import dist as tdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["TOPAY_ROOT"])
import numpy as np, torch, torch.distributed as dist
from topay_amd import dist as tdist
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
S, C = 7, 3                     # 7 scenarios x 3 candidates, deliberately not divisible by 2
lo, hi = tdist.shard_range(S, rank, world)
rng = np.random.default_rng(0)
dur = rng.uniform(5, 20, (S, C)); succ = (rng.uniform(size=(S, C)) > 0.3).astype(np.int32)
cost = rng.uniform(100, 900, (S, C)); npc = rng.integers(4, 12, (S, C))
ids = np.arange(lo, hi)
scen = np.repeat(ids, C)
recs = tdist.scenario_records(ids, scen, succ[lo:hi].reshape(-1), cost[lo:hi].reshape(-1), npc[lo:hi].reshape(-1), dur[lo:hi].reshape(-1))
allr = tdist.gather_records(recs, max_rows=(S + world - 1) // world)
if rank == 0:
    np.save(os.environ["OUT"], allr)
    np.savez(os.environ["OUT"] + ".ref", dur=dur, succ=succ, cost=cost)
dist.barrier()
dist.destroy_process_group()
'''


def test_shard_range_covers_everything():
    for n in (1, 7, 8, 1000, 1023):
        for w in (1, 2, 4, 8):
            spans = [tdist.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


def test_two_rank_gather(tmp_path):
    out = str(tmp_path / "recs.npy")
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", OUT=out,
                   TOPAY_ROOT=ROOT)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env))
    for p in procs:
        assert p.wait(timeout=120) == 0
    recs = np.load(out)
    ref = np.load(out + ".ref.npz")
    assert recs.shape == (7, 6) and sorted(recs[:, 0].astype(int)) == list(range(7))
    for row in recs:
        s = int(row[0])
        ok = np.nonzero(ref["succ"][s])[0]
        if len(ok) == 0:
            assert row[2] == 0
        else:
            best = ok[np.argmin(ref["dur"][s][ok])]
            assert int(row[1]) == best and row[2] == 1 and np.isclose(row[5], ref["dur"][s][best])
            assert np.isclose(row[4], ref["cost"][s][best])
