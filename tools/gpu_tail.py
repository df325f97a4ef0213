"""Tail analysis of the tables batch: per-trajectory device time vs N / evals."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topay_amd import api
from harness import workload as wl
S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
tb = wl.TablesBatch(S, 8, base_seed=42, nthreads=0)
opt = api.MomaTrajOptBatch(device=0, lib_path=os.environ.get("TOPAY_LIB"))
slot = {}
for k, s in enumerate(tb.scenarios):
    w = tb.world(s)
    opt.set_map(w.origin, w.res, w.dims, w.min_b, w.max_b, w.esdf2d, w.esdf3d, map_id=k)
    slot[s] = k
map_ids = np.array([slot[s] for s in tb.scen], dtype=np.int32)
opt.set_init_traj(tb.lens, tb.paths, map_ids=map_ids)
N = opt.n_pieces()
for rep in range(2):
    opt.reset(); ok = opt.optimize()
ms, _ = opt.last_kernel_ms()
us = opt.elapsed_us()
st = opt.stats()
ev = st[:, 2] + st[:, 5]
print("kernel ms", ms, "B", len(N), "sum wave-seconds", us.sum() / 1e6, "-> ideal at 2048 slots", us.sum() / 1e6 / 2048)
print("N hist", np.bincount(N))
order = np.argsort(-us)
print("top 20 by time: (ms, N, evals, iters, ok)")
for b in order[:40]:
    print(f"  {us[b]/1e3:8.1f} {N[b]:3d} {ev[b]:5d} {st[b,1]+st[b,4]:5d} {ok[b]}")
for q in (50, 90, 99, 99.9):
    print("pct", q, np.percentile(us, q) / 1e3, "ms")
# time by bucket
for lo, hi in ((1, 7), (8, 10), (11, 13), (14, 16), (17, 21), (22, 26), (27, 32), (33, 42), (43, 64)):
    m = (N >= lo) & (N <= hi)
    if m.any():
        print(f"N {lo}-{hi}: n={m.sum()} wave-s={us[m].sum()/1e6:.2f} max={us[m].max()/1e3:.0f} ms  us/eval={us[m].sum()/ev[m].sum():.1f} mean evals {ev[m].mean():.0f}")

su = opt.start_us()
print("span of whole solve (first start to last end) ms", (su + us).max() / 1e3)
for lo, hi in ((1, 7), (8, 10), (11, 13), (14, 16), (17, 21), (22, 26), (27, 32), (33, 42), (43, 64)):
    m = (N >= lo) & (N <= hi)
    if m.any():
        print(f"N {lo}-{hi}: first start {su[m].min()/1e3:.0f} ms, last start {su[m].max()/1e3:.0f} ms, last end {(su[m]+us[m]).max()/1e3:.0f} ms")

# utilisation timeline: resident trajectories over time (20 bins)
launched = N > 0
t0 = su[launched].min()
s_ = su[launched] - t0
e_ = s_ + us[launched]
T = e_.max()
edges = np.linspace(0, T, 21)
occ = []
for i in range(20):
    a, b_ = edges[i], edges[i + 1]
    overlap = np.clip(np.minimum(e_, b_) - np.maximum(s_, a), 0, None).sum() / (b_ - a)
    occ.append(int(round(overlap)))
print("span ms %.0f; mean resident waves per 5%% time bin:" % (T / 1e3), occ)
np.savez("gpurun_out/tail_data.npz", N=N, lens=tb.lens, evals=ev, ok=ok, us=us, start=su, st=st, scen=tb.scen)

hw = opt.hw_ids()[launched]
slots = np.unique(hw)
print("distinct SIMD slots used:", len(slots), " XCCs", len(np.unique(hw >> 16)), " CUs", len(np.unique(hw >> 4)))
busy = np.array([us[launched][hw == h_].sum() for h_ in slots]) / 1e6
print("busy seconds per used SIMD slot: min %.2f median %.2f max %.2f (span %.2f s)" % (busy.min(), np.median(busy), busy.max(), T / 1e6))
percu = np.bincount((np.unique(hw) >> 4) - (np.unique(hw) >> 4).min())
print("SIMDs used per CU histogram:", np.bincount(np.bincount(np.unique(hw) >> 4)[np.bincount(np.unique(hw) >> 4) > 0]))
