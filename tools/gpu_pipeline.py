"""Timeline of a pipelined sequence of batches (bench.py's issue/finish order) on the device's constant clock."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topay_amd import api
from harness import workload as wl
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 2
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
tb = wl.TablesBatch(S, 8, base_seed=42, nthreads=0)
worlds = [tb.world(s) for s in tb.scenarios]
slot = {s: k for k, s in enumerate(tb.scenarios)}
map_ids = np.array([slot[s] for s in tb.scen], dtype=np.int32)
opts = []
for _ in range(depth):
    o = api.MomaTrajOptBatch(device=0)
    w0 = worlds[0]
    o.build_esdf_batch(w0.origin, w0.res, w0.dims, w0.min_b, w0.max_b, np.stack([w.occ2d for w in worlds]),
                       np.stack([w.occ3d for w in worlds]))
    o.set_init_traj(tb.lens, tb.paths, map_ids=map_ids)
    opts.append(o)
for o in opts:
    o.reset(); o.optimize()
N = opts[0].n_pieces(); m = (N > 0) & (opts[0].elapsed_us() > 0)   # (candidates a diagnostic build left out have no device time)
rec = []
t0 = time.perf_counter()
def fin(i):
    o = opts[i % depth]; o.finish()
    rec.append((i, o.start_us(raw=True)[m].copy(), o.elapsed_us()[m].copy(), time.perf_counter() - t0, o.hw_ids()[m].copy()))
host = []
for i in range(steps):
    o = opts[i % depth]
    if i >= depth: fin(i - depth)
    a = time.perf_counter() - t0
    o.reset(); o.optimize_async()
    host.append((i, a, time.perf_counter() - t0))
for i in range(max(0, steps - depth), steps): fin(i)
total = time.perf_counter() - t0
print(f"{steps} steps depth {depth}: {total*1e3:.0f} ms -> {steps*m.sum()/total:.0f} traj/s")
base = min(r[1].min() for r in rec)
for (i, a, b) in host: print(f"  host: issue {i} called at {a*1e3:.0f} ms returned at {b*1e3:.0f} ms")
for (i, s, u, th, hw_) in rec:
    s = s - base
    print(f"  batch {i}: first start {s.min()/1e3:.0f}, median start {np.median(s)/1e3:.0f}, last start {s.max()/1e3:.0f}, last end {(s+u).max()/1e3:.0f} (host saw finish at {th*1e3:.0f})")
Nm = N[m]
cls = np.where(Nm <= 10, 0, np.where(Nm <= 21, 1, np.where(Nm <= 32, 2, np.where(Nm <= 42, 3, 4))))
for (i, s_, u_, th, hw_) in rec[:8]:
    s0 = s_ - base
    print(f"  batch {i} per class (first start, queue drained = last start, last end) ms:",
          [(int(s0[cls == k].min() / 1e3), int(s0[cls == k].max() / 1e3), int((s0 + u_)[cls == k].max() / 1e3)) if (cls == k).any() else None for k in range(5)])
T = max((r[1] - base + r[2]).max() for r in rec)
nb = 40
edges = np.linspace(0, T, nb + 1)
tot = np.zeros(nb)
for (i, s, u, th, hw_) in rec:
    s = s - base; e = s + u
    tot += np.array([np.clip(np.minimum(e, edges[k + 1]) - np.maximum(s, edges[k]), 0, None).sum() / (edges[k + 1] - edges[k]) for k in range(nb)])
ends = np.array([ (r[1] - base + r[2]).max() for r in rec ]); fs = np.array([(r[1] - base).min() for r in rec])
print("  batch latency ms (first start to last end):", [int(round(v / 1e3)) for v in ends - fs], " cadence ms:", [int(round(v / 1e3)) for v in np.diff(ends)])
fine = np.linspace(0, T, 401); totf = np.zeros(400)
for (i, s, u, th, hw_) in rec:
    s = s - base; e = s + u
    si = np.searchsorted(fine, s); ei = np.searchsorted(fine, e)
    d = np.zeros(402); np.add.at(d, si, 1); np.add.at(d, ei, -1); totf += np.cumsum(d)[1:401]
print("  max resident (fine bins):", int(totf.max()))
print(f"  total resident waves per {T/nb/1e3:.0f} ms bin:", [int(round(v)) for v in tot])

# gaps per SIMD slot: time a slot spends without a candidate between two candidates (any batch)
allS = np.concatenate([r[1] - base for r in rec]); allE = allS + np.concatenate([r[2] for r in rec])
allH = np.concatenate([r[4] for r in rec]); allC = np.concatenate([cls for _ in rec]); allB = np.concatenate([np.full(m.sum(), r[0]) for r in rec])
order = np.lexsort((allS, allH))
H, S_, E_, C_, B_ = allH[order], allS[order], allE[order], allC[order], allB[order]
same = H[1:] == H[:-1]
gap = (S_[1:] - E_[:-1])[same]
cb, ca = C_[:-1][same], C_[1:][same]
bb, ba = B_[:-1][same], B_[1:][same]
print("  slots seen:", len(np.unique(H)), " gaps >1 ms:", int((gap > 1e3).sum()), " total idle slot-seconds in gaps:", round(gap[gap > 0].sum() / 1e6, 1),
      " of", round(len(np.unique(H)) * T / 1e6, 1))
big = gap > 1e3
for k1 in range(5):
    for k2 in range(5):
        q = big & (cb == k1) & (ca == k2)
        if q.any():
            print(f"    class {k1+1} -> class {k2+1}: {int(q.sum())} gaps, mean {gap[q].mean()/1e3:.0f} ms, total {gap[q].sum()/1e6:.1f} slot-s; same batch {int((bb[q]==ba[q]).sum())}")

# Steady-state accounting with the waves of every workgroup counted (a candidate of more than 32 pieces holds two SIMDs, of
# more than 64 four): busy slot-time inside the window from the first start of batch 2 to the last end of the third batch
# from the end, against 1024 x the window.
wv = np.where(Nm <= 32, 1, np.where(Nm <= 64, 2, 4)).astype(float)
if len(rec) >= 6:
    w0 = (rec[2][1] - base).min(); w1 = (rec[-3][1] - base + rec[-3][2]).max()
    busy = 0.0; per_batch = []
    for (i, s, u, th, hw_) in rec:
        s = s - base; e = s + u
        ov = np.clip(np.minimum(e, w1) - np.maximum(s, w0), 0, None)
        busy += (ov * wv).sum()
        per_batch.append(float((u * wv).sum() / 1e6))
    print(f"  steady window {w0/1e3:.0f}..{w1/1e3:.0f} ms: busy {busy/1e6:.1f} slot-s of {1024*(w1-w0)/1e6:.1f} = {busy/(1024*(w1-w0)):.3f}; "
          f"slot-seconds per batch (elapsed x waves): {[round(v, 1) for v in per_batch]}")
