"""How long do the pieces of the per-step record gather take while a persistent solve occupies every SIMD slot?"""
import os, sys, time
import numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from topay_amd import api
from harness import workload as wl
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
dev = torch.device("cuda:0")
tb = wl.TablesBatch(1024, 8, base_seed=42, nthreads=0)
worlds = [tb.world(s) for s in tb.scenarios]
slot = {s: k for k, s in enumerate(tb.scenarios)}
o = api.MomaTrajOptBatch(device=0)
w0 = worlds[0]
o.build_esdf_batch(w0.origin, w0.res, w0.dims, w0.min_b, w0.max_b, np.stack([w.occ2d for w in worlds]), np.stack([w.occ3d for w in worlds]))
o.set_init_traj(tb.lens, tb.paths, map_ids=np.array([slot[s] for s in tb.scen], dtype=np.int32))
o.reset(); o.optimize()
def probe(tag):
    t = [time.perf_counter()]
    buf = torch.full((1024, 7), float("nan"), dtype=torch.float64); t.append(time.perf_counter())
    buf = buf.to(dev); torch.cuda.current_stream().synchronize(); t.append(time.perf_counter())
    out = [torch.empty_like(buf)]; t.append(time.perf_counter())
    dist.all_gather(out, buf); torch.cuda.current_stream().synchronize(); t.append(time.perf_counter())
    rows = torch.cat(out); torch.cuda.current_stream().synchronize(); t.append(time.perf_counter())
    rows = rows.cpu().numpy(); t.append(time.perf_counter())
    names = ["host full", "to(device)", "empty_like", "all_gather", "cat", "cpu()"]
    print(tag, {n: round((b - a) * 1e3, 2) for n, a, b in zip(names, t[:-1], t[1:])})
probe("idle GPU (first call: communicator setup)")
probe("idle GPU")
o.reset(); o.optimize_async(); time.sleep(0.2)
probe("during the bulk of a solve")
probe("during the bulk of a solve (2)")
o.finish()
dist.destroy_process_group()
