"""Overhead of the sharded protocol itself: one rank through sharded_nn_graph vs the direct call, with the seconds per protocol
part.  Backend: nccl (RCCL, one rank on this GPU) unless ISOCON_DIST_BACKEND=gloo."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import numpy as np, torch, torch.distributed as dist
from isocon_amd import synth
from isocon_amd.store import SeqStore
from isocon_amd.dist import sharded_nn_graph
backend = os.environ.get("ISOCON_DIST_BACKEND", "nccl")
if backend == "nccl":
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    dev = torch.device("cuda", 0)
else:
    dist.init_process_group("gloo", rank=0, world_size=1)
    dev = torch.device("cpu")
accs, seqs, _ = synth.make_reads(50000, 2500, 10, 30001)
seqs = sorted(dict.fromkeys(seqs), key=len)
st = SeqStore(seqs)
st.nn_graph()
sharded_nn_graph(st, dist=dist, device=dev)
for rep in range(10):
    t = time.time(); b0, r0, c0, s0 = st.nn_graph(); t1 = time.time() - t
    laps = {}
    t = time.time(); b1, r1, c1, stats = sharded_nn_graph(st, dist=dist, device=dev, return_stats=True, laps=laps); t2 = time.time() - t
    k = sum(x.get("kernel_ms", 0.0) for x in stats)
    print("%s: direct %.1f ms (kernels %.1f) | sharded protocol %.1f ms (kernels %.1f) -> overhead %.1f ms; identical %s" %
          (backend, t1 * 1e3, s0["kernel_ms"], t2 * 1e3, k, t2 * 1e3 - k, bool((b0 == b1).all() and (c0 == c1).all())))
    print("   parts (ms):", {a: round(b * 1e3, 2) for a, b in laps.items()}, " kernels per phase:", [round(x.get("kernel_ms", 0.0), 1) for x in stats])
dist.destroy_process_group()
