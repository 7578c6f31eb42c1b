"""Exercises the N > 1 code path of bench.py / isocon_amd.dist through RCCL (backend "nccl") with however many ranks
torchrun starts on this box (one GPU -> one rank): process-group set-up with device_id, all_reduce(MIN) on int32,
all_gather on int64 / int32, all_reduce(MAX) on float64, barrier -- and the sharded graph against the single call.
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 scripts/nccl_selfcheck.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

local_rank = int(os.environ.get("LOCAL_RANK", "0"))
os.environ.setdefault("ISOCON_GPU_DEVICE", str(local_rank))
import torch
import torch.distributed as dist

torch.cuda.set_device(local_rank)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
from isocon_amd import dist as D
from isocon_amd import synth
from isocon_amd.store import SeqStore

accs, seqs, _ = synth.make_reads(6000, 1200, 4, 77)
seqs = sorted(dict.fromkeys(seqs), key=len)
store = SeqStore(seqs)
best1, rp1, cols1, _ = store.nn_graph()
best, rp, cols, stats = D.sharded_nn_graph(store, dist=dist, return_stats=True)
assert (best == best1).all() and (rp == rp1).all() and (cols == cols1).all(), "sharded graph differs"
a = np.arange(0, 4000, 7, dtype=np.uint32) % store.n
b = (a + 3) % store.n
ed = D.sharded_ed_pairs(store, a, b, dist=dist)
assert (ed == store.ed_pairs(a, b)).all()
ops, ptr, res = D.sharded_sg_trace(store, a[:200], b[:200], -2, dist=dist)
ops1, ptr1, res1 = store.sg_trace(a[:200], b[:200], np.full(200, -2, dtype=np.int8))
assert (ops == ops1).all() and (ptr == ptr1).all() and (res == res1).all()
t = torch.tensor([1.5 + dist.get_rank()], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
if dist.get_rank() == 0:
    print("nccl selfcheck ok: world %d, %d edges, max %.1f" % (dist.get_world_size(), len(cols), float(t.item())))
dist.destroy_process_group()
