#!/bin/bash
# N>1 path at full C3 size on a 1-GPU box (2 ranks share GPU 0, gloo collectives): functional + host-overhead check
export ISOCON_DIST_BACKEND=gloo ISOCON_GPU_DEVICE=0
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --steps 3 --warmup 1
