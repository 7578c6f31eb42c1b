#!/bin/bash
# functional test of bench.py's N>1 path on a 1-GPU box: 2 ranks share GPU 0, collectives over gloo
export ISOCON_DIST_BACKEND=gloo ISOCON_GPU_DEVICE=0
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --reads 8000 --length 1200 --isoforms 5 --seed 9
