#!/bin/bash
# Occupancy / chunk-size experiments on the 64-row table kernel: step time and per-phase kernel times of the C3 bench under
#   base | nn_lds_pad=10000 (2 workgroups per CU) | nn_lds_pad=40000 (1 per CU) | nn_list_waves=4 | chunks of 1024 / 3072 pairs (variant builds)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() {  # name, env...
  name=$1; shift
  env "$@" python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        p=json.loads(ln); k=p['roofline']['step_kernels_ms']
        print('%-28s step %.2f ms | '%('$name', p['ms_per_step']) + ' | '.join('%s %.2f'%(a.split(' (')[0],b) for a,b in k.items()) + ' | digest ok %s' % p['config'].get('graph_equals_reference_loop_fixture'))
"
}
run base X=1
run lds_pad_10000_2wg_per_cu ISOCON_DEBUG_VARIANT=nn_lds_pad=10000
run lds_pad_40000_1wg_per_cu ISOCON_DEBUG_VARIANT=nn_lds_pad=40000
run list_waves_4 ISOCON_DEBUG_VARIANT=nn_list_waves=4
run chunk_1024 ISOCON_LIB=$R/isocon_amd/lib/libisocon_hip_chunk1024.so
run chunk_3072 ISOCON_LIB=$R/isocon_amd/lib/libisocon_hip_chunk3072.so
