#!/bin/bash
# How much is there to gain from keeping a table for more pairs (lanes of a draining workgroup take the next chunk of the
# SAME entry)?  Upper bound by experiment: chunks of 4 096 / 8 192 pairs (the list builder with 3 / 1 waves per workgroup so that its staging
# buffer fits LDS; its own time suffers, the table launches show what longer chunks are worth).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() { name=$1; shift; env "$@" python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        p=json.loads(ln); k=p['roofline']['step_kernels_ms']
        print('%-22s step %.2f ms | '%('$name', p['ms_per_step']) + ' | '.join('%s %.2f'%(a.split(' (')[0],b) for a,b in k.items()) + ' | digest ok %s' % p['config'].get('graph_equals_reference_loop_fixture'))
"; }
run chunk2048_4waves X=1
run chunk2048_1wave ISOCON_LIB=$R/isocon_amd/lib/libisocon_hip_c2048w1.so
run chunk4096_3waves ISOCON_LIB=$R/isocon_amd/lib/libisocon_hip_c4096w3.so
run chunk8192_1wave ISOCON_LIB=$R/isocon_amd/lib/libisocon_hip_c8192w1.so
