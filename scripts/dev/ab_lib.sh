# scripts/dev/ab_lib.sh NAME : bench.py's NN-graph step with the shipped library and with isocon_amd/lib/libisocon_hip_NAME.so (scripts/dev/build_variant.sh), twice each
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() { name=$1; shift; env "$@" python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        p=json.loads(ln); k=p['roofline']['step_kernels_ms']
        print('%-12s step %.2f ms | '%('$name', p['ms_per_step']) + ' | '.join('%s %.2f'%(a.split(' (')[0],b) for a,b in k.items()) + ' | digest ok %s' % p['config'].get('graph_equals_reference_loop_fixture'))
"; }
run base X=1
run $1 ISOCON_LIB=$R/isocon_amd/lib/libisocon_hip_$1.so
run base X=1
run $1 ISOCON_LIB=$R/isocon_amd/lib/libisocon_hip_$1.so
