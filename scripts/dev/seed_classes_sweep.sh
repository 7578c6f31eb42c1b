R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() { name=$1; shift; env "$@" python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        p=json.loads(ln); k=p['roofline']['step_kernels_ms']
        print('%-16s step %.2f ms | '%('$name', p['ms_per_step']) + ' | '.join('%s %.2f'%(a.split(' (')[0],b) for a,b in k.items()) + ' | aligned %d | digest ok %s' % (p['roofline']['pairs_aligned'], p['config'].get('graph_equals_reference_loop_fixture')))
"; }
run classes4_default X=1
run classes2 ISOCON_DEBUG_VARIANT=nn_seed_classes=2
run classes1 ISOCON_DEBUG_VARIANT=nn_seed_classes=1
