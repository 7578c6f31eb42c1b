#!/bin/bash
# quick look: kernel-trace stats + SQ counters of a short bench run (GPU box)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-q}
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pq_$TAG -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/pq_$TAG.json 2> $R/gpurun_out/pq_$TAG.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/pqc_$TAG -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > /dev/null 2> $R/gpurun_out/pqc_$TAG.err
find $R/gpurun_out/pq_$TAG -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-200 | head -14
python3 - <<PY
import csv,glob,collections
f=glob.glob('$R/gpurun_out/pqc_$TAG/*/*_counter_collection.csv')[0]
agg=collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if 'isocon::k_' in r['Kernel_Name']:
        e=agg.setdefault(r['Dispatch_Id'],{'name':r['Kernel_Name'].split('(')[0].replace('void isocon::','')[:40],'grid':r['Grid_Size'],'dur':int(r['End_Timestamp'])-int(r['Start_Timestamp'])})
        e[r['Counter_Name']]=e.get(r['Counter_Name'],0)+float(r['Counter_Value'])
for k,v in agg.items(): print(" ".join("%s=%s"%(a,("%.4g"%b if isinstance(b,float) else b)) for a,b in v.items()))
PY
