#!/bin/bash
# SQ counters for the NN scan kernel (one bench step).  Usage: pmc_sq.sh TAG
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-sq}
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$TAG -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/pmc_$TAG.err
python3 - <<PY
import csv,glob,collections
f=glob.glob('$R/gpurun_out/pmc_$TAG/*/*_counter_collection.csv')[0]
agg=collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if 'k_nn_scan' in r['Kernel_Name']:
        agg.setdefault((r['Dispatch_Id'],r['Grid_Size'],r['VGPR_Count'],r['SGPR_Count']),{})[r['Counter_Name']]=float(r['Counter_Value'])
for k,v in agg.items(): print(k, {a:"%.4g"%b for a,b in v.items()})
PY
