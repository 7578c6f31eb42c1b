#!/bin/bash
# LDS counters for the NN main-pass kernel (one bench step).  Usage: pmc_lds.sh TAG
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-lds}
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE "SQ_[A-Z_]*LDS[A-Z_]*" | sort -u | tr '\n' ' ' > $R/gpurun_out/lds_counters.txt
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$TAG -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/pmc_$TAG.err
python3 - <<PY
import csv,glob,collections
f=glob.glob('$R/gpurun_out/pmc_$TAG/*/*_counter_collection.csv')[0]
agg=collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if 'k_nn_scan' in r['Kernel_Name']:
        agg.setdefault((r['Kernel_Name'][:40],r['Dispatch_Id'],r['Grid_Size'],r['VGPR_Count'],r['LDS_Block_Size']),{})[r['Counter_Name']]=float(r['Counter_Value'])
for k,v in agg.items(): print(k, {a:"%.4g"%b for a,b in v.items()})
PY
cat $R/gpurun_out/lds_counters.txt
