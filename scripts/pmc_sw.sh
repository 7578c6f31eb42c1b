#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-sw}
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$TAG -- python3 $R/scripts/dev/quick_sw.py > /dev/null 2> $R/gpurun_out/pmc_$TAG.err
python3 - <<PY
import csv,glob,collections
f=glob.glob('$R/gpurun_out/pmc_$TAG/*/*_counter_collection.csv')[0]
agg=collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if 'k_sg_forward' in r['Kernel_Name']:
        agg.setdefault((r['Dispatch_Id'],r['Grid_Size'],r['VGPR_Count'],r['SGPR_Count'], int(r['End_Timestamp'])-int(r['Start_Timestamp'])),{})[r['Counter_Name']]=float(r['Counter_Value'])
for k,v in list(agg.items())[-1:]: print(k, {a:"%.4g"%b for a,b in v.items()})
PY
