#!/bin/bash
# PMC counters of the bound kernel k_qgram_mm (GPU box): matrix-pipe busy cycles, LDS activity, waits
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_mm -- python3 $R/scripts/dev/step_laps.py > /dev/null 2> $R/gpurun_out/pmc_mm.err
python3 - <<PY
import csv,glob,collections
f=sorted(glob.glob('$R/gpurun_out/pmc_mm/*/*_counter_collection.csv'))[-1]
agg=collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if 'k_qgram_mm' in r['Kernel_Name'] or 'scan_refill' in r['Kernel_Name']:
        e=agg.setdefault(r['Dispatch_Id'],{'name':r['Kernel_Name'].split('(')[0][-30:],'dur':int(r['End_Timestamp'])-int(r['Start_Timestamp'])})
        e[r['Counter_Name']]=e.get(r['Counter_Name'],0)+float(r['Counter_Value'])
for k,v in list(agg.items())[-2:]: print(" ".join("%s=%s"%(a,("%.4g"%b if isinstance(b,float) else b)) for a,b in v.items()))
PY
