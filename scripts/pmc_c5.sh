#!/bin/bash
# GPU box: SQ counters of the step-1 graph of config C5 at full size (200 000 ONT-profile reads, 1-5 kb): the wide-band stages
# k_nn_scan_refill<16, W> dominate it.  Usage (via gpurun): bash scripts/pmc_c5.sh TAG [n_reads]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-c5}
N=${2:-200000}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv \
    -d $R/gpurun_out/pmc_$TAG -- python3 $R/scripts/time_c5_nn.py $N > $R/gpurun_out/pmc_$TAG.log 2> $R/gpurun_out/pmc_$TAG.err
python3 - <<PY > $R/gpurun_out/${TAG}_sq_summary.txt
import csv, glob, collections
f = sorted(glob.glob('$R/gpurun_out/pmc_$TAG/*/*_counter_collection.csv'))[-1]
agg = collections.OrderedDict()
seen = set()
for r in csv.DictReader(open(f)):
    if 'isocon::k_' not in r['Kernel_Name']:
        continue
    name = r['Kernel_Name'].split('(')[0].replace('void ', '')
    a = agg.setdefault(name, {'n': 0, 'ns': 0, 'vgpr': r.get('VGPR_Count', '?'), 'c': collections.Counter()})
    key = r['Dispatch_Id']
    if key not in seen:
        seen.add(key); a['n'] += 1; a['ns'] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    a['c'][r['Counter_Name']] += float(r['Counter_Value'])
print("SQ counters of scripts/time_c5_nn.py $N under rocprofv3 --pmc (per kernel, all its dispatches; VALU peak = 1.2288e12 wave-instr/s)")
print(open('$R/gpurun_out/pmc_$TAG.log').read().strip())
for name, a in sorted(agg.items(), key=lambda kv: -kv[1]['ns']):
    ms = a['ns'] / 1e6
    v = a['c'].get('SQ_INSTS_VALU', 0.0)
    print("%-46s dispatches=%-4d ms=%-10.2f VGPRs=%s" % (name, a['n'], ms, a['vgpr']))
    print("    " + " ".join("%s=%.5g" % kv for kv in sorted(a['c'].items())))
    if ms > 0:
        print("    VALU wave-instr/s = %.4g = %.3f of the 2-cycle peak; waves resident per SIMD = %.2f" % (v / (ms / 1e3), v / (ms / 1e3) / 1.2288e12,
              a['c'].get('SQ_WAVE_CYCLES', 0.0) * 4 / (ms / 1e3 * 2.4e9 * 1024)))
PY
tail -30 $R/gpurun_out/${TAG}_sq_summary.txt
