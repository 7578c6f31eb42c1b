#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_sw -- python3 $R/scripts/dev/quick_sw.py > /dev/null 2> $R/gpurun_out/prof_sw.err
cat $R/gpurun_out/prof_sw/*/*_kernel_stats.csv | cut -c1-150
