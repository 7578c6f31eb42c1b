#!/bin/bash
# Runs on the GPU box (via gpurun): bench + rocprofv3 kernel trace + three PMC passes (SQ, FETCH_SIZE, WRITE_SIZE -- separate
# passes with --kernel-trace only, as MI355X_MICROARCH.md prescribes).  Outputs under gpurun_out/; turn them into the
# tracked profiles/ files with `python scripts/summarize_profile.py TAG` afterwards (in the build container).
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
cd /tmp
python3 $R/bench.py --steps 5 --warmup 1 > $R/gpurun_out/bench_$TAG.json 2> $R/gpurun_out/bench_$TAG.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/bench_prof_$TAG.json 2> $R/gpurun_out/prof_$TAG.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq_$TAG -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/bench_sq_$TAG.json 2> $R/gpurun_out/pmc_sq_$TAG.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch_$TAG -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/pmc_fetch_$TAG.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write_$TAG -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/pmc_write_$TAG.err
find $R/gpurun_out -name "*.csv" -newer $R/gpurun_out/bench_$TAG.json | head -20
cat $R/gpurun_out/bench_$TAG.json
