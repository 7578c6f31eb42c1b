#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
for mode in seg noq; do
  if [ $mode = noq ]; then export ISOCON_NN_NO_QGRAM=1; fi
  rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_IFETCH --kernel-trace --output-format csv -d $R/gpurun_out/dg1_$mode -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $R/gpurun_out/dg1_$mode.err
  rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $R/gpurun_out/dg2_$mode -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $R/gpurun_out/dg2_$mode.err
done
