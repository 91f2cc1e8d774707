#!/bin/bash
# round 6, call 31: hardware-queue placement of the cfg3 step's kernels with and without a cfg2 trainer run earlier in the process
mkdir -p gpurun_out/r6c31
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for h in "-" "train"; do
  hh=$h; [ "$h" = "-" ] && hh=""
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r6c31/tr_$h -o t -- python $R/tools/r6/history_probe.py cfg3 fp32 4 $hh 2>&1 | grep "history="
  python $R/tools/r6/queue_summary.py $(find $R/gpurun_out/r6c31/tr_$h -name "*kernel_trace.csv" | head -1)
  rm -rf $R/gpurun_out/r6c31/tr_$h
done
