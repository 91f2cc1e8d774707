#!/bin/bash
# rocprofv3 kernel trace of the replayed cfg2 step for two builds of the library -> gpurun_out/tlab/{A,B}_timeline.txt
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/tlab
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for arm in A B; do
  lib=$1; [ $arm = B ] && lib=$2
  NLAM_LIB=$GRAFT_REPO_ROOT/$lib rocprofv3 --kernel-trace --output-format csv -d $OUT/tr$arm -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path > $OUT/${arm}_bench.json 2>/dev/null
  python $GRAFT_REPO_ROOT/tools/step_timeline.py $(find $OUT/tr$arm -name "*kernel_trace.csv" | head -1) > $OUT/${arm}_timeline.txt
  rm -rf $OUT/tr$arm
  head -1 $OUT/${arm}_timeline.txt
done
