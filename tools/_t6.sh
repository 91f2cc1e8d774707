mkdir -p gpurun_out/r2l
cd /tmp && export TMPDIR=/tmp
for mode in fact plain; do
  if [ $mode = plain ]; then export NLAM_FACTORISE_MIN_EDGES_WIDE=1073741824; else unset NLAM_FACTORISE_MIN_EDGES_WIDE; fi
  rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2l/tr_$mode -o t -- python $GRAFT_REPO_ROOT/bench.py --config cfg4 --steps 6 --warmup 2 --no-cpu-baseline --no-gpu-baseline --no-roofline > /dev/null 2>&1
  python $GRAFT_REPO_ROOT/tools/step_timeline.py $(find $GRAFT_REPO_ROOT/gpurun_out/r2l/tr_$mode -name "*kernel_trace.csv" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/r2l/timeline_cfg4_$mode.txt
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/r2l/tr_$mode
done
cd $GRAFT_REPO_ROOT; for mode in fact plain; do head -1 gpurun_out/r2l/timeline_cfg4_$mode.txt; grep "^# " gpurun_out/r2l/timeline_cfg4_$mode.txt | sed -n 3,22p; done
