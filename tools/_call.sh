mkdir -p gpurun_out/dot && cd gpurun_out/dot
NLAM_DEFER_X=1 DEBUG_HIP_GRAPH_DOT_PRINT=1 AMD_LOG_LEVEL=0 python $GRAFT_REPO_ROOT/bench.py --config cfg2 --steps 3 --warmup 1 --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path > run.log 2>&1
ls -la . /tmp | head -40
find / -name "*.dot" -mmin -5 2>/dev/null | head
