cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof4p4 -o t -- python $GRAFT_REPO_ROOT/bench.py --config cfg4p --steps 20 --warmup 3 --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path > /dev/null 2>&1
