run() { echo "[$1 $2] $(env $1 python bench.py --config $2 --steps $3 --warmup 10 --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*')"; }
run "NLAM_FORK_SPREAD=0" cfg2 200
run "NLAM_FORK_SPREAD=1" cfg2 200
run "NLAM_FORK_SPREAD=0" cfg4 100
run "NLAM_FORK_SPREAD=1" cfg4 100
run "NLAM_FORK_SPREAD=0" cfg4p 100
run "NLAM_FORK_SPREAD=1" cfg4p 100
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
NLAM_FORK_SPREAD=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_sp -o t -- python bench.py --config cfg2 --steps 30 --warmup 5 --no-cpu-baseline --no-gpu-baseline --no-roofline --no-data-path > gpurun_out/sp_prof.log 2>&1
