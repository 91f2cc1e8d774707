#!/bin/bash
# round 6, call 38: the N > 1 control flow on one GPU (NLAM_BENCH_DRYRUN: two ranks over gloo on cuda:0; numbers meaningless) after the stream
# placement / also-leg changes; plus the gloo DDP tests on this box
export NLAM_BENCH_DRYRUN=1
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/dry2.json 2> gpurun_out/dry2.err
echo "rc=$?"; tail -c 1500 gpurun_out/dry2.json; echo; grep -i "error\|Traceback\|warn" gpurun_out/dry2.err | head -10
