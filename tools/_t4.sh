mkdir -p gpurun_out/r2k
for w in m2g m2m; do for thr in 0 1073741824; do echo "== $w d=256 factorise_min_wide=$thr"; NLAM_FACTORISE_MIN_EDGES_WIDE=$thr python tools/kernel_bench.py $w 8 256 2>&1 | grep -v amdgpu.ids; done; done
echo "== m2g d=128"; NLAM_FACTORISE_MIN_EDGES_WIDE=0 python tools/kernel_bench.py m2g 8 128 2>&1 | grep -v amdgpu.ids
echo "== m2g d=128 plain"; NLAM_FACTORISE_MIN_EDGES_WIDE=1073741824 python tools/kernel_bench.py m2g 8 128 2>&1 | grep -v amdgpu.ids
