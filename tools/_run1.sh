mkdir -p gpurun_out/coop7
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "coop" 2>&1 | tail -2
timeout 200 python tools/coop_probe.py 2>&1 | grep -v amdgpu > gpurun_out/coop7/coop_probe.txt
grep -A3 "fwd:" gpurun_out/coop7/coop_probe.txt
timeout 200 python tools/coop_bench.py 2>&1 | grep -v amdgpu > gpurun_out/coop7/coop_bench.txt
cut -c1-200 gpurun_out/coop7/coop_bench.txt
