cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "fused_output_head or linear" 2>&1 | grep -v amdgpu.ids | tail -2
python tools/head_bench.py 4096 256 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_head_bench.txt
