cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "fused_output_head or fused_heads or headline or failed_step or step_" 2>&1 | grep -v amdgpu.ids | tail -8 > gpurun_out/r06_b_tests.txt
python tools/head_bench.py 4096 256 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_head_bench.txt
python tools/step_ab.py 4096 fused=fuse_heads:1 separate=fuse_heads:0 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_head_step_ab.txt
python tools/step_ab.py 256 fused=fuse_heads:1 separate=fuse_heads:0 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06_head_step_ab.txt
cat gpurun_out/r06_b_tests.txt gpurun_out/r06_head_bench.txt gpurun_out/r06_head_step_ab.txt
