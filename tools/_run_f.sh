cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "linear or step_ or headline or adam or unaligned or legacy or latent" 2>&1 | grep -v amdgpu.ids | tail -4
python tools/step_ab.py 4096 fused=narrow_fused:1 separate=narrow_fused:0 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_narrow_step_ab.txt
python tools/step_ab.py 256 fused=narrow_fused:1 separate=narrow_fused:0 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06_narrow_step_ab.txt
python bench.py --batch 256 --steps 300 --warmup 30 --no-cpu-baseline --no-also --graph 2>/dev/null | cut -c1-200
