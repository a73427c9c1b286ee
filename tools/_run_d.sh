cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "linear_group or step_ or headline or decoder or legacy or h0" 2>&1 | grep -v amdgpu.ids | tail -4
python bench.py --batch 256 --steps 300 --warmup 30 --no-cpu-baseline --no-also --graph 2>/dev/null | cut -c1-200
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also 2>/dev/null | cut -c1-200
