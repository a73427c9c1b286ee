cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prep
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "prep" 2>&1 | tail -3
timeout 600 python tools/prep_bench.py 1000000 > gpurun_out/prep/prep_bench.json 2> gpurun_out/prep/prep_bench.err; cat gpurun_out/prep/prep_bench.json; tail -3 gpurun_out/prep/prep_bench.err
