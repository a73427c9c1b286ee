set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/probe
O=gpurun_out/probe
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
VAME_LIB=tools/libvame_hip_probe.so timeout 300 python tools/probe_clock.py > $O/gru_phase2.log 2>&1
cat $O/gru_phase2.log
timeout 300 python tools/microbench.py 10 > $O/micro.log 2>&1; grep gru $O/micro.log
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; cat $O/bench.json
