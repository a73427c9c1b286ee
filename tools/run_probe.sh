cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/probe
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q 2>&1 | tail -2
timeout 300 python tools/microbench.py 10 2>&1 | grep gru
VAME_LIB=tools/libvame_hip_probe.so timeout 300 python tools/probe_clock.py 2>&1 | grep -v amdgpu.ids | head -9
