cd $GRAFT_REPO_ROOT
make -s -j8 probe > /dev/null 2>&1
python tools/head_probe.py 4096 30 512 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_head_probe.txt
python tools/head_probe.py 4096 30 256 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06_head_probe.txt
python tools/head_probe.py 256 30 512 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06_head_probe.txt
python tools/head_bench.py 4096 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_head_bench.txt
