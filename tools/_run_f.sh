cd $GRAFT_REPO_ROOT
python tools/step_ab.py 4096 one=group_zproj:1 six=group_zproj:0 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_zproj_step_ab.txt
python tools/step_ab.py 256 one=group_zproj:1 six=group_zproj:0 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06_zproj_step_ab.txt
python bench.py --batch 256 --steps 300 --warmup 30 --no-cpu-baseline --no-also --graph 2>/dev/null | cut -c1-200
python bench.py --batch 256 --steps 300 --warmup 30 --no-cpu-baseline --no-also 2>/dev/null | cut -c1-200
