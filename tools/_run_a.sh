cd $GRAFT_REPO_ROOT
python -m pytest tests/test_train_driver_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "graph or optimizer_state or failed_step or option_values or cfg4 or headline" -s 2>&1 | grep -v amdgpu.ids | tail -15 > gpurun_out/r06_a_tests.txt
AB_H=512 AB_T=60 AB_STEPS=3 AB_ROUNDS=5 python tools/step_ab.py 8192 on=nuc_side:1,bwd_overlap:1 off=nuc_side:0,bwd_overlap:0 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_cfg3_overlap_ab.txt
python bench.py --hidden 512 --time-window 60 --batch 8192 --steps 3 --warmup 1 --no-cpu-baseline --no-also > gpurun_out/r06_cfg3_bench.json 2>/dev/null
python bench.py --batch 256 --steps 200 --warmup 30 --no-cpu-baseline --no-also --graph > gpurun_out/r06_b256_graph.json 2>/dev/null
python bench.py --batch 256 --steps 200 --warmup 30 --no-cpu-baseline --no-also > gpurun_out/r06_b256_eager.json 2>/dev/null
cat gpurun_out/r06_a_tests.txt gpurun_out/r06_cfg3_overlap_ab.txt; cut -c1-300 gpurun_out/r06_cfg3_bench.json; cut -c1-300 gpurun_out/r06_b256_graph.json; cut -c1-300 gpurun_out/r06_b256_eager.json
