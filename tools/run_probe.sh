set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/probe
O=gpurun_out/probe
(for i in $(seq 1 60); do rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk|fclk" | head -3 | tr '\n' ' '; echo; sleep 0.5; done) > $O/clocks.log 2>&1 &
SMI=$!
timeout 120 ./tools/probe_gemm 768 256 122880 1 1 64 > $O/probe.log 2>&1
timeout 120 ./tools/probe_gemm 768 512 122880 1 1 32 >> $O/probe.log 2>&1
timeout 120 ./tools/probe_gemm 122880 768 512 0 0 1 >> $O/probe.log 2>&1
timeout 120 ./tools/probe_gemm 122880 512 768 0 1 1 >> $O/probe.log 2>&1
timeout 120 ./tools/probe_gemm 4096 4096 4096 1 1 1 >> $O/probe.log 2>&1
timeout 200 python tools/torch_mm_ref.py > $O/torch_mm.log 2>&1
kill $SMI
cat $O/probe.log $O/torch_mm.log; tail -5 $O/clocks.log
