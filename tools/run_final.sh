# Round-end verification on the MI355X: GPU tests, smoke, bench (train + embed), rocprofv3 kernel stats, probes.
cd $GRAFT_REPO_ROOT
O=gpurun_out/final2
mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1; tail -2 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; cat $O/bench.json
timeout 600 python bench.py --mode embed > $O/embed.json 2> $O/embed.err; cat $O/embed.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/trace.log 2>&1
cd $GRAFT_REPO_ROOT
find $O/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
find $O/trace -type f ! -name "*kernel_stats.csv" -delete
head -8 $O/kernel_stats.csv | cut -c1-150
./tools/probe_gemm calib > $O/calib.log 2>&1; cat $O/calib.log
for s in "768 256 122880 1 1 64" "768 512 122880 1 1 32" "122880 768 512 0 0 1" "122880 512 768 0 1 1"; do ./tools/probe_gemm $s; done > $O/gemm_probe.log 2>&1
timeout 200 python tools/torch_mm_ref.py > $O/torch_mm.log 2>&1; cat $O/torch_mm.log
VAME_LIB=tools/libvame_hip_probe.so timeout 300 python tools/probe_clock.py > $O/gru_probe.log 2>&1
