# Round profile pass on the MI355X: bench lines, rocprofv3 kernel trace, PMC HBM traffic (separate passes), SQ counters.
# usage (GPU box): bash tools/run_profiles.sh <outdir-name>
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-prof}
mkdir -p $O
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-250 $O/bench.json
timeout 300 python bench.py --mode embed > $O/embed.json 2> $O/embed.err; cut -c1-250 $O/embed.json
timeout 300 python bench.py --mode embed --embed-windows 10000000 --no-cpu-baseline > $O/embed10m.json 2> $O/embed10m.err; cut -c1-250 $O/embed10m.json
timeout 300 python bench.py --hidden 512 --time-window 60 --batch 8192 --steps 5 --warmup 2 --no-cpu-baseline --dump-kernels > $O/cfg4.json 2> $O/cfg4.err; cut -c1-250 $O/cfg4.json
timeout 200 python bench.py --batch 256 --steps 40 --warmup 10 --no-cpu-baseline > $O/b256.json 2>/dev/null; cut -c1-200 $O/b256.json
timeout 300 python tools/hmm_bench.py 1000000 2>/dev/null > $O/hmm.json; cat $O/hmm.json
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/raw_trace -- $B --steps 10 --warmup 3 > $O/trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/raw_fetch -- $B --steps 2 --warmup 1 > $O/fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/raw_write -- $B --steps 2 --warmup 1 > $O/write.log 2>&1
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE"
timeout 600 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/raw_sq -- $B --steps 2 --warmup 1 > $O/sq.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/raw_sq4 -- $B --hidden 512 --time-window 60 --batch 8192 --steps 2 --warmup 1 > $O/sq4.log 2>&1
B4="$B --hidden 512 --time-window 60 --batch 8192 --steps 2 --warmup 1"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/raw_fetch4 -- $B4 > $O/fetch4.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/raw_write4 -- $B4 > $O/write4.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocprof_digest.py trace $O/raw_trace $O/kernel_trace.csv
python tools/rocprof_digest.py stats $O/raw_trace $O/kernel_stats.csv
python tools/rocprof_digest.py pmc $O/raw_fetch $O/raw_write vame_amd/libvame_hip.so $O/pmc_hbm_traffic.json \
  --key "gemm_kernel TN M=768 N=256 K=122880 x6 grouped=>gemm_kernel<128,128,2,2,true,true,5,2>@589824"
python tools/rocprof_digest.py pmc $O/raw_fetch4 $O/raw_write4 vame_amd/libvame_hip.so $O/cfg4_pmc_hbm_traffic.json \
  --key "gru_wide_fwd_kernel<512> x2 streams=>gru_wide_fwd_kernel<512>@262144" --key "gru_wide_bwd_kernel<512> x2 streams=>gru_wide_bwd_kernel<512,false>@262144"
python tools/rocprof_digest.py sq $O/raw_sq $O/pmc_sq_cfg2.json "rocprofv3 --kernel-trace --pmc SQ_* GRBM_GUI_ACTIVE -- python bench.py --steps 2 --warmup 1 (BASELINE configs[1])"
python tools/rocprof_digest.py sq $O/raw_sq4 $O/pmc_sq_cfg4.json "same counters, --hidden 512 --time-window 60 --batch 8192 (BASELINE configs[3])"
rm -rf $O/raw_*
# the headline line once more, now that a traffic summary of this very build exists (bench.py looks under profiles/)
cp $O/pmc_hbm_traffic.json profiles/_this_run_pmc_hbm_traffic.json
cp $O/cfg4_pmc_hbm_traffic.json profiles/_this_run_cfg4_pmc_hbm_traffic.json
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-250 $O/bench.json
timeout 300 python bench.py --hidden 512 --time-window 60 --batch 8192 --steps 5 --warmup 2 --no-cpu-baseline --dump-kernels > $O/cfg4.json 2> $O/cfg4.err; cut -c1-250 $O/cfg4.json
rm -f profiles/_this_run_pmc_hbm_traffic.json profiles/_this_run_cfg4_pmc_hbm_traffic.json
ls -la $O
