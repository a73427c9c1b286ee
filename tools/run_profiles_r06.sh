# Round-6 profile pass on the MI355X (tools/run_profiles.sh without the passes that need the tuning / probe builds -- those kernels did not change): bench line, rocprofv3 kernel trace, PMC HBM traffic (separate FETCH / WRITE passes), SQ counters.
# usage (GPU box): bash tools/run_profiles.sh <outdir-name>      -> gpurun_out/<outdir-name>/  (copy what is to be judged into profiles/)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-prof}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-also"
B4="$B --hidden 512 --time-window 60 --batch 8192 --steps 2 --warmup 1"
BE="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --mode embed --embed-windows 500000"
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw_trace -- $B --steps 10 --warmup 3 > $O/trace.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/raw_fetch -- $B --steps 2 --warmup 1 > $O/fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/raw_write -- $B --steps 2 --warmup 1 > $O/write.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/raw_sq -- $B --steps 2 --warmup 1 > $O/sq.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/raw_fetch4 -- $B4 > $O/fetch4.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/raw_write4 -- $B4 > $O/write4.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/raw_fetche -- $BE > $O/fetche.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/raw_writee -- $BE > $O/writee.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocprof_digest.py trace $O/raw_trace $O/kernel_trace.csv
python tools/rocprof_digest.py stats $O/raw_trace $O/kernel_stats.csv
cp "$(find $O/raw_trace -name '*kernel_stats.csv' | head -1)" $O/rocprof_kernel_stats.csv      # rocprofv3's own --stats summary, untouched
python tools/rocprof_digest.py pmc $O/raw_fetch $O/raw_write vame_amd/libvame_hip.so $O/pmc_hbm_traffic.json \
  --key "gemm_kernel TN M=768 N=256 K=122880 x6 grouped=>gemm_kernel<128,128,2,2,true,true,5,2>@589824" \
  --cycle "gru_ws_bwd_kernel<256,0>@131072=enc-l1,enc-l0" \
  --key "gru_seq_bwd_kernel<256> x2 streams no-dy T=30=>gru_ws_bwd_kernel<256,0>@131072[enc-l1]" --key "gru_seq_bwd_kernel<256> x2 streams dy T=30=>gru_ws_bwd_kernel<256,0>@131072[enc-l0]" --key "gru_seq_fwd_kernel<256> x2 streams gi T=30=>gru_skew_fwd_kernel<256,false>@131072" \
  --key "gru_seq_fwd_kernel<256> x2 streams xin T=30=>gru_skew_fwd_kernel<256,true>@131072" --key "gru_seq_fwd_kernel<256> x4 streams const-gi T=30=>gru_seq_fwd_kernel<256,0,false>@262144"
python tools/rocprof_digest.py pmc $O/raw_fetch4 $O/raw_write4 vame_amd/libvame_hip.so $O/cfg4_pmc_hbm_traffic.json \
  --cycle "gru_wide_skew_fwd_kernel<512>@262144=enc-l0,enc-l1,dec,fut" --cycle "gru_wide_bwd_kernel<512,false>@262144=dec,fut,enc-l1,enc-l0" \
  --key "gru_wide_fwd_kernel<512> x2 streams gi T=60=>gru_wide_skew_fwd_kernel<512>@262144[enc-l1]" --key "gru_wide_fwd_kernel<512> x2 streams const-gi T=60=>gru_wide_skew_fwd_kernel<512>@262144[dec]" \
  --key "gru_wide_fwd_kernel<512> x2 streams const-gi T=15=>gru_wide_skew_fwd_kernel<512>@262144[fut]" --key "gru_wide_bwd_kernel<512> x2 streams no-dy T=60=>gru_wide_bwd_kernel<512,false>@262144[enc-l1]" \
  --key "gru_wide_bwd_kernel<512> x2 streams dy T=60=>gru_wide_bwd_kernel<512,false>@262144[enc-l0]" \
  --key "gemm_kernel TN M=1536 N=512 K=491520 x6 grouped=>gemm_kernel<128,128,2,2,true,true,5,2>@7077888"
python tools/rocprof_digest.py pmc $O/raw_fetche $O/raw_writee vame_amd/libvame_hip.so $O/embed_pmc_hbm_traffic.json \
  --key "gru_seq_fwd_kernel<256> x2 streams gi T=30 embed=>gru_skew_fwd_kernel<256,false>@524288"
python tools/rocprof_digest.py sq $O/raw_sq $O/pmc_sq_cfg2.json "rocprofv3 --kernel-trace --pmc SQ_* GRBM_GUI_ACTIVE -- python bench.py --no-also --steps 2 --warmup 1 (BASELINE configs[1])"
rm -rf $O/raw_*
# the bench line last, with the traffic summaries of this very build visible to it (bench.py looks under profiles/ for a matching source id)
for f in pmc_hbm_traffic cfg4_pmc_hbm_traffic embed_pmc_hbm_traffic; do cp $O/$f.json profiles/_this_run_$f.json; done
timeout 600 python bench.py --dump-kernels > $O/bench.json 2> $O/bench.err; cut -c1-250 $O/bench.json
rm -f profiles/_this_run_*.json
# the other committed lines of a round: stock-config batch, 10 M-window embedding, 100 timed steps, kernel tables, small-batch overlap A/B
python bench.py --batch 256 --steps 200 --warmup 30 --no-cpu-baseline --no-also --graph > $O/b256.json 2>/dev/null
python bench.py --batch 256 --steps 200 --warmup 30 --no-cpu-baseline --no-also > $O/b256_eager.json 2>/dev/null
python bench.py --mode embed --embed-windows 10000000 --no-cpu-baseline > $O/embed10m.json 2>/dev/null
python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-also > $O/bench_100steps.json 2>/dev/null
python tools/step_ab.py 256 join=nuc_join_before_coop:1 nojoin=nuc_join_before_coop:0 inline=nuc_side:0 2>&1 | grep -v amdgpu > $O/b256_overlap.txt
ls -la $O
# round 5: the opt-in split-bf16 weight-gradient contraction (error table vs float64, timing; ablations and the other mappings on the tuning build),
# the MFMA / VALU overlap probe, throughput over batch and hidden sizes, the configs[3] grouped GEMM's traffic over split-K, the stock-batch
# step as a replayed hipGraph (per-dispatch trace), and the kernel names of a whole train_model() + pose_segmentation() run
python tools/shape_table.py both 2>&1 | grep -v amdgpu > $O/shape_table.txt
python tools/narrow_gemms.py 2>&1 | grep -v amdgpu > $O/narrow_gemms.txt
python tools/head_bench.py 4096 256 2>&1 | grep -v amdgpu > $O/head_bench.txt
bash tools/trace_b256.sh ${1:-prof}/b256_graph 256 --graph > /dev/null 2>&1; cp $O/b256_graph/kernel_trace.csv $O/b256_graph_kernel_trace.csv; rm -rf $O/b256_graph
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw_demo -- python $GRAFT_REPO_ROOT/tools/demo_project.py 256 train-only > $O/demo_trace.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocprof_digest.py stats $O/raw_demo $O/train_model_kernel_stats.csv; rm -rf $O/raw_demo
python tools/demo_project.py 256 2>&1 | grep DEMO > $O/demo_project.txt; python tools/demo_project.py 4096 2>&1 | grep DEMO >> $O/demo_project.txt
ls -la $O
