cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B4="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-also --hidden 512 --time-window 60 --batch 8192 --steps 2 --warmup 1"
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/raw_fetch4 -- $B4 > $O/fetch4.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/raw_write4 -- $B4 > $O/write4.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocprof_digest.py pmc $O/raw_fetch4 $O/raw_write4 vame_amd/libvame_hip.so $O/cfg4_pmc_hbm_traffic.json \
  --cycle "gru_wide_skew_fwd_kernel<512>@262144=enc-l0,enc-l1,dec,fut" --cycle "gru_wide_bwd_kernel<512,false>@262144=dec,fut,enc-l1,enc-l0" \
  --key "gru_wide_fwd_kernel<512> x2 streams gi T=60=>gru_wide_skew_fwd_kernel<512>@262144[enc-l1]" --key "gru_wide_fwd_kernel<512> x2 streams const-gi T=60=>gru_wide_skew_fwd_kernel<512>@262144[dec]" \
  --key "gru_wide_fwd_kernel<512> x2 streams const-gi T=15=>gru_wide_skew_fwd_kernel<512>@262144[fut]" --key "gru_wide_bwd_kernel<512> x2 streams no-dy T=60=>gru_wide_bwd_kernel<512,false>@262144[enc-l1]" \
  --key "gru_wide_bwd_kernel<512> x2 streams dy T=60=>gru_wide_bwd_kernel<512,false>@262144[enc-l0]" \
  --key "gemm_kernel TN M=1536 N=512 K=491520 x6 grouped=>gemm_kernel<128,128,2,2,true,true,5,2>@7077888"
rm -rf $O/raw_*
cp $O/cfg4_pmc_hbm_traffic.json profiles/_this_run_cfg4_pmc_hbm_traffic.json
python bench.py --hidden 512 --time-window 60 --batch 8192 --steps 3 --warmup 1 --no-cpu-baseline --no-also > $O/cfg4.json 2>/dev/null; cut -c1-300 $O/cfg4.json
rm -f profiles/_this_run_*.json
python -c "
import json; j=json.load(open('$O/cfg4_pmc_hbm_traffic.json')); print({k:v for k,v in j.items() if k!='kernels'} if isinstance(j,dict) else j)" | cut -c1-1500
