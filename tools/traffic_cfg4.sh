# HBM traffic of the H = 512 GRU launches per launch form (two rocprofv3 --pmc passes over the configs[3]-shape step): bash tools/traffic_cfg4.sh <outdir-name>
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-traffic4}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B4="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-also --hidden 512 --time-window 60 --batch 8192 --steps 2 --warmup 1"
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/raw_fetch4 -- $B4 > $O/fetch4.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/raw_write4 -- $B4 > $O/write4.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocprof_digest.py pmc $O/raw_fetch4 $O/raw_write4 vame_amd/libvame_hip.so $O/cfg4_pmc_hbm_traffic.json \
  --cycle "gru_wide_skew_fwd_kernel<512>@262144=enc-l0,enc-l1,dec,fut" --cycle "gru_wide_bwd_kernel<512,false>@262144=dec,fut,enc-l1,enc-l0" > /dev/null
rm -rf $O/raw_*
python - <<PY
import json
d = json.load(open("$O/cfg4_pmc_hbm_traffic.json"))
for k, v in d["kernels"].items():
    if "gru_wide" in k: print(f"{k:70s} fetch x2 {2 * v['fetch_kb_per_call'] * 1024 / 1e9:6.2f} GB  write {v['write_kb_per_call'] * 1024 / 1e9:6.2f} GB  total {v['hbm_bytes_per_call_corrected'] / 1e9:6.2f} GB")
PY
grep '^{' $O/fetch4.log | cut -c1-160
