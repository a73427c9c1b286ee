# HBM traffic per launch of the configs[3] grouped weight-gradient GEMM at several split-K factors: bash tools/traffic_gemm_cfg3.sh <outdir-name> [split-K ...]
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-traffic_gemm3}
shift
mkdir -p $O
python tools/gemm_group_cfg3.py "$@" > $O/times.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/tg_$c
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/tg_$c -- python $GRAFT_REPO_ROOT/tools/gemm_group_cfg3.py "$@" > /tmp/tg_$c.log 2>&1
done
python - <<PY > $O/traffic.txt
import csv, glob, collections
csv.field_size_limit(1 << 30)
res = collections.defaultdict(lambda: [0, 0.0, 0.0])
for c, idx in (("FETCH_SIZE", 1), ("WRITE_SIZE", 2)):
    for f in glob.glob(f"/tmp/tg_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c and "gemm_kernel" in r["Kernel_Name"]:
                k = r["Grid_Size"]
                res[k][idx] += float(r["Counter_Value"]); res[k][0] += (c == "FETCH_SIZE")
for k, (n, f, w) in sorted(res.items(), key=lambda kv: int(kv[0])):
    n = max(n, 1)
    print(f"gemm_kernel grid={k:>9s} calls={n:3d}  fetch x2 {2 * f / n * 1024 / 1e9:7.2f} GB  write {w / n * 1024 / 1e9:6.2f} GB  total {(2 * f + w) / n * 1024 / 1e9:7.2f} GB per launch (FETCH_SIZE x2: MI355X_MICROARCH.md gfx950 correction)")
PY
cat $O/times.txt $O/traffic.txt
