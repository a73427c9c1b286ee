# FETCH_SIZE / WRITE_SIZE per kernel of one tools/fwd_table.py run (environment selects H / forms / stash): bash tools/traffic_probe.sh <B> <T>
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/tp_$c
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/tp_$c -- python $GRAFT_REPO_ROOT/tools/fwd_table.py $1 $2 > /tmp/tp_$c.log 2>&1
done
python - <<PY
import csv, glob, collections
csv.field_size_limit(1 << 30)
res = collections.defaultdict(lambda: [0, 0.0, 0.0])
for c, idx in (("FETCH_SIZE", 1), ("WRITE_SIZE", 2)):
    for f in glob.glob(f"/tmp/tp_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c and "gru_" in r["Kernel_Name"] and "pack" not in r["Kernel_Name"]:
                k = (r["Kernel_Name"].split("(")[0][-60:], r["Grid_Size"])
                res[k][idx] += float(r["Counter_Value"]); res[k][0] += (c == "FETCH_SIZE")
for k, (n, f, w) in res.items():
    n = max(n, 1)
    print(f"{k[0]:60s} grid={k[1]:8s} calls={n:4d}  fetch x2 {2 * f / n * 1024 / 1e9:7.3f} GB  write {w / n * 1024 / 1e9:7.3f} GB per launch")
PY
