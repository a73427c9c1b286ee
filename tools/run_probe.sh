cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, os
sys.argv = [sys.argv[0], "10"]
sys.path.insert(0, "tools")
import microbench as mb
for T in (8, 15, 30, 60, 120):
    for ns in (2,):
        f, b = mb.bench_gru(256, 4096, T, ns, quiet=True)
        print(f"B=4096 T={T:3d} streams={ns}: fwd {f:8.1f} us ({f/T:6.2f} us/step)  bwd {b:8.1f} us ({b/T:6.2f} us/step)", flush=True)
for B in (2048, 4096, 8192, 16384):
    f, b = mb.bench_gru(256, B, 30, 2, quiet=True)
    print(f"B={B} T=30 streams=2: fwd {f:8.1f} us  bwd {b:8.1f} us   per 4096 rows: fwd {f*4096/B:7.1f} bwd {b*4096/B:7.1f}", flush=True)
PY
