"""Interleaved A/B of two builds of libvame_hip.so on the bench line (one box, alternating processes):
python tools/lib_ab.py <libA.so> <libB.so> [rounds] [bench args ...]   e.g.  python tools/lib_ab.py tools/libvame_hip_old.so vame_amd/libvame_hip.so 3"""
import json, os, subprocess, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
libs = [os.path.abspath(sys.argv[1]), os.path.abspath(sys.argv[2])]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
extra = sys.argv[4:] or ["--no-also", "--no-cpu-baseline", "--steps", "30", "--warmup", "5"]
res = {l: [] for l in libs}
for _ in range(rounds):
    for l in libs:
        code = (f"import sys; sys.path.insert(0, {R!r}); from vame_amd import _lib; _lib._lib = _lib._bind({l!r}); import bench; "
                f"sys.argv = ['bench.py'] + {extra!r}; bench.main()")
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=R).stdout
        j = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])
        r = j["roofline"]
        res[l].append(j["value"])
        print(f"{os.path.basename(l):24s} {j['value']:10.1f} windows/s {j['ms_per_step']:7.3f} ms  dominant {r['achieved']:6.1f} TF @ {r['clock_mhz']} MHz  "
              + " ".join(f"{k.split(' ')[-1] if k.startswith('gemm') else k[:11]}={v['tflops']}" for k, v in r["by_class"].items() if "tflops" in v), flush=True)
for l in libs:
    v = sorted(res[l]); print(os.path.basename(l), "median", v[len(v) // 2])
