"""The profile pipeline (tools/rocprof_digest.py) on synthetic rocprofv3 CSVs: the files under profiles/ are what the roofline numbers
are checked against, so the digest itself is tested."""
import csv
import os
import subprocess
import sys

from conftest import ROOT

DIGEST = os.path.join(ROOT, "tools", "rocprof_digest.py")


def _write(path, header, rows):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(header)
        w.writerows(rows)


def test_kernel_trace_digest_separates_launches_of_one_kernel_by_grid(tmp_path):
    hdr = ["Kind", "Kernel_Name", "Start_Timestamp", "End_Timestamp", "Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z", "Workgroup_Size_X"]
    g = "void gemm_kernel<128, 128, 2, 2, true, true, 5, 2>(GemmParams)"
    rows = [["KERNEL_DISPATCH", g, 1000, 3500, 589824, 1, 1, 256],                       # the grouped launch: 2.5 us here
            ["KERNEL_DISPATCH", g, 4000, 4400, 393216, 1, 1, 256],
            ["KERNEL_DISPATCH", "void gru_seq_bwd_kernel<256, 0>(GruBwdParams)", 5000, 6000, 131072, 1, 1, 512],
            ["KERNEL_DISPATCH", g, 7000, 9700, 589824, 1, 1, 256]]
    _write(str(tmp_path / "raw" / "pid" / "1_kernel_trace.csv"), hdr, rows)
    out_t, out_s = str(tmp_path / "trace.csv"), str(tmp_path / "stats.csv")
    subprocess.run([sys.executable, DIGEST, "trace", str(tmp_path / "raw"), out_t], check=True, capture_output=True)
    subprocess.run([sys.executable, DIGEST, "stats", str(tmp_path / "raw"), out_s], check=True, capture_output=True)
    tr = list(csv.DictReader(open(out_t)))
    assert [r["kernel"] for r in tr] == ["gemm_kernel<128,128,2,2,true,true,5,2>"] * 2 + ["gru_seq_bwd_kernel<256,0>", "gemm_kernel<128,128,2,2,true,true,5,2>"]
    assert [float(r["start_us"]) for r in tr] == [0.0, 3.0, 4.0, 6.0] and [float(r["duration_us"]) for r in tr] == [2.5, 0.4, 1.0, 2.7]
    st = {(r["kernel"], int(r["grid_threads"])): r for r in csv.DictReader(open(out_s))}
    big = st[("gemm_kernel<128,128,2,2,true,true,5,2>", 589824)]
    assert int(big["calls"]) == 2 and float(big["avg_us"]) == 2.6 and float(big["min_us"]) == 2.5 and float(big["max_us"]) == 2.7
    assert int(st[("gemm_kernel<128,128,2,2,true,true,5,2>", 393216)]["calls"]) == 1


def test_vame_alias_shim_installer(tmp_path):
    """tools/install_vame_alias.py writes a `vame` package whose import turns into the vame_amd alias -- checked in a fresh interpreter
    with only that directory added to sys.path (the reference's own script header `import vame` then works unmodified)."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "install_vame_alias.py"), str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    code = ("import sys; sys.path[:0] = [%r, %r]; import vame; import vame_amd; from vame.model.rnn_vae import RNN_VAE; "
            "assert vame.train_model is vame_amd.train_model and vame.__vame_amd_alias__; print('ALIAS_OK')" % (str(tmp_path), ROOT))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert "ALIAS_OK" in r.stdout, r.stderr[-1500:]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "install_vame_alias.py"), str(tmp_path)], capture_output=True, text=True)
    assert r.returncode != 0 and "exists" in r.stderr              # never silently overwrites an existing `vame` package


def test_pmc_digest_labels_the_launch_forms_of_one_kernel(tmp_path):
    """tools/rocprof_digest.py pmc --cycle: launches of one (kernel, grid) that stream different things (the four forward GRU launches
    of a step) are summarised per form, FETCH_SIZE doubled (gfx950), and a bench key can name one form."""
    import json
    hdr = ["Dispatch_Id", "Grid_Size", "Kernel_Name", "Counter_Name", "Counter_Value"]
    k = "void gru_wide_skew_fwd_kernel<512>(GruFwdParams)"
    fetch = [[10 + i, 262144, k, "FETCH_SIZE", v] for i, v in enumerate([100.0, 300.0, 50.0, 10.0] * 2)]      # KB; dispatch order l0, l1, dec, fut
    write = [[10 + i, 262144, k, "WRITE_SIZE", v] for i, v in enumerate([1000.0, 1000.0, 900.0, 200.0] * 2)]
    other = [[99, 589824, "void gemm_kernel<128, 128, 2, 2, true, true, 5, 2>(GemmParams)", "FETCH_SIZE", 7.0]]
    _write(str(tmp_path / "f" / "p" / "1_counter_collection.csv"), hdr, fetch + other)
    _write(str(tmp_path / "w" / "p" / "1_counter_collection.csv"), hdr, write)
    lib = os.path.join(ROOT, "tests", "emu", "libvame_emu.so")
    if not os.path.exists(lib):
        subprocess.run(["make", "-s", "tests/emu/libvame_emu.so"], cwd=ROOT, check=True)
    out = str(tmp_path / "t.json")
    subprocess.run([sys.executable, DIGEST, "pmc", str(tmp_path / "f"), str(tmp_path / "w"), lib, out,
                    "--cycle", "gru_wide_skew_fwd_kernel<512>@262144=enc-l0,enc-l1,dec,fut",
                    "--key", "gru_wide_fwd_kernel<512> x2 streams gi T=60=>gru_wide_skew_fwd_kernel<512>@262144[enc-l1]",
                    "--key", "gemm=>gemm_kernel<128,128,2,2,true,true,5,2>@589824"], check=True, capture_output=True)
    j = json.load(open(out))
    l1 = j["kernels"]["gru_wide_skew_fwd_kernel<512> grid=262144 [enc-l1]"]
    assert l1["calls"] == 2 and l1["fetch_kb_per_call"] == 300.0 and l1["hbm_bytes_per_call_corrected"] == (2 * 300 + 1000) * 1024
    assert j["kernels"]["gru_wide_skew_fwd_kernel<512> grid=262144 [fut]"]["write_kb_per_call"] == 200.0
    assert j["by_bench_key"]["gru_wide_fwd_kernel<512> x2 streams gi T=60"]["hbm_bytes_per_launch_corrected"] == 1600 * 1024
    assert j["by_bench_key"]["gemm"]["fetch_kb_per_call"] == 7.0 and j["source_id"]
