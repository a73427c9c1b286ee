"""The collective path on the MI355X: `init_process_group("nccl", device_id=...)` = RCCL, the all-reduce of the real
flat gradient bucket, train_model() / pose_segmentation() under the process group (barriers, rank-averaged statistics, all-gather)
and the group shutdown -- executed with ONE rank, which is what a 1-GPU box allows.  The N > 1 arithmetic (replicas in lock-step,
shards, failure containment) is covered by the two-rank gloo tests in test_distributed_cpu.py."""
import json
import os
import subprocess
import sys

import pytest

import driver_cases as dc
from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_rccl_path_single_rank(tmp_path_factory, hip):
    root, cfg, g = dc.make_project(tmp_path_factory)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dist_world1_script.py"), str(root)], capture_output=True, text=True,
                       timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("WORLD1_OK ")]
    assert len(line) == 1, r.stdout[-1500:]
    info = json.loads(line[0][len("WORLD1_OK "):])
    assert info["backend"] == "nccl" and info["library"].startswith("RCCL") and info["bucket_bytes"] > 0


def test_bench_refuses_more_ranks_than_gpus(hip):
    """`python bench.py --gpus 2` on a 1-GPU box starts two ranks itself and both refuse (never a silent 1-rank number)."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("box has several GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True,
                       text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "only 1 GPU" in r.stderr


def test_bench_line_contract(hip):
    """`python bench.py` on the MI355X (short: 3 timed steps, no CPU baseline): ONE JSON line with the contract's fields, the roofline
    block measured live (dominant kernel, MFMA classes, the HBM-bound kernels of the step) and the three-region spread."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2", "--no-cpu-baseline"], capture_output=True,
                       text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["warmup"] == 2 and j["dtype"] == "f32" and j["unit"] == "windows/s" and j["vs_baseline"] is None
    assert j["value"] > 50_000 and abs(j["value"] - 4096 * 3 / (j["ms_per_step"] * 3e-3)) / j["value"] < 1e-3
    roof = j["roofline"]
    assert roof["bound"] == "mfma" and roof["peak"] == 157.3 and 0.3 < roof["frac"] < 1.0 and 0.3 < roof["step_frac"] < 1.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    for c in ("gru_seq_fwd_kernel<256>", "gru_seq_bwd_kernel<256>", "gemm_kernel TN", "gemm_kernel NT", "gemm_kernel NN"):
        assert roof["by_class"][c]["tflops"] > 20, c
    for c in ("window_gather_kernel", "head_stream_kernel", "timesum_kernel"):      # (round 6: the streaming output head replaces the MSE launch)
        assert roof["by_class"][c]["bound"] == "hbm" and 0 < roof["by_class"][c]["frac"] < 1, c
    assert j["repeat_spread"]["regions"] == 3 and j["repeat_spread"]["min"] <= j["value"] <= j["repeat_spread"]["max"] * 1.0001
    also = j["also"]
    assert also["configs3_h512_t60_b8192"]["value"] > 5_000 and also["configs4_embed_1gpu"]["value"] > 100_000
    # round 4: the box's sustained shader clock rides on every roofline block (vame_clock_stamp around the dominant launches / the
    # timed region) with the fraction restated against it, and the reference's stock-config batch is a leg of the default line whose
    # per-class table includes the cooperative GRU launches that batch runs on
    for blk in (roof, also["configs3_h512_t60_b8192"]["roofline"], also["batch256"]["roofline"]):
        assert 1500 < blk["clock_mhz"] < 2600 and 1500 < blk["clock_mhz_timed_region"] < 2600, blk["clock_mhz"]
        assert abs(blk["frac_at_clock"] - blk["frac"] * 2400.0 / blk["clock_mhz"]) < 5e-3
        assert abs(blk["step_frac_at_clock"] - blk["step_frac"] * 2400.0 / blk["clock_mhz_timed_region"]) < 5e-3
    assert 1500 < also["configs4_embed_1gpu"]["roofline"]["clock_mhz"] < 2600
    assert all(1500 < v["clock_mhz"] < 2600 for k, v in roof["by_class"].items() if v.get("bound") != "hbm")
    b256 = also["batch256"]
    assert b256["value"] > 30_000 and b256["steps"] == 30 and "batch=256" in b256["config"]["workload"]
    assert {"gru_coop_fwd_kernel<256>", "gru_coop_bwd_kernel<256>"} <= set(b256["roofline"]["by_class"])
    # round 5: narrow contractions (a dimension <= 32) are rows of the HBM table; no negative times; the opt-in split-bf16 weight-gradient
    # contraction is an `also` leg with both step times, its launches priced against bf16 dense / 6, and the headline stays on f32-input MFMA
    assert any(k.startswith("gemm_kernel") and "narrow" in k and v["bound"] == "hbm" for k, v in roof["by_class"].items())
    assert roof["non_mfma_ms_per_step"] >= 0 and "hidden_by_overlaps_ms_per_step" in roof
    assert not any("bf16x6" in k for k in roof["by_class"]) and "bf16x6" not in roof["kernel"]
    sg = also["split_gemm"]
    assert abs(sg["peak"] - 2500.0 / 6) < 0.1 and sg["default_ms_per_step"] == j["ms_per_step"]
    for name in ("one_accumulator", "two_accumulators"):
        leg = sg[name]
        assert leg["value"] > 50_000 and len(leg["launches"]) == 5, leg       # three grouped weight-gradient shapes + the layer-1 projection and its data gradient
        assert all(0 < v["frac"] <= 1 and v["tflops"] > 60 for v in leg["launches"].values()), leg["launches"]
    # ... and the embedding leg / the stock batch as a replayed graph with the options on, beside their default-path lines
    assert sg["embed_one_accumulator"]["value"] > 0.9 * also["configs4_embed_1gpu"]["value"]
    assert sg["batch256_one_accumulator"]["value"] > 0.9 * also["batch256"]["value"] and sg["batch256_one_accumulator"]["ms_per_step"] > 0
