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
