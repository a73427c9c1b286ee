"""Data-parallel path on CPU: 2 gloo ranks run the emulated kernels on disjoint window shards, all-reduce the flat
gradient bucket once, and must hold identical weights equal to a single process that averages the two shard gradients."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_golden


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), VAME_EMU_THREADS="2")
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import harness
    harness.install()
    from model_cases import build_model
    from vame_amd.model.rnn_vae import FusedAdamAMSGrad, allreduce_gradients
    from vame_amd.analysis.pose_segmentation import embed_series
    g = load_golden("step_tiny")
    model, (T, F, Z, H, FS, fut, sp) = build_model(g, "cpu")
    model.train()
    opt = FusedAdamAMSGrad(model, lr=5e-4)
    win = torch.cat([torch.from_numpy(g["x"]), torch.from_numpy(g["xfut"])], 1)
    eps = torch.from_numpy(g["eps"])
    sl = slice(rank * 4, rank * 4 + 4)                           # rank-local batch of 4 windows
    model.loss_step(win[sl].contiguous(), 1.0, beta=1.0, kloss=Z, klmbda=0.1, bsize=4, eps=eps[sl].contiguous())
    gscale = allreduce_gradients(model)
    opt.step(gscale=gscale)
    flat_p, flat_g = model.flat_parameters()
    np.save(os.path.join(out_dir, f"p{rank}.npy"), flat_p.numpy())
    np.save(os.path.join(out_dir, f"g{rank}.npy"), flat_g.numpy() * gscale)
    model.eval()
    emb = load_golden("embed_tiny")
    m2, _ = build_model(emb, "cpu")
    m2.eval()
    shard, (lo, hi) = embed_series(m2, emb["data"][:, :70], batch=16, rank=rank, world=world)   # sharded by window index
    np.save(os.path.join(out_dir, f"e{rank}.npy"), np.concatenate([[lo, hi], shard.numpy().ravel()]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_and_sharded_embedding(emu, tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    p0, p1 = np.load(tmp_path / "p0.npy"), np.load(tmp_path / "p1.npy")
    np.testing.assert_array_equal(p0, p1)                                    # replicas stay in lock-step
    # single-process reference: average of the two shard gradients, same Adam step
    from model_cases import build_model
    from vame_amd.model.rnn_vae import FusedAdamAMSGrad
    g = load_golden("step_tiny")
    model, (T, F, Z, H, FS, fut, sp) = build_model(g, "cpu")
    model.train()
    win = torch.cat([torch.from_numpy(g["x"]), torch.from_numpy(g["xfut"])], 1)
    eps = torch.from_numpy(g["eps"])
    acc = None
    for r in range(world):
        sl = slice(r * 4, r * 4 + 4)
        model.loss_step(win[sl].contiguous(), 1.0, beta=1.0, kloss=Z, klmbda=0.1, bsize=4, eps=eps[sl].contiguous())
        gr = model.flat_parameters()[1].clone()
        acc = gr if acc is None else acc + gr
    np.testing.assert_allclose(np.load(tmp_path / "g0.npy"), (acc / world).numpy(), rtol=1e-6, atol=1e-7)
    model.flat_parameters()[1].copy_(acc)
    opt = FusedAdamAMSGrad(model, lr=5e-4)
    opt.step(gscale=1.0 / world)
    np.testing.assert_allclose(p0, model.flat_parameters()[0].numpy(), rtol=1e-6, atol=1e-7)
    # sharded embedding: the two shards tile [0, N-T) and equal the reference loop's rows
    emb = load_golden("embed_tiny")
    rows = []
    for r in range(world):
        e = np.load(tmp_path / f"e{r}.npy")
        lo, hi = int(e[0]), int(e[1])
        rows.append((lo, hi, e[2:].reshape(hi - lo, -1)))
    assert rows[0][0] == 0 and rows[0][1] == rows[1][0] and rows[1][1] == 70 - 30
    np.testing.assert_allclose(np.concatenate([rows[0][2], rows[1][2]]), emb["latent"][:40], atol=1e-5)


def _failure_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), VAME_EMU_THREADS="2")
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import harness
    harness.install()
    from vame_amd import _lib, ops
    from vame_amd.model.rnn_model import RNN_VAE
    from vame_amd.model.rnn_vae import FusedAdamAMSGrad, allreduce_gradients
    F, Z, H, T, FS, B = 10, 7, 128, 4, 2, 5                       # this shape runs the cooperative (column-split) GRU kernels
    torch.manual_seed(3)
    model = RNN_VAE(2 * T, Z, F, 1, FS, H, H, H, H, 0, 0, 0, False).train()
    opt = FusedAdamAMSGrad(model, lr=5e-4)
    rng = np.random.default_rng(9 + rank)
    win = torch.from_numpy(rng.standard_normal((B, T + FS, F)).astype(np.float32))

    def step():
        model.loss_step(win, 1.0, beta=1.0, kloss=4, klmbda=0.3, bsize=B)
        opt.step(gscale=allreduce_gradients(model))
    step()
    assert model._engine._coop_state is not None and model._engine._coop_state.dirty
    w_good = model.flat_parameters()[0].clone()
    t_good = opt.t
    if rank == 1:
        ops.gru_coop_set_poll_limit(-1)                           # ONLY this rank's cooperative launches report a timeout
    raised = False
    try:
        step()                                                    # the failed step: must be dropped on BOTH ranks
        ops.gru_coop_set_poll_limit(0)
        step()                                                    # the host notices here or one step later -- on both ranks
        step()
        model._engine.check_async_errors()
    except _lib.VameHipError as e:
        raised = "hand-off" in str(e)
    finally:
        ops.gru_coop_set_poll_limit(0)
    w_after = model.flat_parameters()[0].clone()
    np.save(os.path.join(out_dir, f"fail{rank}.npy"), np.concatenate([[float(raised), float(torch.equal(w_after, w_good)), opt.t - t_good],
                                                                      w_after.numpy()]))
    dist.barrier()
    # both ranks raised, so both are here: training continues in lock-step
    step()
    model._engine.check_async_errors()
    np.save(os.path.join(out_dir, f"cont{rank}.npy"), model.flat_parameters()[0].numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_failed_cooperative_launch_on_one_rank_is_dropped_and_raised_on_all(emu, tmp_path):
    """the status word of the cooperative GRU launches rides the gradient all-reduce, so a launch that
    failed on ONE rank drops the optimizer step on EVERY rank (no undefined gradients in the healthy ranks' weights, no
    divergence), every rank raises, and the optimizer's step count excludes the dropped launches."""
    world = 2
    mp.spawn(_failure_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    f0, f1 = np.load(tmp_path / "fail0.npy"), np.load(tmp_path / "fail1.npy")
    for f in (f0, f1):
        assert f[0] == 1.0, "every rank must raise"
        assert f[1] == 1.0, "the failed step (and the ones behind it) never reached the weights"
        assert f[2] == 0.0, "dropped launches are not counted as optimizer steps"
    np.testing.assert_array_equal(f0[3:], f1[3:])
    np.testing.assert_array_equal(np.load(tmp_path / "cont0.npy"), np.load(tmp_path / "cont1.npy"))
    assert np.abs(np.load(tmp_path / "cont0.npy") - f0[3:]).max() > 0


def _bench_line(cmd):
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(VAME_EMU_THREADS="2", OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                                         # rank 0 prints exactly one JSON line
    return json.loads(lines[0])


def _check_two_rank_line(out):
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["value"] > 0
    assert all(np.isfinite(out["config"]["last_loss_terms"]))
    assert out["roofline"] is None                                          # nothing measured off the GPU
    # an N-rank line is complete: rank 0's CPU baseline rides on it and every rank's own time is listed next to the MAX
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and "train steps of B=256" in cb["sample"]
    pr = out["ms_per_step_by_rank"]
    assert len(pr["ranks"]) == 2 and 0 < pr["min"] <= pr["max"] <= out["ms_per_step"] * 1.001
    d = out["distributed"]                                                   # proof that two ranks went through the collective
    assert d["world_size"] == 2 and d["backend"] == "gloo" and d["allreduce_us"] > 0 and d["allreduce_bucket_bytes"] > 0
    assert 0 < d["weak_scaling_eff"] and d["one_rank_leg_ms_per_step"] > 0


BENCH_TINY = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "8", "--hidden", "32", "--time-window", "4", "--cpu-baseline-steps", "1"]


def test_bench_script_under_torchrun_two_ranks(emu):
    """The driver's multi-GPU invocation of bench.py (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`):
    rank / barrier / max-over-ranks / rank-0 JSON plumbing, here with 2 gloo ranks on the host emulator build and a tiny model."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "emu", "harness.py"), os.path.join(ROOT, "bench.py")] + BENCH_TINY
    _check_two_rank_line(_bench_line(cmd))


def test_bench_embed_mode_two_ranks(emu):
    """`bench.py --mode embed --gpus 2` (BASELINE configs[4]: the window index range sharded over ranks, no collective on the data
    path): self-launch, one JSON line, whole-job windows = 2 x per-rank windows."""
    out = _bench_line([sys.executable, os.path.join(ROOT, "tests", "emu", "harness.py"), os.path.join(ROOT, "bench.py"), "--mode", "embed", "--gpus", "2",
                       "--embed-windows", "300", "--hidden", "32", "--time-window", "4"])
    assert out["n_gpus"] == 2 and out["windows"] == 600 and out["value"] > 0 and out["config"]["parallelism"] == "shard2"
    assert out["metric"].endswith("T=4,F=24,h=32") and out["roofline"]["traffic"] is None


def test_bench_script_launches_its_own_ranks(emu):
    """`python bench.py --gpus 2` with no torchrun environment must start the two ranks itself (never a silent 1-rank run), and a
    WORLD_SIZE that contradicts --gpus is an error."""
    import subprocess
    _check_two_rank_line(_bench_line([sys.executable, os.path.join(ROOT, "tests", "emu", "harness.py"), os.path.join(ROOT, "bench.py")] + BENCH_TINY))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "emu", "harness.py"), os.path.join(ROOT, "bench.py")] + BENCH_TINY,
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_collective_path_single_rank_forced(emu, tmp_path_factory):
    """The one-rank run of the whole collective path that `-m gpu` executes through RCCL (tests/test_distributed_gpu.py), here
    through gloo on the host emulator: same script."""
    import json
    import subprocess
    import driver_cases as dc
    root, cfg, g = dc.make_project(tmp_path_factory)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(VAME_EMU_THREADS="4")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dist_world1_script.py"), str(root), "emu"], capture_output=True,
                       text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("WORLD1_OK ")]
    assert len(line) == 1 and json.loads(line[0][len("WORLD1_OK "):])["backend"] == "gloo"


def _driver_worker(rank, world, port, root):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      VAME_EMU_THREADS="2")
    torch.set_num_threads(1)
    import harness
    harness.install()
    import vame_amd as vame
    from vame_amd.model import rnn_vae
    seen = {}
    orig_loss_step = rnn_vae.RNN_VAE.loss_step

    def spy(self, win, *a, **k):                       # record what this rank trained on and with which draws
        seen.setdefault("first_win", win[:2].clone())
        seen.setdefault("eps0", torch.randn(3))       # consumes the rank's stream once, at the first step
        seen["model"] = self
        return orig_loss_step(self, win, *a, **k)
    rnn_vae.RNN_VAE.loss_step = spy
    np.random.seed(0)                                  # same numpy stream on both ranks: disjoint slices of one draw
    vame.train_model(os.path.join(root, "config.yaml"))
    flat = seen["model"].flat_parameters()[0].numpy().copy()
    np.save(os.path.join(root, f"rank{rank}_params.npy"), flat)
    np.save(os.path.join(root, f"rank{rank}_win.npy"), seen["first_win"].numpy())
    np.save(os.path.join(root, f"rank{rank}_eps.npy"), seen["eps0"].numpy())
    vame.pose_segmentation(os.path.join(root, "config.yaml"))
    rnn_vae.shutdown_distributed()


def test_train_and_segment_drivers_two_ranks(emu, tmp_path):
    """vame.train_model() + vame.pose_segmentation() as two gloo ranks (the multi-GPU launch contract: one process per GPU, RANK /
    WORLD_SIZE / MASTER_* from the environment): replicas end with identical weights, they trained on different windows with
    different reparameterisation draws, only rank 0 wrote files, and the sharded embedding equals the single-process one."""
    import json
    import yaml
    g = load_golden("train_model_run")
    cfg = json.loads(str(g["cfg_json"]))
    root = tmp_path
    os.makedirs(root / "data" / "train")
    os.makedirs(root / "model")
    np.save(root / "data" / "train" / "train_seq.npy", g["train_seq"])
    np.save(root / "data" / "train" / "test_seq.npy", g["test_seq"])
    os.makedirs(root / "data" / "vid1")
    np.save(root / "data" / "vid1" / "vid1-PE-seq-clean.npy", g["train_seq"][:, :101])
    cfg.update(project_path=str(root), n_cluster=3, parameterization="kmeans", individual_parameterization=False, video_sets=["vid1"],
               all_data="yes", hmm_trained=False, random_state_kmeans=42, n_init_kmeans=2, max_epochs=7, batch_size=32, model_snapshot=50)
    with open(root / "config.yaml", "w") as f:
        yaml.safe_dump(cfg, f)
    world = 2
    mp.spawn(_driver_worker, args=(world, _free_port(), str(root)), nprocs=world, join=True)
    p0, p1 = np.load(root / "rank0_params.npy"), np.load(root / "rank1_params.npy")
    np.testing.assert_array_equal(p0, p1)                                                   # lock-step replicas
    assert np.abs(np.load(root / "rank0_win.npy") - np.load(root / "rank1_win.npy")).max() > 1e-3      # different windows ...
    assert np.abs(np.load(root / "rank0_eps.npy") - np.load(root / "rank1_eps.npy")).max() > 1e-3      # ... and different draws
    sd = torch.load(root / "model" / "best_model" / "VAME_demo.pkl", map_location="cpu")
    flat_saved = np.concatenate([v.numpy().ravel() for v in sd.values()])
    assert flat_saved.size == sum(v.numel() for v in sd.values())
    losses = np.load(root / "model" / "model_losses" / "train_losses_VAME.npy")
    assert losses.shape == (6,) and np.isfinite(losses).all()
    # n_batches per rank = N // (B * world): both ranks looped the same number of steps (a mismatch would dead-lock the all-reduce)
    out = root / "results" / "vid1" / "VAME" / "kmeans-3"
    lat = np.load(out / "latent_vector_vid1.npy")
    assert lat.shape == (101 - cfg["time_window"], cfg["zdims"])
    from oracle import vame_oracle as vo
    p = {k: v.numpy() for k, v in sd.items()}
    ref = vo.embed_series(p, np.load(root / "data" / "vid1" / "vid1-PE-seq-clean.npy"),
                          vo.Spec(T=30, F=24, Z=30, H=cfg["hidden_size_layer_1"], FS=15), batch=64)
    np.testing.assert_allclose(lat, ref, atol=2e-5)                                         # shards of two ranks, gathered, in order


def test_scale_sweep_script_two_ranks(emu):
    """tools/scale_sweep.sh (one call = the 1 / 2 / ... rank train lines + the sharded embedding line: what the first visit to an 8-GPU node runs,
    SURVEY 8(e)) on the 2-rank gloo harness with a tiny model: three JSON lines, the rank counts and the whole-job window count as asked."""
    import json
    import subprocess
    env = dict(os.environ, VAME_SCALE_PYTHON=f"{sys.executable} {os.path.join(ROOT, 'tests', 'emu', 'harness.py')}", VAME_SCALE_VISIBLE="2",
               VAME_SCALE_EMBED_ARGS="--hidden 32 --time-window 4")
    env = {k: v for k, v in env.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "scale_sweep.sh"), "8", "300", "--"] + BENCH_TINY[2:],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 3, (r.stdout[-800:], r.stderr[-800:])
    assert [ln["n_gpus"] for ln in lines] == [1, 2, 2] and "error" not in "".join(r.stdout.splitlines())
    assert lines[1]["distributed"]["world_size"] == 2 and lines[2]["windows"] == 600 and lines[2]["config"]["parallelism"] == "shard2"
