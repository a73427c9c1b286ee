"""Training driver of the RNN-VAE on MI355X -- drop-in for vame/model/rnn_vae.py.

Kept from the reference (file:line there): the loss functions :35-60, `kl_annealing` :63-81,
`gaussian` :84-91, the epoch loops `train` :94-164 / `test` :167-210 (including their quirks: epoch
means divide by the last batch index, the scheduler steps on the last batch's loss), and the
`train_model(config)` driver :213-413 (Adam-AMSGrad, ReduceLROnPlateau, best-model / snapshot rules,
the eight loss arrays).  What changed is how a step executes: windows are gathered on the device,
forward + loss + backward run as fused HIP kernels (RNN_VAE.loss_step), gradients are all-reduced
with one RCCL call over the flat bucket when several ranks run, and Adam is one fused kernel.
Loss scalars stay on the device; the host syncs once per epoch instead of four times per step.
"""
import os
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist
from torch import nn
from torch.optim.lr_scheduler import ReduceLROnPlateau, StepLR

from .. import _lib, ops
from ..util.auxiliary import read_config
from .dataloader import SEQUENCE_DATASET, DeviceWindowLoader
from .rnn_model import RNN_VAE, RNN_VAE_LEGACY  # noqa: F401  (evaluate.py:20 imports RNN_VAE from here)


# ------------------------------------------------------------------------------------ losses (API)
def reconstruction_loss(x, x_tilde, reduction):
    return nn.functional.mse_loss(x_tilde, x, reduction=reduction)


def future_reconstruction_loss(x, x_tilde, reduction):
    return nn.functional.mse_loss(x_tilde, x, reduction=reduction)


class _NuclearNorm(torch.autograd.Function):
    """lmbda * sum_k sqrt(eig_k(latent^T latent / batch_size)) on the (Z,Z) Gram -- equal to the
    reference's (B,B) SVD form (rnn_vae.py:45-50) at O(B Z^2) instead of O(B^3)."""

    @staticmethod
    def forward(ctx, latent, kloss, lmbda, batch_size):
        lat = latent.detach().to(torch.float32).contiguous()
        B, Z = lat.shape
        dev = lat.device
        G, Minv, loss = torch.empty(Z, Z, device=dev), torch.empty(Z, Z, device=dev), torch.zeros(1, device=dev)
        sk = max(1, min(64, B // 256))
        ws = torch.empty(sk * Z * Z, device=dev) if sk > 1 else None
        ops.gemm(Z, Z, B, ops.Operand(lat, Z), 1, ops.Operand(lat, Z), 1, G, Z, splitk=sk, ws=ws)
        ops.nuclear(G, Z, int(kloss), B, float(lmbda), float(batch_size), loss, 0, Minv)
        ctx.save_for_backward(lat, Minv)
        return loss[0].clone()

    @staticmethod
    def backward(ctx, g):
        lat, Minv = ctx.saved_tensors
        B, Z = lat.shape
        d = torch.empty(B, Z, device=lat.device)
        ops.gemm(B, Z, Z, ops.Operand(lat, Z), 0, ops.Operand(Minv, Z), 1, d, Z)
        return d * g, None, None, None


def cluster_loss(H, kloss, lmbda, batch_size):
    """H = latent.T (Z,B) as in the reference's call sites (rnn_vae.py:126,137,190,197)."""
    return _NuclearNorm.apply(H.T, kloss, lmbda, batch_size)


def kullback_leibler_loss(mu, logvar):
    return -0.5 * torch.mean(1 + logvar - mu.pow(2) - logvar.exp())


def kl_annealing(epoch, kl_start, annealtime, function):
    if epoch > kl_start:
        if function == 'linear':
            new_weight = min(1, (epoch - kl_start) / (annealtime))
        elif function == 'sigmoid':
            new_weight = float(1 / (1 + np.exp(-0.9 * (epoch - annealtime))))
        else:
            raise NotImplementedError('currently only "linear" and "sigmoid" are implemented')
        return new_weight
    return 0


def gaussian(ins, is_training, seq_len, std_n=0.8):
    """Optional input noise (cfg['noise'], off by default): ins + N(0,1) * 0.8 * std over time."""
    if is_training:
        emp_std = ins.std(1, keepdim=True) * std_n          # unbiased std over time per (sample, feature), rnn_vae.py:86
        return ins + torch.randn_like(ins) * emp_std
    return ins


# ------------------------------------------------------------------------------------ optimizer
class FusedAdamAMSGrad(torch.optim.Optimizer):
    """torch.optim.Adam(amsgrad=True) semantics (rnn_vae.py:332) as ONE kernel over the model's flat
    parameter bucket; exposes param_groups so torch LR schedulers drive it unchanged."""

    def __init__(self, model, lr):
        super().__init__(list(model.parameters()), dict(lr=lr, betas=(0.9, 0.999), eps=1e-8))
        self.model = model
        self.flat_p, self.flat_g = model.flat_parameters()
        self.m = torch.zeros_like(self.flat_p)
        self.v = torch.zeros_like(self.flat_p)
        self.vmax = torch.zeros_like(self.flat_p)
        self.t = 0                                           # host mirror of the step count (the device's is authoritative: `state`)
        self.dropped = torch.zeros(1, dtype=torch.int32, device=self.flat_p.device)   # launches the abort word turned into no-ops
        # {learning rate (float bits), updates applied, ticket, -}: the kernel reads lr and the bias-correction step from here and counts itself,
        # so its argument list never changes (vame_adam_amsgrad_f32 `state`; a captured step graph replays it, GraphedTrainStep)
        # (named dev_state: torch.optim.Optimizer owns `self.state`, the per-parameter dict that state_dict() walks)
        self.dev_state = torch.zeros(4, dtype=torch.int32, device=self.flat_p.device)
        self._lr_on_device = None
        self._hooked = None

    def sync_lr(self):
        """Write the current learning rate to the device word the kernel reads (one tiny fill, only when a scheduler changed it)."""
        lr = float(self.param_groups[0]["lr"])
        if lr != self._lr_on_device:
            self.dev_state.view(torch.float32)[0:1].fill_(lr)
            self._lr_on_device = lr

    def _forget_dropped_steps(self):
        """Runs when a device-side failure is raised (ops.CoopState.on_error; the host is synchronised there): the launches that
        did nothing must not count as steps of the bias correction, should the caller catch the error and go on training."""
        self.t -= int(self.dropped.item())
        self.dropped.zero_()

    @torch.no_grad()
    def step(self, closure=None, gscale=1.0):
        flat_p, flat_g = self.model.flat_parameters()
        if flat_p is not self.flat_p:
            raise RuntimeError("the model was moved after the optimizer was built")
        self.t += 1
        self.sync_lr()
        g = self.param_groups[0]
        eng = self.model._engine
        flag = eng.abort_flag() if eng is not None else None
        if flag is not None and self._hooked is not eng._coop_state:
            eng._coop_state.on_error.append(self._forget_dropped_steps)
            self._hooked = eng._coop_state
        ops.adam_amsgrad(flat_p, flat_g, self.m, self.v, self.vmax, flat_p.numel(), g["lr"], self.t, gscale=gscale,
                         beta1=g["betas"][0], beta2=g["betas"][1], eps=g["eps"], abort_flag=flag,
                         dropped=self.dropped if flag is not None else None, state=self.dev_state)

    def zero_grad(self, set_to_none=False):
        pass  # every backward overwrites the whole bucket

    def state_dict(self):
        """The moments live in three flat buffers beside the parameter bucket, not in torch's per-parameter `state`: a checkpoint
        carries them (and the step count, read back from the device word the kernel counts in) under "fused"."""
        sd = super().state_dict()
        sd["fused"] = dict(m=self.m.clone(), v=self.v.clone(), vmax=self.vmax.clone(), t=int(self.dev_state[1].item()))
        return sd

    def load_state_dict(self, state_dict):
        state_dict = dict(state_dict)
        fused = state_dict.pop("fused", None)
        super().load_state_dict(state_dict)
        if fused is not None:
            for dst, key in ((self.m, "m"), (self.v, "v"), (self.vmax, "vmax")):
                dst.copy_(fused[key].to(dst.device))
            self.t = int(fused["t"])
            self.dev_state[1:2].fill_(self.t)
        self._lr_on_device = None                            # param_groups may carry another learning rate: upload it before the next launch


class GraphedTrainStep:
    """One training step -- window gather -> RNN_VAE.loss_step (forward, losses, BPTT, weight gradients) -> fused Adam -- captured ONCE as a
    hipGraph and replayed per batch.  At the reference's stock batch (256, vame/initialize_project/new.py:111) a step is ~160 short launches
    and the host needs as long to enqueue them as the GPU to run them; a replay is one call.  What makes the step replayable: every launch
    takes only device pointers and shape constants -- the cooperative GRU launches keep their epoch on the device, the reparameterisation
    draws eps from a device-side Philox counter, Adam reads the learning rate and its step number from the device, the loss kernel
    accumulates the epoch's statistics on the device -- and the window starts are uploaded into a fixed buffer before each replay.
    Single rank only (the gradient all-reduce stays outside a graph).  Results are bit-identical to the eager step
    (tests/test_train_driver_gpu.py::test_graphed_step_is_bit_identical_to_eager)."""

    MEASURE_STEPS = 20

    def __init__(self, model, optimizer, loader, acc, warmup=3, choice=None, **loss_kwargs):
        """choice: None = always replay (hip_graph: true).  A dict = hip_graph "auto": the first MEASURE_STEPS steps after the warm-up run eagerly and the
        next MEASURE_STEPS as replays, each span between two events; the faster form is kept and remembered in the dict under the loader's batch size
        (both forms give the same bits, so the mix changes nothing but the time)."""
        assert isinstance(optimizer, FusedAdamAMSGrad) and not _dist_active()
        self.model, self.opt, self.loader, self.acc, self.kw = model, optimizer, loader, acc, loss_kwargs
        self.dev = model.flat_parameters()[0].device
        self.graph = None
        self.terms = None
        self._seen = None
        self.warmup = warmup
        self.choice = choice
        self._ev, self._n = [], 0

    def _eager(self):
        win = self.loader.gather_static()
        terms = self.model.loss_step(win, acc=self.acc, **self.kw)
        self.opt.step()
        return terms

    def _measure(self):
        """hip_graph "auto": which form this call takes while the decision is open -- MEASURE_STEPS eager steps, then MEASURE_STEPS replays (the
        capture in between is outside both spans), then the comparison (the one host synchronisation of the protocol)."""
        n, N = self._n, self.MEASURE_STEPS
        self._n += 1
        if n in (0, N, N + 1, 2 * N + 1):
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self._ev.append(ev)
        if n < N:
            return "eager"
        if n <= 2 * N:                                   # (n == N: this call captures and replays once, untimed)
            return "graph"
        self._ev[3].synchronize()
        t_eager, t_graph = self._ev[0].elapsed_time(self._ev[1]), self._ev[2].elapsed_time(self._ev[3])
        self.choice[self.loader.B] = "graph" if t_graph <= t_eager else "eager"
        self.choice[("ms_per_step", self.loader.B)] = (t_eager / N, t_graph / N)
        return self.choice[self.loader.B]

    def _capture(self):
        eng = self.model._ensure_engine(touch=False)
        self.opt.sync_lr()                               # outside the capture: a learning-rate fill baked into the graph would undo every later change
        self.graph = torch.cuda.CUDAGraph()
        eng.capturing = True
        try:
            # "relaxed": the launches' host side makes harmless runtime queries (occupancy, kernel attributes) that a stricter mode refuses
            with torch.cuda.graph(self.graph, capture_error_mode="relaxed"):
                self.terms = self._eager()
        finally:
            eng.capturing = False
        self.opt.t -= 1                                  # (the capture ran the host part of one step without executing it)
        self._seen = (eng, ops.ALLOC_GEN[0], self.model._flat_p.data_ptr())

    def __call__(self, starts):
        """One step on the windows starting at `starts`; returns the step's [rec, fut, kl, kmeans] (a device view that the next step overwrites)."""
        self.loader.upload_starts(starts)
        eng = self.model._engine
        if self.warmup > 0 or self.dev.type != "cuda":   # the first steps run eagerly: workspaces, plans and caches settle before the capture
            self.warmup -= 1
            return self._eager()
        if self.choice is not None:
            mode = self.choice.get(self.loader.B)
            if mode is None:
                mode = self._measure()
            if mode == "eager":
                return self._eager()
        if self.graph is not None and self._seen != (eng, ops.ALLOC_GEN[0], self.model._flat_p.data_ptr()):
            # a buffer the captured launches point into was reallocated since (another batch size grew a workspace, the model moved):
            # run this step eagerly -- which settles the buffers for this shape again -- and capture afresh on the next call
            self.graph = None
            return self._eager()
        if self.graph is None:
            self._capture()
        if eng is not None:
            eng.poll_async_errors()
        self.opt.sync_lr()
        self.opt.t += 1
        self.graph.replay()
        if eng is not None:
            if eng._coop_state is not None:
                eng._coop_state.dirty = True
            eng.snapshot_async_errors()
        return self.terms


def _world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _dist_active():
    """True when this package runs under a process group (several ranks, or one rank with VAME_AMD_FORCE_DIST=1 -- the
    single-GPU way to execute the RCCL path: init, collectives, barriers, shutdown)."""
    return dist.is_available() and dist.is_initialized()


def _rank_mean(t):
    """Average a small statistics tensor over ranks so every rank takes the same scheduler / checkpoint decisions."""
    _, world = _world()
    if _dist_active():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        t = t / world
    return t


def allreduce_gradients(model):
    """One RCCL all-reduce (SUM) of the flat fp32 gradient bucket over xGMI; the 1/world factor is folded
    into the Adam kernel.  2,618,476 floats = 10.5 MB at the default model size."""
    _, world = _world()
    if _dist_active():
        n = model.flat_parameters()[1].numel()
        bucket, eng = model._flat_g_comm, model._engine
        # the status word of this rank's cooperative launches rides the same all-reduce (slot n): afterwards it is non-zero on
        # EVERY rank if any rank failed, and it is what the optimizer kernel tests -- so a failed step is dropped everywhere and
        # every rank raises, instead of one rank's undefined gradients reaching the healthy ranks' weights
        eng.share_status(bucket[n:n + 1])
        dist.all_reduce(bucket, op=dist.ReduceOp.SUM)
        eng.snapshot_async_errors(reduced=True)
    elif getattr(model, "_engine", None) is not None:
        model._engine.unshare_status()          # (a process group that has been shut down since: back to the rank-local status word)
    return 1.0 / world


# ------------------------------------------------------------------------------------ epoch loops
def _to_windows(item, keep, dev):
    """Accept either device windows (B,L,F) from DeviceWindowLoader or the reference loader's (B,F,2T) items."""
    if item.dim() == 3 and item.is_floating_point() and item.dtype == torch.float32 and item.device.type == dev.type:
        return item
    return item.permute(0, 2, 1)[:, :keep, :].to(dtype=torch.float32).to(dev).contiguous()


def _graphed_step_for(model, optimizer, loader, noise, hip_graph, kw):
    """The cached GraphedTrainStep of this (model, optimizer, loader, loss arguments), or None where the step runs eagerly: several ranks (the
    all-reduce stays outside a graph), options that draw with torch inside the step (input noise, encoder dropout), a loader that is not the
    device batcher, hip_graph = False, or -- hip_graph = None, "auto" -- a batch above 1024, where the step is GPU-bound anyway.  Up to 1024 "auto" times
    both forms in the first epoch and keeps the faster (GraphedTrainStep._measure): which one wins differs between boxes (2.04 vs 2.11 ms at batch 256)."""
    from .dataloader import DeviceWindowLoader
    dev = model.flat_parameters()[0].device
    if (hip_graph is False or dev.type != "cuda" or _dist_active() or noise == True or model.spec.dropout > 0  # noqa: E712
            or not isinstance(optimizer, FusedAdamAMSGrad) or not isinstance(loader, DeviceWindowLoader) or loader.world != 1):
        return None
    if hip_graph in (None, "auto") and loader.B > 1024:
        return None
    cache = optimizer.__dict__.setdefault("_graphed_steps", {})
    key = (id(loader), tuple(sorted((k, v) for k, v in kw.items())))
    g = cache.get(key)
    if g is None:
        if len(cache) >= 8:                  # (KL annealing makes a few distinct weights, then one for the rest of the run)
            cache.pop(next(iter(cache)))
        acc = optimizer.__dict__.setdefault("_epoch_acc", torch.zeros(6, device=dev, dtype=torch.float64))
        # "auto": decided by timing both forms once per batch size (kept across the loss-weight changes of KL annealing, which make new step objects)
        choice = optimizer.__dict__.setdefault("_graph_choice", {}) if hip_graph in (None, "auto") else None
        g = cache[key] = GraphedTrainStep(model, optimizer, loader, acc, choice=choice, **kw)
    return g


def train(train_loader, epoch, model, optimizer, anneal_function, BETA, kl_start, annealtime, seq_len, future_decoder,
          future_steps, scheduler, mse_red, mse_pred, kloss, klmbda, bsize, noise, hip_graph=None):
    model.train()
    dev = model.flat_parameters()[0].device
    seq_len_half = int(seq_len / 2)
    keep = seq_len_half + (future_steps if future_decoder else 0)
    kl_weight = kl_annealing(epoch, kl_start, annealtime, anneal_function)
    # total, rec, fut, kl, kmeans summed over the epoch's batches + the last batch's total: filled by the step's own loss kernel
    # (vame_loss_finish_f32: total = rec + fut + BETA * kl_weight * kl + kl_weight * kmeans, rnn_vae.py:129-150) -- no torch op per step
    acc = torch.zeros(6, device=dev, dtype=torch.float64)
    wts = (1.0, 1.0, BETA * kl_weight, kl_weight)
    idx = -1
    graphed = _graphed_step_for(model, optimizer, train_loader, noise, hip_graph,
                                dict(kl_weight=kl_weight, beta=BETA, kloss=kloss, klmbda=klmbda, bsize=bsize, mse_red=mse_red, mse_pred=mse_pred, weights=wts))
    if graphed is not None:
        # host-bound regime (the stock batch 256): the whole step is one replayed hipGraph, the host only draws and uploads the window starts
        acc = graphed.acc
        acc.zero_()
        for idx in range(len(train_loader)):
            graphed(train_loader.draw_starts())
    else:
        for idx, data_item in enumerate(train_loader):
            win = _to_windows(data_item, keep, dev)
            enc_in = gaussian(win[:, :seq_len_half, :], True, seq_len_half) if noise == True else None  # noqa: E712
            model.loss_step(win, kl_weight, beta=BETA, kloss=kloss, klmbda=klmbda, bsize=bsize, mse_red=mse_red,
                            mse_pred=mse_pred, enc_in=enc_in, weights=wts, acc=acc)
            gscale = allreduce_gradients(model)
            optimizer.step(gscale=gscale) if isinstance(optimizer, FusedAdamAMSGrad) else optimizer.step()
    if idx < 1:
        raise ValueError("train(): need at least 2 batches per epoch (the reference divides by the last batch index, "
                         "rnn_vae.py:158,164); lower batch_size or provide more data")
    acc = _rank_mean(acc)                                                      # identical statistics (and decisions) on all ranks
    scheduler.step(float(acc[5]))
    train_loss, mse_loss, fut_loss, kullback_loss, kmeans_losses = [float(v) for v in acc[:5].cpu()]
    if getattr(model, "_engine", None) is not None:
        model._engine.check_async_errors(all_ranks=True)
    if future_decoder:
        print('Train loss: {:.3f}, MSE-Loss: {:.3f}, MSE-Future-Loss {:.3f}, KL-Loss: {:.3f}, Kmeans-Loss: {:.3f}, weight: {:.2f}'.format(
            train_loss / idx, mse_loss / idx, fut_loss / idx, BETA * kl_weight * kullback_loss / idx, kl_weight * kmeans_losses / idx, kl_weight))
    else:
        print('Train loss: {:.3f}, MSE-Loss: {:.3f}, KL-Loss: {:.3f}, Kmeans-Loss: {:.3f}, weight: {:.2f}'.format(
            train_loss / idx, mse_loss / idx, BETA * kl_weight * kullback_loss / idx, kl_weight * kmeans_losses / idx, kl_weight))
    return kl_weight, train_loss / idx, kl_weight * kmeans_losses / idx, kullback_loss / idx, mse_loss / idx, fut_loss / idx


def test(test_loader, epoch, model, optimizer, BETA, kl_weight, seq_len, mse_red, kloss, klmbda, future_decoder, bsize):
    model.eval()
    dev = model.flat_parameters()[0].device
    seq_len_half = int(seq_len / 2)
    acc6 = torch.zeros(6, device=dev, dtype=torch.float64)    # total, rec, (fut: not part of the test loss), kl, kmeans, last total
    wts = (1.0, 0.0, BETA * kl_weight, kl_weight)
    idx = -1
    with torch.no_grad():
        for idx, data_item in enumerate(test_loader):
            win = _to_windows(data_item, seq_len_half, dev)
            model.loss_step(win, kl_weight, beta=BETA, kloss=kloss, klmbda=klmbda, bsize=bsize, mse_red=mse_red,
                            backward=False, weights=wts, acc=acc6)
    acc = acc6[[0, 1, 3, 4]]
    if idx < 1:
        raise ValueError("test(): need at least 2 test batches of batch_size/4 (rnn_vae.py:207-210 divides by the last index)")
    test_loss, mse_loss, kullback_loss, kmeans_losses = [float(v) for v in _rank_mean(acc).cpu()]
    if getattr(model, "_engine", None) is not None:
        model._engine.check_async_errors(all_ranks=True)
    print('Test loss: {:.3f}, MSE-Loss: {:.3f}, KL-Loss: {:.3f}, Kmeans-Loss: {:.3f}'.format(
        test_loss / idx, mse_loss / idx, BETA * kl_weight * kullback_loss / idx, kl_weight * kmeans_losses / idx))
    return mse_loss / idx, test_loss / idx, kl_weight * kmeans_losses


# ------------------------------------------------------------------------------------ driver
def _maybe_init_distributed():
    """One process per GPU under torchrun (RANK/WORLD_SIZE/LOCAL_RANK in the env): RCCL over xGMI.  VAME_AMD_FORCE_DIST=1 creates
    the group for WORLD_SIZE=1 as well (RANK / MASTER_ADDR / MASTER_PORT must be set): the whole collective path then runs on one GPU."""
    forced = os.environ.get("VAME_AMD_FORCE_DIST", "0") not in ("", "0")
    if (int(os.environ.get("WORLD_SIZE", "1")) > 1 or forced) and not dist.is_initialized():
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")
        import atexit
        atexit.register(shutdown_distributed)      # the group outlives train_model(): pose_segmentation() reuses it
    return _world()


def shutdown_distributed():
    """Tear down the process group this package created (also registered with atexit)."""
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def train_model(config):
    config_file = Path(config).resolve()
    cfg = read_config(config_file)
    legacy = cfg['legacy']
    model_name = cfg['model_name']
    pretrained_weights = cfg['pretrained_weights']
    pretrained_model = cfg['pretrained_model']
    fixed = cfg['egocentric_data']
    rank, world = _maybe_init_distributed()
    is_main = rank == 0

    print("Train Variational Autoencoder - model name: %s \n" % model_name)
    pp = cfg['project_path']
    if is_main and not os.path.exists(os.path.join(pp, 'model', 'best_model', "")):
        os.makedirs(os.path.join(pp, 'model', 'best_model', 'snapshots', ""), exist_ok=True)
        os.makedirs(os.path.join(pp, 'model', 'model_losses', ""), exist_ok=True)

    dev = _lib.device()                  # raises without an MI355X: there is no CPU path
    if dev.type == "cuda":
        print("Using HIP device:", torch.cuda.get_device_name(dev), "| ranks:", world)

    SEED = 19
    TRAIN_BATCH_SIZE = cfg['batch_size']
    TEST_BATCH_SIZE = int(cfg['batch_size'] / 4)
    EPOCHS = cfg['max_epochs']
    ZDIMS = cfg['zdims']
    BETA = cfg['beta']
    SNAPSHOT = cfg['model_snapshot']
    LEARNING_RATE = cfg['learning_rate']
    NUM_FEATURES = cfg['num_features']
    if fixed == False:  # noqa: E712
        NUM_FEATURES = NUM_FEATURES - 2
    TEMPORAL_WINDOW = cfg['time_window'] * 2
    FUTURE_DECODER = cfg['prediction_decoder']
    FUTURE_STEPS = cfg['prediction_steps']
    noise = cfg['noise']
    scheduler_step_size = cfg['scheduler_step_size']
    MSE_REC_REDUCTION = cfg['mse_reconstruction_reduction']
    MSE_PRED_REDUCTION = cfg['mse_prediction_reduction']
    KMEANS_LOSS = cfg['kmeans_loss']
    KMEANS_LAMBDA = cfg['kmeans_lambda']
    KL_START = cfg['kl_start']
    ANNEALTIME = cfg['annealtime']
    anneal_function = cfg['anneal_function']
    optimizer_scheduler = cfg['scheduler']

    BEST_LOSS = 999999
    convergence = 0
    print('Latent Dimensions: %d, Time window: %d, Batch Size: %d, Beta: %d, lr: %.4f\n' % (
        ZDIMS, cfg['time_window'], TRAIN_BATCH_SIZE, BETA, LEARNING_RATE))
    train_losses, test_losses, kmeans_losses, kl_losses, weight_values, mse_losses, fut_losses = [], [], [], [], [], [], []

    torch.manual_seed(SEED)
    if dev.type == "cuda":
        torch.cuda.manual_seed(SEED)
    RNN = RNN_VAE_LEGACY if legacy else RNN_VAE                          # rnn_vae.py:294-297
    model = RNN(TEMPORAL_WINDOW, ZDIMS, NUM_FEATURES, FUTURE_DECODER, FUTURE_STEPS, cfg['hidden_size_layer_1'],
                    cfg['hidden_size_layer_2'], cfg['hidden_size_rec'], cfg['hidden_size_pred'], cfg['dropout_encoder'],
                    cfg['dropout_rec'], cfg['dropout_pred'], cfg['softplus']).to(dev)
    # optional key, absent from the reference's config.yaml: kernel-choice / scheduling options of this build (vame_amd.engine.ENGINE_DEFAULTS)
    model.engine_options = dict(cfg.get('vame_amd_engine') or {})
    if world > 1:
        # identical weights on every rank (same seed above), but independent draws afterwards: eps of the reparameterisation,
        # input noise and dropout masks must not repeat across the ranks' batches
        torch.manual_seed(SEED + 1000 * (rank + 1))
        if dev.type == "cuda":
            torch.cuda.manual_seed(SEED + 1000 * (rank + 1))

    if pretrained_weights:
        cand = os.path.join(pp, 'model', 'best_model', pretrained_model + '_' + cfg['Project'] + '.pkl')
        loaded = False
        for path in (cand, pretrained_model):
            try:
                print("Loading pretrained weights from %s\n" % path)
                model.load_state_dict(torch.load(path, map_location=dev))
                KL_START, ANNEALTIME, loaded = 0, 1, True
                break
            except (FileNotFoundError, IsADirectoryError, RuntimeError, OSError) as e:
                print("No usable file at %s (%s)\n" % (path, type(e).__name__))
        if not loaded:
            print("Could not load pretrained model. Check file path in config.yaml.")

    data_dir = os.path.join(pp, "data", "train", "")
    if _dist_active() and not is_main:
        dist.barrier()                      # let rank 0 create seq_mean/std first
    trainset = SEQUENCE_DATASET(data_dir, data='train_seq.npy', train=True, temporal_window=TEMPORAL_WINDOW)
    if _dist_active() and is_main:
        dist.barrier()
    testset = SEQUENCE_DATASET(data_dir, data='test_seq.npy', train=False, temporal_window=TEMPORAL_WINDOW)
    keep = TEMPORAL_WINDOW // 2 + (FUTURE_STEPS if FUTURE_DECODER else 0)
    train_loader = DeviceWindowLoader(trainset, TRAIN_BATCH_SIZE, keep, dev, rank, world)
    test_loader = DeviceWindowLoader(testset, TEST_BATCH_SIZE, TEMPORAL_WINDOW // 2, dev, 0, 1)

    optimizer = FusedAdamAMSGrad(model, lr=LEARNING_RATE)
    if optimizer_scheduler:
        print('Scheduler step size: %d, Scheduler gamma: %.2f\n' % (scheduler_step_size, cfg['scheduler_gamma']))
        scheduler = ReduceLROnPlateau(optimizer, 'min', factor=cfg['scheduler_gamma'], patience=cfg['scheduler_step_size'],
                                      threshold=1e-3, threshold_mode='rel')
    else:
        scheduler = StepLR(optimizer, step_size=scheduler_step_size, gamma=1, last_epoch=-1)

    print("Start training... ")
    best_dir = os.path.join(pp, "model", "best_model")
    loss_dir = os.path.join(pp, 'model', 'model_losses')
    for epoch in range(1, EPOCHS):
        print("Epoch: %d" % epoch)
        weight, train_loss, km_loss, kl_loss, mse_loss, fut_loss = train(
            train_loader, epoch, model, optimizer, anneal_function, BETA, KL_START, ANNEALTIME, TEMPORAL_WINDOW, FUTURE_DECODER,
            FUTURE_STEPS, scheduler, MSE_REC_REDUCTION, MSE_PRED_REDUCTION, KMEANS_LOSS, KMEANS_LAMBDA, TRAIN_BATCH_SIZE, noise,
            hip_graph=cfg.get('vame_amd_hip_graph', 'auto'))           # optional key: true / false / auto (default: batches up to 1024)
        current_loss, test_loss, test_list = test(test_loader, epoch, model, optimizer, BETA, weight, TEMPORAL_WINDOW,
                                                  MSE_REC_REDUCTION, KMEANS_LOSS, KMEANS_LAMBDA, FUTURE_DECODER, TEST_BATCH_SIZE)
        train_losses.append(train_loss); test_losses.append(test_loss); kmeans_losses.append(km_loss)
        kl_losses.append(kl_loss); weight_values.append(weight); mse_losses.append(mse_loss); fut_losses.append(fut_loss)

        if weight > 0.99 and current_loss <= BEST_LOSS:
            BEST_LOSS = current_loss
            print("Saving model!")
            if is_main:
                torch.save(model.state_dict(), os.path.join(best_dir, model_name + '_' + cfg['Project'] + '.pkl'))
            convergence = 0
        else:
            convergence += 1

        if epoch % SNAPSHOT == 0 and is_main:
            print("Saving model snapshot!\n")
            torch.save(model.state_dict(), os.path.join(best_dir, 'snapshots', model_name + '_' + cfg['Project'] + '_epoch_' + str(epoch) + '.pkl'))

        if convergence > cfg['model_convergence']:
            print('Finished training...')
            print('Model converged. Please check your model with vame.evaluate_model(). \n'
                  'You can also re-run vame.trainmodel() to further improve your model. \n'
                  'Make sure to set _pretrained_weights_ in your config.yaml to "true" \n'
                  'and plug your current model name into _pretrained_model_. \n'
                  'Hint: Set "model_convergence" in your config.yaml to a higher value. \n\n'
                  'Next: \nUse vame.pose_segmentation() to identify behavioral motifs in your dataset!')
            break

        if is_main:
            for nm, arr in (('train_losses_', train_losses), ('test_losses_', test_losses), ('kmeans_losses_', kmeans_losses),
                            ('kl_losses_', kl_losses), ('weight_values_', weight_values), ('mse_train_losses_', mse_losses),
                            ('mse_test_losses_', current_loss), ('fut_losses_', fut_losses)):
                np.save(os.path.join(loss_dir, nm + model_name), arr)
        print("\n")

    if _dist_active():
        dist.barrier()                      # rank 0's checkpoint / loss files are complete before any rank goes on (pose_segmentation)
    if convergence < cfg['model_convergence']:
        print('Finished training...')
        print('Model seems to have not reached convergence. You may want to check your model \n'
              'with vame.evaluate_model(). If your satisfied you can continue. \n'
              'Use vame.pose_segmentation() to identify behavioral motifs! \n'
              'OPTIONAL: You can re-run vame.train_model() to improve performance.')
