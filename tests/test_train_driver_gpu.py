"""vame.train_model() / vame.pose_segmentation() / evaluate / generative / create_trainset end to end on a synthetic project -- on the MI355X (the public entry points of the drop-in boundary).
The checks live in driver_cases.py."""
import pytest

import driver_cases as dc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def project(tmp_path_factory, hip):
    return dc.make_project(tmp_path_factory)


def test_reference_import_names_resolve_to_the_build(hip):
    """`import vame`, `from vame.model.rnn_vae import RNN_VAE`, ... through the opt-in alias (vame_amd/compat.py).
    Every test below drives the build through `import vame` in the reference's call order (examples/demo.py:48-56)."""
    dc.check_alias_surface()


def test_train_model_files_and_losses(project):
    dc.check_train_model_files_and_losses(project)


def test_pose_segmentation_outputs(project):
    dc.check_pose_segmentation_outputs(project)


def test_pose_segmentation_hmm(project):
    dc.check_pose_segmentation_hmm(project)


def test_evaluate_model_outputs(project):
    dc.check_evaluate_model_outputs(project)


def test_generative_model_modes(project):
    dc.check_generative_model_modes(project)


def test_train_model_options(project, tmp_path, capsys):
    dc.check_train_model_options(project, tmp_path, capsys)


def test_pose_segmentation_prompts(project, tmp_path, monkeypatch):
    dc.check_pose_segmentation_prompts(project, tmp_path, monkeypatch)


def test_train_model_legacy_topology(project, tmp_path):
    dc.check_train_model_legacy_topology(project, tmp_path)


def test_create_trainset_files(tmp_path, hip):
    dc.check_create_trainset_files(tmp_path)


def test_read_config_contract(tmp_path, hip):
    dc.check_read_config_contract(tmp_path)


def test_parameterization_with_gpu_kmeans_option(hip):
    dc.check_parameterization_with_gpu_kmeans_option()



@pytest.mark.parametrize("engine_options", [{}, dict(split_wgrad=1, split_proj=1)], ids=["default", "split-bf16"])
def test_graphed_step_is_bit_identical_to_eager(hip, engine_options):
    """round 5: the stock-batch train step (gather -> loss_step -> fused Adam) captured once as a hipGraph and replayed (rnn_vae.GraphedTrainStep)
    against the same ten steps enqueued launch by launch: the weights after every step, the Adam state and the device-side counters (Philox
    step, Adam step, cooperative launch epoch) are the same bits; the loss terms agree to the rounding of their atomic sums."""
    import numpy as np
    import torch
    from vame_amd.model.dataloader import DeviceWindowLoader
    from vame_amd.model.rnn_model import RNN_VAE
    from vame_amd.model.rnn_vae import FusedAdamAMSGrad, GraphedTrainStep
    T, F, Z, H, FS, B, N = 30, 12, 30, 256, 15, 256, 40000

    class DS:
        data_points, X, temporal_window = N, np.empty((F, 1)), 2 * T

        @staticmethod
        def normalised_f32():
            return np.random.default_rng(5).standard_normal((F, N)).astype(np.float32)
    dev = torch.device("cuda", 0)
    kw = dict(kl_weight=0.5, beta=1.0, kloss=Z, klmbda=0.1, bsize=B, weights=(1.0, 1.0, 0.5, 0.5))
    starts = np.random.default_rng(6).integers(0, N - 2 * T, size=(10, B))
    runs = []
    for graphed in (False, True):
        torch.manual_seed(19)
        model = RNN_VAE(2 * T, Z, F, 1, FS, H, H, H, H, 0, 0, 0, False).to(dev).train()
        model.engine_options = dict(engine_options)          # (second case: the opt-in split-bf16 contractions inside the captured step)
        opt = FusedAdamAMSGrad(model, lr=5e-4)
        loader = DeviceWindowLoader(DS(), B, T + FS, dev)
        acc = torch.zeros(6, dtype=torch.float64, device=dev)
        eng = model._ensure_engine()
        eng.seed_rng(777)
        g = GraphedTrainStep(model, opt, loader, acc, warmup=3 if graphed else 10 ** 9, **kw)
        snaps, terms = [], []
        for i in range(10):
            if i == 3:
                opt.param_groups[0]["lr"] = 4e-4               # a cut exactly between the last eager step and the capture: its fill must stay OUTSIDE the graph
            if i == 6:
                opt.param_groups[0]["lr"] = 1.6e-4             # a scheduler's cut between two replays: the kernel reads the rate from the device word
            t = g(starts[i])
            terms.append(t.cpu().numpy().copy())
            snaps.append(model.flat_parameters()[0].cpu().numpy().copy())
        assert (g.graph is not None) == graphed
        # an inference pass with a larger batch between two training steps grows the engine's workspaces: the captured launches would point
        # into freed buffers -- the replayer notices (ops.ALLOC_GEN), runs one step eagerly and captures again
        model.eval()
        with torch.no_grad():
            model(torch.randn(600, T, F, generator=torch.Generator().manual_seed(1)).to(dev))
        model.train()
        old_graph = g.graph
        for i in range(3):
            t = g(starts[i])
            terms.append(t.cpu().numpy().copy())
            snaps.append(model.flat_parameters()[0].cpu().numpy().copy())
        assert not graphed or (g.graph is not None and g.graph is not old_graph)
        assert eng._coop_state is not None                     # batch 256: the cooperative GRU kernels (their epoch counter lives on the device)
        eng.check_async_errors()
        runs.append(dict(snaps=snaps, terms=terms, acc=acc.cpu().numpy(), m=opt.m.cpu().numpy(), vmax=opt.vmax.cpu().numpy(),
                         adam=opt.dev_state.cpu().numpy()[1:3], rng=eng._rng.cpu().numpy(), epoch=eng._coop_state.epoch.cpu().numpy(), t=opt.t))
    a, b = runs
    for i in range(13):
        np.testing.assert_array_equal(a["snaps"][i], b["snaps"][i], err_msg=f"weights after step {i}")
        np.testing.assert_allclose(a["terms"][i], b["terms"][i], rtol=1e-5)
    for k in ("m", "vmax", "adam", "rng", "epoch"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    assert a["t"] == b["t"] == 13 and a["adam"].tolist() == [13, 0] and a["rng"][1] == 13
    np.testing.assert_allclose(a["acc"], b["acc"], rtol=1e-6)
    assert not np.array_equal(a["snaps"][0], a["snaps"][9])
    step5, step6 = (np.abs(a["snaps"][i + 1] - a["snaps"][i]).mean() for i in (4, 6))
    assert step6 < 0.6 * step5                                 # 1.6e-4 / 4e-4 = 0.4 of the update size


def test_hip_graph_auto_times_both_forms_and_keeps_one(hip):
    """round 6: `vame_amd_hip_graph: auto` decides by measurement (GraphedTrainStep._measure): 20 eager steps and 20 replays after the warm-up, each span
    between two events, the faster form kept per batch size.  Whatever it picks, the trajectory is the eager one's bits."""
    import numpy as np
    import torch
    from vame_amd.model.dataloader import DeviceWindowLoader
    from vame_amd.model.rnn_model import RNN_VAE
    from vame_amd.model.rnn_vae import FusedAdamAMSGrad, GraphedTrainStep
    T, F, Z, H, FS, B, N = 30, 12, 30, 256, 15, 256, 40000

    class DS:
        data_points, X, temporal_window = N, np.empty((F, 1)), 2 * T

        @staticmethod
        def normalised_f32():
            return np.random.default_rng(5).standard_normal((F, N)).astype(np.float32)
    dev = torch.device("cuda", 0)
    kw = dict(kl_weight=0.5, beta=1.0, kloss=Z, klmbda=0.1, bsize=B, weights=(1.0, 1.0, 0.5, 0.5))
    steps = 3 + 2 * GraphedTrainStep.MEASURE_STEPS + 2 + 4
    starts = np.random.default_rng(6).integers(0, N - 2 * T, size=(steps, B))
    finals, choice = [], {}
    for auto in (False, True):
        torch.manual_seed(19)
        model = RNN_VAE(2 * T, Z, F, 1, FS, H, H, H, H, 0, 0, 0, False).to(dev).train()
        opt = FusedAdamAMSGrad(model, lr=5e-4)
        loader = DeviceWindowLoader(DS(), B, T + FS, dev)
        acc = torch.zeros(6, dtype=torch.float64, device=dev)
        model._ensure_engine().seed_rng(777)
        g = GraphedTrainStep(model, opt, loader, acc, warmup=3 if auto else 10 ** 9, choice=choice if auto else None, **kw)
        for i in range(steps):
            g(starts[i])
        model._engine.check_async_errors()
        finals.append(model.flat_parameters()[0].cpu().numpy().copy())
    assert choice.get(B) in ("graph", "eager"), choice
    t_eager, t_graph = choice[("ms_per_step", B)]
    assert 0.5 < t_eager < 20 and 0.5 < t_graph < 20, (t_eager, t_graph)
    assert (choice[B] == "graph") == (t_graph <= t_eager)
    np.testing.assert_array_equal(finals[0], finals[1])
    print(f"hip_graph auto at batch {B}: eager {t_eager:.3f} ms/step, replay {t_graph:.3f} ms/step -> {choice[B]}")
