"""Model-level parity through the host emulator: the HIP kernel sources + Python engine vs the
reference's golden vectors (tiny model).  The same bodies run on the GPU in test_model_gpu.py."""
import pytest

from model_cases import check_legacy_padded_hidden, check_decoder_inputs, check_padded_hidden_sizes, check_fused_heads_match, check_adam_trajectory, check_optimizer_checkpoint, check_failed_step_leaves_no_sums, check_engine_option_validation, check_coop_failure_is_contained, check_odd_dims_vs_oracle, check_eval_and_submodules, check_evaluate_and_generative_cores, check_h0_view, check_device_window_loader, check_legacy_step, check_model_options, check_noise_input, check_stale_backward_guard, check_step


@pytest.mark.parametrize("name,kw,mse", [("step_tiny", 1.0, "sum"), ("step_tiny", 0.25, "sum"), ("step_tiny_oddB", 1.0, "sum"),
                                         ("step_tiny_nofut", 1.0, "sum"), ("step_tiny_softplus", 1.0, "sum"),
                                         ("step_tiny_mean", 1.0, "mean"), ("step_h64", 0.5, "sum"),
                                         ("step_tiny_mean_kw025", 0.25, "mean"), ("step_h64_mean", 0.5, "mean")])
def test_fused_step_matches_reference(emu, name, kw, mse):
    check_step("cpu", name, kw, mse)


def test_autograd_path_matches_reference(emu):
    check_step("cpu", "step_tiny", 1.0, via_autograd=True)


def test_hidden_sizes_not_multiple_of_32(emu):
    """nn.GRU takes any hidden_size; the kernels run on a zero-padded parameter image (vame_amd/padding.py)."""
    check_padded_hidden_sizes("cpu")


def test_legacy_topology_with_padded_hidden_size(emu):
    check_legacy_padded_hidden("cpu")


def test_decoders_over_arbitrary_inputs(emu):
    """Decoder.forward(inputs, z) with inputs that are not z tiled over time (rnn_model.py:99-109)."""
    check_decoder_inputs("cpu")


def test_eval_and_submodules(emu):
    check_eval_and_submodules("cpu")


def test_evaluate_and_generative_cores(emu):
    check_evaluate_and_generative_cores("cpu")


def test_legacy_model_matches_reference(emu):
    check_legacy_step("cpu")


def test_decoder_h0_view(emu):
    check_h0_view("cpu")


def test_noise_option_separate_encoder_input(emu):
    check_noise_input("cpu")


@pytest.mark.parametrize("name,kw", [("step_tiny", 1.0), ("step_tiny_oddB", 1.0), ("step_h64", 0.5)])
def test_stepwise_large_h_path_matches_reference(emu, name, kw):
    """The per-step GEMM + gate-kernel path used for H > 256, forced on small models."""
    check_step("cpu", name, kw, stepwise=True)


def test_three_step_adam_trajectory_matches_reference(emu):
    check_adam_trajectory("cpu")


def test_optimizer_state_dict_round_trip_resumes_the_trajectory(emu):
    check_optimizer_checkpoint("cpu")


def test_failed_step_leaves_no_partial_loss_sums(emu):
    check_failed_step_leaves_no_sums("cpu")


def test_engine_option_values_are_validated(emu):
    check_engine_option_validation("cpu")


def test_unaligned_feature_and_latent_dims(emu):
    check_odd_dims_vs_oracle("cpu")
    check_odd_dims_vs_oracle("cpu", F=12, Z=30, H=64, T=6, FS=3, B=33)


def test_more_than_32_features_keeps_the_separate_head_launches(emu):
    """num_features > 32: outside the streaming output head's (and the fused layer-0 input projection's) range -- the step falls back to the
    separate contractions; whole train step against the numpy oracle."""
    from vame_amd import ops
    calls, orig = [], ops.head_stream
    ops.head_stream = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        check_odd_dims_vs_oracle("cpu", F=36, Z=7, H=32, T=5, FS=2, B=9)
    finally:
        ops.head_stream = orig
    assert not calls


def test_latent_width_above_64(emu):
    """zdims > 64 leaves the LDS-resident nuclear-norm kernel for the state-buffer one; everything else is width-agnostic."""
    check_odd_dims_vs_oracle("cpu", F=12, Z=72, H=32, T=5, FS=2, B=90)


def test_small_batch_cooperative_path_vs_oracle(emu):
    """H = 128 with a tiny batch goes through the column-split GRU kernels (engine._coop_ok): full step vs the numpy oracle."""
    check_odd_dims_vs_oracle("cpu", F=10, Z=7, H=128, T=4, FS=2, B=5)


def test_split_bf16_layer1_projection_option_vs_oracle(emu):
    """engine option split_proj: the second encoder layer's input projections (1024 x 192 x 128) and their data gradients (1024 x 128 x 192)
    on the split-bf16 contraction, whole train step against the numpy oracle at the unchanged tolerance."""
    from vame_amd import ops
    seen, orig = [], ops.gemm
    ops.gemm = lambda *a, **k: (seen.append((a[0], a[1], a[2], k.get("split"))), orig(*a, **k))[1]
    try:
        check_odd_dims_vs_oracle("cpu", F=12, Z=7, H=64, T=16, FS=4, B=64, engine_options=dict(split_proj=0))
    finally:
        ops.gemm = orig
    assert sorted(c[:3] for c in seen if c[3] is not None) == [(1024, 128, 192)] * 2 + [(1024, 192, 128)] * 2


@pytest.mark.parametrize("name", ["step_tiny_dropout", "step_tiny_hsizes"])
def test_reference_model_options(emu, name):
    check_model_options("cpu", name)


def test_stale_backward_is_refused(emu):
    check_stale_backward_guard("cpu")


def test_device_window_loader_matches_reference_batcher(emu, tmp_path):
    check_device_window_loader("cpu", tmp_path)


def test_cooperative_launch_failure_is_contained(emu):
    check_coop_failure_is_contained("cpu")


def test_wide_hidden_size_model_step_vs_oracle(emu):
    """256 < H <= 512: persistent two-blocks-per-wave forward (gru_wide.hip) + step-wise BPTT on its fragment stash, whole train step."""
    check_odd_dims_vs_oracle("cpu", F=12, Z=7, H=320, T=3, FS=2, B=5)


def test_headline_hidden_size_takes_the_grouped_dz_path(emu):
    """H = 256 is where K = 3H and 2H split eight ways: the decoders' contributions to dz leave as shared-output grouped launches."""
    from vame_amd import ops
    calls, orig = [], ops.gemm_group
    ops.gemm_group = lambda *a, **k: (calls.append((a[2], len(a[3]))), orig(*a, **k))[1]
    try:
        check_odd_dims_vs_oracle("cpu", F=12, Z=7, H=256, T=2, FS=2, B=3)
    finally:
        ops.gemm_group = orig
    assert (768, 4) in calls and (512, 2) in calls


def test_fused_output_heads_match_the_default_path(emu):
    check_fused_heads_match("cpu")


def test_round4_advisor_findings(emu, monkeypatch):
    """(1) a Parameter OBJECT replaced after the engine was built (module surgery) is noticed -- the old object still points into the flat
    bucket, so comparing addresses alone kept the stale weights; (2) the epoch-end error check enters its all-reduce on every rank, also on one
    that made no cooperative launch; (4) malformed GRU option words are refused on both sides of the C ABI."""
    import numpy as np
    import torch
    from vame_amd import _lib, ops
    from vame_amd.model.rnn_model import RNN_VAE
    torch.manual_seed(3)
    model = RNN_VAE(8, 5, 6, 1, 2, 32, 32, 32, 32, 0, 0, 0, False).eval()
    x = torch.randn(3, 4, 6)
    base = model(x)[0].clone()
    eng0, bucket0 = model._engine, model._flat_p
    w_old = model.decoder.hidden_to_output.weight
    model.decoder.hidden_to_output.weight = torch.nn.Parameter(torch.zeros_like(w_old))          # a NEW object; the old one still aliases the bucket
    out = model(x)[0]
    assert model._flat_p is not bucket0 and model._engine is not eng0, "the flat bucket was not rebuilt over the live parameters"
    bias = model.decoder.hidden_to_output.bias.detach()
    np.testing.assert_allclose(out.numpy(), np.broadcast_to(bias.numpy(), out.shape), atol=1e-6)   # zero weight: the output is the bias
    assert not np.allclose(out.numpy(), base.numpy())
    # (2)
    eng = model._engine
    assert eng._coop_state is None
    calls = []
    monkeypatch.setattr(eng, "_multi_rank", lambda: True)
    monkeypatch.setattr(torch.distributed, "all_reduce", lambda t, op=None: calls.append(int(t.item())))
    eng.check_async_errors(all_ranks=True)
    assert calls == [0]
    monkeypatch.setattr(torch.distributed, "all_reduce", lambda t, op=None: t.fill_(1))              # another rank reports a timeout
    with pytest.raises(_lib.VameHipError):
        eng.check_async_errors(all_ranks=True)
    eng.check_async_errors()                                                                        # rank-local: nothing to do, no collective
    # (4)
    with pytest.raises(ValueError):
        ops.gru_opt(ops.KERNEL_WS, pace_cp=255)
    with pytest.raises(ValueError):
        ops.gru_opt(16)
    d = torch.zeros(ops.GF["N"], dtype=torch.int64)
    d[ops.GF["OPT"]] = 1 << 24
    assert _lib.lib().vame_gru_seq_fwd_f32(d.data_ptr(), 1, 4, 32, None) == -1 and b"malformed option word" in _lib.lib().vame_last_error()


def test_odd_multiples_of_32_between_256_and_512_run_the_wide_kernels(emu):
    """round 5: hidden sizes 288 / 352 / 416 / 480 (nn.GRU takes any, rnn_model.py:34,91,125) used to fall to the per-step GEMM path at half the speed of
    their neighbours; they now run the two-blocks-per-wave persistent kernels on a zero-padded image of the next multiple of 64 (vame_amd/padding.py) --
    a whole train step (losses + all gradients in the reference's shapes) against the numpy oracle."""
    from vame_amd.padding import pad32

    def wide(eng):
        assert eng.spec.H == 320 and eng._wide(eng.spec.H) and eng.wide_bwd and not eng.force_stepwise
    assert [pad32(h) for h in (256, 288, 320, 352, 416, 480, 512, 544, 100)] == [256, 320, 320, 384, 448, 512, 512, 544, 128]
    check_odd_dims_vs_oracle("cpu", F=12, Z=7, H=288, T=3, FS=2, B=5, expect=wide)
