"""Kernel-level checks of the HIP sources executed through the host emulator (tests/emu) vs the numpy oracle.

These exercise the exact kernel code (index math, MFMA fragment layouts, LDS staging, masking) on
the CPU; the same checks run on the real GPU in test_kernels_gpu.py.
"""
import pytest

from kernel_cases import (check_linear_group, check_head_fused, check_gemm_group_shared_output, check_gru_wide, check_gru_wide_small, check_hmm, check_adam, check_adam_abort_and_mask_scale, check_colsum, check_colsum_batch, check_kmeans, check_gather, check_gemm_cases, check_gemm_group, check_gemm_split, check_gemm_split_rows, check_gemm_pipelined_shapes, check_gru_bwd, check_gru_skew_fwd, check_gru_wide_skew_fwd, check_gru_ws_bwd, check_gru_kernel_option_is_an_argument, check_gru_coop_bwd, check_gru_coop_fwd, check_gru_fwd, check_gru_fwd_fused,
                          check_latent, check_latent_draw, check_loss_finish, check_mse, check_nuclear, check_prep_fill_rules, check_prepare_series_golden,
                          check_prepare_series_vs_oracle)

DEV = "cpu"


@pytest.mark.parametrize("epi", ["0", "1", "2"])
def test_gemm(emu, epi, monkeypatch):
    """every epilogue form of gemm_kernel (the emulator build reads VAME_GEMM_EPI like the A/B tuning build)"""
    monkeypatch.setenv("VAME_GEMM_EPI", epi)
    check_gemm_cases(DEV, small=True)
    check_gemm_group(DEV)


def test_gemm_software_pipelined_loop(emu, monkeypatch):
    for var in ("13", "9", "0"):            # pipelined (+ setprio), pipelined, plain loop on the same shapes
        monkeypatch.setenv("VAME_GEMM_VAR", var)
        for epi in ("1", "2"):
            monkeypatch.setenv("VAME_GEMM_EPI", epi)
            check_gemm_pipelined_shapes(DEV)


@pytest.mark.parametrize("H,B,T", [(32, 5, 4), (64, 40, 3), (96, 5, 3)])
def test_gru_fwd(emu, H, B, T):
    check_gru_fwd(DEV, H, B, T)


@pytest.mark.parametrize("H,B,T", [(32, 5, 4), (64, 40, 3), (96, 5, 3)])
def test_gru_bwd(emu, H, B, T):
    check_gru_bwd(DEV, H, B, T)


@pytest.mark.parametrize("H,B,T", [(128, 70, 3), (256, 37, 2), (128, 32, 1), (192, 40, 2), (64, 33, 2)])
def test_gru_wave_specialised_bwd(emu, H, B, T):
    check_gru_ws_bwd(DEV, H, B, T)


@pytest.mark.parametrize("H,B,T", [(64, 40, 3), (128, 37, 2), (256, 33, 2)])
def test_gru_skewed_fwd(emu, H, B, T):
    check_gru_skew_fwd(DEV, H, B, T)


@pytest.mark.parametrize("H,B,T,force", [(128, 37, 2, True), (384, 33, 2, False)])
def test_gru_wide_skewed_fwd(emu, H, B, T, force):
    check_gru_wide_skew_fwd(DEV, H, B, T, force_wide=force)


def test_gru_kernel_option_is_a_launch_argument(emu):
    check_gru_kernel_option_is_an_argument(DEV)


def test_reparameterisation_draw_and_loss_bookkeeping_kernels(emu):
    """round 5: eps drawn inside the latent kernel (Philox4x32-10, device-side step counter) and the step's loss bookkeeping in one launch"""
    check_latent_draw(DEV, n_rows=6000)
    check_loss_finish(DEV)


def test_elementwise(emu):
    check_gather(DEV)
    check_latent(DEV)
    check_mse(DEV)
    check_colsum(DEV)
    check_colsum_batch(DEV)
    check_adam(DEV)
    check_adam_abort_and_mask_scale(DEV)


@pytest.mark.parametrize("B,Z,k", [(64, 30, 30), (8, 30, 30), (50, 7, 4), (128, 34, 34), (96, 48, 20), (200, 64, 64), (70, 63, 63), (150, 66, 66), (300, 96, 40)])
def test_nuclear(emu, B, Z, k):
    check_nuclear(DEV, B, Z, k)


@pytest.mark.parametrize("H,B,T,I", [(32, 5, 4, 24), (64, 40, 3, 8)])
def test_gru_fwd_fused_input(emu, H, B, T, I):
    check_gru_fwd_fused(DEV, H, B, T, I)


def test_kmeans_next_row_n1(emu):
    check_kmeans(DEV)


def test_prepare_series_matches_reference(emu):
    check_prepare_series_golden("cpu")


def test_prepare_series_vs_oracle(emu):
    check_prepare_series_vs_oracle("cpu")


def test_prep_fill_rules(emu):
    check_prep_fill_rules("cpu")


@pytest.mark.parametrize("H,B,T", [(128, 40, 3), (256, 37, 4), (128, 17, 1), (256, 49, 2)])
def test_gru_coop_fwd(emu, H, B, T):
    check_gru_coop_fwd("cpu", H, B, T)


@pytest.mark.parametrize("H,B,T", [(128, 40, 3), (256, 37, 4), (128, 17, 1), (256, 49, 2)])
def test_gru_coop_bwd(emu, H, B, T):
    check_gru_coop_bwd("cpu", H, B, T)


def test_gemm_group(emu):
    check_gemm_group("cpu")


def test_gemm_group_split_bf16x6(emu):
    """the opt-in error-compensated split-bf16 contraction (index math, LDS image, plane split, k-tile ring) on the host emulator"""
    check_gemm_split("cpu")


def test_gemm_split_bf16x6_row_major_a(emu):
    """the split-bf16 contraction of a row-major A with a weight matrix (input projections, data gradients): row producer, rotated image slots"""
    check_gemm_split_rows("cpu")


@pytest.mark.parametrize("N,K,D,chunk", [(700, 4, 6, 64), (300, 3, 5, 7), (130, 17, 4, 16), (3000, 5, 8, None)])
def test_gaussian_hmm_next_row_n1(emu, N, K, D, chunk):
    check_hmm(DEV, N, K, D, chunk)


@pytest.mark.parametrize("H,B,T", [(320, 37, 3), (512, 5, 2)])
def test_gru_wide_hidden_sizes(emu, H, B, T):
    check_gru_wide(DEV, H, B, T)


@pytest.mark.parametrize("H,B,T", [(128, 37, 3), (256, 5, 2)])
def test_gru_wide_kernels_at_small_hidden_sizes(emu, H, B, T):
    check_gru_wide_small(DEV, H, B, T)


def test_gemm_group_shared_output(emu):
    check_gemm_group_shared_output(DEV)


def test_linear_group_of_a_narrow_input(emu):
    check_linear_group(DEV)


@pytest.mark.parametrize("B,T,F,K,pad", [(7, 9, 24, 64, 2), (3, 50, 10, 128, 0), (5, 31, 32, 192, 2), (290, 30, 24, 64, 2), (3, 20, 24, 640, 2), (5, 7, 32, 1024, 0)])
def test_fused_output_head(emu, B, T, F, K, pad):
    check_head_fused(DEV, B, T, F, K, pad)
