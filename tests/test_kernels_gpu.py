"""The kernel checks of test_kernels_emu.py on the real MI355X through libvame_hip.so (C ABI)."""
import pytest

from kernel_cases import (check_linear_group, check_head_fused, check_gemm_group_shared_output, check_gru_wide, check_gru_wide_small, check_hmm, check_adam, check_adam_abort_and_mask_scale, check_colsum, check_colsum_batch, check_kmeans, check_gather, check_gemm_cases, check_gemm_group, check_gemm_split, check_gemm_split_rows, check_gru_bwd, check_gru_skew_fwd, check_gru_wide_skew_fwd, check_gru_fwd_ring_stress, check_gru_ws_bwd, check_gru_kernel_option_is_an_argument, check_gru_coop_bwd, check_gru_coop_fwd, check_gru_fwd, check_gru_fwd_fused,
                          check_latent, check_latent_draw, check_loss_finish, check_mse, check_nuclear, check_prep_fill_rules, check_prepare_series_golden,
                          check_prepare_series_vs_oracle)

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_gemm(hip):
    check_gemm_cases(DEV, small=False)


@pytest.mark.parametrize("H,B,T", [(32, 5, 4), (64, 40, 3), (96, 70, 5), (128, 70, 5), (160, 64, 6), (192, 40, 4), (224, 33, 3), (256, 100, 30), (256, 4096, 4)])
def test_gru_fwd(hip, H, B, T):
    check_gru_fwd(DEV, H, B, T)


@pytest.mark.parametrize("H,B,T", [(32, 5, 4), (64, 40, 3), (96, 70, 5), (128, 70, 5), (160, 64, 6), (192, 40, 4), (224, 33, 3), (256, 100, 30)])
def test_gru_bwd(hip, H, B, T):
    check_gru_bwd(DEV, H, B, T)


@pytest.mark.parametrize("H,B,T", [(128, 70, 5), (256, 100, 30), (256, 4096, 4), (256, 33, 1), (192, 100, 6), (192, 4096, 3), (64, 70, 5)])
def test_gru_wave_specialised_bwd(hip, H, B, T):
    check_gru_ws_bwd(DEV, H, B, T)


def test_gru_fwd_weight_ring_repeat_stress(hip):
    """200 launches of the headline forward shape (H = 256, batch 4096, T = 30, per-step gi) beside a saturating HBM
    copy stream, every launch bit-identical to the first."""
    check_gru_fwd_ring_stress(DEV, 256, 4096, 30, launches=200)


@pytest.mark.parametrize("H,B,T", [(64, 40, 3), (128, 70, 5), (192, 40, 4), (256, 100, 30), (256, 4096, 4), (256, 33, 1)])
def test_gru_skewed_fwd(hip, H, B, T):
    check_gru_skew_fwd(DEV, H, B, T)


@pytest.mark.parametrize("H,B,T,force", [(128, 70, 5, True), (256, 100, 6, True), (384, 70, 5, False), (512, 100, 7, False), (512, 8192, 2, False)])
def test_gru_wide_skewed_fwd(hip, H, B, T, force):
    check_gru_wide_skew_fwd(DEV, H, B, T, force_wide=force)


def test_gru_kernel_option_is_a_launch_argument(hip):
    check_gru_kernel_option_is_an_argument(DEV)


def test_reparameterisation_draw_and_loss_bookkeeping_kernels(hip):
    """round 5: eps drawn inside the latent kernel (Philox4x32-10, device-side step counter) and the step's loss bookkeeping in one launch"""
    check_latent_draw(DEV)
    check_loss_finish(DEV)


def test_elementwise(hip):
    check_gather(DEV)
    check_latent(DEV)
    check_mse(DEV)
    check_colsum(DEV)
    check_colsum_batch(DEV)
    check_adam(DEV)
    check_adam_abort_and_mask_scale(DEV)


@pytest.mark.parametrize("B,Z,k", [(64, 30, 30), (8, 30, 30), (50, 7, 4), (4096, 30, 30), (128, 34, 34), (96, 48, 20), (4096, 64, 64), (70, 63, 63), (150, 66, 66), (4096, 128, 128), (600, 200, 90)])
def test_nuclear(hip, B, Z, k):
    check_nuclear(DEV, B, Z, k)


@pytest.mark.parametrize("H,B,T,I", [(32, 5, 4, 24), (64, 40, 3, 8), (256, 100, 30, 24), (256, 4096, 3, 24)])
def test_gru_fwd_fused_input(hip, H, B, T, I):
    check_gru_fwd_fused(DEV, H, B, T, I)


def test_kmeans_next_row_n1(hip):
    check_kmeans(DEV)
    check_kmeans(DEV, N=200000, K=15, D=30, n_init=2)


def test_prepare_series_matches_reference(hip):
    check_prepare_series_golden(DEV)


def test_prepare_series_vs_oracle(hip):
    check_prepare_series_vs_oracle(DEV, sizes=(200000, 123457))


def test_prep_fill_rules(hip):
    check_prep_fill_rules(DEV)


@pytest.mark.parametrize("H,B,T", [(128, 40, 3), (256, 37, 4), (128, 17, 1), (256, 49, 2), (256, 256, 30), (256, 500, 30), (128, 1000, 7)])
def test_gru_coop_fwd(hip, H, B, T):
    check_gru_coop_fwd(DEV, H, B, T)


@pytest.mark.parametrize("H,B,T", [(128, 40, 3), (256, 37, 4), (128, 17, 1), (256, 49, 2), (256, 256, 30), (256, 500, 30), (128, 1000, 7)])
def test_gru_coop_bwd(hip, H, B, T):
    check_gru_coop_bwd(DEV, H, B, T)


def test_gemm_group(hip):
    check_gemm_group(DEV, small=False)


def test_gemm_group_split_bf16x6(hip):
    """opt-in split-bf16 (bf16x6 planes, fp32 accumulate) contraction vs float64 incl. the weight-gradient shapes of the headline step"""
    check_gemm_split(DEV, small=False)


def test_gemm_split_bf16x6_row_major_a(hip):
    """the split-bf16 contraction of a row-major A with a weight matrix vs float64 incl. the layer-1 projection / data-gradient shapes"""
    check_gemm_split_rows(DEV, small=False)


@pytest.mark.parametrize("N,K,D,chunk", [(700, 4, 6, 64), (300, 3, 5, 7), (130, 17, 4, 16), (20000, 15, 30, 128), (20000, 15, 30, 512), (3000, 5, 8, None)])
def test_gaussian_hmm_next_row_n1(hip, N, K, D, chunk):
    check_hmm(DEV, N, K, D, chunk)


@pytest.mark.parametrize("H,B,T", [(320, 37, 3), (384, 64, 5), (448, 33, 4), (512, 100, 30), (512, 8192, 3)])
def test_gru_wide_hidden_sizes(hip, H, B, T):
    check_gru_wide(DEV, H, B, T)


@pytest.mark.parametrize("H,B,T", [(128, 37, 3), (192, 64, 5), (256, 100, 30), (256, 4096, 4)])
def test_gru_wide_kernels_at_small_hidden_sizes(hip, H, B, T):
    check_gru_wide_small(DEV, H, B, T)


def test_gemm_group_shared_output(hip):
    check_gemm_group_shared_output(DEV)


def test_linear_group_of_a_narrow_input(hip):
    check_linear_group(DEV)


@pytest.mark.parametrize("B,T,F,K,pad", [(7, 9, 24, 64, 2), (3, 50, 10, 128, 0), (5, 31, 32, 192, 2), (4096, 30, 24, 512, 2), (301, 15, 24, 512, 2), (290, 30, 24, 64, 2), (64, 60, 12, 384, 2), (512, 30, 32, 512, 2), (300, 60, 24, 1024, 2), (64, 60, 12, 768, 2), (40, 31, 32, 640, 0)])
def test_fused_output_head(hip, B, T, F, K, pad):
    check_head_fused(DEV, B, T, F, K, pad)
