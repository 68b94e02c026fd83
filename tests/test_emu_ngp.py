"""The Instant-NGP hot path's kernels -- xr_raymarch / xr_grid / xr_encode / xr_mlp / xr_misc .hip, the SAME sources the GPU
library is built from -- executed on the host by the HIP-on-CPU shim (tests/hip_emu) through the unchanged
`xrnerf_amd.ops` front-end, by re-running the bodies of the GPU parity tests (tests/test_gpu_*.py) at sizes the emulation
finishes in seconds: K1 bit-exact against the reference's own kernels (single and five cascades, overflow, step cap), K2,
compositor forward / backward / inference, K6 / K7, hash-grid gather and binned scatter, SH-4, the fully fused fp32-MFMA
MLP forward and backward, ray generation / Huber / Adam, edge cases.  `emulated_ops` swaps the library handle and the
device-pointer helpers of `ops` for the duration of this module only; the product never runs like this."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'hip_emu'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))


@pytest.fixture(scope='module')
def edev():
    import emulib
    ctx = emulib.emulated_ops()
    dev = ctx.__enter__()
    yield dev
    ctx.__exit__(None, None, None)


@pytest.mark.parametrize('n_rays,calls', [(1000, 3), (4096, 1)])
def test_k1_bit_exact_on_the_host(O, lego, edev, n_rays, calls):
    import test_gpu_raymarch as T
    T.test_k1_bit_exact(O, lego, edev, n_rays, calls)


def test_k1_cascades_overflow_and_k2_on_the_host(O, lego, edev):
    import test_gpu_raymarch as T
    T.test_k1_multi_cascade_bit_exact(O, edev)
    T.test_k1_overflow_and_k2_clip(O, lego, edev)


def test_window_march_equals_one_launch_per_iteration_on_the_host(lego, edev):
    """the series kernels (blockIdx.y = iteration of a refresh window) against one launch per iteration, on the host"""
    import test_gpu_raymarch as T
    T.test_window_march_equals_one_launch_per_iteration(lego, edev)


def test_k1_one_launch_stands_for_a_series_of_chunk_launches(lego, edev):
    """rng_chunk: rays i of ONE launch draw the jitter they would draw as ray i % chunk of launch i / chunk of a series (the
    reference marches a frame in `chunk`-sized launches, advancing its hidden generator once per launch): same samples"""
    import numpy as np
    import torch
    from xrnerf_amd import ops, synthetic as S
    bf = torch.from_numpy(lego['bitfield'])
    o, d, _ = S.training_rays(S.lego_cameras(3), 700, seed=5, H=64, W=64, focal=S.LEGO_FOCAL * 64 / 800)
    to, td = torch.from_numpy(o), torch.from_numpy(d)
    chunk, base = 256, 11
    c1, i1, n1, cnt1 = ops.rays_sampler(to, td, bf, (0.0, 1.0), 0.05, 1.0 / 256, 700 * 64, base, rng_chunk=chunk)
    row = 0
    for k, s0 in enumerate(range(0, 700, chunk)):
        ck, ik, nk, cntk = ops.rays_sampler(to[s0:s0 + chunk].contiguous(), td[s0:s0 + chunk].contiguous(), bf, (0.0, 1.0), 0.05, 1.0 / 256,
                                            chunk * 64, base + k)
        m = int(cntk[1])
        assert np.array_equal(nk[:, 0].numpy(), n1[s0:s0 + chunk, 0].numpy())
        assert np.array_equal(ck[:m].numpy().view(np.uint32), c1[row:row + m].numpy().view(np.uint32))
        row += m
    assert row == int(cnt1[1]) and row > 0


def test_compositor_on_the_host(O, lego, edev):
    import test_gpu_raymarch as T
    T.test_compositor_fwd_bwd_inference(O, lego, edev, 2, 3)          # the config's activations (16-lane groups per ray)
    T.test_compositor_zero_sample_rays(O, edev)
    T.test_fused_compositor_train_equals_k3_huber_k4(O, lego, edev, 3000)
    T.test_fused_compositor_train_equals_k3_huber_k4(O, lego, edev, 1001)


def test_grid_upkeep_raygen_loss_adam_on_the_host(O, lego, edev):
    import test_gpu_raymarch as T
    T.test_k6_grid_samples_bit_exact(O, lego, edev)
    T.test_k7_mark_untrained(O, lego, edev)
    T.test_gen_rays_huber_adam(O, lego, edev)
    T.test_adam_multi_grad_scale_equals_scaling_pass_then_adam(O, edev)


def test_refresh_tail_three_launches_on_the_host(O, lego, edev):
    """K9 + K10 + K11 as three launches (xr_ema_update_bitfield) against the two separate entry points and the oracle, the kernels'
    own sources on the host"""
    import test_gpu_raymarch as T
    T.test_refresh_tail_in_three_launches(O, lego, edev, 8)


@pytest.mark.parametrize('n', [1, 31, 4096])
def test_hashgrid_gather_and_scatter_on_the_host(O, edev, n):
    import test_gpu_tcnn as T
    T.test_hashgrid_fwd_bwd(O, edev, n)


def test_fused_mlp_on_the_host(O, edev):
    import test_gpu_tcnn as T
    T.test_grid_meta_matches_oracle(O, edev)
    T.test_sh4(O, edev)
    from xrnerf_amd import ops
    old = ops.f32_forward()
    try:
        for kind in ('f16x2', 'bf16x3', 'mfma'):        # the fixture of the GPU tests, by hand
            ops.set_f32_forward(kind)
            for n in (1, 32, 33):
                T.test_nerf_mlp_fwd(O, edev, n, kind)
            T.test_nerf_mlp_fwd_asymmetric_weights(O, edev, kind)
    finally:
        ops.set_f32_forward(old)
    for n, nv in ((1, None), (33, None), (300, 250), (70, 0)):   # 8-wave workgroups: ragged tiles, device-side count
        T.test_nerf_mlp_fwd_split_operands_equal_fp32_mfma(edev, n, nv)
    for n in (32, 100):
        T.test_nerf_mlp_bwd(O, edev, n)


def test_density_query_with_the_splat_inside_on_the_host(edev):
    import test_gpu_tcnn as T
    from xrnerf_amd import ops
    for kind in ('f16x2', 'bf16x3', 'mfma'):
        old = ops.f32_forward()
        ops.set_f32_forward(kind)
        try:
            T.test_density_query_with_the_splat_inside_equals_query_plus_splat(edev, 1000, kind)
        finally:
            ops.set_f32_forward(old)


def test_deeper_topologies_on_the_host(O, edev, monkeypatch):
    """(5, 5) -- tiny-cuda-nn's default depth -- and other depths: the streamed fused kernels (weights through LDS layer by layer, the
    activation scratch area, the ReLU bits, the per-layer dW reduction) and the layer-by-layer path, both on the host build"""
    import test_gpu_tcnn as T
    T.test_deeper_topologies_against_the_oracle(O, edev, 5, 5, 300, 270, 'streamed', monkeypatch)
    T.test_deeper_topologies_against_the_oracle(O, edev, 2, 1, 65, None, 'streamed', monkeypatch)
    T.test_deeper_topologies_against_the_oracle(O, edev, 3, 4, 65, None, 'layered', monkeypatch)


def test_mlp_backward_arithmetic_modes_on_the_host(O, edev, monkeypatch):
    """the bf16-split products of the backward (dW, dX chain, and the opt-in split recompute) on the host build"""
    import test_gpu_tcnn as T
    for arith in ('f32', 'b2', 'h2f'):
        monkeypatch.setenv('XR_MLP_BWD_DW', arith)
        T.test_nerf_mlp_bwd(O, edev, 100)
        T.test_nerf_mlp_bwd_live_rows(O, edev, 100, None, 'f32')
    monkeypatch.setenv('XR_MLP_BWD_DW', 'bf16')                 # not a mode: an error, not a silent default
    with pytest.raises(Exception, match='XR_MLP_BWD_DW'):
        T.test_nerf_mlp_bwd(O, edev, 32)
    monkeypatch.delenv('XR_MLP_BWD_DW')
    T.test_nerf_mlp_bwd_split_recompute_differs_by_relu_kinks_only(edev, 1200, monkeypatch)


def test_mlp_backward_on_live_rows_on_the_host(O, edev):
    """the live-row compaction in front of the backward (both precisions), ragged / clipped / single-live-row launches"""
    import test_gpu_tcnn as T
    for n, nv in ((31, None), (100, None), (1500, 1200), (1100, 0)):
        for precision in ('f32', 'f16'):
            T.test_nerf_mlp_bwd_live_rows(O, edev, n, nv, precision)
    for n, nv in ((100, None), (17000, 16500), (900, 0)):
        T.test_shared_live_row_list_through_backward_and_scatter(O, edev, n, nv)


def test_reference_precision_mlp_on_the_host(edev):
    """the fp16-MFMA mode (v_mfma_f32_32x32x16_f16 emulated with its operand layout) against the numpy fp16 statement"""
    import test_gpu_tcnn as T
    for n in (32, 100):
        T.test_nerf_mlp_reference_precision_mode(edev, n)


def test_edge_cases_on_the_host(O, edev):
    import test_gpu_edge_cases as T
    T.test_fully_occupied_grid_hits_the_step_cap(O, edev)
    T.test_single_ray_and_all_miss(O, edev)
    T.test_ragged_sample_counts_through_mlp_and_encode(O, edev)
    T.test_live_row_list_edge_cases(O, edev)


def test_emulation_leaves_the_product_untouched(edev):
    """after this module the product path is back to: no library for host tensors"""
    from xrnerf_amd import ops
    assert ops._on_device(__import__('torch').zeros(1)) is True      # inside the emulation window

