"""-m gpu: parity of the HIP path (through the C-ABI, libstylex_hip.so) against the CPU
oracle and the golden vectors captured from the reference.

Tolerances (stated per north_star): fp32 mode 2e-5 relative to the tensor's max-abs (only the
summation order differs: v_mfma_f32_32x32x2_f32 is an fmaf chain); bf16 mode 4e-2 (operands
rounded to bf16, fp32 accumulate, bf16 output tensors: TOLBF = 4e-2 of the tensor's max-abs, doubled for gradients of
two-conv chains); resampling index arithmetic is exact, values to 1e-6."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import hip_backend as hb  # noqa: E402
import ops  # noqa: E402
import stylex_oracle as so  # noqa: E402
import stylex_train as st  # noqa: E402
from conftest import load_golden  # noqa: E402
from lpips_standin import LPIPSStandIn  # noqa: E402
from test_oracle_vs_golden import build_nets_model, close_stats  # noqa: E402
from test_host_logic_cpu import assert_param_stats, check_evalsurface, make_cfg4_trainer, make_trainer, run_steps  # noqa: E402

DEV = "cuda:0"
TOL32, TOLBF = 2e-5, 4e-2


@pytest.fixture(autouse=True)
def hip_impl():
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    prev = ops.use_impl(ops.HipOps)
    ops.set_precision("fp32")
    hb.load_library()  # fails loudly if the extension is missing
    yield
    ops.set_precision("fp32")
    ops.use_impl(prev)


def close(a, b, tol=TOL32, what=""):
    a = torch.as_tensor(np.asarray(a) if not isinstance(a, torch.Tensor) else a).detach().double().cpu()
    b = b.detach().double().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(1e-3, a.abs().max().item())
    err = (a - b).abs().max().item()
    assert err <= tol * scale, "%s: max err %.3e (scale %.3e, tol %.1e)" % (what, err, scale, tol)


def cl(t):
    return t.to(DEV).contiguous(memory_format=torch.channels_last)


CONV_CASES = [
    # B, C, N, H, W, k, stride, pad
    (2, 16, 24, 8, 8, 3, 1, 1),
    (2, 3, 64, 16, 16, 3, 1, 1),      # first D conv (scalar-gather path, K = 27)
    (3, 6, 10, 9, 7, 3, 1, 1),        # odd sizes, C % 4 != 0
    (2, 64, 64, 16, 16, 3, 2, 1),     # blur-downsample conv
    (2, 8, 12, 8, 8, 1, 2, 0),        # conv_res 1x1 stride 2
    (2, 32, 3, 8, 8, 1, 1, 0),        # toRGB shape
    (1, 128, 160, 12, 12, 3, 1, 1),   # >1 N tile, >1 K chunk
    (2, 40, 72, 10, 10, 3, 1, 1),     # ragged channel chunks
    (5, 20, 36, 6, 6, 3, 2, 1),
    (2, 512, 512, 4, 4, 3, 1, 1),
    (1, 64, 64, 64, 64, 3, 1, 1),     # many M tiles
    (2, 64, 64, 32, 32, 3, 1, 1),     # bf16: LDS-halo kernel, 8x32 tiles
    (2, 32, 32, 16, 16, 3, 1, 1),     # bf16: LDS-halo kernel, 16x16 tiles, N=32 variant
    (1, 40, 96, 40, 48, 3, 1, 1),     # halo kernel with partial tiles and a ragged channel chunk
    (3, 128, 24, 24, 40, 3, 1, 1),    # halo kernel: N < tile, H not a multiple of the tile
    (2, 3, 64, 16, 16, 1, 2, 0),      # first conv_res: RGB padded to one 16-byte slot, 1x1 stride 2
    (4, 256, 136, 8, 8, 3, 1, 1),     # bf16: transpose-read wgrad, 128x128 tiles, ragged N, several pixel splits
    (2, 64, 128, 16, 16, 1, 2, 0),    # conv_res at a real channel ratio
    (3, 72, 64, 5, 7, 3, 2, 1),       # stride 2 with odd sizes, ragged 64x64 channel tiles
    (2, 128, 128, 32, 32, 3, 1, 1),   # bf16: LDS-DMA halo kernel (16x32 px x 128 n tiles), whole tiles
    (1, 136, 160, 20, 40, 3, 1, 1),   # bf16: LDS-DMA kernel, ragged: 8-channel tail, partial N tile, partial pixel tiles
    (1, 128, 256, 16, 32, 3, 1, 1),   # bf16: LDS-DMA kernel, one pixel tile, two N tiles
    (1, 128, 160, 16, 40, 3, 1, 1),   # bf16: LDS-DMA kernel with 128-channel tiles (N not a multiple of 64), ragged N
    (2, 64, 64, 24, 64, 3, 1, 1),     # bf16: LDS-DMA kernel with 64-channel tiles (two blocks per CU), ragged rows
    (1, 128, 64, 16, 40, 3, 1, 1),    # bf16: 64-channel tiles, 8 chunks, ragged columns (the 128->64 data gradient)
    (2, 64, 192, 31, 31, 5, 1, 2),    # round 6: 5x5 (LPIPS-AlexNet's second layer, bf16 speed mode) on the generic kernel
    (1, 8, 12, 9, 7, 5, 1, 2),        # 5x5, odd sizes, one 8-channel slot
    (3, 192, 384, 15, 15, 3, 1, 1),   # round 6: 15 x 15 (LPIPS-AlexNet taps) as one partial 16 x 16 tile of the LDS-halo kernel
    (2, 256, 256, 14, 14, 3, 1, 1),   # 14 x 14 (ResNet-18 layer3)
    (2, 64, 64, 12, 13, 3, 1, 1),     # the smallest image the halo kernel takes
]


@pytest.mark.against_definition
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_triad_vs_oracle(case, prec):
    """Conv fwd, data-gradient and weight-gradient against F.conv2d + autograd on the CPU."""
    B, C, N, H, W, k, s, p = case
    g = torch.Generator().manual_seed(hash(case) % 2 ** 31)
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(N, C, k, k, generator=g) / (C * k * k) ** 0.5
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    y_ref = F.conv2d(xr, wr, None, s, p)
    r = torch.randn(y_ref.shape, generator=g)
    (y_ref * r).sum().backward()
    ops.set_precision(prec)
    tol = TOL32 if prec == "fp32" else TOLBF
    xd, wd = cl(x).requires_grad_(), w.to(DEV).requires_grad_()
    y = ops.conv2d(xd, wd, None, s, p)
    (y * cl(r)).sum().backward()
    close(y_ref, y, tol, "fwd")
    close(xr.grad, xd.grad, tol, "dgrad")
    close(wr.grad, wd.grad, tol, "wgrad")


@pytest.mark.against_definition
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 16, 24, 8, 8), (2, 64, 64, 32, 32), (2, 32, 32, 16, 32)])
def test_fused_conv_epilogue_and_scales(shape, prec):
    """The fused forms of the C-ABI (modulation in_scale, demod out_scale, bias, transposed noise,
    residual merge, lrelu) and the scaled data/weight gradients against plain torch."""
    B, C, N, H, W = shape
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(N, C, 3, 3, generator=g) / (C * 9) ** 0.5
    s_in = torch.rand(B, C, generator=g) + 0.5
    s_out = torch.rand(B, N, generator=g) + 0.5
    bias, nw, nb = (torch.randn(N, generator=g) for _ in range(3))
    S = max(H, W)
    inoise = torch.rand(B, S, S, generator=g)
    res = torch.randn(B, N, H, W, generator=g)
    P = hb.BF16_ACT if prec == "bf16" else hb.F32
    adt = hb.act_dtype(P)
    tol = TOL32 if prec == "fp32" else TOLBF
    cl_ = lambda t: cl(t).to(adt)
    ref = F.conv2d(x * s_in[:, :, None, None], w, None, 1, 1) * s_out[:, :, None, None] + bias.view(1, -1, 1, 1)
    ref = ref + inoise[:, :W, :H].transpose(1, 2)[:, None] * nw.view(1, -1, 1, 1) + nb.view(1, -1, 1, 1)
    ref = F.leaky_relu((ref + res) * 0.5, 0.2)
    d = lambda t: t.to(DEV)
    y = hb.conv2d_fwd(cl_(x), d(w), 1, 1, P, bias=d(bias), lrelu=True, in_scale=d(s_in), out_scale=d(s_out),
                      noise=d(inoise), noise_w=d(nw), noise_b=d(nb), residual=cl_(res), res_scale=0.5)
    close(ref, y, tol, "fused fwd")
    dy = torch.randn(B, N, H, W, generator=g)
    xr = x.clone().requires_grad_()
    wr = w.clone().requires_grad_()
    z = F.conv2d(xr * s_in[:, :, None, None], wr, None, 1, 1) * s_out[:, :, None, None]
    (z * dy).sum().backward()
    dx = hb.conv2d_bwd_data(cl_(dy), d(w), tuple(x.shape), 1, 1, P, in_scale=d(s_out), out_scale=d(s_in))
    close(xr.grad, dx, tol, "scaled dgrad")
    dw = hb.conv2d_bwd_weight(cl_(x), cl_(dy), tuple(w.shape), 1, 1, P, x_scale=d(s_in), dy_scale=d(s_out))
    close(wr.grad, dw, tol, "scaled wgrad")


@pytest.mark.against_definition
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_fast_fused_path_matches_differentiable_path(prec):
    """The once-differentiable fused Functions (one forward kernel; fused backward bookkeeping) give the
    same outputs and first-order gradients as the fully differentiable composition."""
    ops.set_precision(prec)
    tol = 2e-5 if prec == "fp32" else 2e-2
    g = torch.Generator().manual_seed(3)
    B, C, N, H = 2, 16, 24, 16
    x0 = torch.randn(B, C, H, H, generator=g)
    w0 = torch.randn(N, C, 3, 3, generator=g) / 12
    b0 = torch.randn(N, generator=g)
    res0 = torch.randn(B, N, H // 2, H // 2, generator=g)
    st0 = torch.randn(B, C, generator=g) * 0.5
    inoise = torch.rand(B, 32, 32, 1, generator=g).to(DEV)
    nw0, nb0 = torch.randn(N, generator=g), torch.randn(N, generator=g)
    wr0 = torch.randn(3, C, 1, 1, generator=g)

    def run(fast):
        prev = ops.set_fast(fast)
        try:
            leaves = [t.clone().to(DEV).requires_grad_() for t in (x0, w0, b0, res0, st0, nw0, nb0, wr0)]
            x, w, b, res, st_, nw, nb, wr = leaves
            xc = x.contiguous(memory_format=torch.channels_last)
            y1 = ops.conv2d(xc, w, b, 1, 1, lrelu=True)                                  # conv+bias+lrelu

            y3 = ops.conv2d(xc, w, b, 2, 1, residual=res, res_scale=0.7)                  # merge epilogue
            y4 = ops.modconv_noise_act(xc, st_, w, inoise, nw, nb)                        # G conv
            y5 = ops.modulated_conv2d(xc, st_, wr, demod=False)                           # toRGB
            loss = (y1 * y1).mean() + (y3 * y3).mean() + (y4 * y4).mean() + (y5 * y5).mean()
            loss.backward()
            return [y1, y3, y4, y5], [t.grad for t in leaves]
        finally:
            ops.set_fast(prev)

    ys_s, gs_s = run(False)
    ys_f, gs_f = run(True)
    for i, (a, b_) in enumerate(zip(ys_s, ys_f)):
        close(a, b_, tol, "fast fwd %d" % i)
    for nm, a, b_ in zip(("x", "w", "bias", "res", "style", "nw", "nb", "w_rgb"), gs_s, gs_f):
        close(a, b_, tol * 5, "fast grad " + nm)


@pytest.mark.against_definition
@pytest.mark.parametrize("shape", [(2, 64, 128, 64, 64), (1, 128, 64, 48, 80), (2, 64, 64, 32, 32), (1, 128, 256, 64, 96)])
def test_s2d_downsample_path_matches_strided_conv(shape):
    """blur -> space-to-depth -> 3x3/s1 halo conv with skipped zero taps == blur -> 3x3/s2 conv (+bias+res)*c,
    outputs and all gradients (bf16 mode: this path only exists on the bf16 kernels).  The first two shapes send the
    data gradient through the LDS-DMA kernel's space-to-depth mode (>= 16x32 half-resolution image; the second with
    ragged tiles), the third keeps it on the register-staged kernel."""
    ops.set_precision("bf16")
    g = torch.Generator().manual_seed(8)
    B, C, N, H, W = shape
    x0 = torch.randn(B, C, H, W, generator=g)
    w0 = torch.randn(N, C, 3, 3, generator=g) / 24
    b0 = torch.randn(N, generator=g)
    r0 = torch.randn(B, N, H // 2, W // 2, generator=g)

    def run(fast):
        prev = ops.set_fast(fast)
        try:
            x, w, b, r = (t.clone().to(DEV).requires_grad_() for t in (x0, w0, b0, r0))
            y = ops.blur_down(x.contiguous(memory_format=torch.channels_last), w, b, r, 0.7)
            (y.float() ** 2).mean().backward()
            return y, [t.grad for t in (x, w, b, r)]
        finally:
            ops.set_fast(prev)

    # float64 CPU reference of the same op
    xr, wr, br, rr = (t.clone().double().requires_grad_() for t in (x0, w0, b0, r0))
    yr = (F.conv2d(so.blur3x3_reflect(xr), wr, br, stride=2, padding=1) + rr) * 0.7
    (yr ** 2).mean().backward()
    for fast in (False, True):
        y, gs = run(fast)
        close(yr, y, TOLBF, "y fast=%s" % fast)
        for nm, a, b_ in zip(("x", "w", "bias", "res"), (xr.grad, wr.grad, br.grad, rr.grad), gs):
            close(a, b_, TOLBF * 2, "grad %s fast=%s" % (nm, fast))


@pytest.mark.against_definition
@pytest.mark.parametrize("tile_mode", ["0", "1"])
@pytest.mark.parametrize("shape", [(3, 64, 64, 24, 64), (1, 128, 128, 32, 32), (2, 64, 256, 8, 96), (5, 192, 64, 16, 32),
                                   (3, 128, 128, 16, 16), (1, 64, 64, 32, 48)])
def test_s2d_data_gradient_all_subpositions_kernel(shape, tile_mode, monkeypatch):
    """conv_s2d_dgrad.hip (round 5: one staged gradient halo for all four sub-positions) against the fp64 definition — the
    gradient of conv2d(x, w, stride 2, pad 1) w.r.t. x, stored space-to-depth — and against the per-sub-position kernel
    it replaces (STYLEX_S2D_DGRAD=0): same bf16 operands and, where that was the LDS-DMA kernel (>= 16 x 32 images), the same K
    order, so the two must agree to the last bit there.  Shapes (B, C, N, half-res H, W): ragged tile lists (3, 5 images over 256 blocks), one / several
    channel groups, 2-8 K stages, tiles on every image border, 16-pixel-wide tiles (the last two); tile_mode 1 = the 8-wave
    blocks (64-channel groups, one block per CU) instead of the 4-wave default."""
    B, C, N, H, W = shape
    monkeypatch.setenv("STYLEX_S2D_DGRAD_TILE", tile_mode)
    g = torch.Generator().manual_seed(21)
    w = (torch.randn(N, C, 3, 3, generator=g) / 24).bfloat16().float().to(DEV)
    dy = torch.randn(B, N, H, W, generator=g).bfloat16().to(DEV).contiguous(memory_format=torch.channels_last)
    _, wb2 = hb.pack_weight_s2d(w)
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("STYLEX_S2D_DGRAD", mode)
        outs[mode] = hb.conv2d_bwd_data(dy, None, (B, 4 * C, H, W), 1, 1, hb.BF16_ACT, packed=wb2, w_shape=(N, 4 * C, 3, 3),
                                        s2d_c=C).float().cpu()
    x = torch.zeros(B, C, 2 * H, 2 * W, dtype=torch.float64, requires_grad=True)
    (gx,) = torch.autograd.grad(F.conv2d(x, w.double().cpu(), stride=2, padding=1), x, dy.double().cpu())
    ref = gx.view(B, C, H, 2, W, 2).permute(0, 3, 5, 1, 2, 4).reshape(B, 4 * C, H, W)
    scale = float(ref.abs().max())
    assert float((outs["1"].double() - ref).abs().max()) / scale < 6e-3  # bf16 output rounding (2^-8 of the value)
    if W % 32 == 0 and H >= 16:
        assert torch.equal(outs["1"], outs["0"]), float((outs["1"] - outs["0"]).abs().max())
    else:  # the old path of 16-wide / 8-row images is the register-staged kernel: another K order
        assert float((outs["1"] - outs["0"]).abs().max()) / scale < 6e-3


@pytest.mark.against_definition
@pytest.mark.parametrize("tile_mode", ["0", "1", "2"])
@pytest.mark.parametrize("shape", [(3, 64, 128, 24, 64), (1, 128, 256, 32, 32), (5, 32, 256, 16, 16), (2, 256, 512, 16, 48),
                                   (3, 8, 64, 16, 64), (2, 40, 64, 16, 16)])
def test_s2d_forward_pipelined_kernel(shape, tile_mode, monkeypatch):
    """conv_s2d_fwd.hip (round 5: the stride-2 forward as one pipelined K loop over the sub-position phases) against the
    fp64 definition conv2d(x, w, stride 2, pad 1) in its three forms — bias only, (conv + bias + residual tensor) * c, and the
    one-launch block tail (conv + bias + 1x1 conv of the block input) * c — and against the kernels it replaces
    (STYLEX_S2D_FWD=0).  Shapes (B, C_res, C = N, half-res H, W): 32- and 16-pixel-wide tiles, ragged tile lists, tiles on
    every image border, 1-4 channel groups, the 64-channel tile with a ragged residual segment (8 / 40 channels: the padded RGB
    input of the first block); tile_mode 0 = by tile count (here: 256 x 128 tiles on 8 waves), 1 = the 8-wave blocks
    with 256 x 256 / 512 x 128 tiles, 2 = 256 x 128 tiles in 4-wave blocks (the default of launches with more than 256 tiles)."""
    B, CR, C, H, W = shape
    monkeypatch.setenv("STYLEX_S2D_FWD_TILE", tile_mode)
    g = torch.Generator().manual_seed(22)
    w = (torch.randn(C, C, 3, 3, generator=g) / (3 * C ** 0.5)).bfloat16().float().to(DEV)
    wres = (torch.randn(C, CR, generator=g) / CR ** 0.5).bfloat16().to(DEV)
    bias = torch.randn(C, generator=g).to(DEV)
    x = torch.randn(B, C, 2 * H, 2 * W, generator=g).bfloat16().float().to(DEV)
    xs = cl(torch.randn(B, CR, H, W, generator=g).bfloat16().to(DEV))
    r = cl(torch.randn(B, C, H, W, generator=g).bfloat16().to(DEV))
    x2 = cl(x.view(B, C, H, 2, W, 2).permute(0, 3, 5, 1, 2, 4).reshape(B, 4 * C, H, W).bfloat16())
    wf2, _ = hb.pack_weight_s2d(w)
    ws = (C, 4 * C, 3, 3)
    y0 = F.conv2d(x.double(), w.double(), bias.double(), stride=2, padding=1)
    refs = {"bias": y0, "residual": (y0 + r.double()) * 0.7, "merged": (y0 + F.conv2d(xs.double(), wres.double()[:, :, None, None])) * 0.7}
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("STYLEX_S2D_FWD", mode)
        hb._S2D_RES_OK.clear()
        outs[mode, "bias"] = hb.conv2d_fwd(x2, None, 1, 1, hb.BF16_ACT, bias=bias, packed=wf2, w_shape=ws, s2d_c=C)
        outs[mode, "residual"] = hb.conv2d_fwd(x2, None, 1, 1, hb.BF16_ACT, bias=bias, residual=r, res_scale=0.7, packed=wf2, w_shape=ws,
                                               s2d_c=C)
        if hb.s2d_res_supported(tuple(x2.shape), C, C, CR):
            outs[mode, "merged"] = hb.conv2d_s2d_res_fwd(x2, wf2, xs, wres, bias, C, C, 0.7)
    hb._S2D_RES_OK.clear()
    assert ("1", "merged") in outs  # the new kernel takes the block tail at every shape of this test
    for (mode, form), y in outs.items():
        ref = refs[form]
        err = float((y.double() - ref).abs().max() / ref.abs().max())
        assert err < 8e-3, (mode, form, err)  # bf16 output rounding (2^-8 of the value) + fp32 summation order
    for form in ("bias", "residual"):  # same operands, fp32 accumulation in another order: at most one bf16 ulp apart
        d = float((outs["1", form].double() - outs["0", form].double()).abs().max() / refs[form].abs().max())
        assert d < 8e-3, (form, d)


@pytest.mark.parametrize("case", [(2, 64, 128, 32, 32, 3, 1, 1, 0), (2, 64, 128, 16, 16, 1, 1, 0, 0), (4, 512, 512, 4, 4, 3, 1, 1, 0),
                                  (2, 8, 64, 32, 32, 3, 1, 1, 0), (2, 64, 64, 32, 32, 3, 1, 1, 64)])
def test_weight_gradient_output_stage_is_the_two_launches_it_replaces(case):
    """stylex_conv2d_bwd_weight_ex / _s2d (round 5): dw = (accumulate ? dw : 0) + out_scale * sum inside the reduce launch must
    be BIT-identical to the plain weight gradient followed by an in-place multiply and autograd's add into an existing
    gradient — for every weight-gradient kernel family (pipelined LDS-DMA, 1x1 / transpose-read, <= 8 px, padded RGB, the
    folded stride-2 form) and for the bias sums that ride the same launch."""
    B, C, N, H, W, k, s, p, s2d = case
    g = torch.Generator().manual_seed(31)
    xc = 4 * C if s2d else C
    x = cl(torch.randn(B, xc, H, W, generator=g).bfloat16().to(DEV))
    dy = cl(torch.randn(B, N, (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1, generator=g).bfloat16().to(DEV))
    old = torch.randn(N, C, k, k, generator=g).to(DEV)
    oldb = torch.randn(N, generator=g).to(DEV)
    sc = 0.7071067811865476
    if s2d:
        plain = hb.conv2d_bwd_weight_s2d(x, dy, (N, C, 3, 3), hb.BF16_ACT)
        scaled = hb.conv2d_bwd_weight_s2d(x, dy, (N, C, 3, 3), hb.BF16_ACT, out_scale=sc)
        acc = hb.conv2d_bwd_weight_s2d(x, dy, (N, C, 3, 3), hb.BF16_ACT, out_scale=sc, accumulate_into=old.clone())
        assert torch.equal(scaled, plain * sc) and torch.equal(acc, old + plain * sc)
        return
    plain, pb = hb.conv2d_bwd_weight(x, dy, (N, C, k, k), s, p, hb.BF16_ACT, want_bias_sum=True)
    scaled, sb = hb.conv2d_bwd_weight(x, dy, (N, C, k, k), s, p, hb.BF16_ACT, want_bias_sum=True, out_scale=sc)
    acc, ab = hb.conv2d_bwd_weight(x, dy, (N, C, k, k), s, p, hb.BF16_ACT, want_bias_sum=True, out_scale=sc,
                                   accumulate_into=old.clone(), accumulate_bias_into=oldb.clone())
    assert torch.equal(scaled, plain * sc) and torch.equal(acc, old + plain * sc)
    assert (pb is None) == (sb is None) == (ab is None)
    if pb is not None:
        assert torch.equal(sb, pb * sc) and torch.equal(ab, oldb + pb * sc)
    only_w = hb.conv2d_bwd_weight(x, dy, (N, C, k, k), s, p, hb.BF16_ACT, accumulate_into=old.clone())
    assert torch.equal(only_w, old + plain)


@pytest.mark.parametrize("case", [(64, 64, 64, True), (64, 128, 32, True), (256, 256, 16, True), (64, 64, 2, False), (3, 64, 64, True)])
def test_twice_used_block_accumulates_in_the_reduce_launch(case):
    """A DiscriminatorBlock applied to two inputs in ONE graph (the encoder of a generator phase: E(x) and E(G(x)), reference
    stylex_train.py:1383-1395): the second node to run adds its weight / bias gradients into the first node's tensors inside
    its reduce launches (ops._gacc_*, stylex_conv2d_bwd_weight_ex accumulate) instead of handing the engine a second tensor
    per parameter.  Same two rounded fp32 operations as the engine's add: every gradient bit-identical to the path without
    it (STYLEX_GRAD_ACC=0), and fewer add launches."""
    from torch.profiler import ProfilerActivity, profile

    cin, cout, size, down = case
    ops.set_precision("bf16")
    torch.manual_seed(11)
    blk = st.DiscriminatorBlock(cin, cout, downsample=down).to(DEV)
    with torch.no_grad():
        for p in blk.parameters():
            if p.dim() == 1:
                p.normal_(0, 0.1)
    g = torch.Generator(device=DEV).manual_seed(12)
    xa = torch.randn(3, cin, size, size, device=DEV, generator=g)
    xb = torch.randn(3, cin, size, size, device=DEV, generator=g)
    results, adds = [], []
    keep = ops._GACC_ON
    try:
        for on in (False, True):
            ops._GACC_ON = on
            blk.zero_grad()
            ops.set_fast(True)
            try:
                with profile(activities=[ProfilerActivity.CPU]) as prof:
                    ya, yb = blk(xa), blk(xb)
                    ((ya.float() ** 2).mean() + 0.5 * (yb.float() ** 3).mean()).backward()
                    torch.cuda.synchronize()
            finally:
                ops.set_fast(False)
            adds.append(sum(1 for e in prof.events() if e.name in ("aten::add_", "aten::add")))
            results.append([p.grad.clone() for p in blk.parameters()])
    finally:
        ops._GACC_ON = keep
    for (name, _), a, b in zip(blk.named_parameters(), *results):
        assert torch.equal(a, b), (name, float((a - b).abs().max()))
    if cin != 3:  # (the padded-RGB block hands the engine slices of its gradients: it keeps the engine's adds)
        assert adds[1] <= adds[0] - 3, adds  # at least the block's three or four weight gradients


@pytest.mark.parametrize("case", [(64, 64, 32, True), (128, 128, 8, True), (64, 64, 16, False)])
def test_twice_used_block_on_two_streams(case):
    """The same double use with the two forward passes on DIFFERENT HIP streams (E(x) on the caller's stream, E(G(x)) on a
    Trainer._fork side stream): autograd replays every node on its forward stream, so the node that adds into the first
    node's gradient tensors runs on another stream than their producer, behind the engine's back (round-5 ADVICE, medium).
    ops._gacc_get makes its stream wait for the producer's and joins it into the caller's stream at the end of the pass.
    The producer's stream is held up by a long sleep kernel enqueued right before backward(): without the wait the adds
    run BEFORE the tensors they add into are written.  Gradients must equal the path without in-launch accumulation bit
    for bit, read on the caller's stream right after backward() as any PyTorch program would."""
    cin, cout, size, down = case
    ops.set_precision("bf16")
    torch.manual_seed(13)
    blk = st.DiscriminatorBlock(cin, cout, downsample=down).to(DEV)
    with torch.no_grad():
        for p in blk.parameters():
            if p.dim() == 1:
                p.normal_(0, 0.1)
    g = torch.Generator(device=DEV).manual_seed(14)
    xa = torch.randn(3, cin, size, size, device=DEV, generator=g)
    xb = torch.randn(3, cin, size, size, device=DEV, generator=g)
    side = torch.cuda.Stream()
    main = torch.cuda.current_stream()
    results = []
    keep = ops._GACC_ON
    try:
        for on in (False, True, True):
            ops._GACC_ON = on
            blk.zero_grad()
            ops.set_fast(True)
            try:
                ya = blk(xa)  # first use: caller's stream (its backward nodes run LAST and add into the other use's tensors)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    yb = blk(xb)  # second use: side stream (its backward nodes run first and produce the tensors)
                    lb = 0.5 * (yb.float() ** 3).mean()
                main.wait_stream(side)
                loss = (ya.float() ** 2).mean() + lb
                with torch.cuda.stream(side):
                    torch.cuda._sleep(200_000_000)  # ~0.1 s: everything the backward enqueues on `side` starts late
                loss.backward()
                results.append([p.grad.clone() for p in blk.parameters()])  # on the caller's stream, no synchronize
            finally:
                ops.set_fast(False)
            torch.cuda.synchronize()
    finally:
        ops._GACC_ON = keep
    for (name, _), a, b, c in zip(blk.named_parameters(), *results):
        assert torch.equal(a, b) and torch.equal(a, c), (name, float((a - b).abs().max()), float((a - c).abs().max()))


@pytest.mark.against_definition
@pytest.mark.parametrize("case", [(4, 128, 128, 8, 8, 3, 1, 1), (4, 256, 128, 16, 16, 3, 2, 1), (4, 128, 256, 32, 32, 1, 1, 0),
                                  (3, 128, 128, 2, 2, 3, 1, 1), (5, 128, 128, 4, 8, 3, 1, 1)])
def test_small_grid_weight_gradient_dma_kernel(case, monkeypatch):
    """conv_wgrad_tr_dma_kernel (round 5: LDS-DMA staging for the 128 x 128 tile of the general weight-gradient kernel — <= 8 px
    3x3 layers, the small stride-2 down convs, 1x1 convs at any power-of-two grid) against the fp64 gradient of F.conv2d and
    against the register-staged kernel it replaces (STYLEX_WGRAD_TR_DMA=0): same bf16 operands, fp32 accumulation."""
    B, C, N, H, W, k, s, p = case
    g = torch.Generator().manual_seed(41)
    x = cl(torch.randn(B, C, H, W, generator=g).bfloat16().to(DEV))
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    dy = cl(torch.randn(B, N, Ho, Wo, generator=g).bfloat16().to(DEV))
    w = torch.zeros(N, C, k, k, dtype=torch.float64, device=DEV, requires_grad=True)
    (ref,) = torch.autograd.grad(F.conv2d(x.double(), w, stride=s, padding=p), w, dy.double())
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("STYLEX_WGRAD_TR_DMA", mode)
        outs[mode] = hb.conv2d_bwd_weight(x, dy, (N, C, k, k), s, p, hb.BF16_ACT).double()
    scale = float(ref.abs().max())
    for mode in ("1", "0"):
        assert float((outs[mode] - ref).abs().max()) / scale < 2e-5, mode
    assert float((outs["1"] - outs["0"]).abs().max()) / scale < 2e-5


@pytest.mark.against_definition
def test_conv_bias_lrelu_and_second_order():
    """conv+bias+lrelu, then a gradient-penalty style double backward through it."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 6, 8, 8, generator=g)
    w1 = torch.randn(10, 6, 3, 3, generator=g) * 0.2
    b1 = torch.randn(10, generator=g) * 0.1
    w2 = torch.randn(4, 10, 3, 3, generator=g) * 0.2
    b2 = torch.randn(4, generator=g) * 0.1

    def net(xx, p, conv):
        h = conv(xx, p[0], p[1], 1, 1, True)
        return conv(h, p[2], p[3], 2, 1, False)

    def ref_conv(xx, w, b, s, p, act):
        y = F.conv2d(xx, w, b, s, p)
        return F.leaky_relu(y, 0.2) if act else y

    outs = []
    for dev, conv in (("cpu", ref_conv), (DEV, ops.conv2d)):
        ps = [t.clone().to(dev).requires_grad_() for t in (w1, b1, w2, b2)]
        xx = (x.clone().to(dev) if dev == "cpu" else cl(x)).requires_grad_()
        y = net(xx, ps, conv)
        (gx,) = torch.autograd.grad(y.sum(), xx, create_graph=True)
        pen = ((gx.reshape(2, -1).norm(2, dim=1) - 1) ** 2).mean()
        pen.backward()
        outs.append((y, gx, pen, [q.grad for q in ps]))
    (y0, g0, p0, gr0), (y1, g1, p1, gr1) = outs
    close(y0, y1, what="y")
    close(g0, g1, what="gx")
    close(p0, p1, 1e-4, "penalty")
    for a, b, nm in zip(gr0, gr1, ("w1", "b1", "w2", "b2")):
        if a is None or b is None:  # None == structurally zero gradient
            other = b if a is None else a
            assert other is None or float(other.abs().max()) == 0.0, nm
            continue
        close(a, b, 2e-4, "second-order grad " + nm)


@pytest.mark.against_definition
@pytest.mark.parametrize("cn", [(128, 256), (64, 64), (128, 160)])
def test_dma_halo_kernel_bias_lrelu_epilogue(cn):
    """The LDS-DMA forward kernel's own epilogue (bias + LeakyReLU / ReLU) and its use as data gradient, for the
    64-channel tile variant (N % 64 == 0) and the 128-channel one (N = 160)."""
    ops.set_precision("bf16")
    try:
        g = torch.Generator().manual_seed(21)
        B, (C, N), H, W = 2, cn, 24, 64
        x = torch.randn(B, C, H, W, generator=g)
        w = torch.randn(N, C, 3, 3, generator=g) / (C * 9) ** 0.5
        b = torch.randn(N, generator=g)
        for act, ref_act in ((True, lambda t: F.leaky_relu(t, 0.2)), ("relu", F.relu), (False, lambda t: t)):
            xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
            yr = ref_act(F.conv2d(xr, wr, br, 1, 1))
            r = torch.randn(yr.shape, generator=g)
            (yr * r).sum().backward()
            xd, wd, bd = cl(x).requires_grad_(), w.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
            prev = ops.set_fast(True)
            try:
                y = ops.conv2d(xd, wd, bd, 1, 1, lrelu=act)
                (y.float() * cl(r)).sum().backward()
            finally:
                ops.set_fast(prev)
            close(yr, y, TOLBF, "dma fwd act=%s" % act)
            # gradients: a bf16 pre-activation at the rounding threshold flips its (Leaky)ReLU mask, which moves single
            # gradient entries by O(1) of their size — compared in the L2 sense (the no-activation case is tight)
            for nm, a, b_ in (("dgrad", xr.grad, xd.grad), ("wgrad", wr.grad, wd.grad), ("bias", br.grad, bd.grad)):
                rel = float((a.double() - b_.detach().double().cpu()).norm() / a.double().norm())
                assert rel <= (6e-2 if act == "relu" else 3e-2 if act else 1e-2), "dma %s act=%s: relative L2 error %.3e" % (nm, act, rel)
    finally:
        ops.set_precision("fp32")


@pytest.mark.parametrize("shape", [(3, 32, 20, 12), (2, 512, 4, 4), (2, 64, 33, 7), (2, 8, 16, 16), (1, 128, 64, 64),
                                   (2, 256, 40, 40)])
def test_torgb_streaming_kernels(shape):
    """to-RGB (RGBBlock :611/:621 = Conv2DMod(C, 3, 1, demod=False)) on the streaming kernels: output, input
    gradient, style gradient and weight gradient against the fp32 per-sample-weight formulation of :647-667."""
    B, C, H, W = shape
    g = torch.Generator().manual_seed(C + H)
    x = torch.randn(B, C, H, W, generator=g)
    style = torch.randn(B, C, generator=g) * 0.5
    w = torch.randn(3, C, 1, 1, generator=g) / C ** 0.5
    r = torch.randn(B, 3, H, W, generator=g)
    xr, sr, wr = x.clone().requires_grad_(), style.clone().requires_grad_(), w.clone().requires_grad_()
    wmod = wr[None] * (sr[:, None, :, None, None] + 1)  # [B, 3, C, 1, 1]
    yr = F.conv2d(xr.reshape(1, B * C, H, W), wmod.reshape(B * 3, C, 1, 1), groups=B).reshape(B, 3, H, W)
    (yr * r).sum().backward()
    ops.set_precision("bf16")
    prev = ops.set_fast(True)
    try:
        xd, sd, wd = cl(x).requires_grad_(), style.to(DEV).requires_grad_(), w.to(DEV).requires_grad_()
        assert hb.torgb_ok(ops._act(xd))
        y = ops.modulated_conv2d(xd, sd, wd, demod=False)
        assert tuple(y.shape) == (B, 3, H, W)
        (y.float() * r.to(DEV)).sum().backward()
    finally:
        ops.set_fast(prev)
        ops.set_precision("fp32")
    close(yr, y, TOLBF, "to-RGB out")
    close(xr.grad, xd.grad, TOLBF, "to-RGB gx")
    close(sr.grad, sd.grad, TOLBF, "to-RGB gstyle")
    close(wr.grad, wd.grad, TOLBF, "to-RGB gw")


def test_torgb_rejects_unsupported_channel_counts():
    x = cl(torch.randn(1, 24, 4, 4)).to(torch.bfloat16)
    assert not hb.torgb_ok(x)
    with pytest.raises(hb.StylexHipError):
        hb.torgb_fwd(x, torch.ones(1, 24, device=DEV), torch.ones(3, 24, 1, 1, device=DEV))


def test_bf16_strip_blur_vs_oracle():
    """bf16 tensors take the column-strip blur kernels (forward and adjoint, plain and space-to-depth output):
    compare with the oracle on the bf16-rounded input; sizes exercise a partial last strip and both borders."""
    ops.set_precision("bf16")
    try:
        g = torch.Generator().manual_seed(11)
        for (B, C, H, W) in ((2, 8, 10, 12), (1, 16, 8, 8), (2, 64, 36, 20)):
            x = torch.randn(B, C, H, W, generator=g).bfloat16().float()
            r = torch.randn(B, C, H, W, generator=g).bfloat16().float()
            xr = x.clone().double().requires_grad_()
            yr = so.blur3x3_reflect(xr)
            (yr * r.double()).sum().backward()
            xd = x.to(DEV).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_()
            y = ops.blur3x3(xd)
            assert y.dtype == torch.bfloat16
            (y.float() * r.to(DEV)).sum().backward()
            close(yr, y, 1e-2, "strip blur fwd %s" % ((B, C, H, W),))
            close(xr.grad, xd.grad, 1e-2, "strip blur adjoint %s" % ((B, C, H, W),))
            # space-to-depth output form and its adjoint
            y2 = hb.blur3x3_s2d_fwd(xd.detach())
            want = F.pixel_unshuffle(yr.detach().float(), 2).view(B, C, 4, H // 2, W // 2).transpose(1, 2).reshape(
                B, 4 * C, H // 2, W // 2)
            close(want, y2, 1e-2, "strip blur s2d fwd")
            r2 = F.pixel_unshuffle(r, 2).view(B, C, 4, H // 2, W // 2).transpose(1, 2).reshape(B, 4 * C, H // 2, W // 2)
            gx2 = hb.blur3x3_s2d_bwd(r2.to(DEV).bfloat16().contiguous(memory_format=torch.channels_last))
            close(xr.grad, gx2, 1e-2, "strip blur s2d adjoint")
    finally:
        ops.set_precision("fp32")


def test_resampling_exact_index_rules():
    g = load_golden("ops")
    x = torch.from_numpy(g["up/x"])
    xd = cl(x).requires_grad_()
    for nm, fn in (("up", ops.upsample2x), ("blur", ops.blur3x3)):
        xd.grad = None
        y = fn(xd)
        close(g[nm + "/y"], y, 1e-6, nm + " fwd vs reference")
        (y * cl(torch.from_numpy(g[nm + "/r"]))).sum().backward()
        close(g[nm + "/gx"], xd.grad, 1e-6, nm + " adjoint vs reference")
    # index rule: a one-hot input lights up exactly the outputs the rule names, with exact weights
    for n in (2, 3, 8):
        for i in range(n):
            e = torch.zeros(1, 4, n, n)
            e[0, :, i, i] = 1.0
            yu = ops.upsample2x(cl(e)).cpu()
            assert torch.equal(yu, so.upsample2x_bilinear_explicit(e)), (n, i)
            yb = ops.blur3x3(cl(e)).cpu()
            assert torch.equal(yb, so.blur3x3_reflect_explicit(e)), (n, i)
    # odd channel count exercises the scalar path
    x3 = torch.randn(2, 3, 6, 10)
    close(so.upsample2x_bilinear(x3), ops.upsample2x(cl(x3)), 1e-6, "up C=3")
    close(so.blur3x3_reflect(x3), ops.blur3x3(cl(x3)), 1e-6, "blur C=3")


def test_bias_act_and_noise_act_and_sumsq():
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 12, 8, 8, generator=g)
    b = torch.randn(12, generator=g)
    xd, bd = cl(x).requires_grad_(), b.to(DEV).requires_grad_()
    xr, br = x.clone().requires_grad_(), b.clone().requires_grad_()
    y = ops.conv2d  # noqa
    from ops import _BiasAct

    yd = _BiasAct.apply(xd, bd)
    yr = F.leaky_relu(xr + br.view(1, -1, 1, 1), 0.2)
    r = torch.randn(yr.shape, generator=g)
    (yd * cl(r)).sum().backward()
    (yr * r).sum().backward()
    close(yr, yd, 1e-6)
    close(xr.grad, xd.grad, 1e-6)
    close(br.grad, bd.grad, 1e-5)
    inoise = torch.rand(2, 16, 16, 1, generator=g)
    nw, nb = torch.randn(12, generator=g), torch.randn(12, generator=g)
    from cpu_ops import CpuOracleOps

    close(CpuOracleOps.noise_act(x, inoise, nw, nb), ops.noise_act(cl(x), inoise.to(DEV), nw.to(DEV), nb.to(DEV)), 1e-6)
    v = torch.randn(5, 3 * 64 * 64 + 3, generator=g)
    close(v.pow(2).sum(1), ops.rowwise_sumsq(v.to(DEV)), 1e-5)


@pytest.mark.against_definition
@pytest.mark.parametrize("tag", ["mod3", "mod1", "mod512"])
def test_conv2dmod_vs_reference_golden(tag):
    g = load_golden("ops")
    ci, co, k, demod, hw, b, wseed = (int(v) for v in g[tag + "/cfg"])
    torch.manual_seed(wseed)
    conv = st.Conv2DMod(ci, co, k, demod=bool(demod)).to(DEV)
    x = cl(torch.from_numpy(g[tag + "/x"])).requires_grad_()
    y = torch.from_numpy(g[tag + "/y"]).to(DEV).requires_grad_()
    o = conv(x, y)
    (o * cl(torch.from_numpy(g[tag + "/r"]))).sum().backward()
    close(g[tag + "/out"], o, what="out")
    close(g[tag + "/gx"], x.grad, what="gx")
    close(g[tag + "/gy"], y.grad, 1e-4, "gstyle")
    if tag + "/gw" in g.files:
        close(g[tag + "/gw"], conv.weight.grad, what="gw")
    else:
        close(g[tag + "/gw_slice"], conv.weight.grad[:4, :4], what="gw slice")
        close_stats(g[tag + "/gw_stats"], conv.weight.grad.cpu())


@pytest.mark.parametrize("size", [16, 32])
def test_network_parity_vs_reference_golden(size):
    g = load_golden("nets_%d" % size)
    prev = ops.use_impl(ops.HipOps)
    m = build_nets_model(g, cls=st.StylEx).to(DEV)
    ops.use_impl(prev)
    w, inoise, x = (torch.from_numpy(g[n]).to(DEV) for n in ("w", "inoise", "x"))
    rgb, coords = m.G(w, inoise, get_style_coords=True)
    close(g["rgb"], rgb, what="G rgb")
    close(g["coords"], coords, what="style coords")
    close(g["d_out"], m.D(x), what="D")
    close(g["enc_out"], m.encoder(x), what="encoder")
    close(g["d_of_g"], m.D(rgb), 1e-4, "D(G)")


def test_gp_and_pl_double_backward_vs_reference_golden():
    g = load_golden("losses")
    s, cap, fmax = (int(v) for v in g["config"])
    torch.manual_seed(int(g["seed"]))
    m = st.StylEx(s, network_capacity=cap, fmap_max=fmax, rank=0)
    x = cl(torch.from_numpy(g["gp/x"])).requires_grad_()
    gp = st.gradient_penalty(x, m.D(x))
    close(g["gp/value"], gp, 1e-4, "gp value")
    m.D.zero_grad()
    gp.backward()
    grads = dict(m.D.named_parameters())
    for n, gs in zip(g["gp/grad_names"], g["gp/grad_stats"]):
        pr = grads[str(n)]  # a structurally-zero gradient (conv biases under a pure GP loss) may come back as None
        close_stats(gs, (torch.zeros_like(pr) if pr.grad is None else pr.grad).cpu(), 5e-4)
    close(g["gp/grad_fc_w"], m.D.fc.weight.grad, 2e-4, "gp grad fc")
    close(g["gp/grad_b0_res_w"], m.D.blocks[0].conv_res.weight.grad, 2e-4, "gp grad conv_res")
    w = torch.from_numpy(g["pl/w"]).to(DEV).requires_grad_()
    img = m.G(w, torch.from_numpy(g["pl/inoise"]).to(DEV))
    torch.manual_seed(int(g["pl/noise_seed"]))
    pl = st.calc_pl_lengths(w, img)
    close(g["pl/lengths"], pl, 1e-4, "pl lengths")
    m.G.zero_grad()
    ((pl - 0.3) ** 2).mean().backward()
    grads = dict(m.G.named_parameters())
    for n, gs in zip(g["pl/grad_names"], g["pl/grad_stats"]):
        close_stats(gs, grads[str(n)].grad.cpu(), 2e-3, head_atol=1e-5)
    close(g["pl/grad_w"], w.grad, 5e-4, "pl grad w")


def assert_trajectory(rows, gold, tag, floor=2e-3, factor=4.0, cap=5e-2):
    """Multi-step loss scalars against the reference's, inside a band tied to what the REFERENCE leaves around itself.
    The untrained GAN amplifies rounding differences from call to call and D's Adam steps are lr * sign(gradient) per
    element; tests/golden/steps_envelope.npz (oracle/make_golden.py gen_steps_envelope) holds, for every step fixture,
    the per-call spread of the reference's own scalars when only its CPU summation order (2 threads) or its inputs by a
    relative 1e-6 (two seeds: the size of an fp32 kernel's summation-order difference) change.  The bound per call is
    `factor` x that measured spread, never below `floor` (2e-3 = north_star's 1e-3 with the margin MIOpen's algorithm
    choice for the frozen nets needs: call 3 of gae2_alt measured 2.4e-3 on one box and 1e-3 on another) and never
    above `cap` — instead of the free 4x-per-call growth this test used in round 3."""
    rows, gold = np.asarray(rows, dtype=np.float64), np.asarray(gold, dtype=np.float64)
    spread = load_golden("steps_envelope")["spread_" + tag]
    dev = np.nanmax(np.abs(rows - gold) / np.maximum(np.abs(gold), 1e-2), axis=1)
    print("trajectory %s: HIP-vs-reference deviation per call %s | reference-vs-itself spread %s"
          % (tag, " ".join("%.1e" % v for v in dev), " ".join("%.1e" % v for v in spread)))
    for k in range(len(gold)):
        tol = min(cap, max(floor, factor * float(spread[min(k, len(spread) - 1)])))
        np.testing.assert_allclose(rows[k], gold[k], rtol=tol, atol=tol, equal_nan=True,
                                   err_msg="train() call %d (band %.1e = max(%.0e, %g x reference spread %.1e))"
                                           % (k, tol, floor, factor, float(spread[min(k, len(spread) - 1)])))


@pytest.mark.parametrize("tag", ["gae1_alt", "gae2_alt", "gae2_noalt", "gae2_pl", "gae2_aug"])
def test_trainer_step_parity_gpu(tag, tmp_path):
    """Trainer.train() on the HIP path reproduces the reference's loss scalars (1e-3, north_star)."""
    g = load_golden("steps_" + tag)
    tr, n = make_trainer(g, tmp_path, device=torch.device(DEV))
    rows = run_steps(tr, n)
    gold = g["scalars"]
    np.testing.assert_allclose(rows[0], gold[0], rtol=2e-4, atol=2e-5, equal_nan=True)
    assert_trajectory(rows, gold, tag)
    # all 226 parameter tensors after the last step vs the reference's: sums to 2e-3 of the abs-sum; single elements to
    # n * lr_D (an element whose gradient is summation-order noise takes a +-lr Adam step in either direction)
    assert_param_stats(tr, g, head_atol=n * 3e-4)


def test_eval_ema_truncation_surface_vs_reference_golden_gpu(tmp_path):
    """N3 on the HIP path: EMA / reset_parameter_averaging, truncate_style, generate_truncated and the three image
    grids of evaluate() against the reference Trainer's own outputs (tests/golden/evalsurface_16.npz)."""
    check_evalsurface(tmp_path, device=torch.device(DEV), tol=2e-3, head_atol=3 * 3e-4)


def test_config4_mobilenet_pl_step_parity_gpu(tmp_path):
    """BASELINE config 4 in miniature on the HIP path: Trainer(classifier_name='mobilenet') with the MobileNetV2
    checkpoint on the GPU, R1 + path-length step (both double backwards), vs the reference's Trainer + its own
    MobileNet wrapper (tests/golden/steps_cfg4.npz)."""
    g = load_golden("steps_cfg4")
    tr, n, batches = make_cfg4_trainer(g, tmp_path, device=torch.device(DEV))
    close(g["logits_batch0"], tr.classifier.classify_images(batches[0].to(DEV)), 1e-4, "MobileNetV2 logits")
    rows = run_steps(tr, n)
    gold = g["scalars"]
    np.testing.assert_allclose(rows[0], gold[0], rtol=2e-4, atol=2e-5, equal_nan=True)
    assert_trajectory(rows, gold, "cfg4")
    assert_param_stats(tr, g, head_atol=n * 3e-4)


def test_newarch_on_hip(tmp_path):
    """N4: the conditional-discriminator architecture (reference stylex_train_new.py) on the HIP path: init / conditional
    D / W-with-probabilities parity and three Trainer.train() calls vs the reference's own module."""
    import stylex_train_new as stn
    from test_host_logic_cpu import check_newarch_init

    check_newarch_init(torch.device(DEV))
    g = load_golden("steps_newarch")
    tr, n = make_trainer(g, tmp_path, device=torch.device(DEV), trainer_cls=stn.Trainer)
    rows = run_steps(tr, n)
    gold = g["scalars"]
    np.testing.assert_allclose(rows[0], gold[0], rtol=2e-4, atol=2e-5, equal_nan=True)
    assert_trajectory(rows, gold, "newarch")
    assert_param_stats(tr, g, head_atol=n * 3e-4)


def test_block_level_goldens_on_hip():
    """GeneratorBlock (transposed noise, toRGB + upsample), DiscriminatorBlock and StyleVectorizer of tests/golden/
    ops.npz (captured from the reference's modules with their own weights) through the HIP modules."""
    g = load_golden("ops")

    def load_sd(mod, prefix):
        mod.load_state_dict({k[len(prefix):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(prefix)})
        return mod.to(DEV)

    blk = load_sd(st.GeneratorBlock(20, 8, 12, upsample=True, upsample_rgb=True), "gblock/sd/")
    xo, rgb, sc = blk(*(torch.from_numpy(g["gblock/" + n]).to(DEV) for n in ("x", "prev", "istyle", "inoise")))
    close(g["gblock/xo"], xo, what="gblock x")
    close(g["gblock/rgb"], rgb, what="gblock rgb")
    close(g["gblock/coords"], sc, what="gblock coords")
    with torch.no_grad():  # the fused fast path (one kernel per conv) must agree as well
        xo2, rgb2, _ = blk(*(torch.from_numpy(g["gblock/" + n]).to(DEV) for n in ("x", "prev", "istyle", "inoise")))
    close(g["gblock/xo"], xo2, what="gblock x (fused)")
    close(g["gblock/rgb"], rgb2, what="gblock rgb (fused)")
    dblk = load_sd(st.DiscriminatorBlock(6, 10, downsample=True), "dblock/sd/")
    x = torch.from_numpy(g["dblock/x"]).to(DEV)
    close(g["dblock/y"], dblk(x), what="dblock")
    with torch.no_grad():
        close(g["dblock/y"], dblk(x), what="dblock (fused)")
    torch.manual_seed(int(g["svec/seed"]))
    sv = st.StyleVectorizer(24, 8, lr_mul=0.1).to(DEV)
    close(g["svec/w"], sv(torch.from_numpy(g["svec/z"]).to(DEV)), what="style vectorizer")
    g32 = load_golden("nets_32")
    prev = ops.use_impl(ops.HipOps)
    m = build_nets_model(g32, cls=st.StylEx).to(DEV)
    ops.use_impl(prev)
    close(g32["s_out"], m.S(torch.from_numpy(g32["w"]).to(DEV)[:, 0]), what="S(w) of nets_32")


def test_bf16_step_band_vs_reference_golden(tmp_path):
    """The benchmarked mode (bf16 MFMA operands + bf16 activation tensors, fused Adam) against the fp32 reference
    trajectory steps_gae2_alt.  The band has two parts, both with a stated origin (round-4 VERDICT item 3c: no free cap):
    (1) the arithmetic — every conv rounds its two operands to bf16 (relative 2^-9 each); a loss scalar sits behind ~30
    chained convs of forward (+ as many of backward for the penalty), a random walk of sqrt(60)*2^-8 = 3e-2, so 5e-2 of
    max(1,|x|) for d_loss / g_loss / rec / kl on the first call, growing 1.3x per call (a bf16-rounded Adam trajectory:
    measured 1.6e-2, 3.5e-2, 4.7e-2, 3.6e-2, 7.6e-2 for calls 0-4 outside the KL column); the gradient penalty is a
    squared norm of a double-rounded gradient: 1e-1.  (2) the trajectory's own sensitivity — tests/golden/
    bf16_band_gae2_alt.npz (tools/gen_bf16_band.py, measured on the MI355X) holds the spread of the bf16 path against
    ITSELF when one loss is scaled by 1 +- 1e-6 or an arithmetic-neutral switch is flipped: < 1 % everywhere except the KL
    column, which this fixture amplifies to 18 % two calls later (2.02 vs 2.39 at the third call, golden 2.06).  Band =
    (1) + 2 x (2), per call and scalar.  g_loss is evaluated on the UPDATED discriminator: until round 3 this test
    excluded it — that was the stale operand cache under the fused Adam; the bf16 mode gives 3.89 vs 3.88."""
    g = load_golden("steps_gae2_alt")
    band = load_golden("bf16_band_gae2_alt")
    ops.set_precision("bf16")
    try:
        tr, n = make_trainer(g, tmp_path, device=torch.device(DEV))
        rows = run_steps(tr, n)
    finally:
        ops.set_precision("fp32")
    gold = g["scalars"]
    spread = np.asarray(band["spread"])
    print("bf16 rows\n", rows, "\ngolden\n", gold, "\nself-spread\n", spread)
    assert np.isfinite(rows[:, :5]).all()
    assert spread.shape == (n, 4) and float(spread[:, :3].max()) < 0.05, "the fixture's premise: only KL is hypersensitive"
    for i in range(n):
        tol = 5e-2 * 1.3 ** i + 2.0 * spread[i]
        scale = np.maximum(1.0, np.abs(gold[i, :4]))
        assert (np.abs(rows[i, :4] - gold[i, :4]) <= tol * scale).all(), (i, tol, rows[i], gold[i])
    assert abs(rows[0, 4] - gold[0, 4]) <= 1e-1 * max(1.0, abs(gold[0, 4])), (rows[0, 4], gold[0, 4])


def test_loss_curve_short_horizon_vs_reference():
    """Config-1 shape (64 px, B=4, GAE=2): the HIP fp32 path follows the reference's own loss trajectory
    (tests/golden/curve_64.npz) while the trajectory is still determined by the arithmetic — the first two
    train() calls agree to 1e-3 (north_star's bound).  Beyond that the untrained GAN is chaotic: the reference's
    own CPU ops re-run with 4 threads instead of 8 leave the 1e-3 band at step 4 and reach 10 % at step 8
    (tools/curve_check.py), so a longer horizon cannot be asserted for ANY implementation."""
    import importlib.util
    import os

    spec = importlib.util.spec_from_file_location(
        "curve_check", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "curve_check.py"))
    cc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cc)
    got, ref = cc.run(n=3, precision="fp32")
    err = np.abs(got[:, :4] - ref[:, :4]) / np.maximum(1.0, np.abs(ref[:, :4]))
    assert np.isfinite(got[:, :4]).all()
    assert err[:2].max() <= 1e-3, err
    assert err[2].max() <= 2e-2, err  # third call: losses of order 1e4, still within 2 %


def test_config2_full_size_bf16_tracks_fp32(tmp_path):
    """BASELINE config 2 whole, as bench.py builds it: 256 px, batch 32, GAE 2, ResNet-18 classifier — two train() calls
    (the first is a gradient-penalty step) through every full-size kernel path of the step, in the bf16 speed mode AND
    in the fp32 parity mode from the same seeds.  Asserted: everything finite, and the first call of the speed mode
    inside the bf16 band of the parity mode for EVERY scalar — d_loss, g_loss (evaluated on the discriminator D's
    optimiser step has just updated: 1020 vs 1038; a stale operand cache gave 9.3 here until round 3), rec, kl within
    5e-2, the penalty within 1e-1.  The second call already shows the untrained GAN's excursions and is only required
    to be finite."""
    import argparse
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench

    res = {}
    try:
        for prec in ("bf16", "fp32"):
            ops.set_precision(prec)
            hb.pack_cache_clear()
            a = argparse.Namespace(batch=32, image_size=256, gae=2, classifier="resnet", workdir=str(tmp_path / prec),
                                   precision=prec)
            bench.seed_all(42)
            tr = bench.build_trainer(a, torch.device(DEV), 0, 1)
            rows = []
            for _ in range(2):
                tr.train()
                rows.append([tr.d_loss, tr.g_loss, tr.total_rec_loss, tr.total_kl_loss, tr.last_gp_loss])
            torch.cuda.synchronize()
            assert all(bool(torch.isfinite(p).all()) for p in tr.StylEx.parameters())
            res[prec] = np.array(rows, dtype=np.float64)
            del tr
            torch.cuda.empty_cache()
    finally:
        ops.set_precision("fp32")
    print("config 2, two calls:", res)
    assert np.isfinite(res["bf16"]).all() and np.isfinite(res["fp32"]).all(), res
    lo, ref = res["bf16"][0], res["fp32"][0]
    scale = np.maximum(1.0, np.abs(ref))
    assert (np.abs(lo[:4] - ref[:4]) <= 5e-2 * scale[:4]).all(), (lo, ref)
    assert abs(lo[4] - ref[4]) <= 1e-1 * scale[4], (lo, ref)


def test_config2_bf16_tracks_fp32_call_by_call(tmp_path):
    """The benchmarked mode against the fp32 parity mode over 20 train() calls of BASELINE config 2 (256 px, batch 32, GAE 2,
    ResNet-18, lr 2e-4) — round-5 VERDICT item 6: an assertion instead of "second call finite".

    What can be asserted.  tests/golden/bf16_band_config2.npz (tools/gen_bf16_band_config2.py, measured on the MI355X) holds
    the fp32 path against ITSELF at this size when ONE parameter element starts one unit in the last place away (a relative
    6e-8): by the second call the scalars differ by up to 8e-4, by the seventh by 5 % — the first Adam steps move every weight
    by +-lr whatever the size of its gradient, so a rounding-level difference in a near-zero gradient becomes an lr-sized
    difference in a weight, an amplification of ~1e4 per call.  A bf16 rounding (2^-9) is past saturation after ONE call: two
    free-running trajectories cannot be held to any band beyond call 0 (the record is in the fixture: 31 % in d_loss at the
    second call, both finite and of the same magnitude for all 20), for ANY reduced-precision implementation.

    What is asserted instead: the bf16 arithmetic at 20 successive states of the REAL trajectory.  Before every call the bf16
    Trainer takes over the fp32 Trainer's complete state (parameters, EMA copies, both Adam states, step counter, path-length
    mean, and the random streams), both run that one call, and every scalar of the call must agree inside the one-call
    arithmetic band: 5e-2 of max(1, |x|) for d, rec, kl (operands rounded to bf16, ~60 chained convs: sqrt(60) * 2^-8 = 3e-2; measured
    <= 2e-2, 5e-3, 7e-3), 1e-1 for the gradient penalty (a squared norm of a twice-rounded gradient; measured 3e-3); g_loss has its own
    statement below (it sits behind the call's own D step).  The fixture's premise (the amplification) is asserted as well."""
    import argparse
    import os
    import random
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench

    fx = load_golden("bf16_band_config2")
    spread = np.asarray(fx["spread"])
    assert spread.shape[1] == 5 and float(spread[1, :4].max()) > 1e-4 and float(spread[4:, :4].max()) > 1e-2, \
        "the fixture's premise: a one-ulp start difference is amplified to the per-cent level within a few calls"
    calls = 20

    def build(prec):
        ops.set_precision(prec)
        hb.pack_cache_clear()
        a = argparse.Namespace(batch=32, image_size=256, gae=2, classifier="resnet", workdir=str(tmp_path / prec), precision=prec)
        bench.seed_all(42)
        return bench.build_trainer(a, torch.device(DEV), 0, 1)

    def rng_get():
        return torch.get_rng_state(), torch.cuda.get_rng_state(), np.random.get_state(), random.getstate()

    def rng_set(st):
        torch.set_rng_state(st[0])
        torch.cuda.set_rng_state(st[1])
        np.random.set_state(st[2])
        random.setstate(st[3])

    def one_call(tr, prec):
        ops.set_precision(prec)
        hb.pack_cache_clear()
        tr.train()
        row = [tr.d_loss, tr.g_loss, tr.total_rec_loss, tr.total_kl_loss, tr.last_gp_loss]
        torch.cuda.synchronize()
        return np.array(row, dtype=np.float64)

    try:
        ref, low = build("fp32"), build("bf16")
        rows = []
        for i in range(calls):
            with torch.no_grad():
                low.StylEx.load_state_dict(ref.StylEx.state_dict())
            low.StylEx.G_opt.load_state_dict(ref.StylEx.G_opt.state_dict())
            low.StylEx.D_opt.load_state_dict(ref.StylEx.D_opt.state_dict())
            hb.mark_updated(list(low.StylEx.parameters()))
            low.steps, low.pl_mean = ref.steps, ref.pl_mean
            st0 = rng_get()
            a = one_call(ref, "fp32")
            st1 = rng_get()
            rng_set(st0)
            b = one_call(low, "bf16")
            rng_set(st1)
            rows.append((a, b))
    finally:
        ops.set_precision("fp32")
    np.set_printoptions(precision=4, suppress=True, linewidth=160)
    dev = np.array([np.abs(b - a) / np.maximum(1.0, np.abs(a)) for a, b in rows])
    print("fp32 rows (d, g, rec, kl, gp):\n", np.array([a for a, _ in rows]), "\nbf16 from the same state, relative deviation:\n", dev)
    assert np.isfinite(np.array([b for _, b in rows])).all() and np.isfinite(np.array([a for a, _ in rows])).all()
    out = os.path.join(root, "gpurun_out")
    if os.path.isdir(out):  # the record behind the bands (copied to profiles/ by the round's tooling)
        with open(os.path.join(out, "config2_bf16_call_by_call.txt"), "w") as f:
            f.write("# 20 train() calls of config 2; fp32 scalars (d, g, rec, kl, gp), then the bf16 mode's relative deviation from the same state\n")
            f.write(np.array2string(np.array([a for a, _ in rows])) + "\n" + np.array2string(dev) + "\n")
    # d, rec, kl, gp: computed from the common state by one forward (+ one input gradient for gp): the arithmetic band
    band = np.array([5e-2, 0.0, 5e-2, 5e-2, 1e-1])
    for j in (0, 2, 3, 4):
        assert (dev[:, j] <= band[j]).all(), (j, np.argwhere(dev[:, j] > band[j]).ravel(), dev[:, j].max())
    # g_loss is evaluated on the discriminator AFTER the call's own D step, i.e. behind one Adam step taken from bf16 gradients
    # on one side and fp32 gradients on the other.  The fixture shows what such a step does at these states: a one-ulp start
    # difference moves g by 3e-3 one call later (calls 2-3, where the losses are of order 1e5 - 1e7) — the step amplifies by
    # ~5e4, so a bf16-level gradient difference saturates there.  Asserted: every call inside 0.5, and the typical call (the
    # median) inside the one-call arithmetic band of 1e-1.
    assert (dev[:, 1] <= 0.5).all() and float(np.median(dev[:, 1])) <= 1e-1, (dev[:, 1].max(), float(np.median(dev[:, 1])))


def test_config4_full_size_bf16_tracks_fp32_and_is_deterministic(tmp_path):
    """BASELINE config 4 at FULL size, as `bench.py --image-size 128 --classifier mobilenet --pl-every 16 --start-step
    5024` builds it: 128 px, batch 32, GAE 2, MobileNetV2 classifier, starting on step 5024 — a call that carries BOTH
    the gradient penalty (5024 % 4 == 0) and the path-length penalty (> 5000, % 16 == 0; the double backward through
    the generator) — followed by a plain call.  Asserted: everything finite; the first call of the bf16 speed mode inside
    the bf16 band of the fp32 parity mode for every scalar (5e-2; the two penalties 1e-1) and pl_mean likewise; and the
    speed mode run twice from the same seeds gives bit-identical scalars and parameters (fixed-order reductions, also
    through the path-length step)."""
    import argparse
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench

    def run(prec, tag):
        ops.set_precision(prec)
        hb.pack_cache_clear()
        a = argparse.Namespace(batch=32, image_size=128, gae=2, classifier="mobilenet", workdir=str(tmp_path / tag),
                               precision=prec, pl_every=16)
        bench.seed_all(42)
        tr = bench.build_trainer(a, torch.device(DEV), 0, 1)
        assert type(tr.classifier).__name__ == "MobileNet" and tr.pl_every == 16
        tr.steps, tr.pl_mean = 5024, 1.0
        rows = []
        for _ in range(2):
            tr.train()
            rows.append([tr.d_loss, tr.g_loss, tr.total_rec_loss, tr.total_kl_loss, tr.last_gp_loss, tr.pl_mean])
        torch.cuda.synchronize()
        assert all(bool(torch.isfinite(p).all()) for p in tr.StylEx.parameters())
        sums = [float(p.detach().double().sum()) for p in tr.StylEx.parameters()]
        del tr
        torch.cuda.empty_cache()
        return np.array(rows, dtype=np.float64), sums

    try:
        run("bf16", "warm")  # the first Trainer of a process orders its double backward differently
        lo, sums_a = run("bf16", "a")
        lo2, sums_b = run("bf16", "b")
        ref, _ = run("fp32", "f")
    finally:
        ops.set_precision("fp32")
    print("config 4, two calls (d, g, rec, kl, gp, pl_mean):", lo, ref)
    assert np.isfinite(lo).all() and np.isfinite(ref).all(), (lo, ref)
    assert lo[0][5] != 1.0, "the first call must have been a path-length step (pl_mean moved)"
    assert np.array_equal(lo, lo2) and sums_a == sums_b, "bf16 mode: two identically seeded runs must be bit-identical"
    scale = np.maximum(1.0, np.abs(ref[0]))
    assert (np.abs(lo[0][:4] - ref[0][:4]) <= 5e-2 * scale[:4]).all(), (lo[0], ref[0])
    assert (np.abs(lo[0][4:] - ref[0][4:]) <= 1e-1 * scale[4:]).all(), (lo[0], ref[0])


def test_100_call_trajectory_vs_reference_golden(tmp_path):
    """X1 — north_star's "loss curves matching the CPU reference to 1e-3 over 100 steps", asserted on EVERY call: the
    HIP fp32 path against the reference's own 100-call trajectory in a regime the reference holds against itself
    (tests/golden/curve_64_calm.npz: config-1 shape, 64 px / capacity 16 / B=4 / GAE 2, lr 1e-8, steps 4960-5059).
    The 100 calls pin everything train() schedules by step count — gradient penalty every 4th call, the path-length
    penalty and pl_mean EMA at 5024 / 5056, reset_parameter_averaging at 5002, noise / encoder alternation, the rec / KL
    cadence — and the RNG draw order of all of it (one draw out of place changes every later scalar by O(1))."""
    from test_oracle_vs_golden import assert_calm_rows

    g = load_golden("curve_64_calm")
    tr, n = make_trainer(g, tmp_path, device=torch.device(DEV))
    assert n == 100 and tr.steps == 4960
    theta0 = {k: p.detach().double().cpu().reshape(-1) for k, p in tr.StylEx.named_parameters()}
    rows = run_steps(tr, n)
    assert tr.steps == 5060
    gold = g["scalars"]
    dev = np.nanmax(np.abs(rows - gold) / np.maximum(np.abs(gold), 1.0), axis=1)
    print("100-call trajectory: max deviation per call, worst %.2e at call %d; median %.2e" % (dev.max(), int(dev.argmax()), np.median(dev)))
    assert_calm_rows(rows, g)
    # ---- what the parameters DID over the 100 calls (round-4 VERDICT: at lr 1e-8 a bound of n * 3e-8 on single elements is
    # wider than their total movement, so the scalars alone cannot see a wrong gradient or optimiser step after the first
    # calls).  The fixture stores, per tensor, (sum, abs-sum, first 8 values) after call 100; the start values are the
    # seeded init (init goldens), so the MOVEMENT of the first 8 elements and of the tensor sum is known for both runs.
    # A dropped / doubled Adam step, a gradient with the wrong sign or scale on any tensor, or a missing lr group moves
    # these by O(movement); summation-order noise moves them by a +-lr random walk on the noise-dominated elements only.
    params = dict(tr.StylEx.named_parameters())
    mh, mr, dsum_h, dsum_r, untouched = [], [], [], [], 0
    for name, gs in zip(g["param_names"], g["param_stats"]):
        t0, t1 = theta0[str(name)], params[str(name)].detach().double().cpu().reshape(-1)
        k = min(8, t0.numel())
        ref_move = torch.from_numpy(np.asarray(gs[2:2 + k], dtype=np.float64)) - t0[:k]
        if float(ref_move.abs().max()) == 0.0:  # a tensor the reference did not touch in this window
            untouched += 1
            assert float((t1[:k] - t0[:k]).abs().max()) == 0.0, (str(name), "moved although the reference left it alone")
            continue
        mh.append(t1[:k] - t0[:k])
        mr.append(ref_move)
        dsum_h.append(float(t1.sum() - t0.sum()))
        dsum_r.append(float(gs[0]) - float(t0.sum()))
    mh, mr = torch.cat(mh), torch.cat(mr)
    rel = float((mh - mr).norm() / mr.norm())
    big = mr.abs() > 0.3 * float(mr.abs().median())
    agree = float(((mh[big] > 0) == (mr[big] > 0)).double().mean())
    dsum_h, dsum_r = np.array(dsum_h), np.array(dsum_r)
    rel_sum = float(np.linalg.norm(dsum_h - dsum_r) / np.linalg.norm(dsum_r))
    print("parameter movement over the 100 calls: %d tensors moved (%d untouched), median |move| %.2e, max %.2e; HIP vs "
          "reference: relative L2 of the per-element moves %.3f, sign agreement %.3f, relative L2 of the per-tensor sum "
          "moves %.3f" % (len(dsum_r), untouched, float(mr.abs().median()), float(mr.abs().max()), rel, agree, rel_sum))
    # measured on the HIP fp32 path: 0.001 / 1.000 / 0.000 (profiles/r05_f_parity_blocks.txt); a dropped optimiser step of
    # one network is >= 0.1, a wrong-sign gradient 2.0
    assert rel < 0.02 and agree > 0.99 and rel_sum < 0.02, (rel, agree, rel_sum)
    assert_param_stats(tr, g, head_atol=n * 3e-8)


def test_loss_curve_inside_reference_envelope():
    """X1 (north_star: loss curves vs the CPU reference) as a statistical statement.  The untrained GAN is chaotic:
    the reference's OWN trajectory moves when only the summation order of its CPU kernels changes — tests/golden/
    curve_64_envelope.npz holds the curve_64 run repeated with 4 and 2 intra-op threads (oracle/make_golden.py::
    gen_envelope): 1e-7 at step 0, 3e-4 at step 3, 2e-2 at step 6, > 0.1 from step 8 on (saturated).  The HIP fp32
    path differs from the reference by a different summation order too (MFMA chains, fused epilogues), with a larger
    step-0 difference (1.5e-6).  Asserted: until the envelope saturates, the HIP-vs-reference error of every step
    stays within 3x the reference-vs-reference error reached up to three steps later (the head start a 10x larger
    initial perturbation buys at the measured ~5x growth per step), and all 12 steps stay finite."""
    import importlib.util
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("curve_check", os.path.join(root, "tools", "curve_check.py"))
    cc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cc)
    env = load_golden("curve_64_envelope")
    n = len(env["scalars_t4"])
    got, ref = cc.run(n=n, precision="fp32")
    assert np.isfinite(got[:, :4]).all()
    rel = lambda a: (np.abs(a[:, :4] - ref[:, :4]) / np.maximum(1.0, np.abs(ref[:, :4]))).max(1)  # noqa: E731
    e_hip = rel(got)
    e_ref = np.maximum(rel(env["scalars_t4"]), rel(env["scalars_t2"]))  # reference vs reference, per step
    head, c, sat = 3, 3.0, 0.2
    checked = 0
    for k in range(n):
        bound = e_ref[:min(k + head, n - 1) + 1].max()
        if bound >= sat:  # the reference no longer agrees with itself to 20 %: nothing left to assert but finiteness
            break
        assert e_hip[k] <= c * bound, "step %d: HIP error %.3e vs %gx envelope %.3e\n%s\n%s" % (k, e_hip[k], c, bound, e_hip, e_ref)
        checked += 1
    assert checked >= 5, (checked, e_ref)


# ---- full-size, size-independent properties at BASELINE.json's 256 px / B=32 shapes ------------


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("case", [(32, 64, 64, 256, 3, 1, 1), (32, 128, 128, 128, 3, 2, 1), (32, 3, 64, 256, 3, 1, 1),
                                  (32, 512, 512, 16, 3, 1, 1)])
def test_full_size_adjoint_identities(case, prec):
    """<conv(x,w), r> == <x, dgrad(r,w)> == <w, wgrad(x,r)> — holds for any size, checks the three
    kernels against each other on the real benchmark shapes where the CPU oracle is too slow."""
    B, C, N, S, k, s, p = case
    ops.set_precision(prec)
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(B, C, S, S, device=DEV, generator=g).contiguous(memory_format=torch.channels_last).requires_grad_()
    w = (torch.randn(N, C, k, k, device=DEV, generator=g) / (C * k * k) ** 0.5).requires_grad_()
    y = ops.conv2d(x, w, None, s, p)
    r = torch.randn(y.shape, device=DEV, generator=g).contiguous(memory_format=torch.channels_last)
    lhs = (y.double() * r.double()).sum()
    gx, gw = torch.autograd.grad(y, (x, w), r)
    a = (x.detach().double() * gx.double()).sum()
    b = (w.detach().double() * gw.double()).sum()
    tol = 1e-4 if prec == "fp32" else 2e-2
    scale = float(y.detach().double().abs().mul(r.double().abs()).sum()) ** 0.5 + abs(float(lhs))
    assert abs(float(lhs - a)) <= tol * scale and abs(float(lhs - b)) <= tol * scale, (float(lhs), float(a), float(b))
    # spot-check one output pixel block against a direct fp64 evaluation
    xs = x.detach()[:1, :, :10, :10].double().cpu()
    ys = F.conv2d(xs, w.detach().double().cpu(), None, s, p)
    hh = ys.shape[2] - 2  # rows not touched by the crop boundary
    close(ys[:, :, :hh, :hh], y.detach()[:1, :, :hh, :hh], TOL32 * 5 if prec == "fp32" else TOLBF, "spot")


def test_full_size_resampling_roundtrip():
    """Linearity + constants: blur and bilinear x2 preserve constants exactly-ish and commute with scaling."""
    x = torch.rand(32, 64, 128, 128, device=DEV).contiguous(memory_format=torch.channels_last)
    ones = torch.ones_like(x)
    assert float((ops.blur3x3(ones) - 1).abs().max()) <= 1e-6
    assert float((ops.upsample2x(ones) - 1).abs().max()) <= 1e-6
    close(ops.upsample2x(x * 3.0), ops.upsample2x(x) * 3.0, 1e-6)
    # adjoint identity at full size
    r = torch.rand(32, 64, 256, 256, device=DEV).contiguous(memory_format=torch.channels_last)
    xr = x.clone().requires_grad_()
    (ops.upsample2x(xr) * r).sum().backward()
    lhs = float((ops.upsample2x(x).double() * r.double()).sum())
    rhs = float((x.double() * xr.grad.double()).sum())
    assert abs(lhs - rhs) <= 1e-6 * abs(lhs)


def test_attfind_batched_engine_on_hip_vs_reference_notebook_golden():
    """the batched AttFind sweep on the HIP kernels (fp32 mode) reproduces the datasets the
    reference notebook's extraction cell writes (golden made by executing that cell on the reference StylEx)."""
    import attfind
    from test_attfind_cpu import build, check

    g = load_golden("attfind_16")
    m, clf, images, noise = build(g, device=DEV)
    m = m.to(DEV)
    out = attfind.attfind_extraction(m, clf, images, len(images), noise, shift_size=float(g["shift_size"]), chunk=64)
    check(out, g, 2e-4)


# ---- round 2: cross-layer fusions of the DiscriminatorBlock backward ----------------------------------------


@pytest.mark.against_definition
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("case", [(2, 64, 64, 32, 64),    # LDS-DMA halo kernel (C >= 64, >= 16x32 px)
                                  (2, 128, 64, 16, 16),   # register-staged halo kernel (16 px)
                                  (4, 64, 64, 8, 8),      # generic implicit GEMM (+ split-K epilogue)
                                  (2, 32, 24, 16, 32)])   # N % 64 != 0
def test_dgrad_gate_epilogue_matches_unfused(case, prec):
    """STYLEX_EPI_GATE (activation derivative of the layer below in the data gradient's store) == data gradient followed
    by the stand-alone activation-derivative kernel, on every kernel family the data gradient can take."""
    B, C, N, H, W = case
    ops.set_precision(prec)
    P = {"fp32": hb.F32, "bf16": hb.BF16_ACT}[prec]
    adt = hb.act_dtype(P)
    g = torch.Generator(device=DEV).manual_seed(5)
    dy = torch.randn(B, N, H, W, device=DEV, generator=g).to(adt).contiguous(memory_format=torch.channels_last)
    gate = torch.randn(B, C, H, W, device=DEV, generator=g).to(adt).contiguous(memory_format=torch.channels_last)
    w = torch.randn(N, C, 3, 3, device=DEV, generator=g) / (9 * C) ** 0.5
    fused = hb.conv2d_bwd_data(dy, w, (B, C, H, W), 1, 1, P, gate=gate)
    plain = hb.bias_act_bwd(hb.conv2d_bwd_data(dy, w, (B, C, H, W), 1, 1, P), gate)
    # bf16: the unfused path rounds dx to bf16 before the 0.2 multiply, the fused one after: one bf16 ulp
    close(plain.float(), fused.float(), 1e-6 if prec == "fp32" else 8e-3, "gated dgrad")


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_blur_adjoint_gate_and_add_at_even(prec):
    ops.set_precision(prec)
    adt = hb.act_dtype({"fp32": hb.F32, "bf16": hb.BF16_ACT}[prec])
    g = torch.Generator(device=DEV).manual_seed(6)
    for (B, C, H, W) in [(2, 64, 32, 32), (2, 8, 6, 10), (1, 3, 8, 8)]:
        dy = torch.randn(B, C, H, W, device=DEV, generator=g).to(adt).contiguous(memory_format=torch.channels_last)
        gate = torch.randn(B, C, H, W, device=DEV, generator=g).to(adt).contiguous(memory_format=torch.channels_last)
        tol = 1e-6 if prec == "fp32" else 8e-3
        close(hb.bias_act_bwd(hb.blur3x3_bwd(dy), gate).float(), hb.blur3x3_bwd_gate(dy, gate).float(), tol, "blur gate")
        if C % 4 == 0:
            dy2 = hb.blur3x3_s2d_fwd(dy)  # any tensor in the space-to-depth layout
            close(hb.bias_act_bwd(hb.blur3x3_s2d_bwd(dy2), gate).float(), hb.blur3x3_s2d_bwd(dy2, gate=gate).float(), tol,
                  "s2d blur gate")
        src = torch.randn(B, C, (H + 1) // 2, (W + 1) // 2, device=DEV, generator=g).to(adt).contiguous(memory_format=torch.channels_last)
        want = dy.float().clone()
        want[:, :, ::2, ::2] += src.float()
        got = hb.add_at_even_(dy.clone(memory_format=torch.channels_last), src)
        close(want, got.float(), 1e-6 if prec == "fp32" else 8e-3, "add_at_even")


@pytest.mark.against_definition
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("case", [(3, 64, 64, True), (64, 64, 64, True), (64, 128, 32, True), (32, 48, 16, True),
                                  (64, 64, 8, True), (64, 64, 2, False)])
def test_fused_dblock_matches_composable_path(case, prec):
    """ops._DBlockFast (whole DiscriminatorBlock as one autograd node, gate / add-at-even fusions) against the
    composable double-differentiable path of the same module: output and every gradient."""
    import os

    cin, cout, size, down = case
    ops.set_precision(prec)
    torch.manual_seed(11)
    blk = st.DiscriminatorBlock(cin, cout, downsample=down).to(DEV)
    g = torch.Generator(device=DEV).manual_seed(12)
    x = torch.randn(3, cin, size, size, device=DEV, generator=g)
    results = []
    for fused in (False, True):
        os.environ["STYLEX_DBLOCK"] = "1" if fused else "0"
        try:
            blk.zero_grad()
            xr = x.clone().requires_grad_()
            ops.set_fast(fused)
            y = blk(xr)
            r = torch.randn(y.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(13)).to(y.dtype)
            (y.float() * r.float()).sum().backward()
        finally:
            ops.set_fast(False)
            os.environ.pop("STYLEX_DBLOCK", None)
        results.append((y.detach().float(), xr.grad.float(), [p.grad.float().clone() for p in blk.parameters()]))
    (y0, gx0, gp0), (y1, gx1, gp1) = results
    tol = 2e-5 if prec == "fp32" else 2e-2
    close(y0, y1, tol, "dblock out")
    close(gx0, gx1, tol, "dblock gx")
    for (name, _), a, b in zip(blk.named_parameters(), gp0, gp1):
        close(a, b, tol, "dblock grad " + name)


@pytest.mark.parametrize("prec,size", [("fp32", 64), ("bf16", 64), ("bf16", 256)])
def test_gradient_penalty_tangent_pass_matches_double_backward(prec, size):
    """gp_tangent (D(real) and ||dD/dx|| as one first-order node; the penalty's parameter gradient through a tangent
    pass of the gated-linear network) against what it replaces: autograd's double backward through the composable ops
    (reference gradient_penalty :296-303).  Loss = hinge-like term on D(real) + 10 * mean((norm - 1)^2): outputs, norms
    and EVERY parameter gradient.  fp32: rounding only; bf16: the band of two bf16 evaluation orders."""
    import gp_tangent

    ops.set_precision(prec)
    torch.manual_seed(21)
    D = st.DiscriminatorE(size, network_capacity=16, fmap_max=512).to(DEV)
    with torch.no_grad():
        for p in D.parameters():  # biases away from zero, weights as initialised
            if p.dim() == 1:
                p.normal_(0, 0.1)
    b = 4 if size <= 64 else 2
    real = torch.rand(b, 3, size, size, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5))
    a = torch.tensor([1.0, 0.0, 1.0, 1.0][:b], device=DEV) / b  # hinge derivative: some samples inactive

    def loss_of(out, norms):
        return (out * a).sum() + 10 * ((norms - 1) ** 2).mean()

    res = {}
    os.environ["STYLEX_GP_TANGENT"] = "2"  # the tangent path in any precision
    # same primal arithmetic on both sides: the fused block's residual conv as a second K segment of the stride-2 conv
    # (round 4) skips one bf16 rounding that the composable double-backward path has — a 0.4 % difference of block 0's
    # output that flips gates further down and moves the gradient of a 2x2 px layer by 10 %: noise between two bf16
    # evaluation orders, not what this test compares
    os.environ["STYLEX_RES_FOLD"] = "0"
    try:
        for mode in ("double", "tangent"):
            D.zero_grad()
            if mode == "double":
                ops.set_fast(False)
                x = real.clone().requires_grad_()
                out = D(x)
                norms = st.gradient_norms(x, out)
            else:
                assert gp_tangent.supported(D, real)
                ops.set_fast(True)
                out, norms = gp_tangent.d_real_with_norms(D, real)
            ops.set_fast(False)
            loss_of(out.float(), norms).backward()
            res[mode] = (out.detach().float(), norms.detach(), {n: p.grad.clone() for n, p in D.named_parameters()})
    finally:
        ops.set_fast(False)
        os.environ.pop("STYLEX_GP_TANGENT", None)
        os.environ.pop("STYLEX_RES_FOLD", None)
        ops.set_precision("fp32")
    tol = 2e-4 if prec == "fp32" else 4e-2
    close(res["double"][0], res["tangent"][0], 2e-5 if prec == "fp32" else 2e-2, "D(real)")
    close(res["double"][1], res["tangent"][1], tol, "gradient norms")
    for n in res["double"][2]:
        close(res["double"][2][n], res["tangent"][2][n], tol, "grad " + n)


def test_graph_replay_matches_eager(tmp_path):
    """Whole-step HIP-graph replay (Trainer(graphs=True)): the same seeds through the eager enqueue and through the
    captured graphs (static input buffers, device-side layer split, three graphs per step shape, both shapes) give the
    same loss trajectory.  fp32 mode, gp_every=2 so that both step shapes are captured and replayed within 8 calls;
    the two paths run the same kernels — what differs is the summation order of the style gradient (torch.where
    instead of expand) and nothing else, so the bound is the 5-step bound of the golden tests."""
    g = load_golden("steps_gae2_alt")
    rows = {}
    # the frozen classifier / LPIPS run on MIOpen: pin its algorithm choice, which otherwise depends on what the process
    # ran before (find cache) and on whether a capture is open — logits then differ at 1e-7 between the two paths and
    # the untrained GAN amplifies that past any bound within a few steps (same pin as tests/test_hip_determinism_gpu.py)
    prev_det = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    for graphs in (False, True):
        tr, _ = make_trainer(g, tmp_path, device=torch.device(DEV))
        tr2 = st.Trainer(name="g%d" % graphs, base_dir=str(tmp_path), image_size=tr.image_size,
                         network_capacity=tr.network_capacity, fmap_max=tr.fmap_max, batch_size=tr.batch_size,
                         gradient_accumulate_every=2, lr=2e-4, ttur_mult=1.5, rec_scaling=1, kl_scaling=1,
                         classifier=tr.classifier, lpips_fn=tr.lpips_fn, classifier_name="resnet", evaluate_every=10 ** 9,
                         save_every=10 ** 9, device=torch.device(DEV), graphs=True, graph_warmup=1 if graphs else 10 ** 9,
                         gp_every=2)
        import random as _r

        torch.manual_seed(42)
        np.random.seed(42)
        _r.seed(42)
        gd = torch.Generator().manual_seed(7)
        tr2.loader = st.cycle([torch.rand(2, 3, tr.image_size, tr.image_size, generator=gd) for _ in range(8)])
        tr2.save = lambda *a, **k: None
        tr2.evaluate = lambda *a, **k: None
        tr2.init_StylEx()
        rows[graphs] = run_steps(tr2, 8)
        if graphs:
            assert sorted(tr2._graph_cache) == [False, True], tr2._graph_cache.keys()
    torch.backends.cudnn.deterministic = prev_det
    a, b = rows[False][:, :5], rows[True][:, :5]
    print("eager\\n", a, "\\ngraph\\n", b)
    assert np.isfinite(b).all()
    np.testing.assert_allclose(b[:4], a[:4], rtol=1e-4, atol=1e-5)  # eager + warm passes: same arithmetic
    np.testing.assert_allclose(b, a, rtol=5e-3, atol=5e-3)


@pytest.mark.parametrize("shape", [(5, 24, 40, 3), (64, 512, 512, 3), (3, 64, 32, 3), (7, 130, 77, 1)])
def test_modcoeff_kernels_vs_torch_composition(shape):
    """csrc/style_coeffs.hip (`demod_coeff` / `bwd_style` of SURVEY §8(b)): s1 = style + 1 and the demodulation
    coefficient d = rsqrt(s1^2 @ sum_k W^2 + eps) (reference :650-656), forward and first-order backward (gradients
    reaching d AND s1), against the ATen composition in float64."""
    B, C, O, k = shape
    g = torch.Generator().manual_seed(5)
    style = torch.randn(B, C, generator=g) * 0.5
    w = torch.randn(O, C, k, k, generator=g) / (C * k * k) ** 0.5
    r1, r2 = torch.randn(B, C, generator=g), torch.randn(B, O, generator=g)
    sr, wr = style.double().requires_grad_(), w.double().requires_grad_()
    s1r = sr + 1
    dr = torch.rsqrt((s1r * s1r) @ wr.pow(2).sum(dim=(2, 3)).t() + 1e-8)
    ((s1r * r1.double()).sum() + (dr * r2.double()).sum()).backward()
    sd, wd = style.to(DEV).requires_grad_(), torch.nn.Parameter(w.to(DEV))
    prev = ops.set_fast(True)
    try:
        s1, d = ops.mod_coeffs(sd, wd, True, 1e-8)
        assert type(s1.grad_fn).__name__.startswith("_ModCoeffs"), "the fused kernels must serve the fast path"
        ((s1 * r1.to(DEV)).sum() + (d * r2.to(DEV)).sum()).backward()
    finally:
        ops.set_fast(prev)
    close(s1r, s1, 1e-6, "s1")
    close(dr, d, 2e-6, "d")
    close(sr.grad, sd.grad, 2e-5, "grad style")
    close(wr.grad, wd.grad, 2e-5, "grad weight")
    # only s1 used downstream (no demodulation gradient): the direct gradient passes through
    sd2 = style.to(DEV).requires_grad_()
    prev = ops.set_fast(True)
    try:
        s1b, _ = ops.mod_coeffs(sd2, wd, True, 1e-8)
        (s1b * r1.to(DEV)).sum().backward()
    finally:
        ops.set_fast(prev)
    close(r1, sd2.grad, 1e-6, "grad style via s1 only")


def test_style_affines_fused_node_vs_three_linears():
    """ops._StyleAffines (round 6): to_style1 / to_style2 / to_rgb.to_style of a GeneratorBlock as one GEMM each way against the
    three nn.Linear modules (reference stylex_train.py:682-688, :609) — outputs, the gradient to the style vector and all six
    parameter gradients in fp32 (only the summation order of the style-vector gradient changes: 1e-6); the concatenated weight
    is rebuilt after an in-place edit of a bias (AttFind's procedure) and after an optimiser-style stamp."""
    torch.manual_seed(21)
    blk = st.GeneratorBlock(514, 64, 32).to(DEV)
    g = torch.Generator(device=DEV).manual_seed(64)
    w = torch.randn(6, 514, device=DEV, generator=g)
    r = torch.randn(6, 64 + 32 + 32, device=DEV, generator=g)
    lins = (blk.to_style1, blk.to_style2, blk.to_rgb.to_style)

    def fused():
        wa = w.clone().requires_grad_()
        prev = ops.set_fast(True)
        try:
            y = ops.style_affines(wa, *lins, blk.__dict__.setdefault("_aff_cache", {}))
        finally:
            ops.set_fast(prev)
        assert y is not None and len(y) == 4 and torch.equal(y[3], torch.cat(y[:2], dim=1))
        coords, y = y[3], torch.cat(y[:3], dim=1)
        blk.zero_grad()
        ((y * r).sum() + 0.5 * (coords * r[:, :96]).sum()).backward()  # the style coordinates have a consumer as well
        return y.detach(), wa.grad, [p.grad.clone() for lin in lins for p in (lin.weight, lin.bias)]

    def plain():
        wb = w.clone().requires_grad_()
        y = torch.cat([lin(wb) for lin in lins], dim=1)
        blk.zero_grad()
        ((y * r).sum() + 0.5 * (y[:, :96] * r[:, :96]).sum()).backward()
        return y.detach(), wb.grad, [p.grad.clone() for lin in lins for p in (lin.weight, lin.bias)]

    for edit in (None, "bias", "stamp"):
        if edit == "bias":
            with torch.no_grad():
                blk.to_style2.bias[3] += 1.5
        if edit == "stamp":
            with torch.no_grad():
                blk.to_style1.weight.data.mul_(1.01)  # a raw-kernel style update: no version bump ...
            hb.mark_updated([blk.to_style1.weight])  # ... but the stamp the Trainer sets after every optimiser step
        (ya, ga, pa), (yb, gb, pb) = fused(), plain()
        close(yb, ya, 1e-6, "outputs (%s)" % edit)
        close(gb, ga, 1e-6, "style-vector gradient (%s)" % edit)
        for a, b_ in zip(pa, pb):
            close(b_, a, 1e-6, "parameter gradient (%s)" % edit)


def test_mapping_network_fused_node_vs_reference_composition():
    """StyleVectorizer (reference :590-601) on the fused EqualLinear+LeakyReLU node with cached scaled parameters, and
    latent_to_w mapping the two latents of a style-mixing step in one pass: values and all parameter / input gradients
    against the float64 composition F.linear(x, W * lr_mul, b * lr_mul) -> leaky_relu; the cache follows in-place
    parameter updates."""
    import networks
    import stylex_train as st

    torch.manual_seed(11)
    S = networks.StyleVectorizer(66, 3, lr_mul=0.1).to(DEV)
    with torch.no_grad():
        for p in S.parameters():
            p.add_(torch.randn_like(p) * 0.3)
    z1, z2 = torch.randn(5, 66, device=DEV, requires_grad=True), torch.randn(5, 66, device=DEV, requires_grad=True)
    r = torch.randn(10, 66, device=DEV)

    def reference(z):
        x = torch.nn.functional.normalize(z.double(), dim=1)
        for m in S.net:
            if isinstance(m, networks.EqualLinear):
                x = torch.nn.functional.linear(x, m.weight.double() * m.lr_mul, m.bias.double() * m.lr_mul)
            else:
                x = torch.nn.functional.leaky_relu(x, 0.2)
        return x

    for rep in range(2):  # second round: after an in-place parameter update (stale cache = wrong values)
        want = torch.cat((reference(z1), reference(z2)))
        gref = torch.autograd.grad((want * r.double()).sum(), [z1, z2] + list(S.parameters()))
        prev = ops.set_fast(True)
        try:
            (w1, n1), (w2, n2) = st.latent_to_w(S, [(z1, 3), (z2, 4)])
            assert (n1, n2) == (3, 4)
            fn, kinds = w1.grad_fn, set()
            while fn is not None:  # split <- fused layer 3 <- fused layer 2 <- ...
                kinds.add(type(fn).__name__)
                fn = fn.next_functions[0][0] if fn.next_functions else None
            assert "_EqualLinearFastBackward" in kinds, "the fused node must serve the fast path: %s" % kinds
            got = torch.cat((w1, w2))
            ggot = torch.autograd.grad((got * r).sum(), [z1, z2] + list(S.parameters()))
        finally:
            ops.set_fast(prev)
        close(want, got, 2e-6, "mapping network output (round %d)" % rep)
        for a, b, nm in zip(gref, ggot, ["z1", "z2"] + [n for n, _ in S.named_parameters()]):
            close(a, b, 2e-5, "grad " + nm)
        with torch.no_grad():
            for p in S.parameters():
                p.mul_(1.1)
    # outside the fast mode the differentiable composition serves (same values)
    assert ops.fast_enabled() is False
    close(reference(z1), S(z1), 2e-6, "composition path")


@pytest.mark.parametrize("shape", [(3, 5, 7, 7), (2, 8, 12, 10), (4, 3, 9, 13)])
def test_frozen_tail_kernels_vs_aten(shape):
    """csrc/frozen_ew.hip against the ATen ops they replace in the frozen classifier: eval BatchNorm (as an affine map)
    -> (+ residual) -> ReLU, and BatchNorm -> ReLU -> MaxPool2d(3, 2, 1); values, input / residual gradients, odd sizes
    (scalar path: H*W % 4 != 0) and windows clipped by the border.  Ties inside a pooling window (incl. all-negative
    windows) follow ATen's first-maximum rule."""
    from frozen_resnet import _AffineAct, _AffineReluPool

    b, c, h, w = shape
    g = torch.Generator().manual_seed(h * 31 + w)
    x = torch.randn(b, c, h, w, generator=g)
    x[:, :, : h // 2] = torch.round(x[:, :, : h // 2] * 2) / 2  # many exact ties and many all-negative windows
    res = torch.randn(b, c, h, w, generator=g)
    sc, sh = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.3
    sc[0] = -sc[0]  # a negative BatchNorm scale flips the order inside the pooling window
    xr, rr = x.clone().requires_grad_(), res.clone().requires_grad_()
    aff = xr * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    y_ref = torch.relu(aff + rr)
    p_ref = torch.nn.functional.max_pool2d(torch.relu(aff), 3, 2, 1)
    gy, gp = torch.randn(y_ref.shape, generator=g), torch.randn(p_ref.shape, generator=g)
    gxa, gra = torch.autograd.grad((y_ref * gy).sum(), [xr, rr], retain_graph=True)
    gxp, = torch.autograd.grad((p_ref * gp).sum(), [xr])
    xd, rd = x.to(DEV).requires_grad_(), res.to(DEV).requires_grad_()
    y = _AffineAct.apply(xd, sc.to(DEV), sh.to(DEV), rd, True)
    gxd, grd = torch.autograd.grad((y * gy.to(DEV)).sum(), [xd, rd])
    close(y_ref, y, 1e-6, "affine + residual + relu")
    close(gxa, gxd, 1e-6, "its input gradient")
    close(gra, grd, 1e-6, "its residual gradient")
    y2 = _AffineAct.apply(xd, sc.to(DEV), sh.to(DEV), None, False)  # downsample branch: no residual, no relu
    close(aff, y2, 1e-6, "affine only")
    close(gy * sc.view(1, -1, 1, 1), torch.autograd.grad((y2 * gy.to(DEV)).sum(), [xd])[0], 1e-6, "affine-only gradient")
    p = _AffineReluPool.apply(xd, sc.to(DEV), sh.to(DEV))
    close(p_ref, p, 1e-6, "affine + relu + maxpool")
    close(gxp, torch.autograd.grad((p * gp.to(DEV)).sum(), [xd])[0], 1e-6, "its input gradient")
    with torch.no_grad():  # no index plane without a gradient consumer
        close(p_ref, _AffineReluPool.apply(x.to(DEV), sc.to(DEV), sh.to(DEV)), 1e-6, "affine + relu + maxpool (no grad)")


@pytest.mark.parametrize("case", [(2, 64, 3, 7, 2, 3, 32, 32), (3, 64, 3, 11, 4, 2, 64, 64), (2, 16, 3, 11, 4, 2, 37, 45),
                                  (1, 8, 4, 3, 1, 1, 9, 70), (2, 24, 3, 7, 2, 3, 33, 130), (1, 64, 3, 5, 4, 0, 21, 17),
                                  (32, 64, 3, 7, 2, 3, 224, 224), (32, 64, 3, 11, 4, 2, 256, 256)])
def test_image_gradient_kernel_of_the_frozen_stems(case, monkeypatch):
    """stylex_conv_image_grad (round 6): the input gradient of the K x K / stride-S stem convolutions of the frozen classifier
    (ResNet conv1) and LPIPS-AlexNet against torch.nn.grad.conv2d_input in float64 — every residue class of the stride,
    odd image sizes (ragged last class rows / columns), 3 and 4 image channels, pad 0, and the two shapes the training step
    runs (B = 32).  fp32 FMA chain against an fp64 sum of N * (K / S)^2 <= 1024 terms: 2e-5 of the largest element; a
    second call is bit-identical (fixed order)."""
    B, N, C, K, S, pad, H, W = case
    monkeypatch.setenv("STYLEX_IMAGE_GRAD", "2")  # every stride through the autograd hook (default: stride 4 only, where it wins)
    g = torch.Generator(device=DEV).manual_seed(81)
    Ho, Wo = (H + 2 * pad - K) // S + 1, (W + 2 * pad - K) // S + 1
    gy = torch.randn(B, N, Ho, Wo, device=DEV, generator=g)
    w = torch.randn(N, C, K, K, device=DEV, generator=g) / (K * K * C) ** 0.5
    got = hb.conv_image_grad(gy, w, (H, W), S, pad)
    want = torch.nn.grad.conv2d_input((B, C, H, W), w.double(), gy.double(), stride=S, padding=pad)
    close(want, got, 2e-5, "image gradient")
    assert torch.equal(got, hb.conv_image_grad(gy, w, (H, W), S, pad))
    # through autograd: the module-level hook (frozen_resnet.first_conv) against plain F.conv2d
    from frozen_resnet import first_conv

    if B <= 3:
        bias = torch.randn(N, device=DEV, generator=g)
        x = torch.randn(B, C, H, W, device=DEV, generator=g)
        xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
        ya, yb = first_conv(xa, w, bias, S, pad), F.conv2d(xb, w, bias, S, pad)
        assert ya.grad_fn is not None and "FirstConv" in type(ya.grad_fn).__name__
        assert torch.equal(ya, yb)
        (ya * gy).sum().backward()
        (yb * gy).sum().backward()
        close(xb.grad, xa.grad, 2e-5, "stem input gradient through autograd")


def test_frozen_classifier_fused_tails_match_plain_module():
    """ResNet.classify_images (reference resnet_classifier.py:57-71) with the fused elementwise tails (default on the
    GPU) against the plain nn.Module (STYLEX_FROZEN_FUSE=0): logits and the gradient reaching the images, fp32, the
    same library convolutions on both sides (tolerance = the rounding of bn(x) as x * s + t)."""
    import resnet_classifier as rc

    clf = rc.ResNet(None, 0, output_size=2, image_size=64)
    with torch.no_grad():  # non-trivial BatchNorm statistics
        for m in clf.model.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.2)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.2)
    torch.manual_seed(3)
    img = torch.rand(4, 3, 64, 64, device=DEV)
    coef = torch.tensor([[1.0, -2.0]])
    outs = {}
    for mode in ("0", "1"):
        os.environ["STYLEX_FROZEN_FUSE"] = mode
        try:
            x = img.clone().requires_grad_()
            logits = clf.classify_images(x)
            gx, = torch.autograd.grad((logits * coef.to(DEV)).sum(), [x])
            outs[mode] = (logits.detach().double().cpu(), gx.double().cpu())
        finally:
            os.environ.pop("STYLEX_FROZEN_FUSE", None)
    assert type(clf._net(img)).__name__ == "FusedTailResNet", "the fused tails must be the default on the GPU"
    # ground truth: the same module in float64 on the CPU.  An fp32 rounding difference can flip the gate of a ReLU
    # whose input is ~0, which moves the image gradient far more than the rounding itself — so the fused path is held
    # to the error the PLAIN fp32 module shows against float64, not to a rounding-sized bound against the plain module
    import copy

    m64 = copy.deepcopy(clf.model).double().cpu()
    x64 = img.double().cpu().requires_grad_()
    xin = F.interpolate(x64, size=[224, 224], mode="bilinear", align_corners=False)
    xin = (xin - clf._mean.double().cpu()) / clf._std.double().cpu()
    l64 = m64(xin)
    g64, = torch.autograd.grad((l64 * coef.double()).sum(), [x64])

    def rel(a, ref):
        return ((a - ref).norm() / ref.norm()).item()

    e_plain = (rel(outs["0"][0], l64.detach()), rel(outs["0"][1], g64))
    e_fused = (rel(outs["1"][0], l64.detach()), rel(outs["1"][1], g64))
    assert e_fused[0] <= max(3 * e_plain[0], 2e-6), ("logits", e_fused, e_plain)
    assert e_fused[1] <= max(3 * e_plain[1], 2e-5), ("image gradient", e_fused, e_plain)
    close(outs["0"][0], outs["1"][0].float(), 2e-5, "logits")
    with torch.no_grad():  # BatchNorm tensors changed in place after the fused module was built: it must follow
        before = clf.classify_images(img)
        clf.model.bn1.bias.add_(0.5)
        after_fused = clf.classify_images(img)
        os.environ["STYLEX_FROZEN_FUSE"] = "0"
        try:
            after_plain = clf.classify_images(img)
        finally:
            os.environ.pop("STYLEX_FROZEN_FUSE", None)
        clf.model.bn1.bias.sub_(0.5)
    assert (before - after_fused).abs().max() > 1e-4
    close(after_plain, after_fused, 2e-5, "logits after an in-place BatchNorm update")


@pytest.mark.parametrize("size", [224, 96])
def test_frozen_classifier_bf16_data_gradient(size, monkeypatch):
    """Round 6: the frozen ResNet-18 in the bf16 speed mode, a pass whose input gradient is wanted (the classifier on generated
    images, reference stylex_train.py:1390, 421-438): the forward stays on the library's fp32 convolutions — the logits of
    the fp32 path — and the data gradient runs on this library's bf16 kernels gated by the signs of the fp32 activations
    (frozen_resnet._ResNetBodyHybrid).  Against the all-fp32 backward (STYLEX_FROZEN_BWD_BF16=0) on the same input: relative L2
    error of the image gradient below 3e-2, cosine above 0.999 (17 chained bf16 data gradients, no gate flips)."""
    from frozen_resnet import FusedTailResNet
    from tv_models import ResNet18

    torch.manual_seed(5)
    net = ResNet18()
    net.fc = torch.nn.Linear(512, 2)
    with torch.no_grad():
        for mod in net.modules():  # non-trivial BatchNorm statistics
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.running_mean.normal_(0, 0.2)
                mod.running_var.uniform_(0.5, 1.5)
                mod.weight.uniform_(0.5, 1.5)
                mod.bias.normal_(0, 0.2)
    net = net.to(DEV).eval()
    for p in net.parameters():
        p.requires_grad = False
    fused = FusedTailResNet(net)
    g = torch.Generator(device=DEV).manual_seed(59)
    x0 = torch.randn(4, 3, size, size, device=DEV, generator=g)
    r = torch.randn(4, 2, device=DEV, generator=g)
    def both():
        res = {}
        ops.set_precision("bf16")
        try:
            for mode in ("0", "1"):
                monkeypatch.setenv("STYLEX_FROZEN_BWD_BF16", mode)
                x = x0.clone().requires_grad_()
                y = fused(x)
                (y * r).sum().backward()
                res[mode] = (y.detach().clone(), x.grad.detach().double())
        finally:
            ops.set_precision("fp32")
        return res

    res = both()
    # the same library convolutions in both modes (the library's own run-to-run noise is ~1e-7: atomics in its algorithms)
    close(res["0"][0], res["1"][0], 1e-5, "the forward must not change")
    g32, g16 = res["0"][1], res["1"][1]
    rel = float((g16 - g32).norm() / g32.norm())
    cos = float((g16 * g32).sum() / (g16.norm() * g32.norm()))
    print("frozen classifier @%d: bf16 data gradient vs fp32: rel L2 %.3e, cosine %.6f" % (size, rel, cos))
    assert rel < 3e-2 and cos > 0.999, (rel, cos)
    # a checkpoint loaded into the classifier LATER must reach the folded bf16 weights of the backward as well
    with torch.no_grad():
        net.layer2[0].conv1.weight.mul_(1.7)
        net.layer3[1].conv2.weight.neg_()
    res = both()
    g32, g16 = res["0"][1], res["1"][1]
    rel2 = float((g16 - g32).norm() / g32.norm())
    assert rel2 < 3e-2, ("stale folded weights after an in-place change of the classifier's parameters", rel2)


@pytest.mark.parametrize("case", [(4, 3, 256, 256, 224, 224, "nchw"), (4, 3, 256, 256, 224, 224, "nhwc"), (2, 3, 32, 32, 224, 224, "nchw"),
                                  (2, 3, 64, 64, 224, 224, "nhwc"), (2, 4, 40, 56, 33, 97, "nchw"), (1, 3, 224, 224, 224, 224, "nchw"),
                                  (3, 1, 7, 5, 3, 11, "slice")])
def test_resize_normalize_kernels_vs_aten(case):
    """stylex_resize_norm_fwd / _bwd (round 6: the classifier's input path, reference resnet_classifier.py:56-71) against
    F.interpolate(mode='bilinear', align_corners=False) + (x - mean) / std and their autograd backward: down- and up-scaling,
    non-square, identity, dense / channels_last / sliced inputs (read through strides).  Same index rule and the same fp32
    products as ATen's kernel: 1e-6 of the largest element forward, 1e-5 backward (ATen's backward scatters with atomics in
    another order); a second backward call is bit-identical."""
    B, C, H, W, Ho, Wo, layout = case
    g = torch.Generator(device=DEV).manual_seed(61)
    x = torch.randn(B, C + (2 if layout == "slice" else 0), H, W, device=DEV, generator=g)
    if layout == "nhwc":
        x = x.contiguous(memory_format=torch.channels_last)
    if layout == "slice":
        x = x[:, 1:1 + C]
    mean = torch.randn(C, device=DEV, generator=g)
    std = torch.rand(C, device=DEV, generator=g) + 0.5
    gy = torch.randn(B, C, Ho, Wo, device=DEV, generator=g)
    xr = x.detach().clone().requires_grad_()
    want = (F.interpolate(xr, size=[Ho, Wo], mode="bilinear", align_corners=False) - mean.view(1, -1, 1, 1)) / std.view(1, -1, 1, 1)
    (want * gy).sum().backward()
    got = hb.resize_norm_fwd(x, (Ho, Wo), mean, std)
    close(want, got, 1e-6, "resize + normalise")
    gx = hb.resize_norm_bwd(gy, (H, W), std)
    close(xr.grad, gx, 1e-5, "adjoint")
    assert torch.equal(gx, hb.resize_norm_bwd(gy, (H, W), std))
    close(F.interpolate(x, size=[Ho, Wo], mode="bilinear", align_corners=False), hb.resize_norm_fwd(x, (Ho, Wo)), 1e-6, "resize only")


def test_resnet_wrapper_vs_reference_golden_on_hip(tmp_path):
    """A16 on the GPU: ResNet.classify_images (the classifier of the headline benchmark configuration) against what
    the REFERENCE's own wrapper class produced (tests/golden/resnet_wrapper.npz from stylex/resnet_classifier.py:29-71
    on seeded weights): logits and the gradient reaching the images, with the fused elementwise tails (the GPU
    default) and with the plain module.  Tolerance: the library's fp32 convolutions (Winograd on MIOpen) against the
    CPU's direct ones through 20 layers — 1e-4 of the logits in the L2 sense (measured 2.5e-7); the image gradient can
    additionally see a ReLU whose input is ~0 flip its gate under that rounding, and ONE flipped gate moves the gradient
    of this 3-image batch by half a percent (measured 5.8e-3 on one box, 1e-3 on others, logits unchanged) — hence
    2e-2, still two orders below what a wrong resize mode or normalisation produces (O(1))."""
    from test_host_logic_cpu import make_resnet_wrapper
    from test_oracle_vs_golden import resnet_wrapper_cases

    def rel(a, ref):
        ref = torch.as_tensor(np.asarray(ref)).double()
        return ((a.detach().double().cpu() - ref).norm() / ref.norm()).item()

    g = load_golden("resnet_wrapper")
    for tag, size, norm, x, coef, logits, gx in resnet_wrapper_cases(g):
        clf = make_resnet_wrapper(g, tmp_path, size, norm)
        assert next(clf.model.parameters()).is_cuda
        for fuse in ("1", "0"):
            os.environ["STYLEX_FROZEN_FUSE"] = fuse
            try:
                xd = x.to(DEV).requires_grad_(True)
                out = clf.classify_images(xd)
                got, = torch.autograd.grad((out * coef.to(DEV)).sum(), xd)
            finally:
                os.environ.pop("STYLEX_FROZEN_FUSE", None)
            e = (rel(out, logits), rel(got, gx))
            assert e[0] <= 1e-4 and e[1] <= 2e-2, (tag, "fuse=" + fuse, e)
    assert type(clf._net(xd)).__name__ == "FusedTailResNet"


def test_lpips_distance_kernels_vs_published_formula():
    """LPIPS-AlexNet forward on the fused tap kernels (default on the GPU) against the published composition
    (STYLEX_LPIPS_FUSE=0: unit-normalise, squared difference, non-negative 1x1 weights, spatial mean, sum over taps):
    distances and the gradient reaching BOTH images, at an image size whose taps are not multiples of the block."""
    from lpips_alex import LPIPS

    torch.manual_seed(5)
    net = LPIPS().to(DEV)
    a, b = torch.rand(3, 3, 96, 80, device=DEV) * 2 - 1, torch.rand(3, 3, 96, 80, device=DEV) * 2 - 1
    outs = {}
    for mode in ("0", "1"):
        os.environ["STYLEX_LPIPS_FUSE"] = mode
        try:
            x0, x1 = a.clone().requires_grad_(), b.clone().requires_grad_()
            d = net(x0, x1)
            assert d.shape == (3, 1, 1, 1)
            g0, g1 = torch.autograd.grad((d.flatten() * torch.tensor([1.0, -0.5, 2.0], device=DEV)).sum(), [x0, x1])
            x1b = b.clone().requires_grad_()  # the train step's case: only the generated image carries a gradient
            g1b, = torch.autograd.grad(net(a, x1b).sum(), [x1b])
            outs[mode] = (d.detach(), g0, g1, g1b)
            if mode == "1":
                assert "_LpipsDistance" in type(d.grad_fn.next_functions[0][0]).__name__ + type(d.grad_fn).__name__
        finally:
            os.environ.pop("STYLEX_LPIPS_FUSE", None)
    close(outs["0"][0], outs["1"][0], 5e-6, "LPIPS distance")
    close(outs["0"][1], outs["1"][1], 2e-5, "gradient to image 0")
    close(outs["0"][2], outs["1"][2], 2e-5, "gradient to image 1")
    close(outs["0"][3], outs["1"][3], 2e-5, "one-sided gradient")


@pytest.mark.parametrize("shape", [(3, 64, 63, 63), (2, 192, 31, 31), (4, 384, 15, 15), (2, 256, 15, 15), (1, 8, 1, 1), (2, 72, 5, 7)])
def test_lpips_nhwc_tap_kernels_vs_published_formula(shape):
    """stylex_lpips_tap_nhwc_fwd / _bwd (round 6: the LPIPS tap on bf16 channels_last features) against the published formula
    (lpips 0.1.4: normalize_tensor, squared difference, lin, spatial average) evaluated in float64 on the same bf16 values —
    the five AlexNet tap shapes, a single pixel, a ragged last block and a ragged channel octet walk (72 = 9 slots).  Values
    1e-5; gradients carry the bf16 rounding of their storage (2^-8 of the element, bounded here by 1e-2 of the tensor's
    largest element); a second call is bit-identical."""
    B, C, H, W = shape
    g = torch.Generator(device=DEV).manual_seed(57)
    f0 = torch.randn(B, C, H, W, device=DEV, generator=g).relu()
    f1 = (f0 + 0.3 * torch.randn(B, C, H, W, device=DEV, generator=g)).relu()
    f0[:, 0] += 0.5  # no all-zero pixel (0 / 0 in the reference's backward as well)
    f1[:, 0] += 0.5
    f0, f1 = cl(f0.bfloat16()), cl(f1.bfloat16())
    lin = torch.rand(C, device=DEV, generator=g) / C
    gout = torch.randn(B, device=DEV, generator=g)
    a, b_ = f0.double().requires_grad_(), f1.double().requires_grad_()
    n0 = a / (a.pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
    n1 = b_ / (b_.pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
    want = (((n0 - n1) ** 2) * lin.double().view(1, -1, 1, 1)).sum(1).mean(dim=(1, 2))
    (want * gout.double()).sum().backward()
    got, norms = hb.lpips_taps_nhwc_fwd([f0], [f1], [lin], keep_norms=True)
    close(want, got, 1e-5, "distance")
    close(f0.double().pow(2).sum(1).sqrt().reshape(B, -1), norms[0][0], 1e-6, "norms")
    g0, g1 = hb.lpips_tap_nhwc_bwd(f0, f1, lin, norms[0][0], norms[0][1], gout, True, True)
    assert g0.dtype == torch.bfloat16 and hb.is_cl(g0) and hb.is_cl(g1)
    close(a.grad, g0, 1e-2, "gradient to f0")
    close(b_.grad, g1, 1e-2, "gradient to f1")
    again, _ = hb.lpips_taps_nhwc_fwd([f0], [f1], [lin], keep_norms=False)
    assert torch.equal(got, again)
    only1 = hb.lpips_tap_nhwc_bwd(f0, f1, lin, norms[0][0], norms[0][1], gout, False, True)
    assert only1[0] is None and torch.equal(only1[1], g1)


@pytest.mark.parametrize("shape", [(2, 64, 63, 63), (3, 192, 15, 15), (1, 8, 1, 1), (2, 72, 7, 7), (4, 512, 7, 7), (2, 64, 56, 56)])
def test_layout_bridge_kernels(shape):
    """stylex_nchw_f32_to_nhwc_bf16 / stylex_nhwc_bf16_to_nchw_f32 (round 6: the hand-over between the library's fp32 NCHW
    tensors and this library's bf16 channels_last ones) against the ATen conversions they replace — exact (one rounding to
    bf16, RNE, as Tensor.to does), with and without the fused ReLU / gate, ragged pixel and channel tiles."""
    B, C, H, W = shape
    g = torch.Generator(device=DEV).manual_seed(60)
    x = torch.randn(B, C, H, W, device=DEV, generator=g)
    for relu in (False, True):
        y = hb.nchw_to_cl_bf16(x, relu=relu)
        want = (x.relu() if relu else x).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        assert y.dtype == torch.bfloat16 and hb.is_cl(y) and torch.equal(y, want), relu
    gy = cl(torch.randn(B, C, H, W, device=DEV, generator=g).bfloat16())
    gate = hb.nchw_to_cl_bf16(x, relu=True)
    assert torch.equal(hb.cl_bf16_to_nchw(gy), gy.float().contiguous())
    got = hb.cl_bf16_to_nchw(gy, gate=gate)
    assert got.is_contiguous() and torch.equal(got, (gy.float() * (x > 0)).contiguous())
    # relu_gate_add: (y > 0) ? a + b : 0 with ONE rounding of the fp32 sum
    gb = cl(torch.randn(B, C, H, W, device=DEV, generator=g).bfloat16())
    want = ((gy.float() + gb.float()) * (gate.float() > 0)).to(torch.bfloat16)
    assert torch.equal(hb.relu_gate_add(gy, gb, gate), want)
    assert torch.equal(hb.relu_gate_add(gy, None, gate), (gy.float() * (gate.float() > 0)).to(torch.bfloat16))


@pytest.mark.parametrize("shape", [(2, 64, 63, 63), (3, 192, 31, 31), (1, 8, 3, 3), (2, 16, 8, 11), (1, 24, 4, 7)])
def test_maxpool_3x3_stride_2_on_bf16_nhwc_vs_aten(shape):
    """stylex_maxpool3s2_nhwc_fwd / _bwd (round 6: LPIPS-AlexNet's two pools on the bf16 channels_last taps) against
    F.max_pool2d(x, 3, 2) and its autograd: the maxima exactly, the gradient exactly — including WHICH element of a window
    with equal values receives it (post-ReLU taps are full of equal zeros: ATen takes the first maximum in scan order) —
    on the two LPIPS shapes, the smallest window, odd / even sizes whose last rows and columns no window covers, and with
    NaNs (a NaN is the maximum of its window)."""
    import lpips_alex

    B, C, H, W = shape
    g = torch.Generator(device=DEV).manual_seed(64)
    x = torch.randn(B, C, H, W, device=DEV, generator=g).relu()  # many ties at zero
    x = (x * 4).round() / 4  # and ties among the positive values
    if H * W > 20:
        x[0, 0, 1, 1] = float("nan")
        x[-1, -1, H - 2, W - 2] = float("nan")
    xb = cl(x.to(torch.bfloat16))
    x1 = xb.clone().requires_grad_()
    x2 = xb.clone().requires_grad_()
    y1 = lpips_alex._MaxPool3s2CL.apply(x1)
    y2 = F.max_pool2d(x2, 3, 2)
    assert y1.shape == y2.shape and y1.dtype == torch.bfloat16 and hb.is_cl(y1)
    assert torch.equal(torch.nan_to_num(y1.float(), nan=-7.0), torch.nan_to_num(y2.float(), nan=-7.0))
    gy = cl(torch.randn(y2.shape, device=DEV, generator=g).to(torch.bfloat16))
    y1.backward(gy)
    y2.backward(gy)
    assert x1.grad.dtype == torch.bfloat16 and hb.is_cl(x1.grad)
    assert torch.equal(x1.grad, x2.grad), float((x1.grad.float() - x2.grad.float()).abs().max())


def test_timing_pause_keeps_frozen_network_launches_out_of_the_classes():
    """hb.timing_pause(): conv launches of the calling thread inside the block are not recorded by the timing hook, and an
    autograd node built inside carries the pause into its backward (which runs on the engine's thread) — so that the frozen
    networks' bf16 layers do not dilute the StylEx conv classes of bench.py's roofline record."""
    ops.set_precision("bf16")
    g = torch.Generator(device=DEV).manual_seed(62)
    x = cl(torch.randn(2, 64, 32, 32, device=DEV, generator=g).bfloat16()).requires_grad_()
    w = torch.randn(64, 64, 3, 3, device=DEV, generator=g) / 24
    hb.timing_enable(1)
    try:
        prev = ops.set_fast(True)
        try:
            with hb.timing_pause():
                assert hb.timing_paused()
                y = ops.conv2d(x, w, None, stride=1, padding=1, lrelu="relu")
            assert not hb.timing_paused()
            y.float().sum().backward()
            torch.cuda.synchronize()
            rep = hb.timing_report()
            assert rep["fwd"]["launches"] == 0 and rep["bwd_data"]["launches"] == 0, rep
            y2 = ops.conv2d(x, w, None, stride=1, padding=1, lrelu="relu")
            y2.float().sum().backward()
            torch.cuda.synchronize()
            rep = hb.timing_report()
            assert rep["fwd"]["launches"] == 1 and rep["bwd_data"]["launches"] == 1, rep
        finally:
            ops.set_fast(prev)
    finally:
        hb.timing_enable(0)
        ops.set_precision("fp32")


@pytest.mark.parametrize("case", [(4, 64, 11, 4, 2, 256, 256), (2, 64, 11, 4, 2, 97, 130), (3, 24, 11, 4, 2, 64, 64)])
def test_bf16_stem_on_the_generic_kernel(case):
    """frozen_resnet._StemBf16 (round 6: LPIPS-AlexNet's 11x11 / stride-4 layer in the bf16 mode on the generic implicit-GEMM
    kernel, image padded to one 8-channel slot, bias + ReLU epilogue) against relu(conv2d) in float64 on the same bf16-rounded
    image and weights (1e-2: fp32 accumulation of 363 products + one bf16 rounding of the output), and its input gradient
    (bridge kernel + stylex_conv_image_grad) against autograd through the fp64 definition gated by the kernel's own output."""
    from frozen_resnet import _StemBf16

    B, N, K, S, pad, H, W = case
    ops.set_precision("bf16")
    g = torch.Generator(device=DEV).manual_seed(63)
    x = torch.randn(B, 3, H, W, device=DEV, generator=g)
    w = torch.randn(N, 3, K, K, device=DEV, generator=g) / (3 * K * K) ** 0.5
    b = torch.randn(N, device=DEV, generator=g) * 0.1
    xa = x.clone().requires_grad_()
    y = _StemBf16.apply(xa, w, b, S, pad)
    assert y.dtype == torch.bfloat16 and hb.is_cl(y)
    xr = x.to(torch.bfloat16).double().requires_grad_()
    yr = F.relu(F.conv2d(xr, w.to(torch.bfloat16).double(), b.double(), S, pad))
    close(yr, y, 1e-2, "stem forward")
    gy = cl(torch.randn(*y.shape, device=DEV, generator=g).bfloat16())
    y.backward(gy)
    # the reference gradient with the gate of the KERNEL's output (a pre-activation within rounding of zero may differ in sign)
    z = F.conv2d(xr, w.double(), b.double(), S, pad)  # fp32 weights in the data gradient, as stylex_conv_image_grad uses them
    (z * (gy.double() * (y.double() > 0))).sum().backward()
    close(xr.grad, xa.grad, 2e-5, "stem input gradient")


def test_lpips_bf16_path_tracks_the_fp32_library_path():
    """LPIPS-AlexNet in the bf16 speed mode (round 6, stylex/lpips_alex.py::_taps_bf16: stem on the library + image-gradient
    kernel, the 5x5 and the three 3x3 layers on this library's bf16 conv kernels, taps as bf16 channels_last) against the
    fp32 library path of the same module on the same images: the distance and the gradient to the generated image — the two
    things reconstruction_loss takes from it (reference stylex_train.py:409-418).  Five ReLU layers of bf16 operands:
    the distance within 3e-2; the image gradient within 0.15 in the relative L2 sense with cosine > 0.99 (ReLU gates that
    flip under the bf16 rounding of a pre-activation near zero change single taps; the direction is what trains G)."""
    import stylex_train as st_mod
    from lpips_alex import LPIPS

    torch.manual_seed(3)
    net = LPIPS(net="alex").to(DEV)
    g = torch.Generator(device=DEV).manual_seed(58)
    real = torch.rand(8, 3, 256, 256, device=DEV, generator=g)
    fake = (real + 0.25 * torch.randn(8, 3, 256, 256, device=DEV, generator=g)).clamp(0, 1)
    res = {}
    for prec in ("fp32", "bf16"):
        ops.set_precision(prec)
        x = fake.clone().requires_grad_()
        d = net(st_mod.lpips_normalize(real), st_mod.lpips_normalize(x)).reshape(-1)
        d.sum().backward()
        res[prec] = (d.detach().double(), x.grad.detach().double())
    ops.set_precision("fp32")
    (d32, g32), (d16, g16) = res["fp32"], res["bf16"]
    rel_d = float(((d16 - d32).abs() / d32.abs()).max())
    rel_g = float((g16 - g32).norm() / g32.norm())
    cos = float((g16 * g32).sum() / (g16.norm() * g32.norm()))
    print("LPIPS bf16 vs fp32: distance rel %.3e, gradient rel L2 %.3e, cosine %.5f" % (rel_d, rel_g, cos))
    assert rel_d < 3e-2 and rel_g < 0.15 and cos > 0.99, (rel_d, rel_g, cos)


def test_operand_cache_follows_parameter_versions_and_prepack():
    """The packed operand copies (hip_backend.pack_weight & co.) are cached per Parameter and valid for one version:
    an in-place update (the optimiser step) invalidates them, `prepack` rebuilds them on a side stream, and a consumer on
    another stream gets the NEW values (ordering through the entry's event) — never the stale pack."""
    torch.manual_seed(2)
    w = torch.nn.Parameter(torch.randn(64, 32, 3, 3, device=DEV))
    x = cl(torch.randn(2, 32, 16, 16)).to(torch.bfloat16)

    def conv():
        return hb.conv2d_fwd(x, w, 1, 1, hb.BF16_ACT).float()

    def ref():
        return F.conv2d(x.float(), w.detach().to(torch.bfloat16).float(), padding=1)

    close(ref(), conv(), 2e-2, "first pack")
    n0 = len(hb._PACK_CACHE)
    close(ref(), conv(), 2e-2, "cache hit")
    assert len(hb._PACK_CACHE) == n0
    with torch.no_grad():
        w.mul_(-0.5)  # version bump without prepack: the stale entry must not be served
    close(ref(), conv(), 2e-2, "after an in-place update")
    assert len(hb._PACK_CACHE) == n0, "one entry per (parameter, variant): replaced, not accumulated"
    with torch.no_grad():
        w.add_(1.0)
    hb.prepack([w])
    key = next(k for k, v in hb._PACK_CACHE.items() if v[0]() is w)
    assert hb._PACK_CACHE[key][5] == hb._gen(w) and hb._PACK_CACHE[key][4] != torch.cuda.current_stream().cuda_stream, \
        "prepack must have rebuilt the operand on its side stream"
    close(ref(), conv(), 2e-2, "after prepack")
    hb.prepack_join()
    # the fused Adam updates parameters WITHOUT bumping Parameter._version: only the Trainer's stamp invalidates
    opt = torch.optim.Adam([w], lr=0.5, fused=True)
    w.grad = torch.ones_like(w)
    before = hb._gen(w)
    opt.step()
    if w._version == before[0]:  # (if a later torch bumps the counter the stamp is merely redundant)
        hb.mark_updated([w])
    assert hb._gen(w) != before
    close(ref(), conv(), 2e-2, "after a fused-Adam step + mark_updated")


def test_generator_phase_sees_the_updated_discriminator(tmp_path):
    """Reference :1356 / :1380-1440: D's optimiser step precedes the generator phase, whose loss is evaluated on the
    UPDATED discriminator.  In the bf16 speed mode the optimiser is the fused Adam, which does not bump
    Parameter._version — a cache of packed weights keyed on that counter alone served the generator phase D's
    pre-update weights (found in round 3).  Two identically seeded Trainers, one with the operand cache switched off:
    same losses, in particular the same first g_loss."""
    g = load_golden("steps_gae2_alt")
    rows = {}
    for cache_on in (True, False):
        prev = hb._CACHE_ON
        hb._CACHE_ON = cache_on
        hb.pack_cache_clear()
        try:
            ops.set_precision("bf16")
            (tmp_path / ("c%d" % cache_on)).mkdir()
            tr, n = make_trainer(g, tmp_path / ("c%d" % cache_on), device=torch.device(DEV))
            rows[cache_on] = run_steps(tr, 3)
        finally:
            hb._CACHE_ON = prev
            ops.set_precision("fp32")
    np.testing.assert_allclose(rows[True], rows[False], rtol=1e-6, atol=1e-6, equal_nan=True)


def test_pad_rgb8_kernel_all_input_layouts():
    """stylex_pad_rgb8: a 3-channel image in any of the layouts the Trainer hands to D / the encoder (fp32 NCHW from the
    loader, fp32 channels_last, the bf16 3-of-4-channel view the generator returns) -> bf16 NHWC with 8 channels, the last
    five zero; values are the RNE bf16 rounding of the input (bit-exact)."""
    g = torch.Generator().manual_seed(3)
    x32 = torch.randn(3, 3, 20, 28, generator=g).to(DEV)
    four = torch.randn(3, 4, 20, 28, generator=g).to(DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    for name, x in (("fp32 nchw", x32), ("fp32 nhwc", x32.contiguous(memory_format=torch.channels_last)),
                    ("bf16 view of 4-channel storage", four[:, :3]), ("fp32 transposed view", x32.transpose(2, 3).transpose(2, 3))):
        y = hb.pad_rgb8(x)
        assert y.shape == (3, 8, 20, 28) and y.dtype == torch.bfloat16 and hb.is_cl(y), name
        assert torch.equal(y[:, :3], x.to(torch.bfloat16)), name
        assert float(y[:, 3:].abs().max()) == 0.0, name


PIPE_CASES = [
    # B, C, N, H, W
    (6, 64, 64, 64, 64),      # 32x32 px x 64 n tiles (N = 64), 4 chunks per tile, 24 tiles
    (3, 64, 64, 40, 72),      # ragged rows and columns
    (2, 128, 128, 64, 64),    # 16x32 px x 128 n tiles
    (70, 128, 256, 32, 32),   # 280 tiles > 256 CUs: several tiles per block, two channel tiles per pixel tile
    (1, 256, 384, 24, 40),    # ragged, three channel tiles
    (2, 512, 512, 32, 32),    # 32 chunks per tile
    (1, 128, 64, 16, 40),     # N = 64 with fewer than 32 rows: stays on the per-tile kernel (both arms identical)
]


@pytest.mark.against_definition
@pytest.mark.parametrize("case", PIPE_CASES)
def test_pipelined_conv_kernel_matches_per_tile_kernel(case):
    """conv_pipe.hip (persistent, software-pipelined LDS-DMA kernel; round 3) against the per-tile LDS-DMA kernel it
    replaces (STYLEX_CONV_PIPE=0) on the same inputs: forward with bias + LeakyReLU (+ activation bit mask), plain data
    gradient, data gradient gated by an activation tensor and by a bit mask.  Both kernels accumulate the 16-channel
    chunks in the same order with the same MFMA, so the results are bit-identical; the forward is also checked against
    the fp32 definition."""
    import os

    B, C, N, H, W = case
    ops.set_precision("bf16")
    P = hb.BF16_ACT
    g = torch.Generator(device=DEV).manual_seed(33)
    mk = lambda *sh: torch.randn(*sh, device=DEV, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)  # noqa: E731
    x, dy, gate = mk(B, C, H, W), mk(B, N, H, W), mk(B, C, H, W)
    w = torch.randn(N, C, 3, 3, device=DEV, generator=g) / (9 * C) ** 0.5
    bias = torch.randn(N, device=DEV, generator=g)
    # the gate as the bit mask a forward conv would have written: bit k of byte i = element 8 i + k of the NHWC tensor > 0
    bits = (gate.permute(0, 2, 3, 1).float() > 0).reshape(B, H, W, C // 8, 8).to(torch.int32)
    gmask = (bits * (2 ** torch.arange(8, device=DEV, dtype=torch.int32))).sum(-1).to(torch.uint8).contiguous()
    sh_d = hb.conv_shape((B, C, H, W), tuple(w.shape), 1, 1)

    def run():
        y, m = hb.conv2d_fwd(x, w, 1, 1, P, bias=bias, lrelu=True, want_mask=True)
        outs = [y, hb.conv2d_bwd_data(dy, w, (B, C, H, W), 1, 1, P), hb.conv2d_bwd_data(dy, w, (B, C, H, W), 1, 1, P, gate=gate)]
        if m is not None:
            outs.append(m)
        if hb.conv_mask_supported(sh_d, 1, hb.EPI_GATE_MASK, P):
            outs.append(hb.conv2d_bwd_data(dy, w, (B, C, H, W), 1, 1, P, gate_mask=gmask))
            assert torch.equal(outs[-1], outs[2]), "mask-gated data gradient differs from the tensor-gated one"
        return outs

    os.environ["STYLEX_CONV_LINE64"] = "0"  # the 64 -> 64 kernel of round 4 would take the large 64-channel cases from both arms
    os.environ["STYLEX_CONV_PIPE"] = "0"
    try:
        ref = run()
        os.environ.pop("STYLEX_CONV_PIPE", None)
        got = run()
    finally:
        os.environ.pop("STYLEX_CONV_PIPE", None)
        os.environ.pop("STYLEX_CONV_LINE64", None)
    assert len(ref) == len(got) and len(got) >= 3
    for k, (a, b) in enumerate(zip(ref, got)):
        assert torch.equal(a, b), "output %d: max diff %g" % (k, float((a.float() - b.float()).abs().max()))
    yr = F.leaky_relu(F.conv2d(x.float(), w.to(torch.bfloat16).float(), bias, 1, 1), 0.2)
    close(yr, got[0].float(), 1e-2, "pipe fwd vs fp32 definition")
    torch.cuda.synchronize()


@pytest.mark.against_definition
@pytest.mark.parametrize("case", [(2, 128, 128), (1, 256, 256), (5, 128, 160), (3, 256, 128), (67, 128, 128)])
def test_line64_conv_kernel_matches_pipelined_kernel_and_definition(case):
    """conv_line64.hip (round 4: 64 -> 64 channels at >= 128^2, whole-pixel K stage, weights resident in LDS) against the
    pipelined kernel on the same inputs (STYLEX_CONV_LINE64=0) and against the fp32 definition: forward with bias +
    LeakyReLU + activation bit mask, plain data gradient, data gradient gated by a bit mask.  The two kernels add the
    same products in a different order (tap-major over whole pixels against 16-channel chunks), so results agree to the
    bf16 rounding of the output (<= 1 ulp on a small fraction of elements), not bit for bit; the mask must be the sign
    of the kernel's OWN output; a second run is bit-identical (fixed order, no atomics).  Cases: fewer tiles than CUs,
    several tiles per block (67 x 64 = 4288 tiles), non-square images."""
    import os

    B, H, W = case
    C = N = 64
    ops.set_precision("bf16")
    P = hb.BF16_ACT
    g = torch.Generator(device=DEV).manual_seed(35)
    mk = lambda *sh: torch.randn(*sh, device=DEV, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)  # noqa: E731
    x, dy, gate = mk(B, C, H, W), mk(B, N, H, W), mk(B, C, H, W)
    w = torch.randn(N, C, 3, 3, device=DEV, generator=g) / (9 * C) ** 0.5
    bias = torch.randn(N, device=DEV, generator=g)
    bits = (gate.permute(0, 2, 3, 1).float() > 0).reshape(B, H, W, C // 8, 8).to(torch.int32)
    gmask = (bits * (2 ** torch.arange(8, device=DEV, dtype=torch.int32))).sum(-1).to(torch.uint8).contiguous()

    def run():
        hb.timing_enable(True)
        y, m = hb.conv2d_fwd(x, w, 1, 1, P, bias=bias, lrelu=True, want_mask=True)
        outs = [y, m, hb.conv2d_bwd_data(dy, w, (B, C, H, W), 1, 1, P), hb.conv2d_bwd_data(dy, w, (B, C, H, W), 1, 1, P, gate_mask=gmask)]
        torch.cuda.synchronize()
        names = [k["kernel"] for k in hb.timing_kernels()]
        hb.timing_enable(False)
        return outs, names

    os.environ["STYLEX_CONV_LINE64"] = "0"
    try:
        ref, ref_names = run()
    finally:
        os.environ.pop("STYLEX_CONV_LINE64", None)
    got, names = run()
    again, _ = run()
    assert any("line64" in n for n in names) and not any("line64" in n for n in ref_names), (names, ref_names)
    for k in (0, 2, 3):
        a, b = ref[k].float(), got[k].float()
        # one bf16 step at the magnitude of the element; near zero the two summation orders differ by the fp32 rounding of
        # sums of O(1) terms (~1e-6) whatever the size of the result
        # (the gated data gradient rounds twice — the sum, then its gate-scaled value — so a one-step difference of the
        # first rounding can become two steps of the result)
        ulp = (a.abs() * (2.0 ** -6 if k == 3 else 2.0 ** -7)).clamp_min(2e-5)
        d = (a - b).abs()
        assert bool((d <= ulp).all()), "output %d: max diff %g" % (k, float(d.max()))
        assert float((d > 0).float().mean()) < 0.05, "output %d: %g of the elements differ" % (k, float((d > 0).float().mean()))
        assert torch.equal(got[k], again[k]), "output %d is not reproducible" % k
    ybits = (got[0].permute(0, 2, 3, 1).float() > 0).reshape(B, H, W, N // 8, 8).to(torch.int32)
    want_mask = (ybits * (2 ** torch.arange(8, device=DEV, dtype=torch.int32))).sum(-1).to(torch.uint8)
    assert torch.equal(got[1].reshape(-1), want_mask.reshape(-1)), "mask is not the sign of the stored output"
    if B <= 5:
        wb = w.to(torch.bfloat16).float()
        close(F.leaky_relu(F.conv2d(x.float(), wb, bias, 1, 1), 0.2), got[0].float(), 1e-2, "line64 fwd vs fp32 definition")
        dxr = F.conv_transpose2d(dy.float(), wb, None, 1, 1)
        close(dxr, got[2].float(), 1e-2, "line64 dgrad vs fp32 definition")
        close(torch.where(gate.float() > 0, dxr, 0.2 * dxr), got[3].float(), 1e-2, "line64 gated dgrad vs fp32 definition")
    torch.cuda.synchronize()


@pytest.mark.against_definition
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("case", [(2, 64, 32, 64, 64), (2, 32, 64, 32, 32), (2, 64, 64, 16, 16), (3, 16, 24, 8, 8)])
def test_natural_order_noise_plane_epilogue(case, prec):
    """STYLEX_EPI_NOISE_NAT (noise plane pre-transposed once per generator forward, 16-byte loads in the epilogue) gives
    bit-identical results to the reference-order plane on every kernel family (LDS-halo wide / narrow, implicit GEMM)."""
    B, C, N, H, W = case
    ops.set_precision(prec)
    P = {"fp32": hb.F32, "bf16": hb.BF16_ACT}[prec]
    adt = hb.act_dtype(P)
    g = torch.Generator(device=DEV).manual_seed(31)
    x = torch.randn(B, C, H, W, device=DEV, generator=g).to(adt).contiguous(memory_format=torch.channels_last)
    w = torch.randn(N, C, 3, 3, device=DEV, generator=g) / (9 * C) ** 0.5
    s1, d = torch.rand(B, C, device=DEV, generator=g) + 0.5, torch.rand(B, N, device=DEV, generator=g) + 0.5
    inoise = torch.rand(B, 64, 64, 1, device=DEV, generator=g)
    nw, nb = torch.randn(N, device=DEV, generator=g), torch.randn(N, device=DEV, generator=g)
    plane = inoise[:, :, :, 0]
    ref = hb.conv2d_fwd(x, w, 1, 1, P, in_scale=s1, out_scale=d, noise=plane, noise_w=nw, noise_b=nb, lrelu=True)
    nat = ops._natural_noise(inoise)
    got = hb.conv2d_fwd(x, w, 1, 1, P, in_scale=s1, out_scale=d, noise=nat, noise_w=nw, noise_b=nb, lrelu=True, noise_natural=True)
    assert torch.equal(ref, got), float((ref.float() - got.float()).abs().max())
    # the one-entry cache is keyed by tensor identity + version: another noise tensor (possibly at a recycled address)
    # or an in-place refill (static graph input buffer) must never see a stale plane
    other = torch.rand_like(inoise)
    assert torch.equal(ops._natural_noise(other), other[:, :, :, 0].transpose(1, 2))
    inoise2 = inoise.clone()
    first = ops._natural_noise(inoise2)
    assert ops._natural_noise(inoise2) is first
    inoise2.copy_(other)
    assert torch.equal(ops._natural_noise(inoise2), other[:, :, :, 0].transpose(1, 2))
    # and against the definition: value at (h, w) is inoise[b, w, h]
    want = F.leaky_relu(F.conv2d(x.float() * s1[:, :, None, None], w, padding=1) * d[:, :, None, None]
                        + plane[:, :W, :H].transpose(1, 2)[:, None] * nw[None, :, None, None] + nb[None, :, None, None], 0.2)
    close(want, got.float(), 1e-4 if prec == "fp32" else 3e-2, "noise epilogue vs definition")


@pytest.mark.against_definition
@pytest.mark.parametrize("case", [(32, 512, 512, 8, 3), (64, 512, 512, 4, 3), (128, 512, 512, 2, 3), (5, 128, 192, 8, 3),
                                  (64, 512, 512, 8, 1), (128, 512, 512, 2, 1), (3, 64, 64, 4, 3)])
def test_small_spatial_gather_kernel(case):
    """conv_gather.hip (LDS-DMA implicit GEMM for the <= 8x8 px layers, 3x3/s1/p1 and 1x1) against an fp64 evaluation
    of the definition and against the generic kernel it replaces (STYLEX_CONV_GATHER=0): forward with bias + LeakyReLU,
    data gradient plain and with the activation gate.  Partial tiles (M, N not multiples of 128) included."""
    import os

    B, C, N, S, k = case
    ops.set_precision("bf16")
    P = hb.BF16_ACT
    g = torch.Generator(device=DEV).manual_seed(41)
    mk = lambda *sh: torch.randn(*sh, device=DEV, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)  # noqa: E731
    x, dy, gate = mk(B, C, S, S), mk(B, N, S, S), mk(B, C, S, S)
    w = torch.randn(N, C, k, k, device=DEV, generator=g) / (k * k * C) ** 0.5
    bias = torch.randn(N, device=DEV, generator=g)
    pad = (k - 1) // 2

    def run():
        return (hb.conv2d_fwd(x, w, 1, pad, P, bias=bias, lrelu=True),
                hb.conv2d_bwd_data(dy, w, (B, C, S, S), 1, pad, P),
                hb.conv2d_bwd_data(dy, w, (B, C, S, S), 1, pad, P, gate=gate))

    got = run()
    os.environ["STYLEX_CONV_GATHER"] = "0"
    try:
        ref = run()
    finally:
        os.environ.pop("STYLEX_CONV_GATHER", None)
    for nm, a, b in zip(("fwd", "dgrad", "dgrad+gate"), ref, got):
        close(a.float(), b.float(), 1e-2, nm + " vs generic kernel")
    wb = w.to(torch.bfloat16).double()
    want = F.leaky_relu(F.conv2d(x.double(), wb, bias.double(), padding=pad), 0.2)
    close(want, got[0].double(), 1e-2, "fwd vs fp64 definition")
    want_dx = torch.nn.grad.conv2d_input((B, C, S, S), wb, dy.double(), padding=pad)
    close(want_dx, got[1].double(), 1e-2, "dgrad vs fp64 definition")


@pytest.mark.against_definition
@pytest.mark.parametrize("case", [(3, 64, 64, 32, 64), (2, 128, 64, 48, 80), (2, 8, 64, 40, 64), (1, 256, 128, 16, 32)])
def test_activation_bit_mask_paths_are_bit_identical(case):
    """STYLEX_EPI_MASK_OUT / STYLEX_EPI_GATE_MASK / stylex_blur3x3_s2d_bwd_gate_mask: the forward conv writes one bit per
    stored element (y > 0) next to y; the gated data gradient and the gated blur adjoint read those bits instead of the
    gate tensor.  Same arithmetic, so everything must agree BIT FOR BIT with the tensor-gated launches; the mask itself
    is checked against the definition.  Shapes: LDS-DMA kernel (ragged tiles included) and the RGB first-layer kernel
    (C = 8); a shape no mask-capable kernel serves must return None / fall back."""
    B, C, N, H, W = case
    ops.set_precision("bf16")
    P = hb.BF16_ACT
    g = torch.Generator(device=DEV).manual_seed(81)
    mk = lambda *sh: torch.randn(*sh, device=DEV, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)  # noqa: E731
    x = mk(B, C, H, W)
    if C == 8:
        x[:, 3:] = 0
    w = torch.randn(N, C, 3, 3, device=DEV, generator=g) / (9 * C) ** 0.5
    bias = torch.randn(N, device=DEV, generator=g)
    y_ref = hb.conv2d_fwd(x, w, 1, 1, P, bias=bias, lrelu=True)
    y, mask = hb.conv2d_fwd(x, w, 1, 1, P, bias=bias, lrelu=True, want_mask=True)
    assert mask is not None and mask.shape == (B, H, W, N // 8)
    assert torch.equal(y, y_ref)
    bits = (y.permute(0, 2, 3, 1).float() > 0).reshape(B, H, W, N // 8, 8).to(torch.int32)
    want = (bits << torch.arange(8, device=DEV, dtype=torch.int32)).sum(-1).to(torch.uint8)
    assert torch.equal(mask, want)
    if C >= 64:  # the data gradient of a conv whose INPUT is y (gate = y): N -> N2 channels, reads the mask of y
        N2 = 64
        w2 = torch.randn(N2, N, 3, 3, device=DEV, generator=g) / (9 * N) ** 0.5
        dy = mk(B, N2, H, W)
        a = hb.conv2d_bwd_data(dy, w2, (B, N, H, W), 1, 1, P, gate=y)
        b_ = hb.conv2d_bwd_data(dy, w2, (B, N, H, W), 1, 1, P, gate=y, gate_mask=mask)
        assert hb.conv_mask_supported(hb.conv_shape((B, N, H, W), w2.shape, 1, 1), 1, hb.EPI_GATE_MASK, P)
        assert torch.equal(a, b_)
    if H % 2 == 0 and W % 2 == 0 and H >= 8:  # gated adjoint of the space-to-depth blur
        dy2 = hb.blur3x3_s2d_fwd(mk(B, N, H, W))
        assert hb.blur_mask_ok((B, N, H, W), torch.bfloat16)
        assert torch.equal(hb.blur3x3_s2d_bwd(dy2, gate=y), hb.blur3x3_s2d_bwd(dy2, gate_mask=mask))
    # a launch no mask-capable kernel serves: no mask, plain result
    xs, ws_ = mk(2, 64, 8, 8), torch.randn(64, 64, 3, 3, device=DEV, generator=g) / 24
    ys, ms = hb.conv2d_fwd(xs, ws_, 1, 1, P, lrelu=True, want_mask=True)
    assert ms is None and torch.equal(ys, hb.conv2d_fwd(xs, ws_, 1, 1, P, lrelu=True))


@pytest.mark.parametrize("case", [(4, 64, 64, 64, 64), (3, 128, 64, 48, 96), (2, 64, 256, 32, 32), (33, 64, 64, 32, 32)])
def test_bias_gradient_from_the_weight_gradient_kernel(case):
    """stylex_conv2d_bwd_weight_bias: the LDS-DMA weight-gradient kernel also returns db[n] = sum of dy over (b, h, w)
    (one MFMA per k-step against a vector of ones, per-split partials reduced in fixed order): against the fp64 sum of
    the bf16 gradient, with ragged tiles, several output-channel tiles and many splits; dw unchanged by the option.
    A shape another kernel serves reports None."""
    B, C, N, H, W = case
    ops.set_precision("bf16")
    P = hb.BF16_ACT
    g = torch.Generator(device=DEV).manual_seed(91)
    mk = lambda *sh: torch.randn(*sh, device=DEV, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)  # noqa: E731
    x, dy = mk(B, C, H, W), mk(B, N, H, W)
    dw0 = hb.conv2d_bwd_weight(x, dy, (N, C, 3, 3), 1, 1, P)
    dw, db = hb.conv2d_bwd_weight(x, dy, (N, C, 3, 3), 1, 1, P, want_bias_sum=True)
    assert torch.equal(dw, dw0)
    assert db is not None and db.shape == (N,)
    close(dy.double().sum(dim=(0, 2, 3)), db, 2e-5, "bias gradient")
    xs, dys = mk(2, 64, 8, 8), mk(2, 64, 8, 8)
    dws, dbs = hb.conv2d_bwd_weight(xs, dys, (64, 64, 3, 3), 1, 1, P, want_bias_sum=True)
    assert dbs is None and torch.equal(dws, hb.conv2d_bwd_weight(xs, dys, (64, 64, 3, 3), 1, 1, P))


def test_fused_dblock_with_and_without_bit_masks():
    """The fused DiscriminatorBlock with the activation bit masks (default) and with STYLEX_GATE_MASK=0 (gates read from
    the activation tensors): output and every gradient bit-identical, at sizes where both masks are in use (RGB first
    block and a 64 -> 128 block on the LDS-DMA kernel)."""
    import os

    ops.set_precision("bf16")
    os.environ["STYLEX_GATE_MASK_MIN_PIXELS"] = "0"  # the product switches the masks on from 128^2 up
    for cin, cout, size in ((3, 64, 64), (64, 128, 64)):
        torch.manual_seed(21)
        blk = st.DiscriminatorBlock(cin, cout, downsample=True).to(DEV)
        x = torch.randn(3, cin, size, size, device=DEV, generator=torch.Generator(device=DEV).manual_seed(22))
        res = []
        for masks in ("1", "0"):
            os.environ["STYLEX_GATE_MASK"] = masks
            try:
                blk.zero_grad()
                xr = x.clone().requires_grad_()
                prev = ops.set_fast(True)
                try:
                    y = blk(xr)
                    r = torch.randn(y.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(23)).to(y.dtype)
                    (y.float() * r.float()).sum().backward()
                finally:
                    ops.set_fast(prev)
            finally:
                os.environ.pop("STYLEX_GATE_MASK", None)
            res.append([y.detach(), xr.grad] + [p.grad.clone() for p in blk.parameters()])
        for a, b_ in zip(*res):
            assert torch.equal(a, b_)
    os.environ.pop("STYLEX_GATE_MASK_MIN_PIXELS", None)


@pytest.mark.parametrize("shape", [(3, 4, 4, 4), (2, 4, 8, 16), (2, 3, 5, 7), (1, 8, 1, 1), (2, 4, 64, 64)])
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_rgb_up_blur_add_kernel(shape, prec):
    """stylex_rgb_up_blur_add_fwd/_bwd == Blur(Upsample2x(rgb + prev)) of RGBBlock.forward (reference :622-626) and its
    adjoint, against the oracle's float64 composition (clamped bilinear x2, then the reflect 3x3 blur), with and
    without the skip input, odd sizes and the 1x1 image included."""
    B, C, H, W = shape
    g = torch.Generator().manual_seed(71)
    rgb0, prev0 = torch.randn(B, C, H, W, generator=g), torch.randn(B, C, H, W, generator=g)
    dy0 = torch.randn(B, C, 2 * H, 2 * W, generator=g)
    dt = torch.float32 if prec == "fp32" else torch.bfloat16
    tol = 1e-5 if prec == "fp32" else 1.2e-2
    for with_prev in (True, False):
        r, pv = rgb0.to(dt).double().requires_grad_(), prev0.to(dt).double().requires_grad_()
        t = r + pv if with_prev else r
        want = so.blur3x3_reflect(F.interpolate(t, scale_factor=2, mode="bilinear", align_corners=False))
        want.backward(dy0.to(dt).double())
        got = hb.rgb_up_blur_add_fwd(cl(rgb0).to(dt), cl(prev0).to(dt) if with_prev else None)
        close(want, got, tol, "rgb tail fwd prev=%s" % with_prev)
        gin = hb.rgb_up_blur_add_bwd(cl(dy0).to(dt))
        close(r.grad, gin, tol, "rgb tail adjoint prev=%s" % with_prev)


def test_fused_rgb_path_matches_composable_chain():
    """Generator forward / backward with the fused RGB skip path (to-RGB + _RGBTailFast on the 4-channel storage) against
    the composable chain (STYLEX_RGB_TAIL=0: to-RGB, add, Upsample2x, Blur as separate nodes), bf16 speed mode: image and
    every parameter gradient."""
    import os

    from networks import Generator

    ops.set_precision("bf16")
    torch.manual_seed(5)
    G = Generator(64, 32, network_capacity=8).to(DEV)
    styles = torch.randn(3, G.num_layers, 32, device=DEV)
    noise = torch.rand(3, 64, 64, 1, device=DEV)
    gout = torch.randn(3, 3, 64, 64, device=DEV)

    def run():
        G.zero_grad(set_to_none=True)
        st = styles.clone().requires_grad_()
        img = G(st, noise)
        img.backward(gout)
        return img.detach(), st.grad, {n: p.grad.clone() for n, p in G.named_parameters() if p.grad is not None}

    prev = ops.set_fast(True)
    try:
        img_f, gs_f, gp_f = run()
        os.environ["STYLEX_RGB_TAIL"] = "0"
        try:
            img_c, gs_c, gp_c = run()
        finally:
            os.environ.pop("STYLEX_RGB_TAIL", None)
    finally:
        ops.set_fast(prev)
    assert img_f.shape == (3, 3, 64, 64)
    close(img_c, img_f, 2e-2, "image")
    close(gs_c, gs_f, 4e-2, "style grad")
    assert gp_f.keys() == gp_c.keys()
    for n in gp_c:
        close(gp_c[n], gp_f[n], 4e-2, "grad " + n)


@pytest.mark.against_definition
@pytest.mark.parametrize("case", [(64, 512, 512, 16, 16), (32, 512, 512, 8, 8), (64, 512, 512, 4, 4), (5, 128, 192, 16, 8),
                                  (3, 64, 64, 4, 12), (130, 64, 64, 2, 2)])
def test_small_spatial_gather_kernel_stride2(case):
    """conv_gather.hip on the 3x3/s2/p1 down conv of the small DiscriminatorBlocks (output <= 8x8 px): forward with bias +
    residual merge, and the data gradient, whose rows are grouped by output-pixel parity so that every tile gathers its
    1, 2 or 4 live taps only — against the generic kernel (STYLEX_CONV_GATHER=0) and an fp64 evaluation of the
    definition.  Ragged tiles and non-square grids included."""
    import os

    B, C, N, H, W = case
    ops.set_precision("bf16")
    P = hb.BF16_ACT
    g = torch.Generator(device=DEV).manual_seed(43)
    mk = lambda *sh: torch.randn(*sh, device=DEV, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)  # noqa: E731
    x, dy, res = mk(B, C, H, W), mk(B, N, H // 2, W // 2), mk(B, N, H // 2, W // 2)
    w = torch.randn(N, C, 3, 3, device=DEV, generator=g) / (9 * C) ** 0.5
    bias = torch.randn(N, device=DEV, generator=g)

    def run():
        return (hb.conv2d_fwd(x, w, 2, 1, P, bias=bias, residual=res, res_scale=0.7),
                hb.conv2d_bwd_data(dy, w, (B, C, H, W), 2, 1, P))

    got = run()
    os.environ["STYLEX_CONV_GATHER"] = "0"
    try:
        ref = run()
    finally:
        os.environ.pop("STYLEX_CONV_GATHER", None)
    for nm, a, b in zip(("fwd", "dgrad"), ref, got):
        close(a.float(), b.float(), 1e-2, nm + " vs generic kernel")
    wb = w.to(torch.bfloat16).double()
    want = (F.conv2d(x.double(), wb, bias.double(), stride=2, padding=1) + res.double()) * 0.7
    close(want, got[0].double(), 1e-2, "fwd vs fp64 definition")
    want_dx = torch.nn.grad.conv2d_input((B, C, H, W), wb, dy.double(), stride=2, padding=1)
    close(want_dx, got[1].double(), 1e-2, "dgrad vs fp64 definition")


@pytest.mark.against_definition
@pytest.mark.parametrize("case", [(8, 64, 32, 256), (16, 128, 128, 64), (64, 256, 256, 32), (4, 64, 64, 128)])
def test_modulated_wgrad_scale_in_epilogue(case):
    """Weight gradient of a modulated layer on the LDS-DMA kernel: the per-sample modulation s[b][c] of x is a factor of
    each sample's partial sum (splits never cross a sample) and is applied to the accumulators when a split ends —
    against the register-staged kernel that scales x while staging (STYLEX_WGRAD_DMA=0 is read once per process, so the
    reference here is the fp32 definition)."""
    B, C, N, S = case
    ops.set_precision("bf16")
    P = hb.BF16_ACT
    g = torch.Generator(device=DEV).manual_seed(61)
    mk = lambda *sh: torch.randn(*sh, device=DEV, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)  # noqa: E731
    x, dy = mk(B, C, S, S), mk(B, N, S, S)
    s1 = torch.randn(B, C, device=DEV, generator=g) * 0.5 + 1.0
    got = hb.conv2d_bwd_weight(x, dy, (N, C, 3, 3), 1, 1, P, x_scale=s1)
    xs = (x.float() * s1[:, :, None, None])
    want = torch.nn.grad.conv2d_weight(xs.double(), (N, C, 3, 3), dy.double(), padding=1)
    close(want, got.double(), 2e-2, "modulated wgrad")
    plain = hb.conv2d_bwd_weight(x, dy, (N, C, 3, 3), 1, 1, P)
    want0 = torch.nn.grad.conv2d_weight(x.double(), (N, C, 3, 3), dy.double(), padding=1)
    close(want0, plain.double(), 2e-2, "plain wgrad")


@pytest.mark.against_definition
@pytest.mark.parametrize("np_tile", ["1", "2"])
@pytest.mark.parametrize("case", [(3, 128, 128, 32, 32, "plain"), (3, 128, 128, 32, 32, "bias"), (2, 64, 256, 16, 16, "bias"),
                                  (5, 192, 128, 8, 16, "plain"), (2, 64, 256, 64, 64, "bias"), (4, 64, 128, 32, 32, "mod"),
                                  (6, 128, 128, 16, 16, "mod"), (2, 64, 128, 32, 32, "s2d"), (3, 128, 128, 16, 16, "s2d"),
                                  (1, 64, 256, 8, 32, "s2d")])
def test_weight_gradient_pipe_kernel_both_block_tiles(case, np_tile, monkeypatch):
    """conv_wgrad_pipe.hip against the fp64 definition with the block tile FORCED (STYLEX_WGRAD_PIPE_NP): the 128(n) x 64(c)
    tile (NP64 = 2) is what the benchmark's large launches select (>= 48 stages per block, csrc/conv_wgrad_pipe.hip
    wg_np64) and what no small test shape reaches by itself (round-5 VERDICT, weak 1).  Forms: plain, with the bias sums
    riding the launch, with the per-sample modulation applied to the accumulators (a_scale), and the stride-2 conv given
    space-to-depth with the folded [N][C][3][3] result; both tile widths (32 / 16 pixel columns), 1-3 input-channel tiles,
    1-2 output tiles, several K splits.  Operands are bf16 values, so the fp64 gradient differs only by the fp32
    accumulation: 1e-4 of the largest element.  The instantiation that ran is read back from the timing hook."""
    B, C, N, H, W, form = case
    monkeypatch.setenv("STYLEX_WGRAD_PIPE_NP", np_tile)
    ops.set_precision("bf16")
    P = hb.BF16_ACT
    g = torch.Generator(device=DEV).manual_seed(67)
    mk = lambda *sh: torch.randn(*sh, device=DEV, generator=g).to(torch.bfloat16)  # noqa: E731
    tw = 32 if W >= 32 else 16
    if form == "s2d":
        x = mk(B, C, 2 * H, 2 * W)
        x2 = cl(x.view(B, C, H, 2, W, 2).permute(0, 3, 5, 1, 2, 4).reshape(B, 4 * C, H, W))
        dy = cl(mk(B, N, H, W))
        got = hb.conv2d_bwd_weight_s2d(x2, dy, (N, C, 3, 3), P)
        wz = torch.zeros(N, C, 3, 3, dtype=torch.float64, device=DEV, requires_grad=True)
        (want,) = torch.autograd.grad(F.conv2d(x.double(), wz, stride=2, padding=1), wz, dy.double())
        expect = "conv3x3_wgrad_pipe_kernel<%s, %d, false, true>" % (np_tile, tw)
    else:
        x, dy = cl(mk(B, C, H, W)), cl(mk(B, N, H, W))
        s1 = torch.randn(B, C, device=DEV, generator=g) * 0.5 + 1.0 if form == "mod" else None
        if form == "bias":
            got, db = hb.conv2d_bwd_weight(x, dy, (N, C, 3, 3), 1, 1, P, want_bias_sum=True)
            assert db is not None
            close(dy.double().sum(dim=(0, 2, 3)), db, 2e-5, "bias sums")
        else:
            got = hb.conv2d_bwd_weight(x, dy, (N, C, 3, 3), 1, 1, P, x_scale=s1)
        xs = x.double() if s1 is None else x.double() * s1.double()[:, :, None, None]
        want = torch.nn.grad.conv2d_weight(xs, (N, C, 3, 3), dy.double(), padding=1)
        expect = "conv3x3_wgrad_pipe_kernel<%s, %d, %s, false>" % (np_tile, tw, "true" if form == "bias" else "false")
    torch.cuda.synchronize()
    ran = {k["kernel"] for k in hb.timing_kernels() if k["cls"] == "bwd_weight"}
    assert expect in ran, (expect, ran)
    assert got.shape == (N, C, 3, 3)
    close(want, got.double(), 1e-4, "weight gradient %s NP64=%s" % (form, np_tile))


@pytest.mark.against_definition
@pytest.mark.parametrize("case", [(4, 64, 64), (2, 40, 72), (3, 256, 256)])
def test_first_layer_rgb_kernel(case):
    """conv_rgb.hip (3x3 over the padded RGB slot to 64 channels, bias + LeakyReLU: the first conv of every
    DiscriminatorBlock chain) against the generic kernel (STYLEX_CONV_RGB=0) and the fp64 definition; partial tiles."""
    import os

    B, H, W = case
    ops.set_precision("bf16")
    g = torch.Generator(device=DEV).manual_seed(71)
    x = torch.rand(B, 3, H, W, device=DEV, generator=g)
    w = torch.randn(64, 3, 3, 3, device=DEV, generator=g) / 27 ** 0.5
    bias = torch.randn(64, device=DEV, generator=g)
    with torch.no_grad():
        got = ops.conv2d(x, w, bias, 1, 1, lrelu=True)
        os.environ["STYLEX_CONV_RGB"] = "0"
        try:
            ref = ops.conv2d(x, w, bias, 1, 1, lrelu=True)
        finally:
            os.environ.pop("STYLEX_CONV_RGB", None)
    close(ref.float(), got.float(), 1e-2, "rgb kernel vs generic")
    want = F.leaky_relu(F.conv2d(x.to(torch.bfloat16).double(), w.to(torch.bfloat16).double(), bias.double(), padding=1), 0.2)
    close(want, got.double(), 1e-2, "rgb kernel vs fp64 definition")


def test_gradsync_on_one_rank_rccl_is_bit_identical_to_single_gpu():
    """the gradient exchange on the REAL RCCL code path (a 1-rank process group): GradSync with the
    collectives after the backward and with the in-backward bucket launch (overlap=True) both end three train() calls
    with parameters and loss scalars BIT-IDENTICAL to the plain single-GPU Trainer (tools/ddp_overlap_identity.py; a
    subprocess, so that the process group cannot leak into the other tests)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "ddp_overlap_identity.py"), "3"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    rep = json.loads(line)
    assert rep["identical_to_single"] == {"post": True, "overlap": True}, rep
    assert rep["n_params"] > 100


@pytest.mark.timeout(900)
def test_two_ranks_on_one_gpu_exchange_gradients_over_gloo():
    """The N > 1 path as close to hardware as a 1-GPU box allows (tools/ddp_two_ranks_one_gpu.py): two processes, both on
    cuda:0 with the real HIP kernels at config 2's network sizes (ResNet-18 classifier, LPIPS-AlexNet, batch 8 per rank),
    gradients exchanged by a REAL two-rank all-reduce (gloo: RCCL refuses two ranks on one device) from the same hooks /
    launch order / stream joins as the RCCL path.  The trained networks of the replicas are bit-identical after every call
    in both exchange modes, the in-backward bucket launch (4 + 7 buckets) ends with the same bits as the post-backward
    exchange on both ranks — a bucket reduced before one of its gradients was complete would not —, the first-use
    self-check ran and passed, and the exchange really mixed the ranks' gradients.  (This test found the round's to-RGB
    kernel defect: profiles/r06_torgb_contention.txt.)"""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = 29553
    procs = []
    for rank in (0, 1):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2",
                   LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0", TWO_BATCH="8")
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "tools", "ddp_two_ranks_one_gpu.py"), "3"], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=800))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    line = [ln for ln in outs[0][0].splitlines() if ln.startswith("{")][-1]
    rep = json.loads(line)
    out_dir = os.path.join(root, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "two_ranks_one_gpu.json"), "w") as f:
        f.write(line + "\n")
    assert rep["world"] == 2 and len(rep["ranks"]) == 2
    for r in rep["ranks"]:
        assert r["replicas_identical"] == {"post": [True] * 3, "overlap": [True] * 3}, r
        assert r["overlap_equals_post"], r
        assert r["exchange_changed_the_update"], r
        assert r["finite"], r
        assert r["selfcheck"]["still_overlapped"] == [True, True], r
        assert min(r["selfcheck"]["ran"]) >= 1, r
        assert min(r["buckets"]["overlap"]) >= 2, r  # several in-backward launches per phase
        assert r["n_param_values"] > 50_000_000
    assert rep["ranks"][0]["scalars_overlap"] != rep["ranks"][1]["scalars_overlap"]  # the ranks saw different data


@pytest.mark.timeout(900)
def test_bench_py_two_ranks_on_one_gpu():
    """bench.py's N = 2 code path with the real kernels (tools/bench_two_ranks_one_gpu.py: both ranks on cuda:0, gloo): the
    process group, the barriers, the MAX-reduced timing, the instrumented roofline steps with their collectives on every
    rank, one JSON line from rank 0 with n_gpus 2 / dp2 and no CPU baseline (N = 1 only).  Its images/s is not a scaling
    number (two ranks share the GPU and the buckets travel through the host)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "bench_two_ranks_one_gpu.py"), "--steps", "3", "--warmup", "2",
                        "--bench-a-steps", "0", "--fp32-steps", "0", "--roofline-steps", "1"], capture_output=True, text=True,
                       timeout=800, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 only
    rep = json.loads(lines[0])
    assert rep["n_gpus"] == 2 and rep["config"]["parallelism"] == "dp2" and rep["scaling"] == "weak"
    assert rep["config"]["global_batch"] == 2 * rep["config"]["batch_per_gpu"]
    assert rep["value"] > 0 and rep["cpu_baseline"] is None and rep["roofline"]["launches"] > 0


def test_adam_pack_step_matches_torch_fused_adam_and_the_pack_kernels():
    """csrc/adam_pack.hip (round 4): ONE launch = the Adam update of torch._fused_adam_ on the optimiser's own state
    tensors + every cached operand copy of the stepped weights rewritten from the updated values.  Against two reference
    runs from identical state: parameters / moments after 3 steps equal torch's fused Adam to fp32 rounding (same rule;
    the operation order inside an element differs by at most an ulp or two), and every refreshed cache entry — bf16
    operand packs [N][T][C] / [C][T][N] (plain and scaled), space-to-depth packs, the bf16 1x1 GEMM matrix, the tap sum
    of squares — is BIT-identical to what the pack kernels produce from the parameter values the fused step left
    behind; parameters without a gradient are untouched and keep their copies valid."""
    torch.manual_seed(5)
    dev = torch.device(DEV)
    shapes = [(64, 64, 3, 3), (24, 40, 3, 3), (128, 64, 1, 1), (3, 32, 1, 1), (70, 130, 3, 3), (512,), (17, 33), (8, 3, 3, 3)]

    def make():
        torch.manual_seed(6)
        ps = [torch.nn.Parameter(torch.randn(s, device=dev) * 0.1) for s in shapes]
        frozen = torch.nn.Parameter(torch.randn(32, 32, 3, 3, device=dev) * 0.1)  # never gets a gradient
        opt = torch.optim.Adam([{"params": ps[:4] + [frozen], "lr": 2e-4}, {"params": ps[4:], "lr": 1e-5}], betas=(0.5, 0.9),
                               fused=True)
        return ps, frozen, opt

    def grads(ps, k):
        g = torch.Generator(device=dev).manual_seed(100 + k)
        for p in ps:
            p.grad = torch.randn(p.shape, device=dev, generator=g) * (10.0 ** (k - 1))

    def touch(ps, frozen):  # create the cache entries a training step would have created
        c = 1 / 2 ** 0.5
        hb.pack_weight(ps[0], True, True, hb.BF16_ACT)
        hb.pack_weight(ps[0], False, True, hb.BF16_ACT, scale=c)
        hb.pack_weight_s2d(ps[0])
        hb.pack_weight_s2d(ps[0], scale=c)
        hb.pack_weight(ps[1], True, True, hb.BF16_ACT)
        hb.weight_sumsq(ps[1])
        hb._bf16_matrix(ps[2])
        hb._bf16_matrix(ps[2], c)
        hb.pack_weight(ps[3], True, True, hb.BF16_ACT)
        hb.pack_weight(ps[4], True, True, hb.BF16_ACT)
        hb.weight_sumsq(ps[4])
        hb.pack_weight(ps[7], True, True, hb.BF16_ACT)
        hb.pack_weight(frozen, True, True, hb.BF16_ACT)

    hb.pack_cache_clear()
    ref_p, ref_frozen, ref_opt = make()
    ps, frozen, opt = make()
    grads(ref_p, 0), grads(ps, 0)
    ref_opt.step()
    assert hb.adam_pack_step(opt) is False, "the first step initialises torch's state: must fall back"
    opt.step()
    touch(ps, frozen)
    frozen_pack = hb.pack_weight(frozen, True, True, hb.BF16_ACT)[0].clone()
    for k in (1, 2, 3):
        grads(ref_p, k), grads(ps, k)
        ref_opt.step()
        before = {key: (e[1], e[2]) for key, e in hb._PACK_CACHE.items()}
        assert hb.adam_pack_step(opt) is True
        for a, b in zip(ps, ref_p):
            close(b, a, 2e-6, "parameter after step %d" % k)
            sa, sb = opt.state[a], ref_opt.state[b]
            close(sb["exp_avg"], sa["exp_avg"], 2e-6, "exp_avg")
            close(sb["exp_avg_sq"], sa["exp_avg_sq"], 2e-6, "exp_avg_sq")
            assert float(sa["step"]) == float(sb["step"]) == k + 1
        # every entry the step claims to have refreshed: served from the cache (no re-pack) and bit-identical to a fresh pack
        c = 1 / 2 ** 0.5
        checks = [(lambda: hb.pack_weight(ps[0], True, True, hb.BF16_ACT), ps[0], None, "pack"),
                  (lambda: hb.pack_weight(ps[0], False, True, hb.BF16_ACT, scale=c), ps[0], c, "pack"),
                  (lambda: hb.pack_weight_s2d(ps[0]), ps[0], None, "s2d"),
                  (lambda: hb.pack_weight_s2d(ps[0], scale=c), ps[0], c, "s2d"),
                  (lambda: hb.pack_weight(ps[1], True, True, hb.BF16_ACT), ps[1], None, "pack"),
                  (lambda: (hb.weight_sumsq(ps[1]), None), ps[1], None, "wsq"),
                  (lambda: (hb._bf16_matrix(ps[2]), None), ps[2], None, "mat"),
                  (lambda: (hb._bf16_matrix(ps[2], c), None), ps[2], c, "mat"),
                  (lambda: hb.pack_weight(ps[3], True, True, hb.BF16_ACT), ps[3], None, "pack"),
                  (lambda: hb.pack_weight(ps[4], True, True, hb.BF16_ACT), ps[4], None, "pack"),
                  (lambda: (hb.weight_sumsq(ps[4]), None), ps[4], None, "wsq"),
                  (lambda: hb.pack_weight(ps[7], True, True, hb.BF16_ACT), ps[7], None, "pack")]
        for get, p, sc, kind in checks:
            got = [t.clone() if t is not None else None for t in get()]
            ptrs = {t.data_ptr() for pair in before.values() for t in pair if t is not None}
            assert all(t is None or t.data_ptr() not in () for t in got)
            served = get()
            assert any(t is not None and t.data_ptr() in ptrs for t in served), (kind, "not served from the cache")
            fresh = torch.nn.Parameter(p.detach().clone())  # a new Parameter object: never cached, packs from scratch
            if kind == "pack":
                want = hb.pack_weight(fresh, got[0] is not None, got[1] is not None, hb.BF16_ACT, scale=sc)
            elif kind == "s2d":
                want = hb.pack_weight_s2d(fresh, scale=sc)
            elif kind == "wsq":
                want = (hb.weight_sumsq(fresh), None)
            else:
                want = (hb._bf16_matrix(fresh, sc), None)
            for g_, w_ in zip(got, want):
                if g_ is None:
                    continue
                if kind == "wsq":
                    close(w_, g_, 1e-6, "tap sum of squares")
                else:
                    assert torch.equal(g_.view(-1), w_.view(-1)), (kind, sc, tuple(p.shape), "copy differs from a fresh pack")
        assert torch.equal(hb.pack_weight(frozen, True, True, hb.BF16_ACT)[0], frozen_pack)
    hb.pack_cache_clear()


def test_adam_pack_step_follows_load_state_dict():
    """the fused Adam + pack launch caches raw pointers to exp_avg / exp_avg_sq / step.
    Optimizer.load_state_dict installs NEW state tensors while the parameters keep their addresses (Trainer.load with
    save_training_state, or the NaN-restart path) — the cached plan must notice, or the kernel keeps updating freed
    buffers and the reloaded moments are never used.  step, save, step, load, step: parameters and moments equal a
    torch fused-Adam run through the same sequence."""
    import copy

    dev = torch.device(DEV)
    shapes = [(64, 64, 3, 3), (40,), (17, 33)]

    def make():
        torch.manual_seed(6)
        ps = [torch.nn.Parameter(torch.randn(s, device=dev) * 0.1) for s in shapes]
        return ps, torch.optim.Adam(ps, lr=2e-4, betas=(0.5, 0.9), fused=True)

    def grads(ps, k):
        g = torch.Generator(device=dev).manual_seed(200 + k)
        for p in ps:
            p.grad = torch.randn(p.shape, device=dev, generator=g)

    hb.pack_cache_clear()
    ref_p, ref_opt = make()
    ps, opt = make()
    for k in range(2):  # the first step initialises torch's state (fallback), the second runs the fused launch
        grads(ref_p, k), grads(ps, k)
        ref_opt.step()
        if not hb.adam_pack_step(opt):
            opt.step()
    saved, ref_saved = copy.deepcopy(opt.state_dict()), copy.deepcopy(ref_opt.state_dict())
    for k in (2, 3):  # move on: the state to be overwritten by the load differs from the saved one
        grads(ref_p, k), grads(ps, k)
        ref_opt.step()
        assert hb.adam_pack_step(opt) is True
    old_ptrs = {opt.state[p]["exp_avg"].data_ptr() for p in ps}
    opt.load_state_dict(saved)
    ref_opt.load_state_dict(ref_saved)
    assert {opt.state[p]["exp_avg"].data_ptr() for p in ps} != old_ptrs, "load_state_dict is expected to install new tensors"
    for k in (4, 5):
        grads(ref_p, k), grads(ps, k)
        ref_opt.step()
        assert hb.adam_pack_step(opt) is True
        for a, b in zip(ps, ref_p):
            close(b, a, 2e-6, "parameter after load + step")
            sa, sb = opt.state[a], ref_opt.state[b]
            close(sb["exp_avg"], sa["exp_avg"], 2e-6, "exp_avg after load")
            close(sb["exp_avg_sq"], sa["exp_avg_sq"], 2e-6, "exp_avg_sq after load")
            assert float(sa["step"]) == float(sb["step"]) == 2 + (k - 3)
    hb.pack_cache_clear()


def test_attfind_visualisation_cells_on_hip():
    """N1: the notebook's visualisation cells (batched: attfind.change_images / visualize_style /
    visualize_style_by_distance_in_s) on the HIP kernels against the arrays the reference notebook's own cells produced
    (tests/golden/attfind_visualize_64.npz)."""
    import test_attfind_cpu as ta

    g = np.load(ta.VGOLD)
    m, clf, noise = ta.build_visualize(g, device=DEV)
    prev = ta.FLIP_FRACTION[0]
    ta.FLIP_FRACTION[0] = 1e-2  # fp32 MFMA summation order vs the CPU's: a few more bytes sit on a truncation boundary
    try:
        ta.check_visualize(m, clf, noise, g, 1e-3)
    finally:
        ta.FLIP_FRACTION[0] = prev


@pytest.mark.against_definition
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_full_resolution_blocks_vs_cpu_oracle(prec):
    """VERDICT r3 weak point 2 (the full-size bf16 band is HIP-bf16 against HIP-fp32: a bug common to both modes at
    >= 128 px could only be caught by the adjoint identities): the blocks whose kernels exist ONLY at full resolution —
    DiscriminatorBlock 0 and 1 of the 256 px model (padded-RGB first layer, pipelined LDS-DMA convs, activation bit
    masks, blur + space-to-depth stride-2 conv with the residual merge, even-pixel gather / add) and GeneratorBlock 6
    (64 -> 32 @256^2: bilinear x2, modulated convs with the transposed noise plane, to-RGB) — against the independent
    CPU oracle modules (oracle/stylex_oracle.py, the restatement pinned to the reference goldens) run in float64 at
    batch 2: outputs, input gradients and every parameter gradient (90th percentile of the error over the tensor's RMS,
    see `rel`).  Bounds: fp32 1e-4 (measured 1e-7 ... 1.5e-6), except the gradients that reach the generator block's
    FIRST style vector, 3e-3: that quantity is ill-conditioned — the reference's own fp32 arithmetic (the CPU oracle in
    float32) is 2.7e-4 away from float64 there, and 3.4e-4 on conv1.weight where the HIP path is at 1.3e-6; HIP fp32
    measured 5.7e-4 ... 8.9e-4.  bf16: outputs 2e-2 (measured 5e-3 ... 7.5e-3), gradients 1.5e-1 — behind a LeakyReLU
    whose pre-activation carries a bf16-sized error, ~0.4 % of the gates sit on the other side of zero than in float64,
    which alone is a 5 % perturbation of the gradients below it (measured 3e-3 ... 1e-2 for the gradients that do not
    pass a gate, 4e-2 ... 9.8e-2 for those that do); a wrong index rule, tap order, scale or layout gives O(1)."""
    import networks

    ops.set_precision(prec)
    hb.pack_cache_clear()
    torch.manual_seed(11)

    def rel(a, b):
        # 90th percentile of the absolute error over the RMS of the reference.  Not a max / L2 norm: a LeakyReLU whose
        # pre-activation is within rounding of zero takes the other slope in fp32 / bf16 than in float64 — a handful of
        # the 4 M elements, each moving the 576 weight-gradient entries of its pixel and channel by O(1) (measured
        # 2.6e-3 of the tensor's maximum and 6e-4 in the L2 sense from such flips in the fp32 mode, while 90 % of the
        # entries agree to 1e-6); a wrong kernel moves every entry
        b = b.detach().double().cpu().reshape(-1)
        err = (a.detach().double().cpu().reshape(-1) - b).abs()
        k = max(1, int(0.9 * err.numel()))
        return float(err.kthvalue(k).values / max(1e-12, float(b.pow(2).mean().sqrt())))

    def tile_max(a, b):
        # Round-4 VERDICT weak point 2: a percentile cannot see a bug confined to < 10 % of the elements — a ragged last
        # tile, a border row, one channel half of a 128-channel tile.  Such bugs are LOCAL and O(1); rounding and flipped
        # LeakyReLU gates are spread evenly.  So: the RMS error of every kernel-tile-sized block — 32 channels x 4 rows x
        # 32 columns of an activation / gradient tensor, 32 x 32 (n, c) per tap of a conv weight gradient — over the RMS
        # of the whole reference tensor, maximum over the blocks.  A wrong block gives ~1 (0.7 for one wrong row of its
        # four); averaging over <= 4096 elements takes the gate-flip noise down to the tensor's mean error.
        a, b = a.detach().double().cpu(), b.detach().double().cpu()
        rms = max(1e-12, float(b.pow(2).mean().sqrt()))
        e2 = (a - b).pow(2)
        if e2.dim() == 4 and e2.shape[2] >= 32 and e2.shape[3] >= 32:  # activation [B, C, H, W]
            cb = 32 if e2.shape[1] % 32 == 0 else e2.shape[1]
            t = e2.reshape(e2.shape[0], e2.shape[1] // cb, cb, e2.shape[2] // 4, 4, e2.shape[3] // 32, 32).mean(dim=(2, 4, 6))
        elif e2.dim() == 4 and e2.shape[2] == e2.shape[3] and e2.shape[2] <= 3:  # conv weight [N, C, k, k]
            nb = 32 if e2.shape[0] % 32 == 0 else e2.shape[0]
            cb = 32 if e2.shape[1] % 32 == 0 else e2.shape[1]
            t = e2.reshape(e2.shape[0] // nb, nb, e2.shape[1] // cb, cb, -1).mean(dim=(1, 3))
        else:
            t = e2.mean().reshape(1)
        return float(t.max().sqrt()) / rms

    def bound(name):
        if prec == "fp32":
            return 3e-3 if name in ("gw", "g:to_style1.weight", "g:to_style1.bias") else 1e-4
        return 2e-2 if name in ("y", "x", "rgb", "coords") else 1.5e-1

    def tile_bound(name):
        # Measured (profiles/r05_f_parity_blocks.txt).  fp32: rounding 1e-7 ... 1e-6; a block that contains a flipped gate
        # 6e-4 ... 1.4e-3, the generator block's input gradient (noise-shifted pre-activations) 8.3e-3 -> 1e-2 / 3e-2.
        # bf16: the worst block equals the tensor's mean error (outputs 5.5e-3, gradients <= 8.7e-2: flip noise averages
        # out over a block) -> 4e-2 / 3e-1.  A wrong block is >= 0.7.
        if prec == "fp32":
            return 3e-2 if name in ("gx", "gw", "g:to_style1.weight", "g:to_style1.bias") else 1e-2
        return 4e-2 if name in ("y", "x", "rgb", "coords") else 3e-1

    def same_weights(hip_mod, cpu_mod):
        sd = {k: v.clone() for k, v in cpu_mod.state_dict().items()}
        missing = hip_mod.load_state_dict(sd, strict=False)
        assert not [k for k in missing.missing_keys if "blur" not in k and not k.endswith(".f")], missing
        return hip_mod.to(DEV)

    try:
        # ---- discriminator blocks 0 (3 -> 64 @256^2) and 1 (64 -> 128 @128^2)
        for cin, cout, size in ((3, 64, 256), (64, 128, 128)):
            ref = so.ODiscriminatorBlock(cin, cout, downsample=True)
            blk = same_weights(networks.DiscriminatorBlock(cin, cout, downsample=True), ref)
            ref = ref.double()
            x = torch.rand(2, cin, size, size) * 2 - 1
            gy = torch.randn(2, cout, size // 2, size // 2)
            xr = x.double().requires_grad_(True)
            yr = ref(xr)
            yr.backward(gy.double())
            xh = x.to(DEV).requires_grad_(True)
            ops.set_fast(True)
            try:
                yh = blk(xh)
                yh.float().backward(gy.to(DEV))
            finally:
                ops.set_fast(False)
            errs = {"y": rel(yh, yr), "gx": rel(xh.grad, xr.grad)}
            tiles = {"y": tile_max(yh, yr), "gx": tile_max(xh.grad, xr.grad)}
            ref_grads = dict(ref.named_parameters())
            for n_, p_ in blk.named_parameters():
                errs["g:" + n_] = rel(p_.grad, ref_grads[n_].grad)
                tiles["g:" + n_] = tile_max(p_.grad, ref_grads[n_].grad)
            print("DiscriminatorBlock %d->%d @%d %s:" % (cin, cout, size, prec), {k: "%.1e" % v for k, v in errs.items()})
            print("  worst block (RMS error / tensor RMS):", {k: "%.1e" % v for k, v in tiles.items()})
            bad = {k: v for k, v in errs.items() if v > bound(k)}
            assert not bad, ("DiscriminatorBlock %d->%d @%d" % (cin, cout, size), prec, bad)
            bad = {k: v for k, v in tiles.items() if v > tile_bound(k)}
            assert not bad, ("DiscriminatorBlock %d->%d @%d: a block of the tensor is off" % (cin, cout, size), prec, bad)
        # ---- generator block 6 (64 -> 32, upsample 128 -> 256, last block: no rgb upsample)
        ref = so.OGeneratorBlock(514, 64, 32, upsample=True, upsample_rgb=False)
        with torch.no_grad():
            for lin in (ref.to_noise1, ref.to_noise2):
                lin.weight.normal_(0, 0.3)
                lin.bias.normal_(0, 0.1)
        blk = same_weights(networks.GeneratorBlock(514, 64, 32, upsample=True, upsample_rgb=False), ref)
        ref = ref.double()
        x = torch.randn(2, 64, 128, 128)
        prev = torch.randn(2, 3, 256, 256) * 0.3
        w = torch.randn(2, 514) * 0.5
        nz = torch.rand(2, 256, 256, 1)
        gx_o, grgb = torch.randn(2, 32, 256, 256), torch.randn(2, 3, 256, 256)
        ins_r = [t.double().requires_grad_(True) for t in (x, prev, w)]
        xo_r, rgb_r, sc_r = ref(ins_r[0], ins_r[1], ins_r[2], nz.double())
        ((xo_r * gx_o.double()).sum() + (rgb_r * grgb.double()).sum()).backward()
        ins_h = [t.to(DEV).requires_grad_(True) for t in (x, prev, w)]
        ops.set_fast(True)
        try:
            xo_h, rgb_h, sc_h = blk(ins_h[0], ins_h[1], ins_h[2], nz.to(DEV))
            ((xo_h.float() * gx_o.to(DEV)).sum() + (rgb_h.float() * grgb.to(DEV)).sum()).backward()
        finally:
            ops.set_fast(False)
        errs = {"x": rel(xo_h, xo_r), "rgb": rel(rgb_h, rgb_r), "coords": rel(sc_h, sc_r)}
        tiles = {"x": tile_max(xo_h, xo_r), "rgb": tile_max(rgb_h, rgb_r)}
        for name, a, b in zip(("gx", "gprev", "gw"), ins_h, ins_r):
            errs[name] = rel(a.grad, b.grad)
            tiles[name] = tile_max(a.grad, b.grad)
        ref_grads = dict(ref.named_parameters())
        for n_, p_ in blk.named_parameters():
            if p_.grad is not None and n_ in ref_grads:
                errs["g:" + n_] = rel(p_.grad, ref_grads[n_].grad)
                tiles["g:" + n_] = tile_max(p_.grad, ref_grads[n_].grad)
        print("GeneratorBlock 64->32 @256 %s:" % prec, {k: "%.1e" % v for k, v in errs.items()})
        print("  worst block (RMS error / tensor RMS):", {k: "%.1e" % v for k, v in tiles.items()})
        bad = {k: v for k, v in errs.items() if v > bound(k)}
        assert not bad, ("GeneratorBlock 64->32 @256", prec, bad)
        bad = {k: v for k, v in tiles.items() if v > tile_bound(k)}
        assert not bad, ("GeneratorBlock 64->32 @256: a block of the tensor is off", prec, bad)
    finally:
        ops.set_precision("fp32")
        hb.pack_cache_clear()


def test_fused_loss_kernels_match_the_torch_compositions():
    """SURVEY K10 (csrc/losses.hip): hinge / generator hinge (reference :382-387), path lengths (:316), the classifier KL
    (:421-438 with KLDivLoss(batchmean, log_target) :406) and nn.L1Loss (:404-405) as one forward + one backward launch each,
    against the reference's torch composition in float64 on the CPU — values and every input gradient, fp32 tolerance
    2e-6 relative; ragged sizes, both operand layouts of the reconstruction L1 (NCHW real batch against the sliced
    channels-last generated batch), a bf16 operand, and STYLEX_FUSED_LOSSES semantics (the fused node is what ran)."""
    g = torch.Generator().manual_seed(31)
    tol = 2e-6

    def check(fused_fn, ref_fn, inputs, grads_for, what, tol=tol):
        xs = [t.clone().to(DEV).requires_grad_(i in grads_for) for i, t in enumerate(inputs)]
        out = fused_fn(*xs)
        assert type(out.grad_fn).__name__.endswith("Backward") and "_" in type(out.grad_fn).__name__, (what, out.grad_fn)
        w = torch.randn(out.shape, generator=g).to(DEV)
        (out * w).sum().backward()
        rs = [t.clone().double().requires_grad_(i in grads_for) for i, t in enumerate(inputs)]
        ref = ref_fn(*rs)
        (ref * w.double().cpu()).sum().backward()
        close(ref, out, tol, what + " value")
        for i in grads_for:
            assert xs[i].grad.shape == xs[i].shape and xs[i].grad.dtype == xs[i].dtype
            close(rs[i].grad, xs[i].grad, tol if xs[i].dtype == torch.float32 else 1e-2, "%s grad %d" % (what, i))

    for n in (1, 5, 32, 1000):
        real, fake = torch.randn(n, generator=g) * 2, torch.randn(n, generator=g) * 2
        real[0], fake[0] = -1.0, 1.0  # the kink: relu'(0) = 0
        check(ops.hinge_loss, lambda r, f: (F.relu(1 + r) + F.relu(1 - f)).mean(), (real, fake), (0, 1), "hinge n=%d" % n)
        check(ops.gen_hinge_loss, lambda f: f.mean(), (fake,), (0,), "gen hinge n=%d" % n)
    for b, l, d in ((1, 1, 1), (5, 7, 514), (32, 6, 514)):
        pg = torch.randn(b, l, d, generator=g) * 0.1
        check(ops.pl_lengths, lambda t: (t ** 2).sum(dim=2).mean(dim=1).sqrt(), (pg,), (0,), "pl_lengths %s" % ((b, l, d),))
    for b, k in ((1, 2), (32, 2), (33, 5), (300, 17)):
        r, f = torch.randn(b, k, generator=g) * 3, torch.randn(b, k, generator=g) * 3
        check(ops.kl_logits, lambda a, c: F.kl_div(F.log_softmax(c, dim=1), F.log_softmax(a, dim=1), reduction="batchmean",
                                                   log_target=True), (r, f), (0, 1), "kl %s" % ((b, k),), tol=5e-6)
    # L1: same layout, both gradients
    a, b_ = torch.randn(7, 512, generator=g), torch.randn(7, 512, generator=g)
    b_[0, :4] = a[0, :4]  # sign(0) = 0
    check(ops.l1_mean, lambda x, y: (x - y).abs().mean(), (a, b_), (0, 1), "l1 2-d")
    # the reconstruction term: NCHW real batch (no gradient) against the [:, :3] slice of a 4-channel channels-last batch
    real = torch.rand(3, 3, 20, 12, generator=g)
    gen4 = torch.randn(3, 4, 20, 12, generator=g)
    xr = real.to(DEV)
    x4 = gen4.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_()
    gen = x4[:, :3]  # the generator's fp32-mode output: a view with a 4-element pixel stride
    assert not gen.is_contiguous() and not gen.permute(0, 2, 3, 1).is_contiguous()
    out = ops.l1_mean(xr, gen)
    assert type(out.grad_fn).__name__ == "_L1MeanBackward"
    out.backward()
    r4 = gen4.clone().double().requires_grad_()
    ref = (real.double() - r4[:, :3]).abs().mean()
    ref.backward()
    close(ref, out, tol, "l1 strided value")
    close(r4.grad, x4.grad, tol, "l1 strided grad")
    # the bf16 mode's output: a dense channels-last copy of that slice
    x4b = gen4.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_()
    genb = x4b[:, :3].contiguous(memory_format=torch.channels_last)
    assert genb.permute(0, 2, 3, 1).is_contiguous() and not genb.is_contiguous()
    out = ops.l1_mean(xr, genb)
    assert type(out.grad_fn).__name__ == "_L1MeanBackward"
    out.backward()
    rb = gen4.clone().double().requires_grad_()
    refb = (real.double() - rb[:, :3]).abs().mean()
    refb.backward()
    close(refb, out, tol, "l1 channels-last value")
    close(rb.grad, x4b.grad, tol, "l1 channels-last grad")
    # gradient INTO the strided operand, and a bf16 operand
    ar = real.to(DEV).requires_grad_()
    out = ops.l1_mean(ar, gen.detach().contiguous(memory_format=torch.channels_last))
    out.backward()
    rr = real.clone().double().requires_grad_()
    (rr - gen4[:, :3].double()).abs().mean().backward()
    assert ar.grad.stride() == ar.stride()
    close(rr.grad, ar.grad, tol, "l1 grad of the strided operand")
    hb16 = torch.randn(4, 8, 6, 6, generator=g).bfloat16()
    check(ops.l1_mean, lambda x, y: (x - y).abs().mean(), (torch.randn(4, 8, 6, 6, generator=g), hb16.float()), (0, 1), "l1 f32")
    xb = hb16.to(DEV).requires_grad_()
    xa = torch.randn(4, 8, 6, 6, generator=g)
    out = ops.l1_mean(xa.to(DEV), xb)
    out.backward()
    close((xa.double() - hb16.double()).abs().mean(), out, tol, "l1 bf16 value")
    assert xb.grad.dtype == torch.bfloat16
    close(-torch.sign(xa.double() - hb16.double()) / xa.numel(), xb.grad, 1e-2, "l1 bf16 grad")
    # not expressible as a walk (broadcast operand): the torch composition answers
    out = ops.l1_mean(torch.zeros(1, 8, 1, 1, device=DEV).expand(4, 8, 6, 6), xb.detach().float())
    close(hb16.double().abs().mean(), out, 1e-6, "l1 broadcast fallback")
