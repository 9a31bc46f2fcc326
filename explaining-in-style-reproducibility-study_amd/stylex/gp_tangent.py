"""Gradient penalty of the discriminator's real branch without a double backward.

Reference: gradient_penalty (stylex/stylex_train.py:296-303) takes ||d D(x) / d x||_2 per sample with
create_graph=True and lets autograd differentiate that graph again.  D (DiscriminatorE, :857-909: DiscriminatorBlocks,
final conv, flatten, linear) is PIECEWISE LINEAR in x — convolutions, a fixed blur, LeakyReLU — so on the linear piece
that contains x

    D(x + eps v) = D(x) + eps L(v),      L = D with its biases removed and every LeakyReLU replaced by its gate
                                             (1 or 0.2, fixed by the primal pass),   u = dD/dx = L^T 1.

With the penalty P(u) and v = dP/du treated as a constant, dP/dtheta = d/dtheta <u(theta), v> = d/dtheta L(v): the
parameter gradient of the penalty is the FIRST-ORDER gradient of the gated-linear network applied to v (the gates do not
depend on theta almost everywhere — exactly what the double backward computes, LeakyReLU'' = 0).  And because the
backward chain through fixed gates is per-sample linear, the pre-activation gradients g_l of the unit chain (the one
that produced u) serve both terms:

    dLoss/dW_l = wgrad( s_b * x_l + t_l , g_l )        s_b = dLoss/dD(x_b) (hinge term),  t_l = activations of L(v)

Passes over D for the real branch of a penalty step: primal forward, unit data-gradient chain, tangent forward, ONE
weight-gradient pass — 4 instead of the 6 of the double backward (forward, data gradient, [forward + weight gradient]
of the second differentiation, [data + weight gradient] of the primal graph), all on the fused first-order kernels of
ops._DBlockFast (gate epilogues, bit masks) instead of the composable double-differentiable ops.

Used by Trainer._d_compute for the default architecture in the bf16 speed mode (STYLEX_GP_TANGENT=0 restores the double
backward; the fp32 parity mode keeps the reference's formulation).  tests/test_hip_parity.py::
test_gradient_penalty_tangent_pass_matches_double_backward holds it to the double backward.
"""
import math
import os

import torch

import hip_backend as hb
import ops


class _Ctx:
    """Stand-in for an autograd ctx so that _DBlockFast.forward / .backward can be driven by hand."""

    def __init__(self, needs):
        self.needs_input_grad = needs
        self.saved_tensors = ()
        self.keep_gz = True

    def save_for_backward(self, *t):
        self.saved_tensors = t


def supported(D, real):
    """The hand-driven pipeline covers the default discriminator: fused blocks, one logit, no conditioning."""
    import networks

    if os.environ.get("STYLEX_GP_TANGENT", "1") == "0" or not real.is_cuda or ops.impl() is not ops.HipOps:
        return False
    if ops.get_precision() != "bf16" and os.environ.get("STYLEX_GP_TANGENT", "1") != "2":  # 2: any precision (tests)
        return False
    if getattr(D, "conditional", False) or getattr(D, "encoder", False) or D.fc.out_features != 1:
        return False
    if torch.cuda.is_current_stream_capturing() or os.environ.get("STYLEX_DBLOCK", "1") == "0":
        return False
    return all(isinstance(b, networks.DiscriminatorBlock) and b.net[0].weight.shape[0] % 4 == 0 for b in D.blocks) and \
        all(a is None for a in D.attn_blocks) and all(q is None for q in D.quantize_blocks)


def _block_params(blk):
    down = blk.downsample[1] if blk.downsample is not None else None
    return (blk.conv_res.weight, blk.conv_res.bias, blk.net[0].weight, blk.net[0].bias, blk.net[2].weight, blk.net[2].bias,
            down.weight if down is not None else None, down.bias if down is not None else None)


def d_real_with_norms(D, real):
    """(D(real) [B], ||dD/dx||_2 per sample [B]) as ONE first-order autograd node over D's parameters."""
    params = []
    for blk in D.blocks:
        params += [p for p in _block_params(blk)]
    params += [D.final_conv.weight, D.final_conv.bias, D.fc.weight, D.fc.bias]
    layout = tuple(blk.downsample is not None for blk in D.blocks)
    # None placeholders (the last block has no down conv) cannot go through apply as tensors-with-grad; pass them as is
    out, norms = _DRealPenalty.apply(real, layout, *params)
    return out, norms


def _gated_conv3x3(t, w, gate, mask, prec):
    """conv3x3(t, w) * lrelu'(gate) with the gate in the kernel's store.  The gate epilogue lives on the data-gradient
    entry point; a forward conv IS the data gradient of the conv with the transposed, tap-mirrored weight
    (hb.pack_weight_fwd_as_dgrad builds that operand)."""
    wf = hb.pack_weight_fwd_as_dgrad(w, prec)
    b, _, h, wd = t.shape
    n, c = w.shape[0], w.shape[1]
    return hb.conv2d_bwd_data(t, None, (b, n, h, wd), 1, 1, prec, packed=wf, w_shape=(c, n, 3, 3), gate=gate, gate_mask=mask)


def _per_sample_channel_sums(g):
    """[B, C] fp32: sum over the pixels of each sample."""
    if ops._reducible(g.shape[1]) and hb.is_cl(g):
        return hb.act_bwd_reduce(g, None, False, 1.0, want_dx=False, want_sum=True, per_sample=True)[1]
    return g.float().sum(dim=(2, 3))


class _DRealPenalty(torch.autograd.Function):
    @staticmethod
    def forward(ctx, real, layout, *params):
        prec = ops._PRECISION
        nb = len(layout)
        blocks = [params[8 * i:8 * i + 8] for i in range(nb)]
        wf, bf, wfc, bfc = params[8 * nb:8 * nb + 4]
        # ---- pass 1: primal forward on the fused block kernels (what ops._DBlockFast.forward does under autograd)
        x = real.detach()
        ctxs = []
        for (w_res, b_res, w1, b1, w2, b2, w3, b3), down in zip(blocks, layout):
            c = _Ctx((True,) + (False,) * 9)
            x = ops._DBlockFast.forward(c, x, w_res, b_res, w1, b1, w2, b2, w3, b3, down)
            ctxs.append(c)
        xf = x
        yf = hb.conv2d_fwd(ops._cl(xf), wf, 1, 1, prec, bias=bf)
        bsz = yf.shape[0]
        flat = yf.reshape(bsz, -1).float()  # nn.Flatten of the logical NCHW tensor
        out = torch.addmm(bfc.float(), flat, wfc.float().t()).reshape(bsz)
        # ---- pass 2: the unit data-gradient chain  u_b = dD(x_b)/dx_b; its pre-activation gradients are kept
        g_yf = ops._cl(wfc.detach().float().reshape(1, *yf.shape[1:]).expand(bsz, -1, -1, -1).contiguous())
        g = hb.conv2d_bwd_data(g_yf, wf, tuple(xf.shape), 1, 1, prec)
        for c in reversed(ctxs):
            g = ops._DBlockFast.backward(c, g)[0]
        u = g.float().contiguous()
        norms = hb.rowwise_sumsq(u.reshape(bsz, -1)).sqrt()
        ctx.layout, ctx.ctxs, ctx.nb = layout, ctxs, nb
        ctx.save_for_backward(u, norms, xf, g_yf, flat, *params)
        return out, norms

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_out, g_norm):
        prec = ops._PRECISION
        u, norms, xf, g_yf, flat = ctx.saved_tensors[:5]
        params = ctx.saved_tensors[5:]
        nb, layout, ctxs = ctx.nb, ctx.layout, ctx.ctxs
        blocks = [params[8 * i:8 * i + 8] for i in range(nb)]
        wf, bf, wfc, bfc = params[8 * nb:8 * nb + 4]
        bsz = u.shape[0]
        s = g_out.detach().float().reshape(bsz)  # dLoss / dD(x_b)
        cinv = 1 / math.sqrt(2)
        # v = dLoss/du = g_norm_b * u_b / ||u_b||   (d||u|| / du = u / ||u||)
        # a sample whose input gradient is exactly zero: autograd's norm backward uses the subgradient 0 there (the
        # double backward this pass replaces), never 0 / 0
        gn = g_norm.detach().float().reshape(bsz)
        v = u * torch.where(norms > 0, gn / norms.clamp_min(1e-30), torch.zeros_like(gn)).view(bsz, 1, 1, 1)
        adt = hb.act_dtype(prec)
        s4 = s.view(bsz, 1, 1, 1).to(adt)

        def operand(x, t):
            """s_b * x + t: the one weight-gradient operand that carries the hinge and the penalty term."""
            return torch.addcmul(t, x, s4)

        def wsum(rows):
            """sum_b s_b * rows[b]  ([B, C] -> [C]) as an elementwise product + a fixed-order reduction: a library GEMV
            may pick an atomic split reduction for such skinny shapes (run-to-run noise, tests/test_hip_determinism_gpu)."""
            return (rows * s.view(bsz, 1)).sum(0)

        grads = []
        t = v
        # ---- pass 3 (tangent forward through the gated-linear network) and pass 4 (one weight gradient per conv)
        for (w_res, b_res, w1, b1, w2, b2, w3, b3), down, c in zip(blocks, layout, ctxs):
            x, xs, y1, y2, xb, _, _, _, _, m1, m2 = c.saved_tensors
            downsample, s2d, cin, _ = c.cfg
            gz3, gz2, gz1, alg = c.gz
            wsc = cinv if alg else 1.0
            if cin == 3:
                t = hb.pad_rgb8(t) if (prec == hb.BF16_ACT and x.shape[1] == 8) else ops._cl(ops._pad_rgb(ops._cl(t), w1)[0])
                extra = x.shape[1] - 3
                w1p = torch.cat([w1, w1.new_zeros(w1.shape[0], extra, 3, 3)], dim=1)
                wrp = torch.cat([w_res, w_res.new_zeros(w_res.shape[0], extra, 1, 1)], dim=1)
            else:
                t = ops._cl(t)
                w1p, wrp = w1, w_res
            if xs is None:
                xs = x
            ts = hb.subsample2_fwd(t) if downsample else t
            res_gemm = prec == hb.BF16_ACT and os.environ.get("STYLEX_RES_GEMM", "1") != "0"
            res_t = hb.conv1x1_gemm_fwd(ts, wrp, None) if res_gemm else hb.conv2d_fwd(ts, wrp, 1, 0, prec)
            a1 = _gated_conv3x3(t, w1p, y1, m1, prec)
            a2 = _gated_conv3x3(a1, w2, y2, m2, prec)
            n = w2.shape[0]
            if downsample:
                if s2d:
                    ab = hb.blur3x3_s2d_fwd(a2)
                    wf2, _ = hb.pack_weight_s2d(w3)
                    t_out = hb.conv2d_fwd(ab, None, 1, 1, prec, residual=res_t, res_scale=cinv, packed=wf2,
                                          w_shape=(w3.shape[0], 4 * n, 3, 3), s2d_c=n)
                else:
                    ab = hb.blur3x3_fwd(a2)
                    t_out = hb.conv2d_fwd(ab, w3, 2, 1, prec, residual=res_t, res_scale=cinv)
            else:
                ab = None
                t_out = (a2 + res_t) * cinv
            # weight gradients: operand s_b * (primal activation) + (tangent activation), gradient = unit chain's
            gw1 = hb.conv2d_bwd_weight(operand(x, t), gz1, tuple(w1p.shape), 1, 1, prec)
            gw2 = hb.conv2d_bwd_weight(operand(y1, a1), gz2, tuple(w2.shape), 1, 1, prec)
            # (the 1/sqrt(2) owed by the gradients computed from the unscaled gz3 rides their reduce launches)
            gw_res = hb.conv2d_bwd_weight(operand(xs, ts), gz3, tuple(wrp.shape), 1, 0, prec, out_scale=wsc)
            gw3 = gb3 = None
            if downsample:
                if s2d:
                    gw3 = hb.conv2d_bwd_weight_s2d(operand(xb, ab), gz3, tuple(w3.shape), prec, out_scale=wsc)
                else:
                    gw3 = hb.conv2d_bwd_weight(operand(xb, ab), gz3, tuple(w3.shape), 2, 1, prec, out_scale=wsc)
            if cin == 3:
                gw1, gw_res = gw1[:, :3].contiguous(), gw_res[:, :3].contiguous()
            # bias gradients: the hinge term only (the tangent network has no biases)
            gb1 = wsum(_per_sample_channel_sums(gz1))
            gb2 = wsum(_per_sample_channel_sums(gz2))
            gb_res = wsum(_per_sample_channel_sums(gz3)) * wsc
            if downsample:
                gb3 = gb_res
            grads += [gw_res, gb_res, gw1, gb1, gw2, gb2, gw3, gb3]
            t = t_out
        # final conv + linear
        tf_in = ops._cl(t)
        tyf = hb.conv2d_fwd(tf_in, wf, 1, 1, prec)
        tflat = tyf.reshape(bsz, -1).float()
        gwf = hb.conv2d_bwd_weight(operand(ops._cl(xf), tf_in), g_yf, tuple(wf.shape), 1, 1, prec)
        gbf = wsum(g_yf.float().sum(dim=(2, 3)))
        gwfc = (wsum(flat) + tflat.sum(0)).reshape(wfc.shape)
        gbfc = s.sum().reshape(bfc.shape)
        grads += [gwf, gbf, gwfc, gbfc]
        return (None, None) + tuple(grads)
