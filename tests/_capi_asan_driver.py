"""Driver for the ASan / UBSan build of the C-ABI host code (tests/test_capi_asan.py runs it in a subprocess with the
sanitizer runtime preloaded).  No torch, no GPU: buffers are host memory, every kernel launch fails with "no device" —
what runs is everything BEFORE a launch: argument validation, dispatch, eligibility checks of all launchers (dry runs),
tile / split-K / workspace planning, weight-gradient plans, timing bookkeeping.  Prints `asan driver ok <n calls>`."""
import ctypes
import itertools
import os
import sys

lib = ctypes.CDLL(os.environ["STYLEX_HIP_LIB"])
i64 = ctypes.c_int64
vp = ctypes.c_void_p


class Epi(ctypes.Structure):
    _fields_ = [("in_scale", vp), ("bias", vp), ("out_scale", vp), ("noise", vp), ("noise_stride", i64), ("noise_w", vp),
                ("noise_b", vp), ("residual", vp), ("res_scale", ctypes.c_float), ("s2d_c", ctypes.c_int32), ("mask", vp)]


def shape(*v):
    return (i64 * len(v))(*v)


def conv_shape(b, h, w, c, n, k, stride, pad):
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    return shape(b, h, w, c, n, k, k, stride, pad, ho, wo), ho, wo


lib.stylex_conv2d_workspace_bytes.restype = i64
lib.stylex_conv2d_bwd_weight_workspace_bytes.restype = i64
lib.stylex_version.restype = ctypes.c_char_p
calls = 0


def buf(nbytes):
    return (ctypes.c_char * max(16, int(nbytes) + 64))()


def ptr(b):
    return ctypes.cast(b, vp)


assert b"stylex" in lib.stylex_version()
lib.stylex_init(0)  # no device: returns an error code, must not crash
lib.stylex_timing_enable(1)

EPI = dict(BIAS=1, LRELU=2, OSCALE=4, NOISE=8, RESIDUAL=16, RELU=32, GATE=64, NOISE_NAT=128, MASK_OUT=256, GATE_MASK=512)
# the layer shapes of the step at 32 ... 256 px (planning only: no buffers are touched by these queries)
plan_shapes = []
for b in (1, 2, 32, 64, 128):
    for (h, c, n) in ((256, 3, 64), (256, 8, 64), (256, 64, 64), (256, 64, 32), (256, 32, 32), (128, 64, 128), (128, 128, 128),
                      (128, 128, 64), (64, 128, 256), (64, 256, 256), (32, 256, 512), (32, 512, 512), (16, 512, 512),
                      (8, 512, 512), (4, 512, 512), (2, 512, 512), (9, 6, 10), (16, 24, 16)):
        for (k, stride, pad) in ((3, 1, 1), (1, 1, 0), (3, 2, 1), (1, 2, 0)):
            if (h + 2 * pad - k) // stride + 1 < 1:
                continue
            plan_shapes.append(conv_shape(b, h, h, c, n, k, stride, pad)[0])
for sh in plan_shapes:
    for prec in (0, 1, 2):
        for which in (0, 1):
            assert lib.stylex_conv2d_workspace_bytes(sh, which, prec) >= 0
            for flags in (EPI["MASK_OUT"] | EPI["BIAS"] | EPI["LRELU"], EPI["GATE_MASK"], 0):
                lib.stylex_conv_mask_supported(sh, which, flags, prec)
                calls += 1
        assert lib.stylex_conv2d_bwd_weight_workspace_bytes(sh) >= 0
        calls += 3
# space-to-depth form (3x3 / s1 over 4C channels with s2d_c = C)
for b, h, c, n in ((2, 32, 64, 64), (64, 128, 64, 64), (64, 64, 128, 128), (64, 32, 256, 256), (64, 16, 512, 512)):
    sh = conv_shape(b, h, h, 4 * c, n, 3, 1, 1)[0]
    for prec in (1, 2):
        lib.stylex_conv2d_workspace_bytes(sh, 0, prec)
        lib.stylex_conv_mask_supported(sh, 0, EPI["MASK_OUT"], prec)
        calls += 2

# real calls on host buffers (small shapes): every launcher's applicability logic + the launch attempt itself
for (b, h, c, n, k, stride, pad), prec, flagset in itertools.product(
        ((1, 8, 16, 24, 3, 1, 1), (2, 16, 64, 64, 3, 1, 1), (1, 32, 64, 64, 3, 1, 1), (2, 16, 8, 64, 3, 1, 1), (1, 8, 64, 64, 3, 2, 1),
         (2, 8, 32, 8, 1, 1, 0), (1, 4, 512, 512, 3, 1, 1), (1, 16, 16, 16, 1, 2, 0), (1, 32, 128, 128, 3, 1, 1), (1, 128, 64, 64, 3, 1, 1)),
        (0, 1, 2), (0, EPI["BIAS"] | EPI["LRELU"], EPI["BIAS"] | EPI["OSCALE"] | EPI["NOISE"] | EPI["LRELU"])):
    sh, ho, wo = conv_shape(b, h, h, c, n, k, stride, pad)
    es = 2 if prec == 2 else 4
    x, y = buf(b * h * h * c * es), buf(b * ho * wo * n * es)
    w = buf(n * k * k * c * 4)
    wf, wb = buf(n * k * k * c * 4), buf(n * k * k * c * 4)
    lib.stylex_pack_weight(ptr(w), ptr(wf), ptr(wb), shape(n, c, k, k), prec, None)
    epi = Epi()
    sc_in, sc_out, bias = buf(b * c * 4), buf(b * n * 4), buf(n * 4)
    noise = buf(b * h * h * 4)
    epi.in_scale, epi.out_scale, epi.bias = ptr(sc_in), ptr(sc_out), ptr(bias)
    epi.noise, epi.noise_stride, epi.noise_w, epi.noise_b = ptr(noise), h, ptr(bias), ptr(bias)
    wsb = lib.stylex_conv2d_workspace_bytes(sh, 0, prec)
    ws = buf(wsb)
    lib.stylex_conv2d_fwd(ptr(x), ptr(wf), ptr(y), sh, flagset, ctypes.byref(epi), prec, ptr(ws) if wsb else None, wsb, None)
    lib.stylex_conv2d_fwd(ptr(x), ptr(wf), ptr(y), sh, 0, None, prec, None, 0, None)
    wsb = lib.stylex_conv2d_workspace_bytes(sh, 1, prec)
    ws = buf(wsb)
    lib.stylex_conv2d_bwd_data(ptr(y), ptr(wb), ptr(x), sh, 0, None, prec, ptr(ws) if wsb else None, wsb, None)
    epi2 = Epi()
    epi2.residual = ptr(x)
    epi2.res_scale = 0.2
    lib.stylex_conv2d_bwd_data(ptr(y), ptr(wb), ptr(x), sh, EPI["GATE"], ctypes.byref(epi2), prec, None, 0, None)
    wsb = lib.stylex_conv2d_bwd_weight_workspace_bytes(sh)
    ws, dw, db, wrote = buf(wsb), buf(n * c * k * k * 4), buf(n * 4), ctypes.c_int(0)
    lib.stylex_conv2d_bwd_weight(ptr(x), ptr(y), ptr(dw), ptr(ws), i64(wsb), sh, None, None, 0, prec, None)
    lib.stylex_conv2d_bwd_weight(ptr(x), ptr(y), ptr(dw), ptr(ws), i64(wsb), sh, ptr(sc_in), ptr(sc_out), 0, prec, None)
    lib.stylex_conv2d_bwd_weight_bias(ptr(x), ptr(y), ptr(dw), ptr(db), ctypes.byref(wrote), ptr(ws), i64(wsb), sh, None, None, 0,
                                      prec, None)
    lib.stylex_conv2d_bwd_weight_ex(ptr(x), ptr(y), ptr(dw), ptr(db), ctypes.byref(wrote), ptr(ws), i64(wsb), sh, None, None, 0,
                                    ctypes.c_float(0.5), 1, prec, None)
    calls += 10
# invalid arguments must be rejected, not dereferenced
bad = shape(0, -1, 8, 8, 8, 3, 3, 1, 1, 8, 8)
assert lib.stylex_conv2d_fwd(None, None, None, bad, 0, None, 2, None, i64(0), None) != 0
assert lib.stylex_conv2d_bwd_data(None, None, None, bad, 0, None, 2, None, i64(0), None) != 0
assert lib.stylex_conv2d_bwd_weight(None, None, None, None, i64(0), bad, None, None, 0, 2, None) != 0
assert lib.stylex_pack_weight(None, None, None, shape(0, 0, 3, 3), 2, None) != 0
calls += 4

# elementwise / reduction families: chunk planning + argument checks + launch attempt
for b, h, w_, c in ((1, 4, 4, 8), (2, 16, 16, 64), (3, 9, 7, 12), (2, 32, 32, 512), (128, 2, 2, 512)):
    sh4 = shape(b, h, w_, c)
    for adt in (0, 1):
        es = 2 if adt else 4
        lo, hi = buf(b * h * w_ * c * es), buf(b * 4 * h * w_ * c * es)
        lib.stylex_upsample2x_bilinear_fwd(ptr(lo), ptr(hi), sh4, adt, None)
        lib.stylex_upsample2x_bilinear_bwd(ptr(hi), ptr(lo), sh4, adt, None)
        lib.stylex_blur3x3_reflect_fwd(ptr(lo), ptr(hi), sh4, adt, None)
        lib.stylex_blur3x3_reflect_bwd(ptr(lo), ptr(hi), sh4, adt, None)
        lib.stylex_bias_act_bwd(ptr(lo), ptr(lo), ptr(hi), sh4, adt, None)
        nch = lib.stylex_reduce_chunks(sh4)
        assert nch >= 1
        part = buf(b * nch * 3 * c * 4)
        lib.stylex_act_bwd_reduce(ptr(lo), ptr(lo), ptr(hi), ptr(part), sh4, nch, 1, ctypes.c_float(1.0), adt, None)
        lib.stylex_scale_reduce(ptr(lo), ptr(lo), ptr(part), ptr(hi), ptr(part), sh4, nch, adt, None)
        if h % 2 == 0 and w_ % 2 == 0:
            lib.stylex_blur3x3_s2d_fwd(ptr(lo), ptr(hi), sh4, adt, None)
            lib.stylex_subsample2_fwd(ptr(lo), ptr(hi), sh4, adt, None)
            lib.stylex_add_at_even(ptr(lo), ptr(hi), sh4, adt, None)
        calls += 10
    lib.stylex_torgb_chunks(sh4)
n, c = 64, 32
lib.stylex_pack_weight_s2d(ptr(buf(n * c * 9 * 4)), ptr(buf(n * c * 36 * 2)), ptr(buf(n * c * 36 * 2)), shape(n, c, 3, 3), None)
lib.stylex_fold_weight_grad_s2d(ptr(buf(n * c * 36 * 4)), ptr(buf(n * c * 9 * 4)), shape(n, c, 3, 3), None)
lib.stylex_weight_sumsq(ptr(buf(n * c * 9 * 4)), ptr(buf(n * c * 4)), i64(n), i64(c), i64(9), None)

# K10 loss reductions: argument validation + walk geometry
f32 = lambda n: ptr(buf(4 * n))
for n in (1, 5, 32, 1000):
    for mode in (0, 1):
        lib.stylex_hinge_fwd(f32(n), f32(n), f32(1), i64(n), mode, None)
        lib.stylex_hinge_bwd(f32(n), f32(n), f32(1), f32(n), f32(n), i64(n), mode, None)
        calls += 2
assert lib.stylex_hinge_fwd(None, f32(4), f32(1), i64(4), 0, None) != 0 and lib.stylex_hinge_fwd(f32(4), f32(4), f32(1), i64(0), 0, None) != 0
assert lib.stylex_hinge_bwd(f32(4), f32(4), f32(1), None, None, i64(4), 0, None) != 0 and lib.stylex_hinge_fwd(f32(4), f32(4), f32(1), i64(4), 2, None) != 0
for b, l, d in ((1, 1, 1), (5, 7, 514)):
    lib.stylex_pl_lengths_fwd(f32(b * l * d), f32(b), shape(b, l, d), None)
    lib.stylex_pl_lengths_bwd(f32(b * l * d), f32(b), f32(b), f32(b * l * d), shape(b, l, d), None)
    lib.stylex_kl_logits_fwd(f32(b * l), f32(b * l), f32(1), shape(b, l), None)
    lib.stylex_kl_logits_bwd(f32(b * l), f32(b * l), f32(1), f32(b * l), None, shape(b, l), None)
    calls += 4
assert lib.stylex_pl_lengths_fwd(f32(4), f32(1), shape(0, 2, 2), None) != 0 and lib.stylex_kl_logits_fwd(f32(4), f32(4), f32(1), shape(2, 0), None) != 0
assert lib.stylex_kl_logits_bwd(f32(4), f32(4), f32(1), None, None, shape(2, 2), None) != 0
lib.stylex_l1_mean_chunks.restype = i64
for n in (1, 777, 3 * 3 * 20 * 12, 1 << 22):
    nch = lib.stylex_l1_mean_chunks(i64(n))
    assert 1 <= nch <= 1024
    lib.stylex_l1_mean_fwd(f32(n), f32(n), f32(nch), f32(1), i64(n), 0, 1, None, None, None, None)
    lib.stylex_l1_mean_bwd(f32(n), f32(n), f32(1), f32(n), None, i64(n), 0, 0, None, None, None, None)
    calls += 2
n = 3 * 3 * 20 * 12
sh4, st_a, st_b = shape(3, 20, 12, 3), shape(720, 12, 1, 240), shape(960, 48, 4, 1)
lib.stylex_l1_mean_fwd(f32(n), f32(n * 2), f32(8), f32(1), i64(n), 0, 0, sh4, st_a, st_b, None)
lib.stylex_l1_mean_bwd(f32(n), f32(n * 2), f32(1), f32(n), f32(n * 2), i64(n), 0, 0, sh4, st_a, None, None)
assert lib.stylex_l1_mean_fwd(f32(n), f32(n), f32(8), f32(1), i64(n), 0, 0, None, st_a, None, None) != 0  # strides without a shape
assert lib.stylex_l1_mean_fwd(f32(n), f32(n), f32(8), f32(1), i64(n + 1), 0, 0, sh4, st_a, None, None) != 0  # product != n
assert lib.stylex_l1_mean_fwd(f32(n), f32(n), f32(8), f32(1), i64(n), 0, 2, None, None, None, None) != 0  # dtype
assert lib.stylex_l1_mean_bwd(f32(n), f32(n), f32(1), None, None, i64(n), 0, 0, None, None, None, None) != 0
calls += 6

# timing bookkeeping (events cannot be created without a device: the report paths must still be sound)
lau, ms, fl, by = i64(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
for cls in (0, 1, 2, 7):
    lib.stylex_timing_report(cls, ctypes.byref(lau), ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by))
meta, vals = (i64 * (64 * 10))(), (ctypes.c_double * (64 * 3))()
lib.stylex_timing_layers(meta, vals, i64(64))
names, kmeta = ctypes.create_string_buffer(64 * 112), (i64 * (64 * 2))()
lib.stylex_timing_kernels(names, kmeta, vals, i64(64))
lib.stylex_timing_enable(0)
print("asan driver ok", calls, flush=True)
sys.exit(0)
