"""Kernel-level checks through the C ABI on the TEST-ONLY emulator (tests/emu): the MFMA GEMMs in their three
arithmetic modes (tsii_set_gemm_products: 0 = f32-input MFMA, 6 = split-bf16 fp32 class, 3 = split-bf16 2 planes)
-- fragment layouts, LDS swizzle, tile tails, row scales, LDS-staged epilogue -- against float64 numpy."""
import ctypes
import os

import numpy as np
import pytest

from text_segmentation_image_inpainting_amd import _lib


@pytest.fixture(scope="module")
def emu():
    from tests.emu import build_emu
    if not build_emu.available():
        pytest.skip("host clang++ not available")
    return _lib.bind(ctypes.CDLL(build_emu.build()))


def P(a):
    return None if a is None else ctypes.c_void_p(a.ctypes.data)


# Workspaces come with a canary tail right behind the size the library asked for; every test checks its tails on the way out (the
# emulator runs kernels as host code: an overrun lands in the numpy heap silently -- that is how an under-sized weight workspace
# once reached the chip before any test saw it).
_CANARY = np.float32(-12345.678)
_guarded = []


def WS(nbytes):
    n = (int(nbytes) + 3) // 4
    buf = np.zeros(n + 64, np.float32)
    buf[n:] = _CANARY
    _guarded.append((buf, n))
    return buf[:max(n, 1)]


@pytest.fixture(autouse=True)
def _check_workspace_tails():
    _guarded.clear()
    yield
    for buf, n in _guarded:
        assert np.all(buf[n:] == _CANARY), f"a kernel wrote past its {4 * n}-byte workspace"
    _guarded.clear()


# tolerance of a GEMM result relative to max|ref| per mode (K <= 192 here): fp32 class for 0 and 6
MODE_TOL = {0: 1e-5, 6: 1e-5, 3: 2e-4, 1: 2e-2}


@pytest.fixture(params=[6, 0, 3, 1])
def mode(request, emu):
    assert emu.tsii_set_gemm_products(request.param) == 0
    yield request.param
    emu.tsii_set_gemm_products(6)


@pytest.mark.parametrize("M,K,N,bias,masked", [(300, 64, 128, True, True), (257, 96, 192, True, False),
                                               (130, 40, 32, False, True), (70, 6, 5, True, True),
                                               (260, 128, 256, False, False), (200, 36, 136, True, True),
                                               (300, 192, 128, True, True), (210, 132, 256, False, True),
                                               # weight gradient on the 32 x 128 tiles (<= 32 output channels, wide reduction side)
                                               (300, 384, 32, True, True), (1000, 160, 16, False, True), (515, 132, 24, True, False)])
def test_pointwise_gemms(emu, mode, M, K, N, bias, masked):
    L = emu
    tol = MODE_TOL[mode]
    rng = np.random.default_rng(M + K + N)
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = rng.standard_normal((N, K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32) if bias else None
    split = (K // 2 // 4) * 4 if K >= 8 else K // 2
    r0 = (rng.uniform(size=M) > 0.3).astype(np.float32) if masked else None
    r1 = (rng.uniform(size=M) > 0.3).astype(np.float32) if masked else None
    denom = rng.integers(1, 9, size=M).astype(np.float32) if masked else None
    keep = (rng.uniform(size=M) > 0.2).astype(np.float32) if masked else None
    y = np.zeros((M, N), np.float32)
    wws = WS(L.tsii_pw_ws_bytes(N, K))
    assert L.tsii_pw_fwd(P(x), M, K, P(w), N, P(b), P(r0), split, P(r1), P(denom), P(keep), P(y), P(wws), wws.nbytes, None) == 0, L.tsii_last_error()
    xm = x.astype(np.float64).copy()
    if masked:
        xm[:, :split] *= r0[:, None]
        xm[:, split:] *= r1[:, None]
    ref = xm @ w.T.astype(np.float64)
    if masked:
        ref = ref / denom[:, None]
    if bias:
        ref = ref + b
    if masked:
        ref = ref * keep[:, None]
    assert np.abs(y - ref).max() <= tol * np.abs(ref).max()

    dy = rng.standard_normal((M, N)).astype(np.float32)
    inv = (keep / denom).astype(np.float32) if masked else None
    dx = np.zeros((M, K), np.float32)
    wt = WS(L.tsii_pw_ws_bytes(N, K))
    assert L.tsii_pw_bwd_dx(P(dy), M, N, P(w), K, P(inv), P(r0), split, P(r1), P(dx), P(wt), None) == 0, L.tsii_last_error()
    g = dy.astype(np.float64) * (inv[:, None] if masked else 1.0)
    rdx = g @ w.astype(np.float64)
    if masked:
        rdx[:, :split] *= r0[:, None]
        rdx[:, split:] *= r1[:, None]
    assert np.abs(dx - rdx).max() <= tol * np.abs(rdx).max()

    nbytes = L.tsii_pw_bwd_dw_ws_bytes(M, N, K)
    ws = WS(nbytes)
    dw = np.zeros((N, K), np.float32)
    db = np.zeros(N, np.float32)
    assert L.tsii_pw_bwd_dw(P(dy), P(x), M, N, K, P(inv), P(keep), P(r0), split, P(r1), P(dw), P(db), P(ws), nbytes, None) == 0, L.tsii_last_error()
    rdw = g.T @ xm
    assert np.abs(dw - rdw).max() <= tol * np.abs(rdw).max()
    rdb = (dy.astype(np.float64) * (keep[:, None] if masked else 1.0)).sum(0)
    assert np.abs(db - rdb).max() <= 1e-5 * np.abs(rdb).max()


@pytest.mark.parametrize("M,K,N,act,slope", [(300, 64, 128, 2, 0.3), (257, 192, 192, 1, 0.0), (140, 36, 40, 3, 0.0),
                                             (260, 128, 256, 0, 0.0), (300, 384, 32, 2, 0.3), (700, 160, 24, 3, 0.0)])
def test_pointwise_fused_batchnorm(emu, mode, M, K, N, act, slope):
    """K6b at kernel level: BatchNorm(+act) of the producer applied on operand load (forward and dW) and the
    statistics partials of the output (-> tsii_bn_finalize) against float64 numpy."""
    L = emu
    tol = MODE_TOL[mode]
    rng = np.random.default_rng(7 * M + K + N)
    xr = (rng.standard_normal((M, K)) * 2 + 0.5).astype(np.float32)      # raw conv output of the producer
    sc = rng.uniform(0.5, 1.5, K).astype(np.float32)
    sh = rng.standard_normal(K).astype(np.float32)
    w = rng.standard_normal((N, K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    r0 = (rng.uniform(size=M) > 0.3).astype(np.float32)
    denom = rng.integers(1, 9, size=M).astype(np.float32)
    keep = (rng.uniform(size=M) > 0.2).astype(np.float32)
    z = xr.astype(np.float64) * sc + sh
    a = {0: z, 1: np.maximum(z, 0), 2: np.where(z > 0, z, slope * z), 3: np.clip(z, 0, 6)}[act]
    am = a * r0[:, None]
    rows = L.tsii_pw_stat_rows(M)
    part = np.zeros((rows, 4, N), np.float32)
    y = np.zeros((M, N), np.float32)
    wws = WS(L.tsii_pw_ws_bytes(N, K))        # pre-split weights (None: split while staging)
    use_ws = (M + N) % 2 == 0
    assert L.tsii_pw_fwd_bn(P(xr), M, K, P(w), N, P(b), P(r0), K, None, P(denom), P(keep), P(sc), P(sh), act, slope,
                            P(part), P(y), P(wws) if use_ws else None, wws.nbytes if use_ws else 0, None) == 0, L.tsii_last_error()
    ref = (am @ w.T.astype(np.float64) / denom[:, None] + b) * keep[:, None]
    assert np.abs(y - ref).max() <= tol * np.abs(ref).max()
    # statistics of y from the partials
    mean, var = np.zeros(N, np.float32), np.zeros(N, np.float32)
    rm, rv = np.zeros(N, np.float32), np.ones(N, np.float32)
    gamma, beta = rng.uniform(0.5, 1.5, N).astype(np.float32), rng.standard_normal(N).astype(np.float32)
    scale, shift = np.zeros(N, np.float32), np.zeros(N, np.float32)
    nb = L.tsii_bn_finalize_ws_bytes(rows, N)
    ws = WS(nb)
    assert L.tsii_bn_finalize(P(part), rows, N, M, P(mean), P(var), P(rm), P(rv), 0.1, P(gamma), P(beta), 1e-5,
                              P(scale), P(shift), P(ws), nb, None) == 0, L.tsii_last_error()
    y64 = y.astype(np.float64)
    assert np.abs(mean - y64.mean(0)).max() <= 1e-6 * (np.abs(y64).max() + 1)
    assert np.abs(var - y64.var(0)).max() <= 4e-6 * y64.var(0).max()
    assert np.abs(rm - 0.1 * y64.mean(0)).max() <= 1e-6 * (np.abs(y64).max() + 1)
    assert np.abs(rv - (0.9 + 0.1 * y64.var(0, ddof=1))).max() <= 4e-6 * (1 + y64.var(0).max())
    rs = gamma / np.sqrt(y64.var(0) + 1e-5)
    assert np.abs(scale - rs).max() <= 1e-5 * np.abs(rs).max()
    assert np.abs(shift - (beta - y64.mean(0) * rs)).max() <= 1e-5 * (np.abs(beta).max() + np.abs(y64.mean(0) * rs).max())
    # dW with the same load-time transform
    dy = rng.standard_normal((M, N)).astype(np.float32)
    inv = (keep / denom).astype(np.float32)
    nbytes = L.tsii_pw_bwd_dw_ws_bytes(M, N, K)
    ws2 = WS(nbytes)
    dw = np.zeros((N, K), np.float32)
    db = np.zeros(N, np.float32)
    assert L.tsii_pw_bwd_dw_bn(P(dy), P(xr), M, N, K, P(inv), P(keep), P(r0), K, None, P(sc), P(sh), act, slope,
                               P(dw), P(db), P(ws2), nbytes, None) == 0, L.tsii_last_error()
    rdw = (dy.astype(np.float64) * inv[:, None]).T @ am
    assert np.abs(dw - rdw).max() <= tol * np.abs(rdw).max()


@pytest.mark.parametrize("k,cin,h,w", [(7, 3, 10, 14), (5, 3, 8, 8), (3, 4, 6, 10)])
def test_stem_space_to_depth_entry_points(emu, k, cin, h, w):
    """K4b at kernel level: tsii_stem_s2d / tsii_stem_w_fwd / tsii_stem_w_bwd against numpy, and the identity
    conv_s2(x*m, w) == conv_valid_s1(s2d(x*m), w2) they are built on."""
    L = emu
    rng = np.random.default_rng(k * 100 + cin)
    n, cout, pad = 2, 5, (k - 1) // 2
    x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
    m = (rng.uniform(size=(n, h, w, cin)) > 0.3).astype(np.float32)
    wt = rng.standard_normal((cout, cin, k, k)).astype(np.float32)
    h2, w2, ka = (h + 2 * pad) // 2, (w + 2 * pad) // 2, (k + 1) // 2
    x2 = np.zeros((n, h2, w2, 4 * cin), np.float32)
    assert L.tsii_stem_s2d(P(x), P(m), None, 0, None, n, h, w, cin, pad, P(x2), None) == 0, L.tsii_last_error()
    xp = np.pad(x * m, ((0, 0), (pad, pad), (pad, pad), (0, 0)))
    ref = np.zeros_like(x2)
    for py in range(2):
        for px in range(2):
            ref[..., (py * 2 + px) * cin:(py * 2 + px + 1) * cin] = xp[:, py::2, px::2, :]
    assert np.array_equal(x2, ref)
    wk = np.zeros((cout, 4 * cin, ka, ka), np.float32)
    assert L.tsii_stem_w_fwd(P(wt), cout, cin, k, P(wk), None) == 0, L.tsii_last_error()
    # direct stride-2 conv vs valid stride-1 conv of the rearranged operands (float64)
    ho, wo = (h + 2 * pad - k) // 2 + 1, (w + 2 * pad - k) // 2 + 1
    y_ref = np.zeros((n, ho, wo, cout))
    y_s2d = np.zeros((n, ho, wo, cout))
    for oy in range(ho):
        for ox in range(wo):
            patch = xp[:, 2 * oy:2 * oy + k, 2 * ox:2 * ox + k, :].astype(np.float64)           # [n,k,k,cin]
            y_ref[:, oy, ox, :] = np.einsum("nyxc,ocyx->no", patch, wt.astype(np.float64))
            p2 = ref[:, oy:oy + ka, ox:ox + ka, :].astype(np.float64)                             # [n,ka,ka,4cin]
            y_s2d[:, oy, ox, :] = np.einsum("nyxe,oeyx->no", p2, wk.astype(np.float64))
    assert np.abs(y_ref - y_s2d).max() <= 1e-12 * max(1.0, np.abs(y_ref).max())
    # the weight-gradient map is the adjoint selection
    dwk = rng.standard_normal(wk.shape).astype(np.float32)
    dw = np.zeros_like(wt)
    assert L.tsii_stem_w_bwd(P(dwk), cout, cin, k, P(dw), None) == 0, L.tsii_last_error()
    for ky in range(k):
        for kx in range(k):
            e0 = ((ky & 1) * 2 + (kx & 1)) * cin
            assert np.array_equal(dw[:, :, ky, kx], dwk[:, e0:e0 + cin, ky // 2, kx // 2])


def _conv_ref(x, w, s, p, d):
    """float64 NHWC convolution: x [n,h,w,ci], w [co,ci,kh,kw] -> [n,ho,wo,co]"""
    n, h, wd, ci = x.shape
    co, _, kh, kw = w.shape
    ho = (h + 2 * p - d * (kh - 1) - 1) // s + 1
    wo = (wd + 2 * p - d * (kw - 1) - 1) // s + 1
    xp = np.zeros((n, h + 2 * p, wd + 2 * p, ci))
    xp[:, p:p + h, p:p + wd] = x
    y = np.zeros((n, ho, wo, co))
    for ky in range(kh):
        for kx in range(kw):
            patch = xp[:, ky * d:ky * d + (ho - 1) * s + 1:s, kx * d:kx * d + (wo - 1) * s + 1:s]
            y += patch @ w[:, :, ky, kx].T.astype(np.float64)
    return y


@pytest.mark.parametrize("cin,cout,k,s,p,d,h", [(16, 32, 3, 1, 1, 1, 11), (8, 136, 3, 2, 1, 1, 13), (24, 16, 3, 1, 2, 2, 9),
                                                (16, 40, 5, 2, 2, 1, 12), (16, 136, 3, 1, 1, 1, 9), (8, 256, 3, 1, 1, 1, 7),
                                                (128, 16, 3, 1, 1, 1, 6),      # 128-wide tiles of the NT / TN forms, dX with N = 128
                                                (12, 64, 4, 1, 0, 1, 11), (20, 16, 2, 1, 1, 1, 8)])   # c % 4 only: half-chunks straddle taps (s2d stems)
def test_dense_conv_gather_gemms(emu, mode, cin, cout, k, s, p, d, h):
    """K4 entry points on the gather forms of the GEMM kernels (split-bf16 loaders in modes 6 / 3 / 1, f32 MFMA in mode 0):
    forward with the x*mask planes and the count division, dX (stride phases for s = 2), dW -- against float64 numpy."""
    L = emu
    tol = MODE_TOL[mode]
    rng = np.random.default_rng(cin + cout + k + h)
    n = 2
    x = rng.standard_normal((n, h, h, cin)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, k, k)) * 0.2).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    r0 = (rng.uniform(size=(n, h, h)) > 0.25).astype(np.float32)
    r1 = (rng.uniform(size=(n, h, h)) > 0.25).astype(np.float32)
    split = (cin // 2 // 4) * 4
    ho = (h + 2 * p - d * (k - 1) - 1) // s + 1
    denom = rng.integers(1, 9, size=(n, ho, ho)).astype(np.float32)
    keep = (rng.uniform(size=(n, ho, ho)) > 0.2).astype(np.float32)
    geom = (n, h, h, cin, cout, k, k, s, s, p, p, d, d, ho, ho)
    nb = L.tsii_dense_ws_bytes(cin, cout, k, k)
    ws = WS(nb)
    y = np.zeros((n, ho, ho, cout), np.float32)
    assert L.tsii_dense_fwd(P(x), None, P(r0), split, P(r1), P(w), P(b), P(denom), P(keep), *geom, P(y), P(ws), nb, None) == 0, L.tsii_last_error()
    xm = x.astype(np.float64).copy()
    xm[..., :split] *= r0[..., None]
    xm[..., split:] *= r1[..., None]
    ref = (_conv_ref(xm, w, s, p, d) / denom[..., None] + b) * keep[..., None]
    assert np.abs(y - ref).max() <= tol * np.abs(ref).max()
    if mode == 1:       # the gather form really runs on the bf16 loaders (an f32-MFMA fallback would be ~1e-7 here)
        assert np.abs(y - ref).max() >= 1e-5 * np.abs(ref).max()

    dy = rng.standard_normal((n, ho, ho, cout)).astype(np.float32)
    inv = (keep / denom).astype(np.float32)
    g = dy.astype(np.float64) * inv[..., None]
    # dW[co,ci,ky,kx] = sum g[n,oy,ox,co] * xm[n, oy*s - p + ky*d, ox*s - p + kx*d, ci]
    xp = np.zeros((n, h + 2 * p, h + 2 * p, cin))
    xp[:, p:p + h, p:p + h] = xm
    rdw = np.zeros((cout, cin, k, k))
    for ky in range(k):
        for kx in range(k):
            patch = xp[:, ky * d:ky * d + (ho - 1) * s + 1:s, kx * d:kx * d + (ho - 1) * s + 1:s]
            rdw[:, :, ky, kx] = np.einsum("nyxo,nyxi->oi", g, patch)
    nbw = L.tsii_dense_bwd_dw_ws_bytes(n, ho, ho, cin, cout, k, k)
    wsw = WS(nbw)
    dw = np.zeros_like(w)
    db = np.zeros(cout, np.float32)
    assert L.tsii_dense_bwd_dw(P(dy), P(inv), P(keep), P(x), None, P(r0), split, P(r1), *geom, P(dw), P(db), P(wsw), nbw, None) == 0, L.tsii_last_error()
    assert np.abs(dw - rdw).max() <= tol * np.abs(rdw).max()
    rdb = (dy.astype(np.float64) * keep[..., None]).sum((0, 1, 2))
    assert np.abs(db - rdb).max() <= 1e-5 * np.abs(rdb).max()

    # dX = (transposed conv of g) * mask planes
    rdx = np.zeros((n, h + 2 * p, h + 2 * p, cin))
    for ky in range(k):
        for kx in range(k):
            rdx[:, ky * d:ky * d + (ho - 1) * s + 1:s, kx * d:kx * d + (ho - 1) * s + 1:s] += g @ w[:, :, ky, kx].astype(np.float64)
    rdx = rdx[:, p:p + h, p:p + h]
    rdx[..., :split] *= r0[..., None]
    rdx[..., split:] *= r1[..., None]
    dx = np.zeros_like(x)
    ws2 = WS(nb)
    assert L.tsii_dense_bwd_dx(P(dy), P(inv), P(w), None, P(r0), split, P(r1), *geom, P(dx), P(ws2), nb, None) == 0, L.tsii_last_error()
    assert np.abs(dx - rdx).max() <= tol * np.abs(rdx).max()


PC_THREADS = 768      # threads per block of the persistent producer / consumer NT kernel (csrc/gemm_pc.hip)


def _set_pc(emu, opt, cus):
    emu.tsii_emu_set_pc_opt(opt)
    emu.tsii_emu_set_pc_cus(cus)


@pytest.fixture(params=[(16, 8), (0, 0)], ids=["xcd-dealt", "contiguous"])
def pc_opt(request, emu):
    """The two tile orders of the persistent kernel (csrc/gemm_pc.hip `opt` bit 4): dealt round robin with the blocks of an XCD on
    consecutive tiles (the stock library's PC_OPT_DEFAULT; 8 emulated CUs, so the XCD remap of the block index is active) and the
    contiguous ranges of rounds 3-5 (3 CUs) -- through emulator-only hooks."""
    _set_pc(emu, *request.param)
    yield request.param[0]
    _set_pc(emu, -1, 0)


@pytest.fixture(params=[(16, 8)], ids=["xcd-dealt"])
def pc_opt2(request, emu):
    """the stock tile order on 8 emulated CUs (XCD remap active), for the tests that are parametrised widely already"""
    _set_pc(emu, *request.param)
    yield request.param[0]
    _set_pc(emu, -1, 0)


@pytest.mark.parametrize("M,K,N", [(768, 64, 256), (512, 40, 128), (512, 96, 384), (256, 32, 128), (1152, 160, 192), (1664, 72, 224), (2560, 128, 128),
                                   (1280, 32, 256), (640, 224, 512), (512, 256, 32), (512, 256, 48), (3072, 64, 384)])
def test_producer_consumer_gemm(emu, pc_opt, M, K, N):
    """K3p (gemm_pc.hip) at kernel level on the emulator (3 'CUs': several tiles per persistent block, more stages than
    LDS slots, k tails, column blocks past N, both tile shapes): forward with BatchNorm-on-load + statistics, dX with
    and without the K6c BatchNorm-backward reductions, against float64 numpy -- and the 768-thread kernel must be the
    one that ran (full tiles only: other shapes stay on the 4-wave kernels, test_pointwise_gemms)."""
    L = emu
    assert L.tsii_set_gemm_products(6) == 0
    lib = L._lib if hasattr(L, "_lib") else None
    rng = np.random.default_rng(M + 3 * K + N)
    from tests.emu import build_emu
    raw = ctypes.CDLL(build_emu.build())
    raw.hipemu_launches.restype = ctypes.c_long
    before = raw.hipemu_launches(PC_THREADS)
    x = (rng.standard_normal((M, K)) * 2 + 0.5).astype(np.float32)
    w = rng.standard_normal((N, K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, K).astype(np.float32)
    sh = rng.standard_normal(K).astype(np.float32)
    r0 = (rng.uniform(size=M) > 0.3).astype(np.float32)
    r1 = (rng.uniform(size=M) > 0.3).astype(np.float32)
    split = (K // 2 // 8) * 8          # the two mask planes meet on an 8-channel boundary (else the 4-wave kernel keeps the layer)
    denom = rng.integers(1, 9, size=M).astype(np.float32)
    keep = (rng.uniform(size=M) > 0.2).astype(np.float32)
    z = x.astype(np.float64) * sc + sh
    a = np.where(z > 0, z, 0.3 * z)
    am = a.copy()
    am[:, :split] *= r0[:, None]
    am[:, split:] *= r1[:, None]
    rows = L.tsii_pw_stat_rows(M)
    part = np.zeros((rows, 4, N), np.float32)
    y = np.zeros((M, N), np.float32)
    wws = WS(L.tsii_pw_ws_bytes(N, K))
    assert L.tsii_pw_fwd_bn(P(x), M, K, P(w), N, P(b), P(r0), split, P(r1), P(denom), P(keep), P(sc), P(sh), 2, 0.3,
                            P(part), P(y), P(wws), wws.nbytes, None) == 0, L.tsii_last_error()
    ref = (am @ w.T.astype(np.float64) / denom[:, None] + b) * keep[:, None]
    assert np.abs(y - ref).max() <= 1e-5 * np.abs(ref).max()
    assert raw.hipemu_launches(PC_THREADS) == before + (1 if N >= 128 else 0)      # (512, 256, 32): forward on the 4-wave kernel, dX + K6c
                                                                                   # = one-stage tiles of the persistent one
    # the statistics partials: per 128-row block (count, pivot, sum(y - pivot), sum((y - pivot)^2))
    y64 = y.astype(np.float64)
    for rb in range(rows):
        blk = y64[rb * 128:(rb + 1) * 128]
        assert np.all(part[rb, 0] == blk.shape[0])
        d = blk - part[rb, 1].astype(np.float64)
        assert np.abs(part[rb, 2] - d.sum(0)).max() <= 1e-5 * (np.abs(d).sum(0).max() + 1e-30)
        assert np.abs(part[rb, 3] - (d * d).sum(0)).max() <= 1e-5 * (d * d).sum(0).max()
    # plain forward without any epilogue side input
    y2 = np.zeros((M, N), np.float32)
    assert L.tsii_pw_fwd(P(x), M, K, P(w), N, None, None, 0, None, None, None, P(y2), P(wws), wws.nbytes, None) == 0, L.tsii_last_error()
    ref2 = x.astype(np.float64) @ w.T.astype(np.float64)
    assert np.abs(y2 - ref2).max() <= 1e-5 * np.abs(ref2).max()
    # dX, plain and with the BatchNorm-backward reductions of the tensor it feeds (x is that BatchNorm's raw input)
    dy = rng.standard_normal((M, N)).astype(np.float32)
    inv = (keep / denom).astype(np.float32)
    g = dy.astype(np.float64) * inv[:, None]
    rdx = g @ w.astype(np.float64)
    rdx[:, :split] *= r0[:, None]
    rdx[:, split:] *= r1[:, None]
    dx = np.zeros((M, K), np.float32)
    wt = WS(L.tsii_pw_ws_bytes(N, K))
    # dX runs on W^T: its tiled planes pad the reduction (N here) to 32 -- (512, 256, 48) is the shape that once overran this buffer
    assert wt.nbytes >= 3 * K * ((N + 31) // 32 * 32) * 2
    before = raw.hipemu_launches(PC_THREADS)
    assert L.tsii_pw_bwd_dx(P(dy), M, N, P(w), K, P(inv), P(r0), split, P(r1), P(dx), P(wt), None) == 0, L.tsii_last_error()
    pc_dx = raw.hipemu_launches(PC_THREADS) - before          # dX has K output columns: whole 32-blocks, at least 128
    assert pc_dx == (1 if (K >= 128 and K % 32 == 0) else 0)
    assert np.abs(dx - rdx).max() <= 1e-5 * np.abs(rdx).max()
    if K % 4 == 0:
        mean = x.astype(np.float64).mean(0).astype(np.float32)
        var = x.astype(np.float64).var(0).astype(np.float32)
        gamma = rng.uniform(0.5, 1.5, K).astype(np.float32)
        beta = rng.standard_normal(K).astype(np.float32)
        bpart = np.zeros((rows, 2, K), np.float32)
        dx2 = np.zeros((M, K), np.float32)
        assert L.tsii_pw_bwd_dx_bn(P(dy), M, N, P(w), K, P(inv), P(r0), split, P(r1), P(x), P(mean), P(var), P(gamma), P(beta), 1e-5, 2, 0.3,
                                   P(dx2), P(bpart), P(wt), None) == 0, L.tsii_last_error()
        assert np.abs(dx2 - rdx).max() <= 1e-5 * np.abs(rdx).max()
        xh = (x.astype(np.float64) - mean) / np.sqrt(var.astype(np.float64) + 1e-5)
        zz = xh * gamma + beta
        dz = dx2.astype(np.float64) * np.where(zz > 0, 1.0, 0.3)
        s1 = bpart[:, 0].astype(np.float64).sum(0)
        s2 = bpart[:, 1].astype(np.float64).sum(0)
        assert np.abs(s1 - dz.sum(0)).max() <= 2e-5 * np.abs(dz).sum(0).max()
        assert np.abs(s2 - (dz * xh).sum(0)).max() <= 2e-5 * np.abs(dz * xh).sum(0).max()


@pytest.mark.parametrize("opt,cus", [(16 + 32, 8), (16 + 64, 3), (32, 0)], ids=["xcd-dealt-lag1", "dealt-lag2", "lag1"])
@pytest.mark.parametrize("M,K,N", [(768, 64, 256), (1152, 160, 192), (3072, 64, 384)])
def test_producer_consumer_gemm_consumer_lag(emu, opt, cus, M, K, N):
    """`opt` bits 5-6: the second consumer wave of every SIMD starts 1 / 2 stages behind the first (an A/B form, measured without
    effect on the chip): same results, no deadlock with 2-5 LDS stages, with and without the XCD dealing."""
    _set_pc(emu, opt, cus)
    try:
        L = emu
        assert L.tsii_set_gemm_products(6) == 0
        rng = np.random.default_rng(M + K + N + opt)
        x = rng.standard_normal((M, K)).astype(np.float32)
        w = rng.standard_normal((N, K)).astype(np.float32)
        sc = rng.uniform(0.5, 1.5, K).astype(np.float32); sh = rng.standard_normal(K).astype(np.float32)
        wws = WS(L.tsii_pw_ws_bytes(N, K))
        y = np.zeros((M, N), np.float32)
        part = np.zeros((L.tsii_pw_stat_rows(M), 4, N), np.float32)
        assert L.tsii_pw_fwd_bn(P(x), M, K, P(w), N, None, None, 0, None, None, None, P(sc), P(sh), 2, 0.3, P(part), P(y), P(wws), wws.nbytes, None) == 0, L.tsii_last_error()
        z = x.astype(np.float64) * sc + sh
        ref = np.where(z > 0, z, 0.3 * z) @ w.T.astype(np.float64)
        assert np.abs(y - ref).max() <= 1e-5 * np.abs(ref).max()
        dy = rng.standard_normal((M, N)).astype(np.float32)
        dx = np.zeros((M, K), np.float32)
        wt = WS(L.tsii_pw_ws_bytes(N, K))
        assert L.tsii_pw_bwd_dx(P(dy), M, N, P(w), K, None, None, 0, None, P(dx), P(wt), None) == 0, L.tsii_last_error()
        rdx = dy.astype(np.float64) @ w.astype(np.float64)
        assert np.abs(dx - rdx).max() <= 1e-5 * np.abs(rdx).max()
    finally:
        _set_pc(emu, -1, 0)


def test_split_modes_non_finite_operands(emu):
    """Range caveat of the split-bf16 arithmetic (include/tsii_hip.h): an operand that is inf, or finite but beyond the largest
    bf16, splits into (inf, NaN, NaN), so the affected output row is NaN where the f32 MFMA mode gives +-inf (or NaN from
    inf * 0).  Either way the row is lost and every OTHER row is untouched -- that is what this pins."""
    L = emu
    rng = np.random.default_rng(77)
    M, K, N = 256, 64, 128
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = rng.standard_normal((N, K)).astype(np.float32)
    x[5, 3] = np.inf
    x[9, 7] = 3.4e38                  # finite in fp32, rounds to inf in bf16
    ref = np.delete(x, [5, 9], axis=0).astype(np.float64) @ w.T.astype(np.float64)
    try:
        for mode in (0, 6):
            assert L.tsii_set_gemm_products(mode) == 0
            y = np.zeros((M, N), np.float32)
            wws = WS(L.tsii_pw_ws_bytes(N, K))
            assert L.tsii_pw_fwd(P(x), M, K, P(w), N, None, None, 0, None, None, None, P(y), P(wws), wws.nbytes, None) == 0, L.tsii_last_error()
            assert not np.isfinite(y[5]).any(), mode                         # the row holding inf: inf or NaN everywhere
            if mode == 6:
                assert not np.isfinite(y[9]).any()                           # beyond the bf16 range: lost in the split modes only
            else:
                assert np.isfinite(y[9]).all() or np.isinf(y[9]).any()
            rest = np.delete(y, [5, 9], axis=0)
            assert np.isfinite(rest).all()
            assert np.abs(rest - ref).max() <= 1e-5 * np.abs(ref).max()
    finally:
        L.tsii_set_gemm_products(6)


# ---- depth-wise 3x3 marching strips through the C ABI (K2; the stride-1 / dilation-1 geometry runs csrc/dw_lean.h) -------------
def _dw_ref_fwd(x, rmask, w, bias, denom, keep, s, d, p):
    """float64: y = keep ? (sum_t w[t] * (x*rmask)[i_t]) / denom + bias : 0 on NHWC arrays."""
    n, h, wd, c = x.shape
    xm = x.astype(np.float64) * (1.0 if rmask is None else rmask.astype(np.float64)[..., None])
    ho, wo = (h + 2 * p - 2 * d - 1) // s + 1, (wd + 2 * p - 2 * d - 1) // s + 1
    xp = np.zeros((n, h + 2 * p, wd + 2 * p, c))
    xp[:, p:p + h, p:p + wd] = xm
    y = np.zeros((n, ho, wo, c))
    for ky in range(3):
        for kx in range(3):
            y += xp[:, ky * d:ky * d + (ho - 1) * s + 1:s, kx * d:kx * d + (wo - 1) * s + 1:s] * w[:, 0, ky, kx].astype(np.float64)
    if denom is not None:
        y = y / denom.astype(np.float64)[..., None]
    if bias is not None:
        y = y + bias.astype(np.float64)
    if keep is not None:
        y = np.where(keep[..., None] != 0, y, 0.0)
    return y


def _act(z, act, slope):
    if act == 1:
        return np.maximum(z, 0)
    if act == 2:
        return np.where(z > 0, z, z * slope)
    if act == 3:
        return np.clip(z, 0, 6)
    return z


@pytest.mark.parametrize("n,h,wd,c,masked,bias,act", [
    (2, 21, 37, 40, True, True, 2),      # row / column / channel tails, one chunk
    (1, 8, 16, 32, True, False, 3),      # exactly one step, one strip, ReLU6 (finite upper clamp)
    (1, 3, 5, 4, False, True, 0),        # smaller than a step and a strip, no mask planes
    (3, 43, 18, 36, True, True, 1),      # several steps per chunk, strip tail of 2 columns, channel tail
    (1, 70, 33, 8, False, False, 2),     # many steps: both LDS buffers in turn, no mask planes
])
@pytest.mark.parametrize("target", [1536, 1])
def test_depthwise_strip_entry_points(emu, n, h, wd, c, masked, bias, act, target):
    """target = 1: one chunk per image (emulator-only hook), i.e. chunks of several marching steps -- with the product's block
    target small tensors are cut into one-step chunks and the step-to-step hand-over would never run on the CPU."""
    L = emu
    L.tsii_emu_set_strip_target(target)
    try:
        _strip_entry_points(L, n, h, wd, c, masked, bias, act)
    finally:
        L.tsii_emu_set_strip_target(0)


def _strip_entry_points(L, n, h, wd, c, masked, bias, act, d=1, p=None):
    """3x3 / stride 1 / dilation d / padding p (default d: a "same" convolution): input grid [h, wd], output grid [ho, wo]"""
    p = d if p is None else p
    ho, wo = h + 2 * p - 2 * d, wd + 2 * p - 2 * d
    assert ho > 0 and wo > 0
    rng = np.random.default_rng(h * 100 + wd + 7 * d + p)
    slope = 0.3
    x = rng.standard_normal((n, h, wd, c)).astype(np.float32) + 0.5
    w = rng.standard_normal((c, 1, 3, 3)).astype(np.float32)
    b = rng.standard_normal(c).astype(np.float32) if bias else None
    rmask = (rng.uniform(size=(n, h, wd)) > 0.2).astype(np.float32) if masked else None
    cnt = None
    if masked:      # valid-input count of every output pixel: the same taps over the mask plane
        cnt = _dw_ref_fwd(rmask[..., None], None, np.ones((1, 1, 3, 3), np.float32), None, None, None, 1, d, p)[..., 0]
    keep = (cnt > 0).astype(np.float32) if masked else None
    denom = (np.where(cnt > 0, cnt, 1.0) * c).astype(np.float32) if masked else None
    geom = (3, 3, 1, 1, p, p, d, d)
    ws = WS(4 * (9 * c + 16))
    # plain forward
    y = np.full((n, ho, wo, c), np.nan, np.float32)
    assert L.tsii_dw_fwd(P(x), P(rmask), P(w), P(b), P(denom), P(keep), n, h, wd, c, *geom, ho, wo, P(y), P(ws), None) == 0, L.tsii_last_error()
    yr = _dw_ref_fwd(x, rmask, w, b, denom, keep, 1, d, p)
    assert np.abs(y - yr).max() <= 2e-6 * max(1.0, np.abs(yr).max())
    # forward with BatchNorm + activation on load and statistics partials
    sc = (rng.uniform(size=c) + 0.5).astype(np.float32); sh = rng.standard_normal(c).astype(np.float32)
    rows = L.tsii_dw_stat_rows(n, ho, wo, c, 3, 3, 1, 1, d, d)
    assert rows > 0
    part = WS(4 * rows * 4 * c)
    part[:] = np.nan
    y2 = np.full((n, ho, wo, c), np.nan, np.float32)
    assert L.tsii_dw_fwd_bn(P(x), P(rmask), P(w), P(b), P(denom), P(keep), n, h, wd, c, *geom, ho, wo, P(sc), P(sh), act, slope,
                            P(part), P(y2), P(ws), None) == 0, L.tsii_last_error()
    xa = _act(x.astype(np.float64) * sc + sh, act, slope)
    y2r = _dw_ref_fwd(xa, rmask, w, b, denom, keep, 1, d, p)
    assert np.abs(y2 - y2r).max() <= 2e-6 * max(1.0, np.abs(y2r).max())
    pr = part[:rows * 4 * c].reshape(rows, 4, c).astype(np.float64)
    assert np.isfinite(pr).all(), "every partial row the caller sized the buffer for is written"
    cnts, piv, s1, s2 = pr[:, 0], pr[:, 1], pr[:, 2], pr[:, 3]
    m_tot = n * ho * wo
    assert np.all(cnts.sum(0) == m_tot)
    mean = (cnts * piv + s1).sum(0) / m_tot
    ex2 = (s2 + 2 * piv * s1 + cnts * piv * piv).sum(0) / m_tot
    y2d = y2.astype(np.float64).reshape(-1, c)
    assert np.abs(mean - y2d.mean(0)).max() <= 1e-5 * max(1.0, np.abs(y2d).max())
    assert np.abs(ex2 - (y2d ** 2).mean(0)).max() <= 1e-5 * max(1.0, (y2d ** 2).max())
    # statistics only (no producer BatchNorm): the identity transform must be exact
    y3 = np.full((n, ho, wo, c), np.nan, np.float32)
    assert L.tsii_dw_fwd_bn(P(x), P(rmask), P(w), P(b), P(denom), P(keep), n, h, wd, c, *geom, ho, wo, None, None, 0, 0.0,
                            P(part), P(y3), P(ws), None) == 0, L.tsii_last_error()
    assert np.array_equal(y3, y)
    # dX (flipped taps, dy * inv staged, rmask applied to the result) and its K6c form
    dy = rng.standard_normal((n, ho, wo, c)).astype(np.float32)
    inv = (keep / denom).astype(np.float32) if masked else None
    dx = np.full((n, h, wd, c), np.nan, np.float32)
    assert L.tsii_dw_bwd_dx(P(dy), P(inv), P(w), P(rmask), n, h, wd, c, *geom, ho, wo, P(dx), P(ws), None) == 0, L.tsii_last_error()
    wf = w[:, :, ::-1, ::-1]
    dxr = _dw_ref_fwd(dy, inv, wf, None, None, None, 1, d, 2 * d - p)
    if masked:
        dxr = dxr * rmask.astype(np.float64)[..., None]
    assert np.abs(dx - dxr).max() <= 2e-6 * max(1.0, np.abs(dxr).max())
    brows = L.tsii_dw_bwd_stat_rows(n, h, wd, c, *geom)
    assert brows > 0
    bpart = WS(4 * brows * 2 * c)
    bpart[:] = np.nan
    mean_b = rng.standard_normal(c).astype(np.float32); var_b = (rng.uniform(size=c) + 0.5).astype(np.float32)
    gam = (rng.uniform(size=c) + 0.5).astype(np.float32); bet = rng.standard_normal(c).astype(np.float32)
    dx2 = np.full((n, h, wd, c), np.nan, np.float32)
    assert L.tsii_dw_bwd_dx_bn(P(dy), P(inv), P(w), P(rmask), n, h, wd, c, *geom, ho, wo, P(x), P(mean_b), P(var_b), P(gam), P(bet),
                               1e-5, act, slope, P(dx2), P(bpart), P(ws), None) == 0, L.tsii_last_error()
    assert np.array_equal(dx2, dx)
    assert np.isfinite(bpart[:brows * 2 * c]).all(), "every K6c partial row is written"
    xh = (x.astype(np.float64) - mean_b) / np.sqrt(var_b.astype(np.float64) + 1e-5)
    z = xh * gam + bet
    g1 = {0: np.ones_like(z), 1: (z > 0) * 1.0, 2: np.where(z > 0, 1.0, slope), 3: ((z > 0) & (z < 6)) * 1.0}[act]
    dz = dx.astype(np.float64) * g1
    bp = bpart[:brows * 2 * c].reshape(brows, 2, c).astype(np.float64).sum(0)
    # a pre-activation within rounding of a kink may take the other slope: compare with a bound on the flipped mass
    near = (np.abs(z) < 1e-5) | (np.abs(z - 6) < 1e-5)
    slack = (np.abs(dx) * near).reshape(-1, c).sum(0) * 2 + 1e-4 * np.abs(dz).reshape(-1, c).sum(0).max()
    assert np.all(np.abs(bp[0] - dz.reshape(-1, c).sum(0)) <= slack)
    assert np.all(np.abs(bp[1] - (dz * xh).reshape(-1, c).sum(0)) <= slack * max(1.0, np.abs(xh).max()))
    # K6d: the same pass also returns the weight gradient -- dX and the K6c partials bit for bit those above, dW against float64 and
    # against the separate entry point it replaces
    dwb = L.tsii_dw_bwd_dxdw_ws_bytes(n, h, wd, c, *geom)
    if dwb == 0:
        assert d != 1, "K6d always exists at dilation 1; at dilation 2 / 4 / 8 on the ring kernels only (not by phases, not on the small-map kernel)"
        return
    assert dwb == 4 * brows * 9 * c
    wsd = WS(dwb); wsd[:] = np.nan
    bpart3 = WS(4 * brows * 2 * c); bpart3[:] = np.nan
    dx3 = np.full((n, h, wd, c), np.nan, np.float32)
    dw3 = np.full((c, 1, 3, 3), np.nan, np.float32)
    assert L.tsii_dw_bwd_dxdw_bn(P(dy), P(inv), P(w), P(rmask), n, h, wd, c, *geom, ho, wo, P(x), P(mean_b), P(var_b), P(gam), P(bet),
                                 1e-5, act, slope, P(dx3), P(bpart3), P(dw3), P(ws), P(wsd), dwb, None) == 0, L.tsii_last_error()
    assert np.array_equal(dx3, dx)
    assert np.array_equal(bpart3[:brows * 2 * c], bpart[:brows * 2 * c])
    am = _act(z, act, slope) * (1.0 if rmask is None else rmask.astype(np.float64)[..., None])    # the layer's input
    gfull = dy.astype(np.float64) * (1.0 if inv is None else inv.astype(np.float64)[..., None])
    ap = np.zeros((n, h + 2 * p, wd + 2 * p, c)); ap[:, p:p + h, p:p + wd] = am
    dwr = np.zeros((c, 1, 3, 3)); dwabs = np.zeros((c, 1, 3, 3))
    for ky in range(3):
        for kx in range(3):
            prod = ap[:, ky * d:ky * d + ho, kx * d:kx * d + wo] * gfull
            dwr[:, 0, ky, kx] = prod.reshape(-1, c).sum(0)
            dwabs[:, 0, ky, kx] = np.abs(prod).reshape(-1, c).sum(0)
    assert np.all(np.abs(dw3 - dwr) <= 2e-6 * dwabs + 1e-6), np.abs(dw3 - dwr).max()
    # the separate weight-gradient entry point (BatchNorm on load with the equivalent scale / shift): same sums, another order
    isig = 1.0 / np.sqrt(var_b.astype(np.float64) + 1e-5)
    sc2 = (gam * isig).astype(np.float32); sh2 = (bet - mean_b * gam * isig).astype(np.float32)
    nb = L.tsii_dw_bwd_dw_ws_bytes(n, ho, wo, c, 3, 3)
    wsw = WS(nb)
    dw4 = np.full((c, 1, 3, 3), np.nan, np.float32)
    assert L.tsii_dw_bwd_dw_bn(P(dy), P(inv), P(keep), P(x), P(rmask), n, h, wd, c, *geom, ho, wo, P(sc2), P(sh2), act, slope,
                               P(dw4), None, P(wsw), nb, None) == 0, L.tsii_last_error()
    assert np.all(np.abs(dw3 - dw4) <= 2e-5 * dwabs + 1e-5), np.abs(dw3 - dw4).max()
    # K6e: the same pass fed with the gradient w.r.t. the activation of the BatchNorm that FOLLOWS the layer (da2, its raw input y2 and
    # the constants' table of tsii_bn_bwd_reduce): that BatchNorm's backward is applied on load.  Against the two-step route
    # (tsii_bn_bwd_apply, then tsii_dw_bwd_dxdw_bn) and the pieces against float64.
    if d != 1:
        assert L.tsii_dw_bwd_dxdw_fold_ok(n, h, wd, c, *geom) == 0, "K6e: dilation 1 only"
        return
    assert L.tsii_dw_bwd_dxdw_fold_ok(n, h, wd, c, *geom) == 1
    m2 = n * ho * wo
    da2 = rng.standard_normal((n, ho, wo, c)).astype(np.float32)
    y2 = (rng.standard_normal((n, ho, wo, c)) * 1.5 + 0.3).astype(np.float32)
    mean2 = rng.standard_normal(c).astype(np.float32) * 0.2; var2 = (rng.uniform(size=c) + 0.5).astype(np.float32)
    gam2 = (rng.uniform(size=c) + 0.5).astype(np.float32); bet2 = rng.standard_normal(c).astype(np.float32) * 0.3
    act2 = {0: 2, 1: 3, 2: 1, 3: 0}[act]            # another activation than the layer's input BatchNorm has
    xh2 = (y2.astype(np.float64) - mean2) / np.sqrt(var2.astype(np.float64) + 1e-5)
    z2 = xh2 * gam2 + bet2
    g2 = {0: np.ones_like(z2), 1: (z2 > 0) * 1.0, 2: np.where(z2 > 0, 1.0, slope), 3: ((z2 > 0) & (z2 < 6)) * 1.0}[act2]
    dz2 = da2.astype(np.float64) * g2
    # K6c partial rows of that BatchNorm as a producer would leave them: 3 rows that sum to the totals
    rows2 = 3
    tot = np.stack([dz2.reshape(-1, c).sum(0), (dz2 * xh2).reshape(-1, c).sum(0)])        # [2][c]
    split = rng.uniform(size=(rows2, 1, 1)); split /= split.sum(0)
    part2 = (split * tot[None]).astype(np.float32)
    coef = np.full(6 * c, np.nan, np.float32); dgam2 = np.full(c, np.nan, np.float32); dbet2 = np.full(c, np.nan, np.float32)
    rb = L.tsii_bn_bwd_reduce_ws_bytes(rows2, c)
    wsr = WS(rb)
    assert L.tsii_bn_bwd_reduce(P(mean2), P(var2), P(gam2), P(bet2), 1e-5, 1, P(part2), rows2, m2, c, P(dgam2), P(dbet2), P(coef),
                                P(wsr), rb, None) == 0, L.tsii_last_error()
    tot32 = part2.astype(np.float64).sum(0)
    assert np.allclose(dbet2, tot32[0], rtol=1e-6, atol=1e-6) and np.allclose(dgam2, tot32[1], rtol=1e-6, atol=1e-6)
    cf = coef.reshape(6, c).astype(np.float64)
    assert np.allclose(cf[0], mean2) and np.allclose(cf[1], 1 / np.sqrt(var2.astype(np.float64) + 1e-5), rtol=1e-6)
    assert np.allclose(cf[4], tot32[0] / m2, rtol=1e-5, atol=1e-7) and np.allclose(cf[5], tot32[1] / m2, rtol=1e-5, atol=1e-7)
    dy2 = np.full((n, ho, wo, c), np.nan, np.float32)
    assert L.tsii_bn_bwd_apply(P(da2), P(y2), m2, c, P(coef), act2, slope, P(dy2), None) == 0, L.tsii_last_error()
    dy2r = (dz2 - cf[4] - xh2 * cf[5]) * gam2 / np.sqrt(var2.astype(np.float64) + 1e-5)
    near2 = (np.abs(z2) < 1e-5) | (np.abs(z2 - 6) < 1e-5)
    assert np.all((np.abs(dy2 - dy2r) <= 1e-5 * max(1.0, np.abs(dy2r).max())) | near2)
    # two-step route
    dxA = np.full((n, h, wd, c), np.nan, np.float32); dwA = np.full((c, 1, 3, 3), np.nan, np.float32)
    bpA = WS(4 * brows * 2 * c); bpA[:] = np.nan
    assert L.tsii_dw_bwd_dxdw_bn(P(dy2), P(inv), P(w), P(rmask), n, h, wd, c, *geom, ho, wo, P(x), P(mean_b), P(var_b), P(gam), P(bet),
                                 1e-5, act, slope, P(dxA), P(bpA), P(dwA), P(ws), P(wsd), dwb, None) == 0, L.tsii_last_error()
    # folded
    dxB = np.full((n, h, wd, c), np.nan, np.float32); dwB = np.full((c, 1, 3, 3), np.nan, np.float32)
    bpB = WS(4 * brows * 2 * c); bpB[:] = np.nan
    wsd2 = WS(dwb); wsd2[:] = np.nan
    assert L.tsii_dw_bwd_dxdw_bn2(P(da2), P(y2), P(coef), act2, slope, P(inv), P(w), P(rmask), n, h, wd, c, *geom, ho, wo,
                                  P(x), P(mean_b), P(var_b), P(gam), P(bet), 1e-5, act, slope, P(dxB), P(bpB), P(dwB), P(ws), P(wsd2), dwb,
                                  None) == 0, L.tsii_last_error()
    sA = max(1.0, np.abs(dxA).max())
    assert np.abs(dxB - dxA).max() <= 2e-6 * sA, np.abs(dxB - dxA).max()
    gA = np.abs(dy2.astype(np.float64) * (1.0 if inv is None else inv.astype(np.float64)[..., None]))
    apA = np.zeros((n, h + 2 * p, wd + 2 * p, c)); apA[:, p:p + h, p:p + wd] = np.abs(am)
    dwabsA = np.zeros((c, 1, 3, 3))
    for ky in range(3):
        for kx in range(3):
            dwabsA[:, 0, ky, kx] = (apA[:, ky:ky + ho, kx:kx + wo] * gA).reshape(-1, c).sum(0)
    assert np.all(np.abs(dwB - dwA) <= 4e-6 * dwabsA + 1e-6), np.abs(dwB - dwA).max()
    pA = bpA[:brows * 2 * c].reshape(brows, 2, c).astype(np.float64).sum(0); pB = bpB[:brows * 2 * c].reshape(brows, 2, c).astype(np.float64).sum(0)
    assert np.isfinite(bpB[:brows * 2 * c]).all()
    # (a value within rounding of the input BatchNorm's kink may take the other slope in one of the two routes)
    slackB = (np.abs(dxA) * near).reshape(-1, c).sum(0) * 2 + 1e-4 * np.abs(dxA).reshape(-1, c).sum(0).max()
    assert np.all(np.abs(pB[0] - pA[0]) <= slackB) and np.all(np.abs(pB[1] - pA[1]) <= slackB * max(1.0, np.abs(xh).max()))


@pytest.mark.parametrize("n,h,wd,c,d,p,masked,bias,act", [
    (2, 21, 37, 40, 2, 2, True, True, 2),       # odd sizes: the phases have different row / column counts; channel tail
    (1, 32, 64, 32, 8, 8, True, False, 3),      # 4 x 8-pixel phase images (half a strip wide, half a step high), ReLU6
    (1, 33, 70, 8, 4, 4, False, True, 0),       # no mask planes; 70 = 17 full columns + a phase tail
    (2, 40, 48, 36, 2, 1, True, True, 1),       # padding < dilation: output 2 smaller per side than the input, input phase != output phase
    (1, 24, 40, 12, 4, 6, True, True, 2),       # padding > dilation: output larger than the input
    (1, 150, 19, 4, 2, 2, False, False, 2),     # many steps per chunk (both LDS buffers in turn) at target 1
])
@pytest.mark.parametrize("target", [1536, 1])
@pytest.mark.parametrize("phased", [1, 0])
def test_depthwise_dilated_strips_by_phase(emu, n, h, wd, c, d, p, masked, bias, act, target, phased):
    """Dilation 2 / 4 / 8 on the lean strip kernel PER PHASE (csrc/dw_lean.h, PH): every entry point of the d = 1 test at a dilated
    geometry -- forward, BatchNorm on load + statistics partials (all rows of tsii_dw_stat_rows written, counts adding up), dX, dX +
    K6c -- against float64.  The stock library keeps these geometries on the round-3 ring kernels (measured, dw_lean.h: the phased
    form only wins without mask planes and below dilation 8); phased = 1 selects it through an emulator-only switch, phased = 0 is
    the product's path at the same geometries."""
    L = emu
    L.tsii_emu_set_strip_target(target)
    L.tsii_emu_set_ls_phased(phased)
    try:
        _strip_entry_points(L, n, h, wd, c, masked, bias, act, d=d, p=p)
    finally:
        L.tsii_emu_set_strip_target(0)
        L.tsii_emu_set_ls_phased(0)


def _dw_ref_dw(x, dy, d, p):
    """float64 weight gradient [c, 3, 3] and bias gradient [c] of the stride-1 depth-wise 3x3 convolution (NHWC arrays)"""
    n, h, wd, c = x.shape
    ho, wo = dy.shape[1], dy.shape[2]
    xp = np.zeros((n, h + 2 * p, wd + 2 * p, c))
    xp[:, p:p + h, p:p + wd] = x
    g = np.zeros((c, 3, 3))
    for ky in range(3):
        for kx in range(3):
            g[:, ky, kx] = (dy.astype(np.float64) * xp[:, ky * d:ky * d + ho, kx * d:kx * d + wo]).sum((0, 1, 2))
    return g, dy.astype(np.float64).sum((0, 1, 2))


@pytest.mark.parametrize("n,h,wd,c,d,masked,bias,act", [
    (2, 40, 48, 40, 8, False, True, 2),      # 5-row phases, 48 columns (16 idle column lanes), channel tail
    (1, 64, 64, 32, 8, False, False, 3),     # the cfg-3 geometry: 8 x 64 phases exactly, ReLU6
    (1, 48, 30, 8, 16, False, True, 1),      # dilation 16: no strip form existed; 3-row phases
    (1, 60, 33, 36, 17, False, False, 2),    # RFB's dilation 17: phases of 4 and 3 rows
    (1, 64, 64, 16, 29, False, True, 0),     # dilation 29: most tap columns outside the image
    (2, 40, 48, 40, 8, True, True, 2),       # mask planes: the strip kernels, SAME partial-row layout
])
@pytest.mark.parametrize("rows", [1, 0], ids=["row-phase", "strips"])
def test_depthwise_large_dilation_row_phase(emu, n, h, wd, c, d, masked, bias, act, rows):
    """Dilation >= 8 on maps of more than 1024 pixels and at most 64 columns (csrc/dw_rows.h: one row phase of a 32-channel block in
    LDS, no halo): forward, BatchNorm on load + statistics partials, dX, dX + K6c through the C ABI against float64 -- at dilations
    16 / 17 / 29 the fused forms exist ONLY on this kernel -- and the weight gradient with and without BatchNorm on load.  rows = 0
    keeps the geometries on the strip / direct kernels (emulator-only switch): same answers, and for dilation 8 the same row count."""
    L = emu
    L.tsii_emu_set_dw_rows(rows)
    try:
        fused = rows == 1 or d == 8
        if fused and not (masked and d != 8):
            _strip_entry_points(L, n, h, wd, c, masked, bias, act, d=d, p=d)
        else:
            assert L.tsii_dw_stat_rows(n, h, wd, c, 3, 3, 1, 1, d, d) == 0        # no fused form without the row-phase kernel
        if masked:
            return
        rng = np.random.default_rng(n + h + wd + c + d)
        x = rng.standard_normal((n, h, wd, c)).astype(np.float32)
        dy = rng.standard_normal((n, h, wd, c)).astype(np.float32)
        geom = (3, 3, 1, 1, d, d, d, d)
        nb = L.tsii_dw_bwd_dw_ws_bytes(n, h, wd, c, 3, 3)
        ws = WS(nb)
        dwg = np.full((c, 1, 3, 3), np.nan, np.float32); db = np.full(c, np.nan, np.float32)
        assert L.tsii_dw_bwd_dw(P(dy), None, None, P(x), None, n, h, wd, c, *geom, h, wd, P(dwg), P(db), P(ws), nb, None) == 0, L.tsii_last_error()
        gr, br = _dw_ref_dw(x.astype(np.float64), dy, d, d)
        assert np.abs(dwg[:, 0] - gr).max() <= 2e-5 * max(1.0, np.abs(gr).max()) and np.abs(db - br).max() <= 2e-5 * max(1.0, np.abs(br).max())
        if fused:
            sc = (rng.uniform(size=c) + 0.5).astype(np.float32); sh = rng.standard_normal(c).astype(np.float32)
            dwg2 = np.full((c, 1, 3, 3), np.nan, np.float32)
            ws2 = WS(nb)
            assert L.tsii_dw_bwd_dw_bn(P(dy), None, None, P(x), None, n, h, wd, c, *geom, h, wd, P(sc), P(sh), act, 0.3, P(dwg2), None, P(ws2), nb, None) == 0, L.tsii_last_error()
            xa = _act(x.astype(np.float64) * sc + sh, act, 0.3)
            gr2, _ = _dw_ref_dw(xa, dy, d, d)
            assert np.abs(dwg2[:, 0] - gr2).max() <= 2e-5 * max(1.0, np.abs(gr2).max())
    finally:
        L.tsii_emu_set_dw_rows(1)


@pytest.mark.parametrize("n,h,wd,c,masked,bias,act", [
    (2, 21, 37, 40, True, True, 2),      # odd sizes: row / column / channel tails
    (1, 16, 32, 32, True, False, 3),     # two steps, two strips exactly, ReLU6
    (1, 5, 7, 4, False, True, 0),        # smaller than a step and a strip, no mask planes
    (2, 70, 34, 36, True, True, 1),      # many steps (both LDS buffers in turn), strip tail of one column, channel tail
    (1, 64, 64, 8, False, False, 2),     # whole tiles, no planes
])
@pytest.mark.parametrize("target", [1536, 1])
def test_depthwise_stride2_forward_strip(emu, n, h, wd, c, masked, bias, act, target):
    """The stride-2 forward strip (dw_lean_s2.h: 3x3, stride 2, pad 1) through tsii_dw_fwd / tsii_dw_fwd_bn: plain forward,
    BatchNorm + activation on load with the statistics partials, statistics alone (bit-identical output); chunks of one and of
    several marching steps (target, see test_depthwise_strip_entry_points)."""
    L = emu
    L.tsii_emu_set_strip_target(target)
    try:
        rng = np.random.default_rng(h * 100 + wd + 7)
        slope = 0.3
        ho, wo = (h - 1) // 2 + 1, (wd - 1) // 2 + 1
        x = rng.standard_normal((n, h, wd, c)).astype(np.float32) + 0.5
        w = rng.standard_normal((c, 1, 3, 3)).astype(np.float32)
        b = rng.standard_normal(c).astype(np.float32) if bias else None
        rmask = (rng.uniform(size=(n, h, wd)) > 0.25).astype(np.float32) if masked else None
        keep = denom = None
        if masked:
            pm = np.zeros((n, h + 2, wd + 2)); pm[:, 1:-1, 1:-1] = rmask
            cnt = sum(pm[:, ky:ky + 2 * (ho - 1) + 1:2, kx:kx + 2 * (wo - 1) + 1:2] for ky in range(3) for kx in range(3))
            keep = (cnt > 0).astype(np.float32)
            denom = (np.where(cnt > 0, cnt, 1.0) * c).astype(np.float32)
        geom = (3, 3, 2, 2, 1, 1, 1, 1)
        ws = WS(4 * (9 * c + 16))
        before = L.hipemu_launches(256)
        y = np.full((n, ho, wo, c), np.nan, np.float32)
        assert L.tsii_dw_fwd(P(x), P(rmask), P(w), P(b), P(denom), P(keep), n, h, wd, c, *geom, ho, wo, P(y), P(ws), None) == 0, L.tsii_last_error()
        yr = _dw_ref_fwd(x, rmask, w, b, denom, keep, 2, 1, 1)
        assert np.abs(y - yr).max() <= 2e-6 * max(1.0, np.abs(yr).max())
        sc = (rng.uniform(size=c) + 0.5).astype(np.float32); sh = rng.standard_normal(c).astype(np.float32)
        rows = L.tsii_dw_stat_rows(n, ho, wo, c, 3, 3, 2, 2, 1, 1)
        assert rows > 0
        part = WS(4 * rows * 4 * c)
        y2 = np.full((n, ho, wo, c), np.nan, np.float32)
        assert L.tsii_dw_fwd_bn(P(x), P(rmask), P(w), P(b), P(denom), P(keep), n, h, wd, c, *geom, ho, wo, P(sc), P(sh), act, slope,
                                P(part), P(y2), P(ws), None) == 0, L.tsii_last_error()
        xa = _act(x.astype(np.float64) * sc + sh, act, slope)
        y2r = _dw_ref_fwd(xa, rmask, w, b, denom, keep, 2, 1, 1)
        assert np.abs(y2 - y2r).max() <= 2e-6 * max(1.0, np.abs(y2r).max())
        pr = part[:rows * 4 * c].reshape(rows, 4, c).astype(np.float64)
        cnts, piv, s1, s2 = pr[:, 0], pr[:, 1], pr[:, 2], pr[:, 3]
        m_tot = n * ho * wo
        assert np.all(cnts.sum(0) == m_tot)
        mean = (cnts * piv + s1).sum(0) / m_tot
        ex2 = (s2 + 2 * piv * s1 + cnts * piv * piv).sum(0) / m_tot
        y2d = y2.astype(np.float64).reshape(-1, c)
        assert np.abs(mean - y2d.mean(0)).max() <= 1e-5 * max(1.0, np.abs(y2d).max())
        assert np.abs(ex2 - (y2d ** 2).mean(0)).max() <= 1e-5 * max(1.0, (y2d ** 2).max())
        y3 = np.full((n, ho, wo, c), np.nan, np.float32)
        assert L.tsii_dw_fwd_bn(P(x), P(rmask), P(w), P(b), P(denom), P(keep), n, h, wd, c, *geom, ho, wo, None, None, 0, 0.0,
                                P(part), P(y3), P(ws), None) == 0, L.tsii_last_error()
        assert np.array_equal(y3, y)
    finally:
        L.tsii_emu_set_strip_target(0)


@pytest.mark.parametrize("n,h,wd,c,d,masked,bias,act", [
    (2, 32, 32, 40, 8, True, True, 2),     # ImageFill's dilated level geometry; channel tail (40 = 2 x 16 + 8); 2 partial rows per image
    (1, 32, 32, 16, 8, True, False, 3),    # ReLU6 clamp
    (3, 9, 13, 20, 8, False, True, 0),     # odd map smaller than the dilation's reach, no planes
    (1, 20, 40, 8, 17, True, True, 1),     # dilation larger than the map's height: whole tap rows fall outside
])
def test_depthwise_small_map_dilated(emu, n, h, wd, c, d, masked, bias, act):
    """Dilated 3x3 depth-wise stencils on small maps (dw_small.h: the whole map of a 16-channel block in LDS): forward, forward
    with BatchNorm on load + statistics partials (rows laid out as the strip plan's, the extra rows empty), dX, dX with the
    BatchNorm-backward reductions."""
    L = emu
    rng = np.random.default_rng(h * 100 + wd + d)
    slope = 0.3
    x = rng.standard_normal((n, h, wd, c)).astype(np.float32) + 0.5
    w = rng.standard_normal((c, 1, 3, 3)).astype(np.float32)
    b = rng.standard_normal(c).astype(np.float32) if bias else None
    rmask = (rng.uniform(size=(n, h, wd)) > 0.2).astype(np.float32) if masked else None
    keep = denom = None
    if masked:
        pm = np.zeros((n, h + 2 * d, wd + 2 * d)); pm[:, d:-d, d:-d] = rmask
        cnt = sum(pm[:, ky * d:ky * d + h, kx * d:kx * d + wd] for ky in range(3) for kx in range(3))
        keep = (cnt > 0).astype(np.float32)
        denom = (np.where(cnt > 0, cnt, 1.0) * c).astype(np.float32)
    geom = (3, 3, 1, 1, d, d, d, d)
    ws = WS(4 * (9 * c + 16))
    y = np.full((n, h, wd, c), np.nan, np.float32)
    assert L.tsii_dw_fwd(P(x), P(rmask), P(w), P(b), P(denom), P(keep), n, h, wd, c, *geom, h, wd, P(y), P(ws), None) == 0, L.tsii_last_error()
    yr = _dw_ref_fwd(x, rmask, w, b, denom, keep, 1, d, d)
    assert np.abs(y - yr).max() <= 2e-6 * max(1.0, np.abs(yr).max())
    sc = (rng.uniform(size=c) + 0.5).astype(np.float32); sh = rng.standard_normal(c).astype(np.float32)
    rows = L.tsii_dw_stat_rows(n, h, wd, c, 3, 3, 1, 1, d, d)
    if rows > 0:
        part = WS(4 * rows * 4 * c)
        part[:] = np.nan
        y2 = np.full((n, h, wd, c), np.nan, np.float32)
        assert L.tsii_dw_fwd_bn(P(x), P(rmask), P(w), P(b), P(denom), P(keep), n, h, wd, c, *geom, h, wd, P(sc), P(sh), act, slope,
                                P(part), P(y2), P(ws), None) == 0, L.tsii_last_error()
        xa = _act(x.astype(np.float64) * sc + sh, act, slope)
        y2r = _dw_ref_fwd(xa, rmask, w, b, denom, keep, 1, d, d)
        assert np.abs(y2 - y2r).max() <= 2e-6 * max(1.0, np.abs(y2r).max())
        pr = part[:rows * 4 * c].reshape(rows, 4, c).astype(np.float64)
        cnts = pr[:, 0]
        assert np.isfinite(cnts).all() and np.all(cnts.sum(0) == n * h * wd)
        live = cnts > 0
        piv, s1, s2 = np.where(live, pr[:, 1], 0.0), np.where(live, pr[:, 2], 0.0), np.where(live, pr[:, 3], 0.0)
        m_tot = n * h * wd
        mean = (cnts * piv + s1).sum(0) / m_tot
        ex2 = (s2 + 2 * piv * s1 + cnts * piv * piv).sum(0) / m_tot
        y2d = y2.astype(np.float64).reshape(-1, c)
        assert np.abs(mean - y2d.mean(0)).max() <= 1e-5 * max(1.0, np.abs(y2d).max())
        assert np.abs(ex2 - (y2d ** 2).mean(0)).max() <= 1e-5 * max(1.0, (y2d ** 2).max())
    dy = rng.standard_normal((n, h, wd, c)).astype(np.float32)
    inv = (keep / denom).astype(np.float32) if masked else None
    dx = np.full((n, h, wd, c), np.nan, np.float32)
    assert L.tsii_dw_bwd_dx(P(dy), P(inv), P(w), P(rmask), n, h, wd, c, *geom, h, wd, P(dx), P(ws), None) == 0, L.tsii_last_error()
    dxr = _dw_ref_fwd(dy, inv, w[:, :, ::-1, ::-1], None, None, None, 1, d, d)
    if masked:
        dxr = dxr * rmask.astype(np.float64)[..., None]
    assert np.abs(dx - dxr).max() <= 2e-6 * max(1.0, np.abs(dxr).max())
    brows = L.tsii_dw_bwd_stat_rows(n, h, wd, c, *geom)
    if brows > 0:
        bpart = WS(4 * brows * 2 * c)
        bpart[:] = np.nan
        mean_b = rng.standard_normal(c).astype(np.float32); var_b = (rng.uniform(size=c) + 0.5).astype(np.float32)
        gam = (rng.uniform(size=c) + 0.5).astype(np.float32); bet = rng.standard_normal(c).astype(np.float32)
        dx2 = np.full((n, h, wd, c), np.nan, np.float32)
        assert L.tsii_dw_bwd_dx_bn(P(dy), P(inv), P(w), P(rmask), n, h, wd, c, *geom, h, wd, P(x), P(mean_b), P(var_b), P(gam), P(bet),
                                   1e-5, act, slope, P(dx2), P(bpart), P(ws), None) == 0, L.tsii_last_error()
        assert np.array_equal(dx2, dx)
        xh = (x.astype(np.float64) - mean_b) / np.sqrt(var_b.astype(np.float64) + 1e-5)
        z = xh * gam + bet
        g1 = {0: np.ones_like(z), 1: (z > 0) * 1.0, 2: np.where(z > 0, 1.0, slope), 3: ((z > 0) & (z < 6)) * 1.0}[act]
        dz = dx.astype(np.float64) * g1
        bp = bpart[:brows * 2 * c].reshape(brows, 2, c).astype(np.float64)
        assert np.isfinite(bp).all()
        bp = bp.sum(0)
        near = (np.abs(z) < 1e-5) | (np.abs(z - 6) < 1e-5)
        slack = (np.abs(dx) * near).reshape(-1, c).sum(0) * 2 + 1e-4 * np.abs(dz).reshape(-1, c).sum(0).max()
        assert np.all(np.abs(bp[0] - dz.reshape(-1, c).sum(0)) <= slack)
        assert np.all(np.abs(bp[1] - (dz * xh).reshape(-1, c).sum(0)) <= slack * max(1.0, np.abs(xh).max()))
        # K6d on the small-map kernel (dilation 8): the same pass also returns the weight gradient
        dwb = L.tsii_dw_bwd_dxdw_ws_bytes(n, h, wd, c, *geom)
        assert (dwb > 0) == (d == 8), "the small-map kernel has the K6c / K6d forms where the strip plan defines the partial rows"
        if dwb > 0:
            assert dwb == 4 * brows * 9 * c
            wsd = WS(dwb); wsd[:] = np.nan
            bpart3 = WS(4 * brows * 2 * c); bpart3[:] = np.nan
            dx3 = np.full((n, h, wd, c), np.nan, np.float32)
            dw3 = np.full((c, 1, 3, 3), np.nan, np.float32)
            assert L.tsii_dw_bwd_dxdw_bn(P(dy), P(inv), P(w), P(rmask), n, h, wd, c, *geom, h, wd, P(x), P(mean_b), P(var_b), P(gam), P(bet),
                                         1e-5, act, slope, P(dx3), P(bpart3), P(dw3), P(ws), P(wsd), dwb, None) == 0, L.tsii_last_error()
            assert np.array_equal(dx3, dx)
            assert np.array_equal(bpart3[:brows * 2 * c], bpart[:brows * 2 * c])
            assert np.isfinite(wsd[:dwb // 4]).all(), "every weight-gradient partial row is written (an image's other rows: zeros)"
            am = _act(z, act, slope) * (1.0 if rmask is None else rmask.astype(np.float64)[..., None])
            gfull = dy.astype(np.float64) * (1.0 if inv is None else inv.astype(np.float64)[..., None])
            ap = np.zeros((n, h + 2 * d, wd + 2 * d, c)); ap[:, d:d + h, d:d + wd] = am
            for ky in range(3):
                for kx in range(3):
                    prod = ap[:, ky * d:ky * d + h, kx * d:kx * d + wd] * gfull
                    ref = prod.reshape(-1, c).sum(0)
                    assert np.all(np.abs(dw3[:, 0, ky, kx] - ref) <= 2e-6 * np.abs(prod).reshape(-1, c).sum(0) + 1e-6), (ky, kx)


@pytest.mark.parametrize("n,h,wd,c,masked,act", [
    (2, 21, 37, 40, True, 2),        # odd sizes, row / column / channel tails
    (1, 16, 32, 32, False, 3),       # whole steps and strips, no mask planes, ReLU6
    (3, 45, 18, 36, True, 1),        # several steps per chunk, a strip tail of 2 columns
    (1, 6, 5, 4, True, 0),           # smaller than a step and a strip
])
@pytest.mark.parametrize("target", [1536, 1])
def test_depthwise_stride2_dx_with_weight_gradient(emu, n, h, wd, c, masked, act, target):
    """K6d on the stride-2 dX strips (3x3 / stride 2 / padding 1): tsii_dw_bwd_dxdw_bn returns tsii_dw_bwd_dx_bn's dX and K6c partial
    rows bit for bit plus the weight gradient -- against float64 and against the separate tsii_dw_bwd_dw_bn pass."""
    L = emu
    L.tsii_emu_set_strip_target(target)
    try:
        rng = np.random.default_rng(h * 100 + wd + 13)
        slope = 0.3
        ho, wo = (h - 1) // 2 + 1, (wd - 1) // 2 + 1
        x = rng.standard_normal((n, h, wd, c)).astype(np.float32) + 0.5       # raw BatchNorm input of the layer
        w = rng.standard_normal((c, 1, 3, 3)).astype(np.float32)
        dy = rng.standard_normal((n, ho, wo, c)).astype(np.float32)
        rmask = (rng.uniform(size=(n, h, wd)) > 0.2).astype(np.float32) if masked else None
        keep = inv = None
        if masked:
            cnt = _dw_ref_fwd(rmask[..., None], None, np.ones((1, 1, 3, 3), np.float32), None, None, None, 2, 1, 1)[..., 0]
            keep = (cnt > 0).astype(np.float32)
            inv = (keep / (np.where(cnt > 0, cnt, 1.0) * c)).astype(np.float32)
        geom = (3, 3, 2, 2, 1, 1, 1, 1)
        ws = WS(4 * (9 * c + 16))
        brows = L.tsii_dw_bwd_stat_rows(n, h, wd, c, *geom)
        assert brows > 0
        mean_b = rng.standard_normal(c).astype(np.float32); var_b = (rng.uniform(size=c) + 0.5).astype(np.float32)
        gam = (rng.uniform(size=c) + 0.5).astype(np.float32); bet = rng.standard_normal(c).astype(np.float32)
        bpart = WS(4 * brows * 2 * c); bpart[:] = np.nan
        dx = np.full((n, h, wd, c), np.nan, np.float32)
        assert L.tsii_dw_bwd_dx_bn(P(dy), P(inv), P(w), P(rmask), n, h, wd, c, *geom, ho, wo, P(x), P(mean_b), P(var_b), P(gam), P(bet),
                                   1e-5, act, slope, P(dx), P(bpart), P(ws), None) == 0, L.tsii_last_error()
        dwb = L.tsii_dw_bwd_dxdw_ws_bytes(n, h, wd, c, *geom)
        assert dwb == 4 * brows * 9 * c
        wsd = WS(dwb); wsd[:] = np.nan
        bpart3 = WS(4 * brows * 2 * c); bpart3[:] = np.nan
        dx3 = np.full((n, h, wd, c), np.nan, np.float32)
        dw3 = np.full((c, 1, 3, 3), np.nan, np.float32)
        assert L.tsii_dw_bwd_dxdw_bn(P(dy), P(inv), P(w), P(rmask), n, h, wd, c, *geom, ho, wo, P(x), P(mean_b), P(var_b), P(gam), P(bet),
                                     1e-5, act, slope, P(dx3), P(bpart3), P(dw3), P(ws), P(wsd), dwb, None) == 0, L.tsii_last_error()
        assert np.array_equal(dx3, dx)
        assert np.array_equal(bpart3[:brows * 2 * c], bpart[:brows * 2 * c])
        xh = (x.astype(np.float64) - mean_b) / np.sqrt(var_b.astype(np.float64) + 1e-5)
        am = _act(xh * gam + bet, act, slope) * (1.0 if rmask is None else rmask.astype(np.float64)[..., None])
        gfull = dy.astype(np.float64) * (1.0 if inv is None else inv.astype(np.float64)[..., None])
        ap = np.zeros((n, h + 3, wd + 3, c)); ap[:, 1:1 + h, 1:1 + wd] = am
        dwr = np.zeros((c, 1, 3, 3)); dwabs = np.zeros((c, 1, 3, 3))
        for ky in range(3):
            for kx in range(3):
                prod = ap[:, ky:ky + 2 * (ho - 1) + 1:2, kx:kx + 2 * (wo - 1) + 1:2] * gfull
                dwr[:, 0, ky, kx] = prod.reshape(-1, c).sum(0)
                dwabs[:, 0, ky, kx] = np.abs(prod).reshape(-1, c).sum(0)
        assert np.all(np.abs(dw3 - dwr) <= 2e-6 * dwabs + 1e-6), np.abs(dw3 - dwr).max()
        isig = 1.0 / np.sqrt(var_b.astype(np.float64) + 1e-5)
        sc2 = (gam * isig).astype(np.float32); sh2 = (bet - mean_b * gam * isig).astype(np.float32)
        nb = L.tsii_dw_bwd_dw_ws_bytes(n, ho, wo, c, 3, 3)
        wsw = WS(nb)
        dw4 = np.full((c, 1, 3, 3), np.nan, np.float32)
        assert L.tsii_dw_bwd_dw_bn(P(dy), P(inv), P(keep), P(x), P(rmask), n, h, wd, c, *geom, ho, wo, P(sc2), P(sh2), act, slope,
                                   P(dw4), None, P(wsw), nb, None) == 0, L.tsii_last_error()
        assert np.all(np.abs(dw3 - dw4) <= 2e-5 * dwabs + 1e-5), np.abs(dw3 - dw4).max()
        # K6e on the stride-2 strips: the following BatchNorm's backward applied while the slab is committed, against the two-step route
        assert L.tsii_dw_bwd_dxdw_fold_ok(n, h, wd, c, *geom) == 1
        m2 = n * ho * wo
        da2 = rng.standard_normal((n, ho, wo, c)).astype(np.float32)
        y2 = (rng.standard_normal((n, ho, wo, c)) * 1.5 + 0.3).astype(np.float32)
        mean2 = rng.standard_normal(c).astype(np.float32) * 0.2; var2 = (rng.uniform(size=c) + 0.5).astype(np.float32)
        gam2 = (rng.uniform(size=c) + 0.5).astype(np.float32); bet2 = rng.standard_normal(c).astype(np.float32) * 0.3
        act2 = {0: 2, 1: 3, 2: 1, 3: 0}[act]
        xh2 = (y2.astype(np.float64) - mean2) / np.sqrt(var2.astype(np.float64) + 1e-5)
        z2 = xh2 * gam2 + bet2
        g2 = {0: np.ones_like(z2), 1: (z2 > 0) * 1.0, 2: np.where(z2 > 0, 1.0, slope), 3: ((z2 > 0) & (z2 < 6)) * 1.0}[act2]
        dz2 = da2.astype(np.float64) * g2
        part2 = np.stack([dz2.reshape(-1, c).sum(0), (dz2 * xh2).reshape(-1, c).sum(0)]).astype(np.float32).reshape(1, 2, c)
        coef = np.full(6 * c, np.nan, np.float32); dgam2 = np.full(c, np.nan, np.float32); dbet2 = np.full(c, np.nan, np.float32)
        rb = L.tsii_bn_bwd_reduce_ws_bytes(1, c)
        wsr = WS(rb)
        assert L.tsii_bn_bwd_reduce(P(mean2), P(var2), P(gam2), P(bet2), 1e-5, 1, P(part2), 1, m2, c, P(dgam2), P(dbet2), P(coef), P(wsr), rb, None) == 0, L.tsii_last_error()
        dy2 = np.full((n, ho, wo, c), np.nan, np.float32)
        assert L.tsii_bn_bwd_apply(P(da2), P(y2), m2, c, P(coef), act2, slope, P(dy2), None) == 0, L.tsii_last_error()
        dxA = np.full((n, h, wd, c), np.nan, np.float32); dwA = np.full((c, 1, 3, 3), np.nan, np.float32)
        bpA = WS(4 * brows * 2 * c); bpA[:] = np.nan
        assert L.tsii_dw_bwd_dxdw_bn(P(dy2), P(inv), P(w), P(rmask), n, h, wd, c, *geom, ho, wo, P(x), P(mean_b), P(var_b), P(gam), P(bet),
                                     1e-5, act, slope, P(dxA), P(bpA), P(dwA), P(ws), P(wsd), dwb, None) == 0, L.tsii_last_error()
        dxB = np.full((n, h, wd, c), np.nan, np.float32); dwB = np.full((c, 1, 3, 3), np.nan, np.float32)
        bpB = WS(4 * brows * 2 * c); bpB[:] = np.nan
        wsd2 = WS(dwb); wsd2[:] = np.nan
        assert L.tsii_dw_bwd_dxdw_bn2(P(da2), P(y2), P(coef), act2, slope, P(inv), P(w), P(rmask), n, h, wd, c, *geom, ho, wo,
                                      P(x), P(mean_b), P(var_b), P(gam), P(bet), 1e-5, act, slope, P(dxB), P(bpB), P(dwB), P(ws), P(wsd2), dwb,
                                      None) == 0, L.tsii_last_error()
        assert np.abs(dxB - dxA).max() <= 2e-6 * max(1.0, np.abs(dxA).max()), np.abs(dxB - dxA).max()
        gA = np.abs(dy2.astype(np.float64) * (1.0 if inv is None else inv.astype(np.float64)[..., None]))
        apA = np.zeros((n, h + 3, wd + 3, c)); apA[:, 1:1 + h, 1:1 + wd] = np.abs(am)
        for ky in range(3):
            for kx in range(3):
                bound = (apA[:, ky:ky + 2 * (ho - 1) + 1:2, kx:kx + 2 * (wo - 1) + 1:2] * gA).reshape(-1, c).sum(0)
                assert np.all(np.abs(dwB[:, 0, ky, kx] - dwA[:, 0, ky, kx]) <= 4e-6 * bound + 1e-6), (ky, kx)
        pA = bpA[:brows * 2 * c].reshape(brows, 2, c).astype(np.float64).sum(0); pB = bpB[:brows * 2 * c].reshape(brows, 2, c).astype(np.float64).sum(0)
        assert np.isfinite(bpB[:brows * 2 * c]).all()
        zz = xh * gam + bet
        nearB = (np.abs(zz) < 1e-5) | (np.abs(zz - 6) < 1e-5)
        slackB = (np.abs(dxA) * nearB).reshape(-1, c).sum(0) * 2 + 1e-4 * np.abs(dxA).reshape(-1, c).sum(0).max()
        assert np.all(np.abs(pB[0] - pA[0]) <= slackB) and np.all(np.abs(pB[1] - pA[1]) <= slackB * max(1.0, np.abs(xh).max()))
    finally:
        L.tsii_emu_set_strip_target(0)


@pytest.mark.parametrize("pad", [0, 2])
def test_depthwise_stride2_forward_strip_other_paddings(emu, pad):
    """The stride-2 strip with padding 0 (valid) and 2: the kernel's buffer geometry (9 rows, 17 columns per 4 x 8 outputs) does not
    depend on the padding, only the origin of the slab does."""
    L = emu
    L.tsii_emu_set_strip_target(1)
    try:
        n, h, wd, c = 2, 27, 41, 12
        rng = np.random.default_rng(pad + 11)
        ho, wo = (h + 2 * pad - 3) // 2 + 1, (wd + 2 * pad - 3) // 2 + 1
        x = rng.standard_normal((n, h, wd, c)).astype(np.float32)
        w = rng.standard_normal((c, 1, 3, 3)).astype(np.float32)
        b = rng.standard_normal(c).astype(np.float32)
        rmask = (rng.uniform(size=(n, h, wd)) > 0.2).astype(np.float32)
        geom = (3, 3, 2, 2, pad, pad, 1, 1)
        ws = WS(4 * (9 * c + 16))
        y = np.full((n, ho, wo, c), np.nan, np.float32)
        assert L.tsii_dw_fwd(P(x), P(rmask), P(w), P(b), None, None, n, h, wd, c, *geom, ho, wo, P(y), P(ws), None) == 0, L.tsii_last_error()
        xm = x.astype(np.float64) * rmask[..., None]
        xp = np.zeros((n, h + 2 * pad, wd + 2 * pad, c)); xp[:, pad:pad + h, pad:pad + wd] = xm
        yr = np.zeros((n, ho, wo, c))
        for ky in range(3):
            for kx in range(3):
                yr += xp[:, ky:ky + 2 * (ho - 1) + 1:2, kx:kx + 2 * (wo - 1) + 1:2] * w[:, 0, ky, kx].astype(np.float64)
        yr += b.astype(np.float64)
        assert np.abs(y - yr).max() <= 2e-6 * max(1.0, np.abs(yr).max())
    finally:
        L.tsii_emu_set_strip_target(0)


# ---- 1x1 convolution over cat(nearest-x2(low), skip) with the low half computed at low resolution (K7b) -------------------------
@pytest.mark.parametrize("n,h,wd,k,N,stats", [(2, 8, 16, 40, 128, True),      # producer / consumer kernel (256-row tiles)
                                              (1, 16, 32, 64, 256, False),     # producer / consumer kernel, 128 x 256 tiles
                                              (1, 6, 12, 20, 36, True),        # 4-wave kernels, row / column tails
                                              (3, 4, 8, 8, 64, False)])
def test_pointwise_upsampled_addend(emu, pc_opt2, mode, n, h, wd, k, N, stats):
    L = emu
    rng = np.random.default_rng(n * 1000 + h * 10 + k)
    m = n * h * wd
    x = rng.standard_normal((m, k)).astype(np.float32)
    w = rng.standard_normal((N, k)).astype(np.float32)
    z = rng.standard_normal((n, h // 2, wd // 2, N)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    r0 = (rng.uniform(size=m) > 0.2).astype(np.float32)
    keep = (rng.uniform(size=m) > 0.1).astype(np.float32)
    denom = rng.integers(1, 9, size=m).astype(np.float32)
    y = np.full((m, N), np.nan, np.float32)
    nb = L.tsii_pw_ws_bytes(N, k)
    ws = WS(nb)
    rows = L.tsii_pw_stat_rows(m)
    part = WS(4 * rows * 4 * N) if stats else None
    before = L.hipemu_launches(768)
    assert L.tsii_pw_fwd_up(P(x), m, k, P(w), N, P(bias), P(r0), k, None, P(denom), P(keep), P(z), h, wd, P(part), P(y), P(ws), nb, None) == 0, L.tsii_last_error()
    if mode == 6 and N >= 128:
        assert L.hipemu_launches(768) == before + 1, "the producer / consumer kernel should have taken this shape"
    zup = np.repeat(np.repeat(z.astype(np.float64), 2, axis=1), 2, axis=2).reshape(m, N)
    acc = (x.astype(np.float64) * r0[:, None]) @ w.astype(np.float64).T
    ref = np.where(keep[:, None] != 0, (acc + zup) / denom[:, None] + bias, 0.0)
    assert np.abs(y - ref).max() <= MODE_TOL[mode] * np.abs(ref).max(), (np.abs(y - ref).max(), np.abs(ref).max())
    if stats:
        pr = part[:rows * 4 * N].reshape(rows, 4, N).astype(np.float64)
        cnt, piv, s1, s2 = pr[:, 0], pr[:, 1], pr[:, 2], pr[:, 3]
        assert np.all(cnt.sum(0) == m)
        mean = (cnt * piv + s1).sum(0) / m
        yd = y.astype(np.float64)
        assert np.abs(mean - yd.mean(0)).max() <= 1e-5 * max(1.0, np.abs(yd).max())
        ex2 = (s2 + 2 * piv * s1 + cnt * piv * piv).sum(0) / m
        assert np.abs(ex2 - (yd ** 2).mean(0)).max() <= 1e-5 * max(1.0, (yd ** 2).max())
    # the addend's gradient: 2 x 2 sums of dy * inv
    dy = rng.standard_normal((n, h, wd, N)).astype(np.float32)
    inv = (keep / denom).astype(np.float32).reshape(n, h, wd)
    dz = np.full((n, h // 2, wd // 2, N), np.nan, np.float32)
    assert L.tsii_pool2x2_scaled(P(dy), P(inv), n, h // 2, wd // 2, N, P(dz), None) == 0, L.tsii_last_error()
    g = dy.astype(np.float64) * inv[..., None]
    dzr = g.reshape(n, h // 2, 2, wd // 2, 2, N).sum(axis=(2, 4))
    assert np.abs(dz - dzr).max() <= 1e-6 * max(1.0, np.abs(dzr).max())


@pytest.mark.parametrize("n,h,wd,c,act,rows", [(2, 8, 12, 8, 2, 5), (1, 4, 6, 5, 1, 3), (3, 6, 4, 36, 0, 600)])
def test_batchnorm_backward_with_pooled_addend_gradient(emu, n, h, wd, c, act, rows):
    """tsii_bn_act_bwd_pre_pool = tsii_bn_act_bwd_pre followed by tsii_pool2x2_scaled over its dy: same dy / dgamma / dbeta bit
    for bit, pooled sums equal to the second pass's (same association; 1e-6 leaves room for FMA contraction)."""
    L = emu
    rng = np.random.default_rng(n * 100 + c)
    m = n * h * wd
    y = rng.standard_normal((m, c)).astype(np.float32)
    dout = rng.standard_normal((m, c)).astype(np.float32)
    mean = y.mean(0); var = y.var(0)
    gamma = rng.uniform(0.5, 1.5, size=c).astype(np.float32); beta = rng.standard_normal(c).astype(np.float32) * 0.3
    # any partial rows do: both entries reduce the same ones
    part = rng.standard_normal((rows, 2, c)).astype(np.float32)
    inv = (rng.uniform(size=m) > 0.2).astype(np.float32) / rng.integers(1, 9, size=m).astype(np.float32)
    nb = L.tsii_bn_ws_bytes(m, c)
    out = {}
    for pooled in (False, True):
        dy = np.full((m, c), np.nan, np.float32); dg = np.zeros(c, np.float32); db = np.zeros(c, np.float32)
        ws = WS(nb)
        if pooled:
            dz = np.full((n, h // 2, wd // 2, c), np.nan, np.float32)
            assert L.tsii_bn_act_bwd_pre_pool(P(dout), P(y), m, c, P(mean), P(var), P(gamma), P(beta), 1e-5, act, 0.2, 1, P(part), rows,
                                              h, wd, P(inv), P(dy), P(dz), P(dg), P(db), P(ws), nb, None) == 0, L.tsii_last_error()
        else:
            dz = np.full((n, h // 2, wd // 2, c), np.nan, np.float32)
            assert L.tsii_bn_act_bwd_pre(P(dout), P(y), m, c, P(mean), P(var), P(gamma), P(beta), 1e-5, act, 0.2, 1, P(part), rows,
                                         P(dy), P(dg), P(db), P(ws), nb, None) == 0, L.tsii_last_error()
            assert L.tsii_pool2x2_scaled(P(dy), P(inv), n, h // 2, wd // 2, c, P(dz), None) == 0, L.tsii_last_error()
        out[pooled] = (dy, dg, db, dz)
    (dy0, dg0, db0, dz0), (dy1, dg1, db1, dz1) = out[False], out[True]
    assert np.array_equal(dy0, dy1) and np.array_equal(dg0, dg1) and np.array_equal(db0, db1)
    assert np.isfinite(dz1).all() and np.abs(dz1 - dz0).max() <= 1e-6 * max(1.0, np.abs(dz0).max())
    ref = (dy0.astype(np.float64) * inv[:, None]).reshape(n, h // 2, 2, wd // 2, 2, c).sum(axis=(2, 4))
    assert np.abs(dz1 - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max())
    # rows that are not whole even-sized images are refused
    ws = WS(nb)
    assert L.tsii_bn_act_bwd_pre_pool(P(dout), P(y), m, c, P(mean), P(var), P(gamma), P(beta), 1e-5, act, 0.2, 1, P(part), rows,
                                      h + 1, wd, P(inv), P(dy1), P(dz1), P(dg1), P(db1), P(ws), nb, None) != 0
    assert b"bn_act_bwd_pre_pool" in L.tsii_last_error()


@pytest.mark.parametrize("rows", [1, 3, 64, 65, 600, 2048, 2049, 5000])
@pytest.mark.parametrize("c,training", [(8, 1), (40, 1), (33, 0)])
def test_batchnorm_backward_reduction_of_partial_rows(emu, rows, c, training):
    """tsii_bn_bwd_reduce over every row-count regime (<= 64 rows: one launch of the final kernel; 65 .. 2048: the one-launch fp64 kernel of
    round 6; above: level-1 fold + final): dgamma, dbeta and the constants' table against float64."""
    L = emu
    rng = np.random.default_rng(rows * 7 + c)
    m = 12345
    part = rng.standard_normal((rows, 2, c)).astype(np.float32)
    mean = rng.standard_normal(c).astype(np.float32); var = (rng.uniform(size=c) + 0.5).astype(np.float32)
    gam = (rng.uniform(size=c) + 0.5).astype(np.float32); bet = rng.standard_normal(c).astype(np.float32)
    coef = np.full(6 * c + 4, np.nan, np.float32)[:6 * c]; dg = np.full(c, np.nan, np.float32); db = np.full(c, np.nan, np.float32)
    nb = L.tsii_bn_bwd_reduce_ws_bytes(rows, c)
    ws = WS(nb)
    assert L.tsii_bn_bwd_reduce(P(mean), P(var), P(gam), P(bet), 1e-5, training, P(part), rows, m, c, P(dg), P(db), P(coef), P(ws), nb, None) == 0, L.tsii_last_error()
    tot = part.astype(np.float64).sum(0)
    scale = np.abs(part).astype(np.float64).sum(0)
    assert np.all(np.abs(db - tot[0]) <= 1e-6 * scale[0] + 1e-7) and np.all(np.abs(dg - tot[1]) <= 1e-6 * scale[1] + 1e-7)
    cf = coef.reshape(6, c).astype(np.float64)
    assert np.array_equal(cf[0], mean.astype(np.float64)) and np.array_equal(cf[2], gam.astype(np.float64)) and np.array_equal(cf[3], bet.astype(np.float64))
    assert np.allclose(cf[1], 1 / np.sqrt(var.astype(np.float64) + 1e-5), rtol=1e-6)
    if training:
        assert np.all(np.abs(cf[4] - tot[0] / m) <= 1e-6 * scale[0] / m + 1e-9) and np.all(np.abs(cf[5] - tot[1] / m) <= 1e-6 * scale[1] / m + 1e-9)
    else:
        assert not cf[4].any() and not cf[5].any()


def test_pointwise_upsampled_addend_rejects_bad_geometry(emu):
    L = emu
    x = np.zeros((48, 8), np.float32); w = np.zeros((8, 8), np.float32); z = np.zeros((12, 8), np.float32); y = np.zeros((48, 8), np.float32)
    nb = L.tsii_pw_ws_bytes(8, 8); ws = WS(nb)
    # width 6 is not a multiple of 4; height 3 is odd; rows are not whole images
    for hh, ww in ((8, 6), (3, 16), (5, 8)):
        assert L.tsii_pw_fwd_up(P(x), 48, 8, P(w), 8, None, None, 0, None, None, None, P(z), hh, ww, None, P(y), P(ws), nb, None) != 0
        assert b"pw_fwd_up" in L.tsii_last_error()


def test_head_matrix_core_entries_refuse_what_they_do_not_cover(emu):
    """K4d entry points: geometry queries and loud refusals (whole 16 x 64 tiles, 32 / 64 low channels, <= 3 output channels; forward:
    3 skip channels; fused d low: 32 low channels) -- the callers fall back to tsii_head_cat_* on a 0."""
    L = emu
    assert L.tsii_head_cat_low_ok(2, 32, 64, 32, 3, 3) == 1 and L.tsii_head_cat_low_ok(2, 32, 64, 64, 5, 1) == 1
    for bad in ((2, 24, 64, 32, 3, 3), (2, 32, 96, 32, 3, 3), (2, 32, 64, 48, 3, 3), (2, 32, 64, 32, 3, 4), (2, 32, 64, 32, 0, 3)):
        assert L.tsii_head_cat_low_ok(*bad) == 0, bad
    assert L.tsii_head_cat_fwd_low_ok(1, 16, 64, 64, 3, 2) == 1 and L.tsii_head_cat_fwd_low_ok(1, 16, 64, 32, 4, 2) == 0
    assert L.tsii_head_cat_bwd_low_ok(1, 16, 64, 32, 3, 3) == 1 and L.tsii_head_cat_bwd_low_ok(1, 16, 64, 64, 3, 3) == 0
    n, h, w, c1, c2, co = 1, 16, 64, 64, 3, 3
    low = np.zeros((n, h // 2, w // 2, c1), np.float32); skip = np.zeros((n, h, w, c2), np.float32)
    wt = np.zeros((co, c1 + c2, 3, 3), np.float32); dy = np.zeros((n, h, w, co), np.float32)
    dw = np.zeros_like(wt); dlow = np.zeros_like(low); y = np.zeros_like(dy)
    nb = L.tsii_dense_bwd_dw_ws_bytes(n, h, w, c1 + c2, co, 3, 3); ws = WS(nb)
    # 64 low channels have no fused d low
    assert L.tsii_head_cat_bwd_low(P(dy), None, None, P(low), P(skip), c1, c2, None, None, P(wt), n, h, w, co, P(dw), None, P(dlow), P(ws), nb, None) != 0
    assert b"head_cat_bwd_low" in L.tsii_last_error()
    # a map that is not whole tiles
    assert L.tsii_head_cat_fwd_low(P(low), P(skip), c1, c2, None, None, P(wt), None, None, None, n, h + 2, w, co, P(y), None) != 0
    assert b"head_cat_fwd_low" in L.tsii_last_error()
    assert L.tsii_head_cat_bwd_dw_low(P(dy), None, None, P(low), P(skip), c1, c2, None, None, n, h, w + 2, co, P(dw), None, P(ws), nb, None) != 0
    assert b"head_cat_bwd_dw_low" in L.tsii_last_error()


@pytest.mark.parametrize("rows,c", [(3, 8), (256, 40), (1024, 33), (1025, 32), (5000, 36)])
def test_bn_finalize_from_partials(emu, rows, c):
    """tsii_bn_finalize: (count, pivot, sum(y-p), sum((y-p)^2)) rows -> mean / biased variance / running statistics / (scale,
    shift); <= 1024 rows take the one-kernel path, more the two-level one; rows with count 0 are ignored."""
    L = emu
    rng = np.random.default_rng(rows)
    cnt = rng.integers(0, 130, size=(rows, 1)).astype(np.float32) * np.ones((1, c), np.float32)
    cnt[0] = 128
    piv = rng.standard_normal((rows, c)).astype(np.float32) * 3
    mu_b = piv + rng.standard_normal((rows, c)).astype(np.float32) * 0.1
    var_b = rng.uniform(0.5, 2.0, size=(rows, c)).astype(np.float32)
    s1 = (cnt * (mu_b - piv)).astype(np.float32)
    s2 = (cnt * (var_b + (mu_b - piv) ** 2)).astype(np.float32)
    part = np.stack([cnt, piv, s1, s2], axis=1).copy()            # [rows][4][c]
    part[cnt[:, 0] == 0, 1:] = np.nan                              # empty rows may hold anything
    m = int(cnt[:, 0].sum())
    d = lambda a: a.astype(np.float64)
    mean_r = (d(cnt) * d(piv) + d(s1)).sum(0) / m
    ex2_r = (d(s2) + 2 * d(piv) * d(s1) + d(cnt) * d(piv) ** 2)
    ex2_r = np.where(cnt > 0, ex2_r, 0.0).sum(0) / m
    mean_r = np.where(cnt > 0, d(cnt) * d(piv) + d(s1), 0.0).sum(0) / m
    var_r = ex2_r - mean_r ** 2
    gamma = rng.uniform(0.5, 1.5, size=c).astype(np.float32); beta = rng.standard_normal(c).astype(np.float32)
    rm = rng.standard_normal(c).astype(np.float32); rv = rng.uniform(0.5, 1.5, size=c).astype(np.float32)
    rm0, rv0 = rm.copy(), rv.copy()
    mean = np.zeros(c, np.float32); var = np.zeros(c, np.float32); sc = np.zeros(c, np.float32); sh = np.zeros(c, np.float32)
    nb = L.tsii_bn_finalize_ws_bytes(rows, c)
    ws = WS(nb)
    assert L.tsii_bn_finalize(P(part), rows, c, m, P(mean), P(var), P(rm), P(rv), 0.1, P(gamma), P(beta), 1e-5, P(sc), P(sh), P(ws), nb, None) == 0, L.tsii_last_error()
    assert np.abs(mean - mean_r).max() <= 1e-6 * max(1.0, np.abs(mean_r).max())
    assert np.abs(var - var_r).max() <= 1e-5 * var_r.max()
    assert np.allclose(rm, 0.9 * rm0 + 0.1 * mean, rtol=1e-6, atol=1e-7)
    assert np.allclose(rv, 0.9 * rv0 + 0.1 * var * m / (m - 1), rtol=1e-5)
    scr = gamma / np.sqrt(var + 1e-5)
    assert np.allclose(sc, scr, rtol=1e-5) and np.allclose(sh, beta - mean * scr, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("split,has_r1", [(20, False), (20, True), (24, False)])
def test_pointwise_row_scale_split_inside_a_chunk(emu, split, has_r1):
    """Row scale r0 on channels < split, r1 (or the factor 1 of a premultiplied second part) after it.  The persistent
    producer / consumer kernel applies ONE factor per 8-channel chunk, so a split inside a chunk must send the layer to the 4-wave
    kernels -- with r1 == NULL too (it once scaled the chunk's tail by r0: relative error 0.38 on this shape)."""
    L = emu
    assert L.tsii_set_gemm_products(6) == 0
    M, K, N = 256, 64, 128
    rng = np.random.default_rng(split + has_r1)
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = rng.standard_normal((N, K)).astype(np.float32)
    r0 = rng.uniform(0.5, 2.0, size=M).astype(np.float32)
    r1 = rng.uniform(0.5, 2.0, size=M).astype(np.float32) if has_r1 else None
    y = np.zeros((M, N), np.float32)
    ws = WS(L.tsii_pw_ws_bytes(N, K))
    before = L.hipemu_launches(PC_THREADS)
    assert L.tsii_pw_fwd(P(x), M, K, P(w), N, None, P(r0), split, P(r1), None, None, P(y), P(ws), ws.nbytes, None) == 0, L.tsii_last_error()
    assert L.hipemu_launches(PC_THREADS) - before == (1 if split % 8 == 0 else 0)
    xs = x.astype(np.float64).copy()
    xs[:, :split] *= r0[:, None]
    if has_r1:
        xs[:, split:] *= r1[:, None]
    ref = xs @ w.astype(np.float64).T
    assert np.abs(y - ref).max() <= 1e-5 * np.abs(ref).max()


def _act_np(z, act, slope):
    """(a, da/dz) of the library's activation codes: 0 none, 1 ReLU, 2 LeakyReLU(slope), 3 ReLU6"""
    if act == 0:
        return z, np.ones_like(z)
    if act == 1:
        return np.maximum(z, 0), (z > 0).astype(z.dtype)
    if act == 2:
        return np.where(z > 0, z, slope * z), np.where(z > 0, 1.0, slope)
    return np.clip(z, 0, 6), ((z > 0) & (z < 6)).astype(z.dtype)


@pytest.mark.parametrize("m,c", [(1, 4), (3, 8), (5, 36), (127, 4), (130, 64), (1025, 12), (4099, 32), (257, 5), (4099, 1024), (40003, 64)])      # the last two: several rows per reduction lane (the 4-rows-in-flight loops and their tails)
@pytest.mark.parametrize("act,slope", [(0, 0.0), (2, 0.3), (3, 0.0)])
def test_batchnorm_apply_and_backward_kernels_row_tails(emu, m, c, act, slope):
    """tsii_bn_act_fwd / tsii_bn_act_bwd (training and eval) / tsii_bn_act_bwd_pre straight through the C ABI against float64 numpy,
    at row counts around the apply kernels' rows-per-thread blocking (round 5: a thread owns one channel vector and 4 rows; m % 4 in
    {1, 2, 3}, fewer rows than one thread's share, channel counts with and without the 16-byte vector path) -- a dense error in these
    passes would otherwise only show up as 'noise' in whole-network gradient tests."""
    L = emu
    rng = np.random.default_rng(m * 131 + c * 7 + act)
    y = (rng.standard_normal((m, c)) * 1.5 + 0.3).astype(np.float32)
    res = rng.standard_normal((m, c)).astype(np.float32)
    dout = rng.standard_normal((m, c)).astype(np.float32)
    gamma = rng.uniform(0.5, 1.5, size=c).astype(np.float32)
    beta = (rng.standard_normal(c) * 0.3).astype(np.float32)
    eps = 1e-5
    y64 = y.astype(np.float64)
    mean = y64.mean(0).astype(np.float32) if m > 1 else y64[0].astype(np.float32)
    var = (y64.var(0) if m > 1 else np.ones(c)).astype(np.float32)
    rstd = 1.0 / np.sqrt(var.astype(np.float64) + eps)
    xhat = (y64 - mean.astype(np.float64)) * rstd
    z = xhat * gamma + beta
    # ---- forward apply, with and without the residual
    for with_res in (False, True):
        out = np.full((m, c), np.nan, np.float32)
        assert L.tsii_bn_act_fwd(P(y), m, c, P(mean), P(var), P(gamma), P(beta), eps, act, slope, P(res) if with_res else None, P(out), None) == 0, L.tsii_last_error()
        # (the residual is added AFTER the activation: out = act(bn(y)) + residual, models/MobileNetV2.py:49 pattern)
        ref = _act_np(z, act, slope)[0] + (res.astype(np.float64) if with_res else 0.0)
        assert np.isfinite(out).all() and np.abs(out - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max()), (with_res, np.abs(out - ref).max())
    # ---- backward: training (batch statistics) and eval
    a, dadz = _act_np(z, act, slope)
    # keep the comparison away from the activation kinks: an fp32 z within rounding of a kink legitimately takes either derivative
    safe = np.ones_like(z, bool) if act == 0 else (np.abs(z) > 1e-4) & ((np.abs(z - 6) > 1e-4) if act == 3 else True)
    dz = dout.astype(np.float64) * dadz
    dbeta_r, dgamma_r = dz.sum(0), (dz * xhat).sum(0)
    dy_train = (dz - dbeta_r / m - xhat * dgamma_r / m) * gamma * rstd
    dy_eval = dz * gamma * rstd
    nb = L.tsii_bn_ws_bytes(m, c)
    for training, dy_r in ((1, dy_train), (0, dy_eval)):
        dy = np.full((m, c), np.nan, np.float32); dg = np.full(c, np.nan, np.float32); db = np.full(c, np.nan, np.float32)
        ws = WS(nb)
        assert L.tsii_bn_act_bwd(P(dout), P(y), m, c, P(mean), P(var), P(gamma), P(beta), eps, act, slope, training, P(dy), P(dg), P(db), P(ws), nb, None) == 0, L.tsii_last_error()
        if safe.all():
            scale = max(1.0, np.abs(dy_r).max())
            assert np.abs(dy - dy_r).max() <= 2e-5 * scale, (training, np.abs(dy - dy_r).max())
            assert np.abs(dg - dgamma_r).max() <= 2e-5 * max(1.0, np.abs(dgamma_r).max()) and np.abs(db - dbeta_r).max() <= 2e-5 * max(1.0, np.abs(dbeta_r).max())
        else:
            assert np.isfinite(dy).all()
    # ---- the 3-pass form fed with reductions somebody else took (K6c): partial rows that sum to the exact totals
    rows = 3
    part = np.zeros((rows, 2, c), np.float32)
    split = rng.dirichlet(np.ones(rows), size=c).T          # [rows, c] weights summing to 1 per channel
    part[:, 0, :] = (split * dbeta_r).astype(np.float32)
    part[:, 1, :] = (split * dgamma_r).astype(np.float32)
    dy = np.full((m, c), np.nan, np.float32); dg = np.full(c, np.nan, np.float32); db = np.full(c, np.nan, np.float32)
    ws = WS(nb)
    assert L.tsii_bn_act_bwd_pre(P(dout), P(y), m, c, P(mean), P(var), P(gamma), P(beta), eps, act, slope, 1, P(part), rows,
                                 P(dy), P(dg), P(db), P(ws), nb, None) == 0, L.tsii_last_error()
    if safe.all():
        assert np.abs(dy - dy_train).max() <= 5e-5 * max(1.0, np.abs(dy_train).max())
        assert np.abs(dg - dgamma_r).max() <= 5e-5 * max(1.0, np.abs(dgamma_r).max()) and np.abs(db - dbeta_r).max() <= 5e-5 * max(1.0, np.abs(dbeta_r).max())


@pytest.mark.parametrize("n,hw,c", [(2, 37, 128), (1, 300, 48), (2, 64, 320), (1, 50, 32), (3, 1000, 192), (1, 20, 640), (1, 9, 2052), (1, 33, 6)])
def test_scse_backward_one_pass(emu, n, hw, c):
    """tsii_scse_bwd (models/common.py:38-43): dx = g (cse + sse), dcse[n, c] = sum_hw g x, dsse[n, hw] = sum_c g x against float64 --
    the one-pass kernel (channel quads of a pixel in a lane group, channel sums through LDS) on the shapes it takes (c % 4 == 0, up to
    2048 channels) and the three-kernel form on the rest (2052, 6 channels); pixel ranges that do not fill the last iteration."""
    L = emu
    rng = np.random.default_rng(n * 1000 + hw + c)
    g = rng.standard_normal((n, hw, c)).astype(np.float32)
    x = rng.standard_normal((n, hw, c)).astype(np.float32)
    cse = rng.uniform(size=(n, c)).astype(np.float32)
    sse = rng.uniform(size=(n, hw)).astype(np.float32)
    dx = np.full((n, hw, c), np.nan, np.float32); dcse = np.full((n, c), np.nan, np.float32); dsse = np.full((n, hw), np.nan, np.float32)
    nb = L.tsii_gap_ws_bytes(n, hw, c)
    ws = WS(nb)
    assert L.tsii_scse_bwd(P(g), P(x), P(cse), P(sse), n, hw, c, P(dx), P(dcse), P(dsse), P(ws), nb, None) == 0, L.tsii_last_error()
    g64, x64 = g.astype(np.float64), x.astype(np.float64)
    rdx = g64 * (cse.astype(np.float64)[:, None, :] + sse.astype(np.float64)[:, :, None])
    assert np.abs(dx - rdx).max() <= 2e-6 * max(1.0, np.abs(rdx).max())
    rc = (g64 * x64).sum(1); rs = (g64 * x64).sum(2)
    assert np.abs(dcse - rc).max() <= 1e-5 * max(1.0, np.abs(rc).max()) and np.abs(dsse - rs).max() <= 1e-5 * max(1.0, np.abs(rs).max())


@pytest.mark.parametrize("n,hw,c", [(2, 37, 128), (1, 300, 48), (2, 600, 320), (1, 50, 32), (1, 20, 1920), (1, 9, 2052), (1, 33, 6), (2, 5, 4)])
def test_global_average_pool_forward(emu, n, hw, c):
    """tsii_gap_fwd (the squeeze of scSE, models/common.py:13-27): mean over the pixels of every (image, channel) against float64 --
    16-byte loads when c % 4 == 0 (two rows in flight per thread, odd row counts, more channel quads than threads), else the scalar form."""
    L = emu
    rng = np.random.default_rng(n * 77 + hw + c)
    x = (rng.standard_normal((n, hw, c)) + 0.25).astype(np.float32)
    out = np.full((n, c), np.nan, np.float32)
    nb = L.tsii_gap_ws_bytes(n, hw, c)
    ws = WS(nb)
    assert L.tsii_gap_fwd(P(x), n, hw, c, P(out), P(ws), nb, None) == 0, L.tsii_last_error()
    ref = x.astype(np.float64).mean(1)
    assert np.abs(out - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max())
