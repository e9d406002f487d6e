"""bf16 ACTIVATION STORAGE (include/tsii_hip.h, "bf16 activation storage"): every tsii_bf16_* entry point straight through the
C ABI on the TEST-ONLY emulator against float64 numpy evaluated on the SAME bf16-rounded operands.  The definition under test:
operands are the bf16 values in memory, products and sums are fp32-class, the result is rounded once (RNE) when stored; partial
rows (BatchNorm statistics, K6c reductions) describe the rounded values.  Tolerance of a stored tensor: one bf16 ulp of the
reference (2^-8 relative, + an fp32-accumulation allowance); fp32 outputs (weight gradients, partials): 1e-4 class."""
import ctypes

import numpy as np
import pytest

from text_segmentation_image_inpainting_amd import _lib


import torch

from tests.backends import both_backends  # noqa: F401  (the same cases run on the emulator and, -m gpu, on the chip)


class _Arr:
    """a numpy array handed to the C ABI: on the GPU backend it is uploaded for the call and copied back afterwards"""

    def __init__(self, a):
        self.a = a


_MODE = {"gpu": False}


def P(a):
    if a is None:
        return None
    return _Arr(a) if _MODE["gpu"] else ctypes.c_void_p(a.ctypes.data)


class _GpuLib:
    """the bound library with numpy in / numpy out: every array argument is staged through device memory around the call"""

    def __init__(self, lib):
        self._lib = lib

    def __getattr__(self, name):
        fn = getattr(self._lib, name)

        def call(*args):
            staged, conv = [], []
            for x in args:
                if isinstance(x, _Arr):
                    t = torch.from_numpy(x.a.reshape(-1).view(np.uint8)).cuda()
                    staged.append((x.a, t))
                    conv.append(ctypes.c_void_p(t.data_ptr()))
                else:
                    conv.append(x)
            if conv and conv[-1] is None and fn.argtypes and fn.argtypes[-1] is ctypes.c_void_p:
                conv[-1] = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            rc = fn(*conv)
            torch.cuda.synchronize()
            for a, t in staged:
                a.reshape(-1).view(np.uint8)[:] = t.cpu().numpy()
            return rc
        return call


@pytest.fixture(params=[pytest.param("emu"), pytest.param("gpu", marks=pytest.mark.gpu)])
def emu(request):
    """the library under test: the host emulation of the unmodified kernel sources (CPU suite) or libtsii_hip.so on the chip (-m gpu)"""
    if request.param == "gpu":
        if not torch.cuda.is_available():
            pytest.fail("-m gpu tests need a ROCm GPU")
        _MODE["gpu"] = True
        try:
            yield _GpuLib(_lib.lib())
        finally:
            _MODE["gpu"] = False
        return
    from tests.emu import build_emu
    if not build_emu.available():
        pytest.skip("host clang++ not available")
    yield _lib.bind(ctypes.CDLL(build_emu.build()))


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


# ---- bf16 <-> numpy -------------------------------------------------------------------------------------------------
def bf16_bits(a):
    """float32 array -> uint16 bf16 bits, round to nearest even"""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32)
    return ((u + (((u >> 16) & 1) + np.uint32(0x7FFF))) >> 16).astype(np.uint16)


def bf16_val(b):
    """uint16 bf16 bits -> float32 values"""
    return (b.astype(np.uint32) << 16).view(np.float32)


def rb(a):
    """round a float array to bf16 and back (float64 result for reference arithmetic)"""
    return bf16_val(bf16_bits(np.asarray(a, np.float32))).astype(np.float64)


def rand_bf16(rng, shape, scale=1.0, shift=0.0):
    bits = bf16_bits((rng.standard_normal(shape) * scale + shift).astype(np.float32))
    return bits, bf16_val(bits).astype(np.float64)


def close_bf16(got_bits, ref, extra=0.0):
    """stored bf16 tensor vs float64 reference: one bf16 ulp of the element, plus an allowance for the fp32 accumulation"""
    got = bf16_val(got_bits).astype(np.float64)
    tol = np.abs(ref) * 2.0 ** -8 + (2e-6 + extra) * np.abs(ref).max() + 1e-30
    bad = np.abs(got - ref) > tol
    assert not bad.any(), f"{bad.sum()} of {bad.size} elements off; worst {np.abs(got - ref).max():.3e} vs max|ref| {np.abs(ref).max():.3e}"


def act_np(z, act, slope):
    return {0: z, 1: np.maximum(z, 0), 2: np.where(z > 0, z, slope * z), 3: np.clip(z, 0, 6)}[act]


def act_grad_np(z, act, slope):
    return {0: np.ones_like(z), 1: (z > 0) * 1.0, 2: np.where(z > 0, 1.0, slope), 3: ((z > 0) & (z < 6)) * 1.0}[act]


def inbn_np(xv, sc, sh, act, slope):
    """the load-time BatchNorm + activation as the kernels evaluate it: fp32 fma, then the result is rounded to bf16 (it is an MFMA operand)"""
    z = np.float32(xv.astype(np.float32) * sc.astype(np.float32) + sh.astype(np.float32)).astype(np.float64)
    return rb(act_np(z, act, slope))


def stats_from_part(part, m):
    """(mean, biased var) from [rows][4][c] partials (count, pivot, s1, s2)"""
    n, p, s1, s2 = (part[:, i].astype(np.float64) for i in range(4))
    assert n.sum(0).min() == m and n.sum(0).max() == m
    mean = (n * p + s1).sum(0) / m
    ex2 = (s2 + 2 * p * s1 + n * p * p).sum(0) / m
    return mean, ex2 - mean * mean


# ---- 1x1 convolutions -----------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,K,N,bias,act", [(300, 64, 128, True, 2), (257, 96, 192, False, None), (130, 40, 32, True, 1),
                                            (260, 128, 256, False, 3), (200, 136, 48, True, None), (70, 8, 8, True, 0),
                                            # the 256 x 256 direct-to-LDS kernel (plain operands, K and N >= 256: K tails, row / column tails, three row blocks) and the
                                            # 128 x 256 tiles of the fused forms (N % 256 == 0)
                                            (300, 256, 256, True, 2), (520, 264, 512, False, None), (385, 320, 264, True, 1)])
def test_pointwise_forward_dx_dw(emu, M, K, N, bias, act):
    L = emu
    rng = np.random.default_rng(M + 3 * K + 7 * N)
    xb, xv = rand_bf16(rng, (M, K), 1.5, 0.3)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32) if bias else None
    slope = 0.3
    sc = rng.uniform(0.5, 1.5, K).astype(np.float32) if act is not None else None
    sh = rng.standard_normal(K).astype(np.float32) if act is not None else None
    a = inbn_np(xv, sc, sh, act, slope) if act is not None else xv
    wv = rb(w)
    rows = L.tsii_bf16_stat_rows(M)
    assert rows == (M + 127) // 128
    part = np.zeros((rows, 4, N), np.float32)
    y = np.zeros((M, N), np.uint16)
    ws = WS(L.tsii_bf16_pw_ws_bytes(N, K))
    assert L.tsii_bf16_pw_fwd(P(xb), M, K, P(w), N, P(b), P(sc), P(sh), act or 0, slope, P(part), P(y), P(ws), ws.nbytes, None) == 0, L.tsii_last_error()
    ref = a @ wv.T + (b if bias else 0.0)
    close_bf16(y, ref)
    # the partials describe the values AS STORED
    yv = bf16_val(y).astype(np.float64)
    mean, var = stats_from_part(part, M)
    assert np.abs(mean - yv.mean(0)).max() <= 1e-5 * np.abs(yv).max()
    assert np.abs(var - yv.var(0)).max() <= 1e-4 * yv.var(0).max()

    # dX (+ the BatchNorm-backward reductions of the layer that produced x, K6c)
    dyb, dyv = rand_bf16(rng, (M, N))
    dx = np.zeros((M, K), np.uint16)
    ws = WS(L.tsii_bf16_pw_ws_bytes(N, K))
    assert L.tsii_bf16_pw_bwd_dx(P(dyb), M, N, P(w), K, None, None, None, None, None, 0.0, 0, 0.0, P(dx), None, P(ws), ws.nbytes, None) == 0, L.tsii_last_error()
    rdx = dyv @ wv
    close_bf16(dx, rdx)
    if act is not None:
        mean_ = rng.standard_normal(K).astype(np.float32)
        var_ = rng.uniform(0.5, 2.0, K).astype(np.float32)
        gamma = rng.uniform(0.5, 1.5, K).astype(np.float32)
        beta = rng.standard_normal(K).astype(np.float32)
        eps = 1e-5
        bpart = np.zeros((rows, 2, K), np.float32)
        dx2 = np.zeros((M, K), np.uint16)
        assert L.tsii_bf16_pw_bwd_dx(P(dyb), M, N, P(w), K, P(xb), P(mean_), P(var_), P(gamma), P(beta), eps, act, slope, P(dx2), P(bpart),
                                     P(ws), ws.nbytes, None) == 0, L.tsii_last_error()
        assert np.array_equal(dx2, dx)
        xh = (xv - mean_) / np.sqrt(var_.astype(np.float64) + eps)
        z = xh * gamma + beta
        dz = bf16_val(dx).astype(np.float64) * act_grad_np(z, act, slope)
        s1, s2 = dz.sum(0), (dz * xh).sum(0)
        kink = np.abs(z) < 1e-4          # an fp32 z on the other side of the kink than the float64 one
        assert kink.mean() < 1e-3
        assert np.abs(bpart[:, 0].sum(0, dtype=np.float64) - s1).max() <= 1e-4 * np.abs(dz).sum(0).max() + np.abs(dz * kink).sum(0).max()
        assert np.abs(bpart[:, 1].sum(0, dtype=np.float64) - s2).max() <= 1e-4 * np.abs(dz * xh).sum(0).max() + np.abs(dz * xh * kink).sum(0).max()

    # dW / dbias (fp32 outputs)
    nbytes = L.tsii_bf16_pw_bwd_dw_ws_bytes(M, N, K)
    ws = WS(nbytes)
    dw = np.zeros((N, K), np.float32)
    db = np.zeros(N, np.float32)
    assert L.tsii_bf16_pw_bwd_dw(P(dyb), P(xb), M, N, K, P(sc), P(sh), act or 0, slope, P(dw), P(db) if bias else None, P(ws), nbytes, None) == 0, L.tsii_last_error()
    rdw = dyv.T @ a
    assert np.abs(dw - rdw).max() <= 2e-5 * np.abs(rdw).max()
    if bias:
        assert np.abs(db - dyv.sum(0)).max() <= 1e-5 * np.abs(dyv).sum(0).max()


def test_pointwise_rejects_odd_channel_counts(emu):
    L = emu
    x = np.zeros((16, 12), np.uint16)
    w = np.zeros((8, 12), np.float32)
    y = np.zeros((16, 8), np.uint16)
    ws = WS(1024)
    assert L.tsii_bf16_pw_fwd(P(x), 16, 12, P(w), 8, None, None, None, 0, 0.0, None, P(y), P(ws), ws.nbytes, None) != 0
    assert b"multiples of 8" in L.tsii_last_error()


# ---- dense convolutions (implicit GEMM) -------------------------------------------------------------------------------------
def conv_ref(x, w, geom, bias=None):
    """x [n,h,w,cin], w [cout,cin,kh,kw] float64 -> [n,ho,wo,cout]"""
    kh, kw, sh, sw, ph, pw, dh, dw = geom
    n, h, wd, cin = x.shape
    cout = w.shape[0]
    ho = (h + 2 * ph - dh * (kh - 1) - 1) // sh + 1
    wo = (wd + 2 * pw - dw * (kw - 1) - 1) // sw + 1
    xp = np.zeros((n, h + 2 * ph, wd + 2 * pw, cin))
    xp[:, ph:ph + h, pw:pw + wd] = x
    y = np.zeros((n, ho, wo, cout))
    for ky in range(kh):
        for kx in range(kw):
            patch = xp[:, ky * dh: ky * dh + (ho - 1) * sh + 1: sh, kx * dw: kx * dw + (wo - 1) * sw + 1: sw]
            y += patch @ w[:, :, ky, kx].T
    return y + (bias if bias is not None else 0.0)


def conv_dx_ref(dy, w, geom, h, wd):
    kh, kw, sh, sw, ph, pw, dh, dw = geom
    n, ho, wo, cout = dy.shape
    cin = w.shape[1]
    dxp = np.zeros((n, h + 2 * ph, wd + 2 * pw, cin))
    for ky in range(kh):
        for kx in range(kw):
            dxp[:, ky * dh: ky * dh + (ho - 1) * sh + 1: sh, kx * dw: kx * dw + (wo - 1) * sw + 1: sw] += dy @ w[:, :, ky, kx]
    return dxp[:, ph:ph + h, pw:pw + wd]


def conv_dw_ref(dy, x, geom, kshape):
    kh, kw, sh, sw, ph, pw, dh, dw = geom
    n, h, wd, cin = x.shape
    _, ho, wo, cout = dy.shape
    xp = np.zeros((n, h + 2 * ph, wd + 2 * pw, cin))
    xp[:, ph:ph + h, pw:pw + wd] = x
    g = np.zeros(kshape)
    for ky in range(kh):
        for kx in range(kw):
            patch = xp[:, ky * dh: ky * dh + (ho - 1) * sh + 1: sh, kx * dw: kx * dw + (wo - 1) * sw + 1: sw]
            g[:, :, ky, kx] = np.einsum("nyxo,nyxi->oi", dy, patch)
    return g


@pytest.mark.parametrize("n,h,wd,cin,cout,geom,bias", [
    (2, 12, 10, 16, 32, (3, 3, 1, 1, 1, 1, 1, 1), False),       # 3x3 same
    (1, 14, 14, 32, 136, (3, 3, 1, 1, 3, 3, 3, 3), False),      # ASP: dilation 3
    (2, 9, 11, 24, 8, (3, 3, 1, 1, 1, 1, 1, 1), True),          # padded 1-channel head (cout 8)
    (2, 12, 12, 16, 64, (1, 1, 2, 2, 0, 0, 1, 1), False),       # strided 1x1 shortcut
    (2, 13, 13, 16, 32, (2, 2, 1, 1, 0, 0, 1, 1), False),       # space-to-depth stem
    (1, 16, 16, 8, 16, (3, 3, 2, 2, 1, 1, 1, 1), True),         # strided 3x3
    (2, 13, 11, 256, 256, (3, 3, 1, 1, 2, 2, 2, 2), True),      # 256 x 256 tiles of the gathered forms (forward: cout, dX: cin % 256 == 0; 286 rows)
    (1, 9, 9, 256, 256, (3, 3, 1, 1, 1, 1, 1, 1), False),       # ... and their 128 x 256 form (fewer than 256 rows)
])
def test_dense_forward_dx_dw(emu, n, h, wd, cin, cout, geom, bias):
    L = emu
    kh, kw, sh, sw, ph, pw, dh, dw = geom
    ho = (h + 2 * ph - dh * (kh - 1) - 1) // sh + 1
    wo = (wd + 2 * pw - dw * (kw - 1) - 1) // sw + 1
    rng = np.random.default_rng(n + h + 3 * cin + 5 * cout + kh)
    xb, xv = rand_bf16(rng, (n, h, wd, cin))
    w = (rng.standard_normal((cout, cin, kh, kw)) / np.sqrt(cin * kh * kw)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32) if bias else None
    wv = rb(w)
    m = n * ho * wo
    rows = L.tsii_bf16_stat_rows(m)
    part = np.zeros((rows, 4, cout), np.float32)
    y = np.zeros((n, ho, wo, cout), np.uint16)
    ws = WS(L.tsii_bf16_dense_ws_bytes(cin, cout, kh, kw))
    assert L.tsii_bf16_dense_fwd(P(xb), P(w), P(b), n, h, wd, cin, cout, *geom, ho, wo, P(part), P(y), P(ws), ws.nbytes, None) == 0, L.tsii_last_error()
    ref = conv_ref(xv, wv, geom, b)
    close_bf16(y, ref)
    yv = bf16_val(y).astype(np.float64).reshape(m, cout)
    mean, var = stats_from_part(part, m)
    assert np.abs(mean - yv.mean(0)).max() <= 1e-5 * np.abs(yv).max()
    assert np.abs(var - yv.var(0)).max() <= 1e-4 * yv.var(0).max()

    dyb, dyv = rand_bf16(rng, (n, ho, wo, cout))
    dx = np.zeros((n, h, wd, cin), np.uint16)
    ws = WS(L.tsii_bf16_dense_ws_bytes(cin, cout, kh, kw))
    assert L.tsii_bf16_dense_bwd_dx(P(dyb), P(w), n, h, wd, cin, cout, *geom, ho, wo, P(dx), P(ws), ws.nbytes, None) == 0, L.tsii_last_error()
    close_bf16(dx, conv_dx_ref(dyv, wv, geom, h, wd))

    nbytes = L.tsii_bf16_dense_bwd_dw_ws_bytes(n, ho, wo, cin, cout, kh, kw)
    ws = WS(nbytes)
    dwg = np.zeros((cout, cin, kh, kw), np.float32)
    db = np.zeros(cout, np.float32)
    assert L.tsii_bf16_dense_bwd_dw(P(dyb), P(xb), n, h, wd, cin, cout, *geom, ho, wo, P(dwg), P(db) if bias else None, P(ws), nbytes, None) == 0, L.tsii_last_error()
    rdw = conv_dw_ref(dyv, xv, geom, w.shape)
    assert np.abs(dwg - rdw).max() <= 2e-5 * np.abs(rdw).max()
    if bias:
        assert np.abs(db - dyv.reshape(m, cout).sum(0)).max() <= 1e-5 * np.abs(dyv).reshape(m, cout).sum(0).max()


# ---- depth-wise 3x3 ---------------------------------------------------------------------------------------------------------
def dw_ref(x, w, s, p, d):
    """x [n,h,w,c] float64, w [c,3,3] -> [n,ho,wo,c]"""
    n, h, wd, c = x.shape
    ho = (h + 2 * p - 2 * d - 1) // s + 1
    wo = (wd + 2 * p - 2 * d - 1) // s + 1
    xp = np.zeros((n, h + 2 * p, wd + 2 * p, c))
    xp[:, p:p + h, p:p + wd] = x
    y = np.zeros((n, ho, wo, c))
    for ky in range(3):
        for kx in range(3):
            y += xp[:, ky * d: ky * d + (ho - 1) * s + 1: s, kx * d: kx * d + (wo - 1) * s + 1: s] * w[:, ky, kx]
    return y


def dw_dx_ref(dy, w, s, p, d, h, wd):
    n, ho, wo, c = dy.shape
    dxp = np.zeros((n, h + 2 * p, wd + 2 * p, c))
    for ky in range(3):
        for kx in range(3):
            dxp[:, ky * d: ky * d + (ho - 1) * s + 1: s, kx * d: kx * d + (wo - 1) * s + 1: s] += dy * w[:, ky, kx]
    return dxp[:, p:p + h, p:p + wd]


def dw_dw_ref(dy, a, s, p, d):
    n, h, wd, c = a.shape
    _, ho, wo, _ = dy.shape
    xp = np.zeros((n, h + 2 * p, wd + 2 * p, c))
    xp[:, p:p + h, p:p + wd] = a
    g = np.zeros((c, 3, 3))
    for ky in range(3):
        for kx in range(3):
            g[:, ky, kx] = (dy * xp[:, ky * d: ky * d + (ho - 1) * s + 1: s, kx * d: kx * d + (wo - 1) * s + 1: s]).sum((0, 1, 2))
    return g


@pytest.mark.parametrize("n,h,wd,c,s,d,act,bias", [
    (2, 20, 19, 64, 1, 1, 2, False), (1, 33, 17, 24, 1, 2, None, True), (2, 24, 40, 128, 1, 4, 2, False),
    (2, 22, 21, 40, 2, 1, 2, False), (1, 16, 16, 8, 2, 1, None, True), (1, 9, 300, 16, 1, 1, 1, False),
    (3, 70, 6, 256, 1, 2, 3, False), (1, 40, 36, 32, 1, 4, 3, True), (2, 21, 37, 16, 1, 2, 1, True)])
@pytest.mark.parametrize("lean", [1, 0], ids=["strip", "column"])
def test_depthwise_forward_dx_dw(emu, n, h, wd, c, s, d, act, bias, lean):
    """lean = 0: the marching-column kernels of bf16_dw.hip -- the product's path; lean = 1: the stride-1 layers with dilation 1 / 2 / 4
    on the LDS-slab strip kernel in its bf16-storage form (csrc/dw_lean.h H16, dilation by phases), which measured slower on the chip
    (csrc/dwconv.hip) and is kept as a tested A/B form behind an emulator-only switch.  Same contract, same tolerances."""
    L = emu
    if hasattr(L, "tsii_emu_set_hdw_lean"):
        L.tsii_emu_set_hdw_lean(lean)
    elif lean == 1:
        pytest.skip("the stock library has one path per geometry (the switch exists in the emulator build only)")
    try:
        _depthwise_forward_dx_dw(L, n, h, wd, c, s, d, act, bias)
    finally:
        if hasattr(L, "tsii_emu_set_hdw_lean"):
            L.tsii_emu_set_hdw_lean(0)


def _depthwise_forward_dx_dw(L, n, h, wd, c, s, d, act, bias):
    p = d
    geom = (3, 3, s, s, p, p, d, d)
    ho = (h + 2 * p - 2 * d - 1) // s + 1
    wo = (wd + 2 * p - 2 * d - 1) // s + 1
    rng = np.random.default_rng(n + 3 * h + 5 * c + 11 * s + d)
    slope = 0.3
    xb, xv = rand_bf16(rng, (n, h, wd, c), 1.5, 0.2)
    w = (rng.standard_normal((c, 1, 3, 3)) / 3).astype(np.float32)
    b = rng.standard_normal(c).astype(np.float32) if bias else None
    sc = rng.uniform(0.5, 1.5, c).astype(np.float32) if act is not None else None
    sh = rng.standard_normal(c).astype(np.float32) if act is not None else None
    a = inbn_np(xv, sc, sh, act, slope) if act is not None else xv
    w64 = w.astype(np.float64)[:, 0]
    rows = L.tsii_bf16_dw_stat_rows(n, ho, wo, c, 3, 3, s, s, d, d)
    assert rows > 0
    part = np.full((rows, 4, c), np.nan, np.float32)
    y = np.zeros((n, ho, wo, c), np.uint16)
    assert L.tsii_bf16_dw_fwd(P(xb), P(w), P(b), n, h, wd, c, *geom, ho, wo, P(sc), P(sh), act or 0, slope, P(part), P(y), None) == 0, L.tsii_last_error()
    ref = dw_ref(a, w64, s, p, d) + (b if bias else 0.0)
    close_bf16(y, ref, extra=2e-6)
    assert not np.isnan(part).any()
    yv = bf16_val(y).astype(np.float64).reshape(-1, c)
    mean, var = stats_from_part(part, n * ho * wo)
    assert np.abs(mean - yv.mean(0)).max() <= 1e-5 * np.abs(yv).max()
    assert np.abs(var - yv.var(0)).max() <= 1e-4 * yv.var(0).max()
    # the plain form is the same kernel without the partials
    y2 = np.zeros_like(y)
    assert L.tsii_bf16_dw_fwd(P(xb), P(w), P(b), n, h, wd, c, *geom, ho, wo, P(sc), P(sh), act or 0, slope, None, P(y2), None) == 0
    assert np.array_equal(y2, y)

    dyb, dyv = rand_bf16(rng, (n, ho, wo, c))
    dx = np.zeros((n, h, wd, c), np.uint16)
    assert L.tsii_bf16_dw_bwd_dx(P(dyb), P(w), n, h, wd, c, *geom, ho, wo, None, None, None, None, None, 0.0, 0, 0.0, P(dx), None, None) == 0, L.tsii_last_error()
    close_bf16(dx, dw_dx_ref(dyv, w64, s, p, d, h, wd), extra=2e-6)
    brows = L.tsii_bf16_dw_bwd_stat_rows(n, h, wd, c, *geom)
    assert (brows > 0) == (s == 1)
    if act is not None and brows > 0:
        mean_ = rng.standard_normal(c).astype(np.float32)
        var_ = rng.uniform(0.5, 2.0, c).astype(np.float32)
        gamma = rng.uniform(0.5, 1.5, c).astype(np.float32)
        beta = rng.standard_normal(c).astype(np.float32)
        eps = 1e-5
        bpart = np.full((brows, 2, c), np.nan, np.float32)
        dx2 = np.zeros_like(dx)
        assert L.tsii_bf16_dw_bwd_dx(P(dyb), P(w), n, h, wd, c, *geom, ho, wo, P(xb), P(mean_), P(var_), P(gamma), P(beta), eps, act, slope,
                                     P(dx2), P(bpart), None) == 0, L.tsii_last_error()
        assert np.array_equal(dx2, dx) and not np.isnan(bpart).any()
        xh = (xv - mean_) / np.sqrt(var_.astype(np.float64) + eps)
        z = xh * gamma + beta
        dz = bf16_val(dx).astype(np.float64) * act_grad_np(z, act, slope)
        kink = np.abs(z) < 1e-4
        ax = (0, 1, 2)
        assert np.abs(bpart[:, 0].sum(0, dtype=np.float64) - dz.sum(ax)).max() <= 1e-4 * np.abs(dz).sum(ax).max() + np.abs(dz * kink).sum(ax).max()
        assert np.abs(bpart[:, 1].sum(0, dtype=np.float64) - (dz * xh).sum(ax)).max() <= 1e-4 * np.abs(dz * xh).sum(ax).max() + np.abs(dz * xh * kink).sum(ax).max()

    nbytes = L.tsii_bf16_dw_bwd_dw_ws_bytes(n, ho, wo, c, 3, 3, s, s, d, d)
    ws = WS(nbytes)
    dwg = np.zeros((c, 1, 3, 3), np.float32)
    db = np.zeros(c, np.float32)
    assert L.tsii_bf16_dw_bwd_dw(P(dyb), P(xb), n, h, wd, c, *geom, ho, wo, P(sc), P(sh), act or 0, slope, P(dwg), P(db) if bias else None,
                                 P(ws), nbytes, None) == 0, L.tsii_last_error()
    rdw = dw_dw_ref(dyv, a, s, p, d)
    assert np.abs(dwg[:, 0] - rdw).max() <= 2e-5 * np.abs(rdw).max()
    if bias:
        assert np.abs(db - dyv.sum((0, 1, 2))).max() <= 1e-5 * np.abs(dyv).sum((0, 1, 2)).max()


@pytest.mark.parametrize("n,h,wd,c,k", [(2, 17, 20, 64, 3), (1, 12, 9, 24, 5), (2, 20, 33, 128, 9), (1, 5, 4, 8, 9)])
def test_average_pool(emu, n, h, wd, c, k):
    L = emu
    rng = np.random.default_rng(n + h + c + k)
    xb, xv = rand_bf16(rng, (n, h, wd, c))
    y = np.zeros((n, h, wd, c), np.uint16)
    assert L.tsii_bf16_avgpool(P(xb), n, h, wd, c, k, P(y), None) == 0, L.tsii_last_error()
    r = (k - 1) // 2
    xp = np.zeros((n, h + 2 * r, wd + 2 * r, c))
    xp[:, r:r + h, r:r + wd] = xv
    ref = sum(xp[:, i:i + h, j:j + wd] for i in range(k) for j in range(k)) / (k * k)
    close_bf16(y, ref, extra=2e-6)


# ---- streaming kernels ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,C,act,res", [(1000, 64, 2, True), (333, 24, 0, False), (5000, 512, 3, True), (17, 8, 1, False)])
def test_batchnorm_statistics_apply_backward(emu, M, C, act, res):
    L = emu
    rng = np.random.default_rng(M + C)
    slope = 0.3
    yb, yv = rand_bf16(rng, (M, C), 2.0, 5.0)
    rows = L.tsii_bf16_bn_stat_rows(M, C)
    part = np.full((rows, 4, C), np.nan, np.float32)
    assert L.tsii_bf16_bn_stats(P(yb), M, C, P(part), None) == 0, L.tsii_last_error()
    mean, var = stats_from_part(part, M)
    assert np.abs(mean - yv.mean(0)).max() <= 1e-5 * np.abs(yv).max()
    assert np.abs(var - yv.var(0)).max() <= 1e-4 * yv.var(0).max()
    # ... and tsii_bn_finalize accepts them as it does the conv-emitted partials
    gamma = rng.uniform(0.5, 1.5, C).astype(np.float32)
    beta = rng.standard_normal(C).astype(np.float32)
    eps = 1e-5
    mean32, var32, sc, sh = (np.zeros(C, np.float32) for _ in range(4))
    rmean, rvar = np.zeros(C, np.float32), np.ones(C, np.float32)
    ws = WS(L.tsii_bn_finalize_ws_bytes(rows, C))
    assert L.tsii_bn_finalize(P(part), rows, C, M, P(mean32), P(var32), P(rmean), P(rvar), 0.1, P(gamma), P(beta), eps, P(sc), P(sh), P(ws), ws.nbytes, None) == 0
    assert np.abs(mean32 - yv.mean(0)).max() <= 1e-5 * np.abs(yv).max() and np.abs(var32 - yv.var(0)).max() <= 1e-4 * yv.var(0).max()

    rb_, rv_ = rand_bf16(rng, (M, C)) if res else (None, None)
    out = np.zeros((M, C), np.uint16)
    assert L.tsii_bf16_bn_act_fwd(P(yb), M, C, P(sc), P(sh), act, slope, P(rb_), P(out), None) == 0, L.tsii_last_error()
    z = np.float32(yv.astype(np.float32) * sc + sh).astype(np.float64)
    close_bf16(out, act_np(z, act, slope) + (rv_ if res else 0.0), extra=1e-6)

    for training, with_part in ((1, False), (0, False), (1, True)):
        db, dv = rand_bf16(rng, (M, C))
        xh = (yv - mean32) / np.sqrt(var32.astype(np.float64) + eps)
        zz = xh * gamma + beta
        dz = dv * act_grad_np(zz, act, slope)
        s1, s2 = dz.sum(0), (dz * xh).sum(0)
        ref = gamma / np.sqrt(var32.astype(np.float64) + eps) * (dz - (s1 / M + xh * s2 / M if training else 0.0))
        bp = None
        if with_part:       # partial rows as a consumer's dX kernel would leave them
            bp = np.zeros((3, 2, C), np.float32)
            bp[0, 0], bp[1, 0], bp[2, 0] = s1 * 0.25, s1 * 0.5, s1 * 0.25
            bp[0, 1], bp[2, 1] = s2 * 0.5, s2 * 0.5
        dy = np.zeros((M, C), np.uint16)
        dg, dbt = np.zeros(C, np.float32), np.zeros(C, np.float32)
        ws = WS(L.tsii_bf16_bn_ws_bytes(M, C))
        assert L.tsii_bf16_bn_act_bwd(P(db), P(yb), M, C, P(mean32), P(var32), P(gamma), P(beta), eps, act, slope, training, P(bp), 3 if with_part else 0,
                                      P(dy), P(dg), P(dbt), P(ws), ws.nbytes, None) == 0, L.tsii_last_error()
        kink = np.abs(zz) < 1e-5
        assert kink.mean() < 1e-3
        got = bf16_val(dy).astype(np.float64)
        tol = np.abs(ref) * 2.0 ** -8 + 3e-5 * np.abs(ref).max()
        assert (np.abs(got - ref) > tol)[~kink].sum() == 0
        assert np.abs(dbt - s1).max() <= 1e-4 * np.abs(dz).sum(0).max() + np.abs(dz * kink).sum(0).max()
        assert np.abs(dg - s2).max() <= 1e-4 * np.abs(dz * xh).sum(0).max() + np.abs(dz * xh * kink).sum(0).max()


def test_add_concat_bilinear_casts(emu):
    L = emu
    rng = np.random.default_rng(5)
    n, h, w, c = 2, 5, 7, 24
    ab, av = rand_bf16(rng, (n, h, w, c))
    bb, bv = rand_bf16(rng, (n, h, w, c))
    out = np.zeros((n, h, w, c), np.uint16)
    assert L.tsii_bf16_add_act_fwd(P(ab), P(bb), ab.size, 2, 0.3, P(out), None) == 0, L.tsii_last_error()
    s = av + bv
    close_bf16(out, np.where(s > 0, s, 0.3 * s))
    db, dv = rand_bf16(rng, (n, h, w, c))
    dx = np.zeros_like(out)
    assert L.tsii_bf16_act_bwd(P(db), P(out), out.size, 2, 0.3, P(dx), None) == 0
    close_bf16(dx, dv * np.where(bf16_val(out) > 0, 1.0, 0.3))
    # concat / slice
    big = np.zeros((n * h * w, 24 + 16), np.uint16)
    sb, _ = rand_bf16(rng, (n * h * w, 16))
    assert L.tsii_bf16_copy_channels(P(big), n * h * w, 40, 0, P(ab), 24, 1, None) == 0
    assert L.tsii_bf16_copy_channels(P(big), n * h * w, 40, 24, P(sb), 16, 1, None) == 0
    assert np.array_equal(big[:, :24], ab.reshape(-1, 24)) and np.array_equal(big[:, 24:], sb)
    back = np.zeros_like(sb)
    assert L.tsii_bf16_copy_channels(P(big), n * h * w, 40, 24, P(back), 16, 0, None) == 0
    assert np.array_equal(back, sb)
    # bilinear x2 / x4 against the separable definition (align_corners = False), and the adjoint identity <up(x), g> = <x, up^T(g)>
    for scale in (2, 4):
        y = np.zeros((n, h * scale, w * scale, c), np.uint16)
        assert L.tsii_bf16_bilinear_up_fwd(P(ab), n, h, w, c, scale, P(y), None) == 0

        def taps(o, lim):
            sp = max((o + 0.5) / scale - 0.5, 0.0)
            i0 = min(int(sp), lim - 1)
            return i0, min(i0 + 1, lim - 1), sp - i0
        ref = np.zeros((n, h * scale, w * scale, c))
        for oy in range(h * scale):
            y0, y1, ly = taps(oy, h)
            for ox in range(w * scale):
                x0, x1, lx = taps(ox, w)
                ref[:, oy, ox] = (1 - ly) * ((1 - lx) * av[:, y0, x0] + lx * av[:, y0, x1]) + ly * ((1 - lx) * av[:, y1, x0] + lx * av[:, y1, x1])
        close_bf16(y, ref, extra=1e-6)
        gb, gv = rand_bf16(rng, y.shape)
        gx = np.zeros((n, h, w, c), np.uint16)
        assert L.tsii_bf16_bilinear_up_bwd(P(gb), n, h, w, c, scale, P(gx), None) == 0
        lhs, rhs = (ref * gv).sum(), (av * bf16_val(gx)).sum()
        assert abs(lhs - rhs) <= 1e-2 * (np.abs(ref * gv).sum() ** 0.5 + 1)
    # casts
    f = rng.standard_normal(4096).astype(np.float32)
    hb = np.zeros(4096, np.uint16)
    assert L.tsii_bf16_from_f32(P(f), 4096, P(hb), None) == 0
    assert np.array_equal(hb, bf16_bits(f))
    f2 = np.zeros(4096, np.float32)
    assert L.tsii_bf16_to_f32(P(hb), 4096, P(f2), None) == 0
    assert np.array_equal(f2, bf16_val(hb))
    m8 = np.zeros((100, 8), np.uint16)
    lg = rng.standard_normal(100).astype(np.float32)
    assert L.tsii_bf16_channel_from_f32(P(lg), 100, 8, 0, P(m8), None) == 0
    assert np.array_equal(m8[:, 0], bf16_bits(lg)) and not m8[:, 1:].any()
    lg2 = np.zeros(100, np.float32)
    assert L.tsii_bf16_channel_to_f32(P(m8), 100, 8, 0, P(lg2), None) == 0
    assert np.array_equal(lg2, bf16_val(m8[:, 0]))


def test_stem_space_to_depth(emu):
    L = emu
    rng = np.random.default_rng(9)
    n, h, w, c, pad = 2, 10, 14, 3, 1
    x = rng.standard_normal((n, h, w, c)).astype(np.float32)
    h2, w2 = (h + 2 * pad) // 2, (w + 2 * pad) // 2
    out = np.full((n, h2, w2, 16), 0xFFFF, np.uint16)
    assert L.tsii_bf16_stem_s2d(P(x), n, h, w, c, pad, P(out), None) == 0, L.tsii_last_error()
    xp = np.zeros((n, h + 2 * pad, w + 2 * pad, 4), np.float32)
    xp[:, pad:pad + h, pad:pad + w, :c] = x
    ref = xp.reshape(n, h2, 2, w2, 2, 4).transpose(0, 1, 3, 2, 4, 5).reshape(n, h2, w2, 16)
    assert np.array_equal(out, bf16_bits(ref).reshape(out.shape))
