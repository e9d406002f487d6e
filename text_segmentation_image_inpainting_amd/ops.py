"""autograd glue between the nn.Module mirror and the C ABI (include/tsii_hip.h).

Every op works on NHWC-contiguous fp32 tensors ``[N, H, W, C]`` that live on the GPU and
enqueues hand-written HIP kernels on torch's current stream.  PyTorch only provides the
device memory, the stream and the autograd tape; there is no aten compute and no CPU path.
Masks are carried as ``[N, H, W]`` planes (see ``masks.py``); they never require grad
(models/partial_convolution.py:57 runs the mask path under ``no_grad``).
"""
from typing import NamedTuple, Optional

import torch

from . import _lib
from ._lib import call, ptr

ACT_NONE, ACT_RELU, ACT_LEAKY, ACT_RELU6, ACT_SIGMOID = 0, 1, 2, 3, 4

# A/B switches between a fused / matrix-core form and its other spelling.  Plain module attributes (no environment variables: the
# only process-global configuration the package reads is the library's TSII_GEMM_PRODUCTS / TSII_LIBRARY, include/tsii_hip.h);
# tests monkeypatch them, bench.py sets them with --knob NAME=VALUE, tools/ A/B scripts likewise.  Both spellings of every
# switch are tested against the oracle; the defaults are the fused ones.
# K6c (BatchNorm-backward reductions taken by the kernel that produces the incoming gradient): FUSE_BN_BWD in the depth-wise dX
# strip kernel, FUSE_BN_BWD_PW also in the point-wise dX GEMM epilogue.  Measured on MI355X (ImageFill 512^2 bs 32, split-bf16
# GEMMs): PW on vs off = BatchNorm backward 19.9 -> 16.3 ms for +2.0 ms of GEMM epilogue: 90.5 -> 89.3 ms per step.
FUSE_BN_BWD = True
FUSE_BN_BWD_PW = True
# K6d: the depth-wise dX + K6c pass also takes the layer's weight gradient (tsii_dw_bwd_dxdw_bn; stride 1 / dilation 1, bias-free)
FUSE_DW_DXDW = True
# K6e: ... and applies the backward of the BatchNorm that FOLLOWS the layer while it loads (tsii_dw_bwd_dxdw_bn2): that BatchNorm's
# backward only reduces its K6c partial rows (tsii_bn_bwd_reduce) and hands (gradient, raw input, constants) over; stride 1
FUSE_DW_BN2_FOLD = True
# bf16 storage: where the PLAIN dX product runs on the 256 x 256 direct-to-LDS kernel (k, cout >= 256) the K6c epilogue (register-staged
# 128 x 256 tiles) costs more in a microbenchmark than that kernel plus the stand-alone reduction pass (131072 x 512 x 512: 153-165 us against 88 + 52) --
# in the cfg 5 step it does not (profiles/r05x_k6c_unfuse.log: 60.2 ms fused, 61.0 unfused: the reduction pass reads dx and y cold), so: off
BF16_UNFUSE_K6C_ON_LARGE = False
# the gradient of a K7b up-sampled addend taken inside the BatchNorm-backward apply pass (tsii_bn_act_bwd_pre_pool)
FUSE_POOL_BN_BWD = True
# K4d (head weight gradient on the f32 matrix cores)
USE_HEAD_MFMA = True
# ... and its d low taken in the weight-gradient kernel's pass (tsii_head_cat_bwd_low) instead of by the vector-ALU dX kernel
FUSE_HEAD_DLOW = True
# K4b (stems as a space-to-depth stride-1 conv on the vector-gather GEMM)
USE_STEM_S2D = True


class Geom(NamedTuple):
    kh: int
    kw: int
    sh: int
    sw: int
    ph: int
    pw: int
    dh: int
    dw: int

    def out_hw(self, h, w):
        ho = (h + 2 * self.ph - self.dh * (self.kh - 1) - 1) // self.sh + 1
        wo = (w + 2 * self.pw - self.dw * (self.kw - 1) - 1) // self.sw + 1
        return ho, wo


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def make_geom(kernel_size, stride=1, padding=0, dilation=1) -> Geom:
    (kh, kw), (sh, sw), (ph, pw), (dh, dw) = _pair(kernel_size), _pair(stride), _pair(padding), _pair(dilation)
    return Geom(kh, kw, sh, sw, ph, pw, dh, dw)


# ---------------------------------------------------------------------------------------
# bf16 ACTIVATION STORAGE (BASELINE config 5; include/tsii_hip.h "bf16 activation storage")
# ---------------------------------------------------------------------------------------
# The switch decides ONE thing: the dtype the data layer (a net's stem) hands on.  Everything downstream follows the dtype of
# the tensor it receives -- bf16 tensors go to the tsii_bf16_* entry points, fp32 tensors to the fp32 ones -- so forward,
# backward (autograd's own threads) and recomputation agree without further state.  Parameters, their gradients and the
# BatchNorm statistics are fp32 in both modes.  The partial-convolution family (mask planes) has no bf16 form: it refuses.
BF16 = torch.bfloat16
_ACT_STORAGE = torch.float32


_STORAGE_NAMES = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "f32": torch.float32, "fp32": torch.float32, "float32": torch.float32}


def _storage_dtype(dtype):
    d = _STORAGE_NAMES.get(dtype.lower()) if isinstance(dtype, str) else dtype
    if d not in (torch.float32, torch.bfloat16):
        raise ValueError(f"activation storage: torch.float32 / 'f32' or torch.bfloat16 / 'bf16', got {dtype!r}")
    return d


def set_activation_storage(dtype):
    """torch.float32 / "f32" (default) or torch.bfloat16 / "bf16": the storage type of activations / activation gradients of
    the mask-free segmentation layers, applied by the network's stem from the next forward pass on."""
    global _ACT_STORAGE
    _ACT_STORAGE = _storage_dtype(dtype)


class _StorageScope:
    """``with activation_storage("bf16"): ...`` -- sets the storage type and restores the previous one on exit."""

    def __init__(self, dtype):
        self.dtype, self.prev = _storage_dtype(dtype), None

    def __enter__(self):
        self.prev = _ACT_STORAGE
        set_activation_storage(self.dtype)
        return self.dtype

    def __exit__(self, *exc):
        set_activation_storage(self.prev)
        return False


def activation_storage(dtype=None):
    """Without an argument: the current storage type.  With one: a context manager that selects it for the block."""
    if dtype is None:
        return _ACT_STORAGE
    return _StorageScope(dtype)


def _h(t) -> bool:
    return t is not None and t.dtype == torch.bfloat16


def _no_masks(who, *planes):
    if any(p is not None for p in planes):
        raise NotImplementedError(f"{who}: bf16 activation storage exists for the mask-free (segmentation) layers only; "
                                  "the partial-convolution family keeps fp32 storage")


def _ws(nbytes: int, like: torch.Tensor) -> torch.Tensor:
    return torch.empty(max(4, (int(nbytes) + 3) // 4), dtype=torch.float32, device=like.device)


def _c(t: Optional[torch.Tensor]):
    return None if t is None else t.contiguous()


def _al16(*ts) -> bool:
    """All given tensors start on a 16-byte boundary (the fused strip / vector kernels' requirement; torch
    allocations always do, contiguous views at an odd element offset do not)."""
    return all(t is None or t.data_ptr() % 16 == 0 for t in ts)


# ---------------------------------------------------------------------------------------
# K1 mask planes (no autograd)
# ---------------------------------------------------------------------------------------
def mask_channel_sum(mask_nchw: torch.Tensor, channels=None) -> torch.Tensor:
    """plane[n,h,w] = sum over the first ``channels`` channels of an [N,C,H,W] mask (any strides)."""
    _lib.check_device(mask_nchw)
    n, c, h, w = mask_nchw.shape
    c = c if channels is None else channels
    plane = torch.empty((n, h, w), dtype=torch.float32, device=mask_nchw.device)
    sn, sc, sh, sw = mask_nchw.stride()
    call("tsii_mask_channel_sum", ptr(mask_nchw), n, h, w, c, sn, sh, sw, sc, ptr(plane), _lib.stream())
    return plane


def mask_update(p0, a0, p1, a1, g: Geom, post_scale: float, fill_holes: bool,
                want_denom=True, want_mask=True, want_inv=True):
    """K1: (denom, new_mask, inv) planes at the conv's output resolution."""
    _lib.check_device(p0)
    n, h, w = p0.shape
    ho, wo = g.out_hw(h, w)
    mk = lambda want: torch.empty((n, ho, wo), dtype=torch.float32, device=p0.device) if want else None
    denom, new_mask, inv = mk(want_denom), mk(want_mask), mk(want_inv)
    call("tsii_mask_update", ptr(p0), float(a0), ptr(p1), float(a1), n, h, w, *g, ho, wo,
         float(post_scale), int(bool(fill_holes)), ptr(denom), ptr(new_mask), ptr(inv), _lib.stream())
    return denom, new_mask, inv


def plane_upsample2x(p: torch.Tensor) -> torch.Tensor:
    _lib.check_device(p)
    n, h, w = p.shape
    out = torch.empty((n, 2 * h, 2 * w), dtype=torch.float32, device=p.device)
    call("tsii_plane_upsample2x", ptr(p), n, h, w, ptr(out), _lib.stream())
    return out


# ---------------------------------------------------------------------------------------
# K3 point-wise partial convolution
# ---------------------------------------------------------------------------------------
class _PoolHandOver:
    """K7b backward hand-over: the BatchNorm backward that produces dy for a conv with an up-sampled addend also takes the
    addend's gradient (2x2 sums of dy * inv) in the same pass (tsii_bn_act_bwd_pre_pool) and leaves it here; the conv's own
    backward takes it if the dy it receives is that very tensor, and runs tsii_pool2x2_scaled otherwise."""

    __slots__ = ("inv", "h", "w", "dy", "dz", "want")

    def __init__(self, inv, h, w, want=True):
        self.inv, self.h, self.w, self.dy, self.dz = inv, int(h), int(w), None, None
        self.want = bool(want)      # the addend needs a gradient at all: otherwise nobody would take dz and it would stay pinned

    def take(self, gy):
        dy, dz, self.dy, self.dz = self.dy, self.dz, None, None
        if dy is not None and dy.data_ptr() == gy.data_ptr() and dy.shape == gy.shape:
            return dz
        return None


class _FoldHandOver:
    """K6e hand-over: the BatchNorm that follows a depth-wise layer defers its backward APPLY pass to that layer's dX + dW kernel, which
    forms dy = BatchNorm-backward(da, y) while it stages its slab.  _BNLazy.backward leaves (da, y, coef, act, slope) here and returns da
    unchanged as the "gradient" of the conv output; _Depthwise.backward takes it if the gradient it receives is that very tensor --
    anything else (autograd summed a second consumer's gradient into it) cannot be undone and raises."""

    __slots__ = ("pending",)

    def __init__(self):
        self.pending = None

    def take(self, gy):
        pend, self.pending = self.pending, None
        if pend is None:
            return None
        if pend[0].data_ptr() != gy.data_ptr() or pend[0].shape != gy.shape:
            raise RuntimeError("deferred BatchNorm backward (K6e): the depth-wise layer's output has another consumer than its BatchNorm; "
                               "set ops.FUSE_DW_BN2_FOLD = False for this network")
        return pend


def _apply_deferred_bn(pend):
    """The stand-alone apply pass of a deferred BatchNorm backward: dy from (da, y, coef)."""
    da, y, coef, act, slope = pend
    c = y.shape[-1]
    dy = torch.empty_like(y)
    call("tsii_bn_bwd_apply", ptr(da), ptr(y), y.numel() // c, c, ptr(coef), int(act), float(slope), ptr(dy), _lib.stream())
    return dy


class _Pointwise(torch.autograd.Function):
    """x may be the raw output of the previous conv whose BatchNorm(+act) is applied on load (in_scale / in_shift:
    K6b, constants here -- the BatchNorm gradient flows through _BNLazy); want_stats adds the BatchNorm partial sums
    of y as a second, non-differentiable output."""

    @staticmethod
    def forward(ctx, x, w, bias, r0, r1, denom, keep, inv, split, in_scale, in_shift, in_act, in_slope, want_stats, bn=None, up_add=None):
        _lib.check_device(x, bf16_ok=True)
        ctx.bn = bn     # (mean, var, gamma, beta, eps, slot) of the lazily applied producer BatchNorm (K6c) or None
        ctx.pool = None
        ctx.has_up = up_add is not None   # K7b: [n, h/2, w/2, cout] addend, up-sampled x2 onto the accumulator (differentiable)
        x, w = x.contiguous(), w.contiguous()
        n, h, wd, k = x.shape
        cout = w.shape[0]
        assert w.shape[1] == k and w.shape[2] == 1 and w.shape[3] == 1, "point-wise weight must be [Cout,Cin,1,1]"
        m = n * h * wd
        if _h(x):        # bf16 activation storage: mask-free layers only
            _no_masks("1x1 convolution", r0, r1, denom, keep, up_add)
            L = _lib.lib()
            y = torch.empty((n, h, wd, cout), dtype=BF16, device=x.device)
            part = torch.empty((int(L.tsii_bf16_stat_rows(m)), 4, cout), dtype=torch.float32, device=x.device) if want_stats else None
            wbytes = L.tsii_bf16_pw_ws_bytes(cout, k)
            wws = _ws(wbytes, x)
            call("tsii_bf16_pw_fwd", ptr(x), m, k, ptr(w), cout, ptr(bias), ptr(in_scale), ptr(in_shift), int(in_act), float(in_slope),
                 ptr(part), ptr(y), ptr(wws), wbytes, _lib.stream())
            ctx.save_for_backward(x, w, None, None, None, None, in_scale, in_shift)
            ctx.split, ctx.has_bias, ctx.in_cfg = 0, bias is not None, (int(in_act), float(in_slope))
            ctx.set_materialize_grads(False)
            if want_stats:
                ctx.mark_non_differentiable(part)
                return y, part
            return y
        y = torch.empty((n, h, wd, cout), dtype=torch.float32, device=x.device)
        part = None
        wbytes = _lib.lib().tsii_pw_ws_bytes(cout, k)
        wws = _ws(wbytes, x)
        if up_add is not None:
            assert in_scale is None, "the up-sampled addend has no BatchNorm-on-load form"
            up_add = up_add.contiguous()
            assert tuple(up_add.shape) == (n, h // 2, wd // 2, cout), "up_add must be [n, h/2, w/2, cout]"
            ctx.pool = _PoolHandOver(inv, h, wd, want=ctx.needs_input_grad[15]) if FUSE_POOL_BN_BWD else None
            if want_stats:
                part = torch.empty((_lib.lib().tsii_pw_stat_rows(m), 4, cout), dtype=torch.float32, device=x.device)
            call("tsii_pw_fwd_up", ptr(x), m, k, ptr(w), cout, ptr(bias), ptr(r0), int(split), ptr(r1), ptr(denom), ptr(keep),
                 ptr(up_add), h, wd, ptr(part), ptr(y), ptr(wws), wbytes, _lib.stream())
        elif in_scale is None and not want_stats:
            call("tsii_pw_fwd", ptr(x), m, k, ptr(w), cout, ptr(bias), ptr(r0), int(split), ptr(r1),
                 ptr(denom), ptr(keep), ptr(y), ptr(wws), wbytes, _lib.stream())
        else:
            if want_stats:
                rows = _lib.lib().tsii_pw_stat_rows(m)
                part = torch.empty((rows, 4, cout), dtype=torch.float32, device=x.device)
            call("tsii_pw_fwd_bn", ptr(x), m, k, ptr(w), cout, ptr(bias), ptr(r0), int(split), ptr(r1),
                 ptr(denom), ptr(keep), ptr(in_scale), ptr(in_shift), int(in_act), float(in_slope), ptr(part), ptr(y),
                 ptr(wws), wbytes, _lib.stream())
        ctx.save_for_backward(x, w, r0, r1, inv, keep, in_scale, in_shift)
        ctx.split, ctx.has_bias, ctx.in_cfg = int(split), bias is not None, (int(in_act), float(in_slope))
        ctx.set_materialize_grads(False)   # no zero tensors for the statistics output (one tiny fill kernel each otherwise)
        if want_stats:
            ctx.mark_non_differentiable(part)
            return y, part
        return y

    @staticmethod
    def backward(ctx, gy, *_):
        if gy is None:
            return (None,) * 16
        x, w, r0, r1, inv, keep, in_scale, in_shift = ctx.saved_tensors
        gy = gy.contiguous()
        n, h, wd, k = x.shape
        cout = w.shape[0]
        m = n * h * wd
        dx = dw = db = None
        st = _lib.stream()
        if _h(x):
            L = _lib.lib()
            if gy.dtype != BF16:
                raise RuntimeError("bf16 storage: the incoming gradient of a bf16 layer must be bf16")
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)
                wbytes = L.tsii_bf16_pw_ws_bytes(cout, k)
                wt = _ws(wbytes, x)
                if (ctx.bn is not None and FUSE_BN_BWD_PW and load_time_act(*ctx.in_cfg)
                        and not (BF16_UNFUSE_K6C_ON_LARGE and k >= 256 and cout >= 256 and m >= 256)):
                    mean, var, gamma, beta, eps, slot = ctx.bn
                    part = torch.empty((int(L.tsii_bf16_stat_rows(m)), 2, k), dtype=torch.float32, device=x.device)
                    call("tsii_bf16_pw_bwd_dx", ptr(gy), m, cout, ptr(w), k, ptr(x), ptr(mean), ptr(var), ptr(gamma), ptr(beta), float(eps),
                         ctx.in_cfg[0], ctx.in_cfg[1], ptr(dx), ptr(part), ptr(wt), wbytes, st)
                    slot.part = part
                else:
                    call("tsii_bf16_pw_bwd_dx", ptr(gy), m, cout, ptr(w), k, None, None, None, None, None, 0.0, 0, 0.0, ptr(dx), None,
                         ptr(wt), wbytes, st)
            if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
                dw = torch.empty_like(w)
                db = torch.empty(cout, dtype=torch.float32, device=x.device) if ctx.has_bias else None
                nbytes = L.tsii_bf16_pw_bwd_dw_ws_bytes(m, cout, k)
                ws = _ws(nbytes, x)
                call("tsii_bf16_pw_bwd_dw", ptr(gy), ptr(x), m, cout, k, ptr(in_scale), ptr(in_shift), ctx.in_cfg[0], ctx.in_cfg[1],
                     ptr(dw), ptr(db), ptr(ws), nbytes, st)
            return (dx, dw, db) + (None,) * 13
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)   # gradient w.r.t. the (virtual) normalised input when in_scale is set
            wt = _ws(_lib.lib().tsii_pw_ws_bytes(cout, k), x)
            if ctx.bn is not None and FUSE_BN_BWD_PW and k % 4 == 0 and cout % 4 == 0 and load_time_act(*ctx.in_cfg):
                mean, var, gamma, beta, eps, slot = ctx.bn
                part = torch.empty((int(_lib.lib().tsii_pw_stat_rows(m)), 2, k), dtype=torch.float32, device=x.device)
                call("tsii_pw_bwd_dx_bn", ptr(gy), m, cout, ptr(w), k, ptr(inv), ptr(r0), ctx.split, ptr(r1),
                     ptr(x), ptr(mean), ptr(var), ptr(gamma), ptr(beta), float(eps), ctx.in_cfg[0], ctx.in_cfg[1],
                     ptr(dx), ptr(part), ptr(wt), st)
                slot.part = part
            else:
                call("tsii_pw_bwd_dx", ptr(gy), m, cout, ptr(w), k, ptr(inv), ptr(r0), ctx.split, ptr(r1),
                     ptr(dx), ptr(wt), st)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw = torch.empty_like(w)
            db = torch.empty(cout, dtype=torch.float32, device=x.device) if ctx.has_bias else None
            nbytes = _lib.lib().tsii_pw_bwd_dw_ws_bytes(m, cout, k)
            ws = _ws(nbytes, x)
            if in_scale is None:
                call("tsii_pw_bwd_dw", ptr(gy), ptr(x), m, cout, k, ptr(inv), ptr(keep), ptr(r0), ctx.split, ptr(r1),
                     ptr(dw), ptr(db), ptr(ws), nbytes, st)
            else:
                call("tsii_pw_bwd_dw_bn", ptr(gy), ptr(x), m, cout, k, ptr(inv), ptr(keep), ptr(r0), ctx.split, ptr(r1),
                     ptr(in_scale), ptr(in_shift), ctx.in_cfg[0], ctx.in_cfg[1], ptr(dw), ptr(db), ptr(ws), nbytes, st)
        dz = None
        if ctx.has_up and ctx.needs_input_grad[15]:
            # the addend joins the accumulator before the count division: its gradient is the 2x2 sum of dy * inv
            dz = ctx.pool.take(gy) if ctx.pool is not None else None
            if dz is None:
                dz = torch.empty((n, h // 2, wd // 2, cout), dtype=torch.float32, device=x.device)
                call("tsii_pool2x2_scaled", ptr(gy), ptr(inv), n, h // 2, wd // 2, cout, ptr(dz), st)
        return (dx, dw, db) + (None,) * 12 + (dz,)


def pointwise_up_ok(vc: "VirtualCat", cout: int) -> bool:
    """Can a 1x1 convolution over this virtual concatenation run its low half at low resolution (tsii_pw_fwd_up)?"""
    n, h, w, _ = vc.shape
    return vc.low_plane is not None and cout % 4 == 0 and h % 2 == 0 and w % 4 == 0 and n * h * w < 2 ** 31 and _al16(vc.low, vc.skip)


def pconv_pointwise(x, w, bias=None, r0=None, split=0, r1=None, denom=None, keep=None, inv=None, want_stats=False, up_add=None):
    """y = keep ? ((x*rs) @ w^T [+ up2(up_add)]) / denom + bias : 0   (include/tsii_hip.h, K3 / K7b).  ``x`` may be a LazyBN
    (K6b); with ``want_stats`` returns (y, stat_part)."""
    if up_add is not None:
        if isinstance(x, LazyBN):
            x = x.materialize()
        out = _Pointwise.apply(x, w, bias, r0, r1, denom, keep, inv, split, None, None, 0, 0.0, want_stats, None, up_add)
        y = out[0] if want_stats else out
        pool = getattr(y.grad_fn, "pool", None)
        if pool is not None:
            y._tsii_pool = pool      # read by bn_lazy: the BatchNorm backward over y takes the addend's gradient in its pass
        return out
    if isinstance(x, LazyBN):
        if x.token.shape[-1] % 4 == 0:
            x.consumed()
            bn = (x.mean, x.var, x.gamma, x.beta, x.eps, x.slot) if x.slot is not None else None
            return _Pointwise.apply(x.token, w, bias, r0, r1, denom, keep, inv, split, x.scale, x.shift, x.act, x.slope,
                                    want_stats, bn)
        x = x.materialize()
    return _Pointwise.apply(x, w, bias, r0, r1, denom, keep, inv, split, None, None, 0, 0.0, want_stats)


# ---------------------------------------------------------------------------------------
# K2 depth-wise partial convolution
# ---------------------------------------------------------------------------------------
def dw_stat_rows(x_shape, g: "Geom", dtype=torch.float32) -> int:
    """Partial-sum rows of the fused depth-wise forms; 0 = geometry without the LDS-tiled kernel (unfused only)."""
    n, h, wd, c = x_shape
    ho, wo = g.out_hw(h, wd)
    fn = _lib.lib().tsii_bf16_dw_stat_rows if dtype == BF16 else _lib.lib().tsii_dw_stat_rows
    return int(fn(n, ho, wo, c, g.kh, g.kw, g.sh, g.sw, g.dh, g.dw))


class _Depthwise(torch.autograd.Function):
    """See _Pointwise for in_scale / in_shift / want_stats (K6b)."""

    @staticmethod
    def forward(ctx, x, w, bias, rmask, denom, keep, inv, g, in_scale, in_shift, in_act, in_slope, want_stats, bn=None):
        _lib.check_device(x, bf16_ok=True)
        ctx.bn = bn     # (mean, var, gamma, beta, eps, slot) of the lazily applied producer BatchNorm (K6c) or None
        x, w = x.contiguous(), w.contiguous()
        n, h, wd, c = x.shape
        assert w.shape[0] == c and w.shape[1] == 1, "depth-wise weight must be [C,1,kh,kw]"
        ho, wo = g.out_hw(h, wd)
        if _h(x):        # bf16 activation storage
            _no_masks("depth-wise convolution", rmask, denom, keep)
            y = torch.empty((n, ho, wo, c), dtype=BF16, device=x.device)
            part = torch.empty((dw_stat_rows(x.shape, g, BF16), 4, c), dtype=torch.float32, device=x.device) if want_stats else None
            call("tsii_bf16_dw_fwd", ptr(x), ptr(w), ptr(bias), n, h, wd, c, *g, ho, wo, ptr(in_scale), ptr(in_shift), int(in_act),
                 float(in_slope), ptr(part), ptr(y), _lib.stream())
            ctx.save_for_backward(x, w, None, None, None, in_scale, in_shift)
            ctx.g, ctx.has_bias, ctx.in_cfg = g, bias is not None, (int(in_act), float(in_slope))
            ctx.set_materialize_grads(False)
            if want_stats:
                ctx.mark_non_differentiable(part)
                return y, part
            return y
        y = torch.empty((n, ho, wo, c), dtype=torch.float32, device=x.device)
        ws = _ws(4 * c * g.kh * g.kw, x)
        part = None
        if in_scale is None and not want_stats:
            call("tsii_dw_fwd", ptr(x), ptr(rmask), ptr(w), ptr(bias), ptr(denom), ptr(keep), n, h, wd, c, *g,
                 ho, wo, ptr(y), ptr(ws), _lib.stream())
        else:
            if want_stats:
                part = torch.empty((dw_stat_rows(x.shape, g), 4, c), dtype=torch.float32, device=x.device)
            call("tsii_dw_fwd_bn", ptr(x), ptr(rmask), ptr(w), ptr(bias), ptr(denom), ptr(keep), n, h, wd, c, *g,
                 ho, wo, ptr(in_scale), ptr(in_shift), int(in_act), float(in_slope), ptr(part), ptr(y), ptr(ws),
                 _lib.stream())
        ctx.save_for_backward(x, w, rmask, inv, keep, in_scale, in_shift)
        ctx.g, ctx.has_bias, ctx.in_cfg = g, bias is not None, (int(in_act), float(in_slope))
        ctx.set_materialize_grads(False)
        # K6e: will this layer's backward be the one-pass dX + K6c + dW kernel with a form that applies the FOLLOWING BatchNorm's
        # backward on load?  Decided here, where everything it depends on is known; pconv_depthwise hangs the hand-over on y
        ctx.fold = None
        if (FUSE_DW_BN2_FOLD and FUSE_DW_DXDW and FUSE_BN_BWD and bn is not None and bias is None and in_scale is not None
                and ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and load_time_act(int(in_act), float(in_slope)) and _al16(x, y)
                and int(_lib.lib().tsii_dw_bwd_stat_rows(n, h, wd, c, *g)) > 0 and int(_lib.lib().tsii_dw_bwd_dxdw_fold_ok(n, h, wd, c, *g)) == 1):
            ctx.fold = _FoldHandOver()
        if want_stats:
            ctx.mark_non_differentiable(part)
            return y, part
        return y

    @staticmethod
    def backward(ctx, gy, *_):
        if gy is None:
            return (None,) * 14
        x, w, rmask, inv, keep, in_scale, in_shift = ctx.saved_tensors
        g = ctx.g
        gy = gy.contiguous()
        pend = ctx.fold.take(gy) if getattr(ctx, "fold", None) is not None else None
        if pend is not None:
            # the following BatchNorm deferred its apply pass to this backward (K6e); if the one-pass kernel cannot run after all (a
            # switch changed between forward and backward), the stand-alone apply pass runs here
            L = _lib.lib()
            can = (FUSE_DW_DXDW and FUSE_BN_BWD and ctx.bn is not None and ctx.needs_input_grad[0] and ctx.needs_input_grad[1]
                   and not ctx.has_bias and in_scale is not None and load_time_act(*ctx.in_cfg) and _al16(gy, x)
                   and int(L.tsii_dw_bwd_stat_rows(x.shape[0], x.shape[1], x.shape[2], x.shape[3], *g)) > 0
                   and int(L.tsii_dw_bwd_dxdw_fold_ok(x.shape[0], x.shape[1], x.shape[2], x.shape[3], *g)) == 1)
            if not can:
                gy, pend = _apply_deferred_bn(pend), None
        n, h, wd, c = x.shape
        ho, wo = g.out_hw(h, wd)
        st = _lib.stream()
        dx = dw = db = None
        if _h(x):
            L = _lib.lib()
            if gy.dtype != BF16:
                raise RuntimeError("bf16 storage: the incoming gradient of a bf16 layer must be bf16")
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)
                rows = 0
                if ctx.bn is not None and FUSE_BN_BWD and load_time_act(*ctx.in_cfg):
                    rows = int(L.tsii_bf16_dw_bwd_stat_rows(n, h, wd, c, *g))
                if rows > 0:
                    mean, var, gamma, beta, eps, slot = ctx.bn
                    part = torch.empty((rows, 2, c), dtype=torch.float32, device=x.device)
                    call("tsii_bf16_dw_bwd_dx", ptr(gy), ptr(w), n, h, wd, c, *g, ho, wo, ptr(x), ptr(mean), ptr(var), ptr(gamma), ptr(beta),
                         float(eps), ctx.in_cfg[0], ctx.in_cfg[1], ptr(dx), ptr(part), st)
                    slot.part = part
                else:
                    call("tsii_bf16_dw_bwd_dx", ptr(gy), ptr(w), n, h, wd, c, *g, ho, wo, None, None, None, None, None, 0.0, 0, 0.0,
                         ptr(dx), None, st)
            if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
                dw = torch.empty_like(w)
                db = torch.empty(c, dtype=torch.float32, device=x.device) if ctx.has_bias else None
                nbytes = L.tsii_bf16_dw_bwd_dw_ws_bytes(n, ho, wo, c, g.kh, g.kw, g.sh, g.sw, g.dh, g.dw)
                ws = _ws(nbytes, x)
                call("tsii_bf16_dw_bwd_dw", ptr(gy), ptr(x), n, h, wd, c, *g, ho, wo, ptr(in_scale), ptr(in_shift), ctx.in_cfg[0], ctx.in_cfg[1],
                     ptr(dw), ptr(db), ptr(ws), nbytes, st)
            return (dx, dw, db) + (None,) * 11
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            ws = _ws(4 * c * g.kh * g.kw, x)
            rows = 0
            if ctx.bn is not None and FUSE_BN_BWD and load_time_act(*ctx.in_cfg) and _al16(gy, x):   # K6c: strip paths of the dX grid
                rows = int(_lib.lib().tsii_dw_bwd_stat_rows(n, h, wd, c, *g))
            # K6d: the dX pass also takes the weight gradient (both of its operands are on chip there); bias-free layers only
            dwb = 0
            if rows > 0 and FUSE_DW_DXDW and ctx.needs_input_grad[1] and not ctx.has_bias and in_scale is not None:
                dwb = int(_lib.lib().tsii_dw_bwd_dxdw_ws_bytes(n, h, wd, c, *g))
            if dwb > 0:
                mean, var, gamma, beta, eps, slot = ctx.bn
                part = torch.empty((rows, 2, c), dtype=torch.float32, device=x.device)
                dw = torch.empty_like(w)
                wsd = _ws(dwb, x)
                if pend is not None:
                    _, y2, coef, act2, slope2 = pend      # K6e: gy is still the gradient w.r.t. act2(BatchNorm2(y2))
                    call("tsii_dw_bwd_dxdw_bn2", ptr(gy), ptr(y2), ptr(coef), int(act2), float(slope2), ptr(inv), ptr(w), ptr(rmask),
                         n, h, wd, c, *g, ho, wo, ptr(x), ptr(mean), ptr(var), ptr(gamma), ptr(beta), float(eps), ctx.in_cfg[0], ctx.in_cfg[1],
                         ptr(dx), ptr(part), ptr(dw), ptr(ws), ptr(wsd), dwb, st)
                    slot.part = part
                    return (dx, dw, None) + (None,) * 11
                call("tsii_dw_bwd_dxdw_bn", ptr(gy), ptr(inv), ptr(w), ptr(rmask), n, h, wd, c, *g, ho, wo,
                     ptr(x), ptr(mean), ptr(var), ptr(gamma), ptr(beta), float(eps), ctx.in_cfg[0], ctx.in_cfg[1],
                     ptr(dx), ptr(part), ptr(dw), ptr(ws), ptr(wsd), dwb, st)
                slot.part = part
                return (dx, dw, None) + (None,) * 11
            assert pend is None, "K6e: a deferred BatchNorm backward reached a path without the one-pass kernel"
            if rows > 0:
                mean, var, gamma, beta, eps, slot = ctx.bn
                part = torch.empty((rows, 2, c), dtype=torch.float32, device=x.device)
                call("tsii_dw_bwd_dx_bn", ptr(gy), ptr(inv), ptr(w), ptr(rmask), n, h, wd, c, *g, ho, wo,
                     ptr(x), ptr(mean), ptr(var), ptr(gamma), ptr(beta), float(eps), ctx.in_cfg[0], ctx.in_cfg[1],
                     ptr(dx), ptr(part), ptr(ws), st)
                slot.part = part
            else:
                call("tsii_dw_bwd_dx", ptr(gy), ptr(inv), ptr(w), ptr(rmask), n, h, wd, c, *g, ho, wo,
                     ptr(dx), ptr(ws), st)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw = torch.empty_like(w)
            db = torch.empty(c, dtype=torch.float32, device=x.device) if ctx.has_bias else None
            nbytes = _lib.lib().tsii_dw_bwd_dw_ws_bytes(n, ho, wo, c, g.kh, g.kw)
            ws = _ws(nbytes, x)
            if in_scale is None:
                call("tsii_dw_bwd_dw", ptr(gy), ptr(inv), ptr(keep), ptr(x), ptr(rmask), n, h, wd, c, *g, ho, wo,
                     ptr(dw), ptr(db), ptr(ws), nbytes, st)
            else:
                call("tsii_dw_bwd_dw_bn", ptr(gy), ptr(inv), ptr(keep), ptr(x), ptr(rmask), n, h, wd, c, *g, ho, wo,
                     ptr(in_scale), ptr(in_shift), ctx.in_cfg[0], ctx.in_cfg[1], ptr(dw), ptr(db), ptr(ws), nbytes, st)
        return (dx, dw, db) + (None,) * 11


def pconv_depthwise(x, w, bias, rmask, denom, keep, inv, g: Geom, want_stats=False):
    """``x`` may be a LazyBN; with ``want_stats`` returns (y, stat_part or None) -- None when the geometry has no
    fused form (the caller then takes the statistics with the separate pass)."""
    lazy = isinstance(x, LazyBN)
    src = x.token if lazy else x
    if _h(src) and dw_stat_rows(src.shape, g, BF16) <= 0:
        raise NotImplementedError(f"bf16 storage: depth-wise geometry {tuple(g)} on {tuple(src.shape)} has no kernel (3x3; stride 1 any dilation, stride 2 dilation 1)")
    # the fused forms need the marching-strip kernel: supported geometry AND 16-byte aligned operands
    fusable = (lazy or want_stats) and src.is_contiguous() and _al16(src) and dw_stat_rows(src.shape, g, src.dtype) > 0
    if lazy and not fusable:
        x, lazy = x.materialize(), False
    stats = want_stats and fusable
    if lazy:
        x.consumed()
        bn = (x.mean, x.var, x.gamma, x.beta, x.eps, x.slot) if x.slot is not None else None
        out = _Depthwise.apply(x.token, w, bias, rmask, denom, keep, inv, g, x.scale, x.shift, x.act, x.slope, stats, bn)
        y = out[0] if stats else out
        fold = getattr(y.grad_fn, "fold", None)
        if fold is not None:
            y._tsii_fold = fold      # read by bn_lazy: the BatchNorm over y may leave its backward apply pass to this layer's dX kernel (K6e)
    else:
        out = _Depthwise.apply(x, w, bias, rmask, denom, keep, inv, g, None, None, 0, 0.0, stats)
    if want_stats and not stats:
        return out, None
    return out


# ---------------------------------------------------------------------------------------
# K4 dense partial convolution (general path)
# ---------------------------------------------------------------------------------------
class _Dense(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, mfull, r0, r1, denom, keep, inv, split, g, want_stats=False):
        _lib.check_device(x)
        x, w = x.contiguous(), w.contiguous()
        n, h, wd, cin = x.shape
        cout = w.shape[0]
        assert w.shape[1] == cin, "dense partial conv needs groups == 1"
        ho, wo = g.out_hw(h, wd)
        y = torch.empty((n, ho, wo, cout), dtype=torch.float32, device=x.device)
        L = _lib.lib()
        nbytes = L.tsii_dense_ws_bytes(cin, cout, g.kh, g.kw)
        ws = _ws(nbytes, x)
        rows = int(L.tsii_dense_stat_rows(int(mfull is not None), n, h, wd, cin, cout, *g, ho, wo)) if want_stats else 0
        part = None
        if rows > 0:     # implicit-GEMM path: the BatchNorm statistics partials come out of the epilogue (K6b)
            part = torch.empty((rows, 4, cout), dtype=torch.float32, device=x.device)
            call("tsii_dense_fwd_bn", ptr(x), ptr(mfull), ptr(r0), int(split), ptr(r1), ptr(w), ptr(bias),
                 ptr(denom), ptr(keep), n, h, wd, cin, cout, *g, ho, wo, ptr(part), ptr(y), ptr(ws), nbytes, _lib.stream())
        else:
            call("tsii_dense_fwd", ptr(x), ptr(mfull), ptr(r0), int(split), ptr(r1), ptr(w), ptr(bias),
                 ptr(denom), ptr(keep), n, h, wd, cin, cout, *g, ho, wo, ptr(y), ptr(ws), nbytes, _lib.stream())
        ctx.save_for_backward(x, w, mfull, r0, r1, inv, keep)
        ctx.g, ctx.split, ctx.has_bias = g, int(split), bias is not None
        ctx.set_materialize_grads(False)
        if part is not None:
            ctx.mark_non_differentiable(part)
            return y, part
        return y

    @staticmethod
    def backward(ctx, gy, *_):
        if gy is None:
            return (None,) * 12
        x, w, mfull, r0, r1, inv, keep = ctx.saved_tensors
        g = ctx.g
        gy = gy.contiguous()
        n, h, wd, cin = x.shape
        cout = w.shape[0]
        ho, wo = g.out_hw(h, wd)
        L, st = _lib.lib(), _lib.stream()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            nbytes = L.tsii_dense_ws_bytes(cin, cout, g.kh, g.kw)
            ws = _ws(nbytes, x)
            call("tsii_dense_bwd_dx", ptr(gy), ptr(inv), ptr(w), ptr(mfull), ptr(r0), ctx.split, ptr(r1),
                 n, h, wd, cin, cout, *g, ho, wo, ptr(dx), ptr(ws), nbytes, st)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw = torch.empty_like(w)
            db = torch.empty(cout, dtype=torch.float32, device=x.device) if ctx.has_bias else None
            nbytes = L.tsii_dense_bwd_dw_ws_bytes(n, ho, wo, cin, cout, g.kh, g.kw)
            ws = _ws(nbytes, x)
            call("tsii_dense_bwd_dw", ptr(gy), ptr(inv), ptr(keep), ptr(x), ptr(mfull), ptr(r0), ctx.split, ptr(r1),
                 n, h, wd, cin, cout, *g, ho, wo, ptr(dw), ptr(db), ptr(ws), nbytes, st)
        return dx, dw, db, None, None, None, None, None, None, None, None, None


class _StemS2D(torch.autograd.Function):
    """K4b: odd k, stride 2, pad (k-1)/2 conv over <= 4 input channels as a stride-1 valid conv over the space-to-depth
    image (x*mask folded into the rearrangement), on the vector-gather implicit GEMM.  No input gradient (data layer)."""

    @staticmethod
    def forward(ctx, x, w, bias, mfull, r0, r1, denom, keep, inv, split, g, want_stats):
        _lib.check_device(x)
        x, w = x.contiguous(), w.contiguous()
        n, h, wd, cin = x.shape
        cout, k, pad = w.shape[0], g.kh, g.ph
        ka, c4 = (k + 1) // 2, 4 * cin
        h2, w2 = (h + 2 * pad) // 2, (wd + 2 * pad) // 2
        ho, wo = g.out_hw(h, wd)
        st, L = _lib.stream(), _lib.lib()
        x2 = torch.empty((n, h2, w2, c4), dtype=torch.float32, device=x.device)
        call("tsii_stem_s2d", ptr(x), ptr(mfull), ptr(r0), int(split), ptr(r1), n, h, wd, cin, pad, ptr(x2), st)
        wk = torch.empty((cout, c4, ka, ka), dtype=torch.float32, device=x.device)
        call("tsii_stem_w_fwd", ptr(w), cout, cin, k, ptr(wk), st)
        g2 = Geom(ka, ka, 1, 1, 0, 0, 1, 1)
        assert g2.out_hw(h2, w2) == (ho, wo)
        y = torch.empty((n, ho, wo, cout), dtype=torch.float32, device=x.device)
        nbytes = L.tsii_dense_ws_bytes(c4, cout, ka, ka)
        ws = _ws(nbytes, x)
        rows = int(L.tsii_dense_stat_rows(0, n, h2, w2, c4, cout, *g2, ho, wo)) if want_stats else 0
        part = None
        if rows > 0:
            part = torch.empty((rows, 4, cout), dtype=torch.float32, device=x.device)
            call("tsii_dense_fwd_bn", ptr(x2), None, None, 0, None, ptr(wk), ptr(bias), ptr(denom), ptr(keep),
                 n, h2, w2, c4, cout, *g2, ho, wo, ptr(part), ptr(y), ptr(ws), nbytes, st)
        else:
            call("tsii_dense_fwd", ptr(x2), None, None, 0, None, ptr(wk), ptr(bias), ptr(denom), ptr(keep),
                 n, h2, w2, c4, cout, *g2, ho, wo, ptr(y), ptr(ws), nbytes, st)
        ctx.save_for_backward(x2, inv, keep)
        ctx.cfg = (tuple(w.shape), g2, (n, h2, w2, c4, ho, wo), bias is not None)
        ctx.set_materialize_grads(False)
        if part is not None:
            ctx.mark_non_differentiable(part)
            return y, part
        return y

    @staticmethod
    def backward(ctx, gy, *_):
        if gy is None:
            return (None,) * 12
        x2, inv, keep = ctx.saved_tensors
        wshape, g2, (n, h2, w2, c4, ho, wo), has_bias = ctx.cfg
        cout, cin, k = wshape[0], wshape[1], wshape[2]
        gy = gy.contiguous()
        L, st = _lib.lib(), _lib.stream()
        dw = db = None
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            ka = g2.kh
            dwk = torch.empty((cout, c4, ka, ka), dtype=torch.float32, device=gy.device)
            db = torch.empty(cout, dtype=torch.float32, device=gy.device) if has_bias else None
            nbytes = L.tsii_dense_bwd_dw_ws_bytes(n, ho, wo, c4, cout, ka, ka)
            ws = _ws(nbytes, gy)
            call("tsii_dense_bwd_dw", ptr(gy), ptr(inv), ptr(keep), ptr(x2), None, None, 0, None,
                 n, h2, w2, c4, cout, *g2, ho, wo, ptr(dwk), ptr(db), ptr(ws), nbytes, st)
            dw = torch.empty(wshape, dtype=torch.float32, device=gy.device)
            call("tsii_stem_w_bwd", ptr(dwk), cout, cin, k, ptr(dw), st)
        return (None, dw, db) + (None,) * 9


class _DenseH(torch.autograd.Function):
    """bf16 activation storage: dense k x k convolution (groups == 1, no masks) as implicit GEMM on bf16 operands
    (tsii_bf16_dense_*); channel counts are multiples of 8 (pconv_dense pads a 1-channel head)."""

    @staticmethod
    def forward(ctx, x, w, bias, g, want_stats):
        _lib.check_device(x, bf16_ok=True)
        x, w = x.contiguous(), w.contiguous()
        n, h, wd, cin = x.shape
        cout = w.shape[0]
        assert w.shape[1] == cin, "dense conv needs groups == 1"
        ho, wo = g.out_hw(h, wd)
        L = _lib.lib()
        y = torch.empty((n, ho, wo, cout), dtype=BF16, device=x.device)
        part = torch.empty((int(L.tsii_bf16_stat_rows(n * ho * wo)), 4, cout), dtype=torch.float32, device=x.device) if want_stats else None
        nbytes = L.tsii_bf16_dense_ws_bytes(cin, cout, g.kh, g.kw)
        ws = _ws(nbytes, x)
        call("tsii_bf16_dense_fwd", ptr(x), ptr(w), ptr(bias), n, h, wd, cin, cout, *g, ho, wo, ptr(part), ptr(y), ptr(ws), nbytes, _lib.stream())
        ctx.save_for_backward(x, w)
        ctx.g, ctx.has_bias = g, bias is not None
        ctx.set_materialize_grads(False)
        if want_stats:
            ctx.mark_non_differentiable(part)
            return y, part
        return y

    @staticmethod
    def backward(ctx, gy, *_):
        if gy is None:
            return (None,) * 5
        x, w = ctx.saved_tensors
        g = ctx.g
        gy = gy.contiguous()
        n, h, wd, cin = x.shape
        cout = w.shape[0]
        ho, wo = g.out_hw(h, wd)
        L, st = _lib.lib(), _lib.stream()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            nbytes = L.tsii_bf16_dense_ws_bytes(cin, cout, g.kh, g.kw)
            ws = _ws(nbytes, x)
            call("tsii_bf16_dense_bwd_dx", ptr(gy), ptr(w), n, h, wd, cin, cout, *g, ho, wo, ptr(dx), ptr(ws), nbytes, st)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw = torch.empty_like(w)
            db = torch.empty(cout, dtype=torch.float32, device=x.device) if ctx.has_bias else None
            nbytes = L.tsii_bf16_dense_bwd_dw_ws_bytes(n, ho, wo, cin, cout, g.kh, g.kw)
            ws = _ws(nbytes, x)
            call("tsii_bf16_dense_bwd_dw", ptr(gy), ptr(x), n, h, wd, cin, cout, *g, ho, wo, ptr(dw), ptr(db), ptr(ws), nbytes, st)
        return dx, dw, db, None, None


class _StemS2DH(torch.autograd.Function):
    """bf16 activation storage enters here: the fp32 image goes through the space-to-depth rearrangement (K4b) straight into a
    bf16 tensor (3 -> 4 channels per phase, 16 in all) and the stem runs as a stride-1 valid conv on the bf16 implicit GEMM."""

    @staticmethod
    def forward(ctx, x, w, bias, g, want_stats):
        _lib.check_device(x, bf16_ok=True)
        x, w = x.contiguous(), w.contiguous()
        n, h, wd, cin = x.shape
        cout, k, pad = w.shape[0], g.kh, g.ph
        ka = (k + 1) // 2
        h2, w2 = (h + 2 * pad) // 2, (wd + 2 * pad) // 2
        ho, wo = g.out_hw(h, wd)
        st, L = _lib.stream(), _lib.lib()
        x2 = torch.empty((n, h2, w2, 16), dtype=BF16, device=x.device)
        call("tsii_bf16_stem_s2d", ptr(x), n, h, wd, cin, pad, ptr(x2), st)
        w4 = w.new_zeros((cout, 4, k, k))
        w4[:, :cin] = w
        wk = torch.empty((cout, 16, ka, ka), dtype=torch.float32, device=x.device)
        call("tsii_stem_w_fwd", ptr(w4), cout, 4, k, ptr(wk), st)
        g2 = Geom(ka, ka, 1, 1, 0, 0, 1, 1)
        assert g2.out_hw(h2, w2) == (ho, wo)
        y = torch.empty((n, ho, wo, cout), dtype=BF16, device=x.device)
        part = torch.empty((int(L.tsii_bf16_stat_rows(n * ho * wo)), 4, cout), dtype=torch.float32, device=x.device) if want_stats else None
        nbytes = L.tsii_bf16_dense_ws_bytes(16, cout, ka, ka)
        ws = _ws(nbytes, x)
        call("tsii_bf16_dense_fwd", ptr(x2), ptr(wk), ptr(bias), n, h2, w2, 16, cout, *g2, ho, wo, ptr(part), ptr(y), ptr(ws), nbytes, st)
        ctx.save_for_backward(x2)
        ctx.cfg = (tuple(w.shape), g2, (n, h2, w2, ho, wo), bias is not None)
        ctx.set_materialize_grads(False)
        if want_stats:
            ctx.mark_non_differentiable(part)
            return y, part
        return y

    @staticmethod
    def backward(ctx, gy, *_):
        if gy is None:
            return (None,) * 5
        (x2,) = ctx.saved_tensors
        wshape, g2, (n, h2, w2, ho, wo), has_bias = ctx.cfg
        cout, cin, k = wshape[0], wshape[1], wshape[2]
        gy = gy.contiguous()
        L, st = _lib.lib(), _lib.stream()
        dw = db = None
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            ka = g2.kh
            dwk = torch.empty((cout, 16, ka, ka), dtype=torch.float32, device=gy.device)
            db = torch.empty(cout, dtype=torch.float32, device=gy.device) if has_bias else None
            nbytes = L.tsii_bf16_dense_bwd_dw_ws_bytes(n, ho, wo, 16, cout, ka, ka)
            ws = _ws(nbytes, gy)
            call("tsii_bf16_dense_bwd_dw", ptr(gy), ptr(x2), n, h2, w2, 16, cout, *g2, ho, wo, ptr(dwk), ptr(db), ptr(ws), nbytes, st)
            dw4 = torch.empty((cout, 4, k, k), dtype=torch.float32, device=gy.device)
            call("tsii_stem_w_bwd", ptr(dwk), cout, 4, k, ptr(dw4), st)
            dw = dw4[:, :cin].contiguous()
        return None, dw, db, None, None


class _ChannelToF32(torch.autograd.Function):
    """channel `ch` of a bf16 [n,h,w,c] tensor as an fp32 [n,h,w,1] tensor (the logits of a head whose single output channel
    was padded to 8 for the bf16 kernels); backward writes the fp32 gradient back into that channel, zeros elsewhere."""

    @staticmethod
    def forward(ctx, y, ch):
        _lib.check_device(y, bf16_ok=True)
        y = y.contiguous()
        n, h, w, c = y.shape
        out = torch.empty((n, h, w, 1), dtype=torch.float32, device=y.device)
        call("tsii_bf16_channel_to_f32", ptr(y), n * h * w, c, int(ch), ptr(out), _lib.stream())
        ctx.cfg = (n, h, w, c, int(ch))
        return out

    @staticmethod
    def backward(ctx, g):
        n, h, w, c, ch = ctx.cfg
        g = g.contiguous()
        dy = torch.empty((n, h, w, c), dtype=BF16, device=g.device)
        call("tsii_bf16_channel_from_f32", ptr(g), n * h * w, c, ch, ptr(dy), _lib.stream())
        return dy, None


class _Cast(torch.autograd.Function):
    """fp32 <-> bf16 storage of an activation tensor (numel % 8 == 0); the gradient takes the way back."""

    @staticmethod
    def forward(ctx, x, to_bf16):
        _lib.check_device(x, bf16_ok=True)
        x = x.contiguous()
        ctx.to_bf16 = bool(to_bf16)
        out = torch.empty(x.shape, dtype=BF16 if to_bf16 else torch.float32, device=x.device)
        call("tsii_bf16_from_f32" if to_bf16 else "tsii_bf16_to_f32", ptr(x), x.numel(), ptr(out), _lib.stream())
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        out = torch.empty(g.shape, dtype=torch.float32 if ctx.to_bf16 else BF16, device=g.device)
        call("tsii_bf16_to_f32" if ctx.to_bf16 else "tsii_bf16_from_f32", ptr(g), g.numel(), ptr(out), _lib.stream())
        return out, None


def to_storage(x, dtype):
    """An activation tensor in the given storage type (torch.float32 / torch.bfloat16); a no-op when it already is."""
    if x.dtype == dtype:
        return x
    if {x.dtype, dtype} != {torch.float32, BF16}:
        raise ValueError(f"to_storage: {x.dtype} -> {dtype}")
    return _Cast.apply(x, dtype == BF16)


def _dense_bf16(x, w, bias, g: Geom, want_stats):
    """Dense conv in bf16 storage.  x fp32 is the data layer (a stem: converted on the way in); a single output channel is padded
    to 8 (zero weights) and handed on as an fp32 tensor -- the logits stay fp32."""
    cin, cout = x.shape[-1], w.shape[0]
    if x.dtype == torch.float32:
        if not _bf16_stem_form(x, w, g):
            raise NotImplementedError("bf16 storage: an fp32 input can only enter through a stem (odd k, stride 2, <= 4 channels, no gradient)")
        return _StemS2DH.apply(x, w, bias, g, want_stats)
    if cout % 8 != 0:
        if cout != 1 or want_stats:
            raise NotImplementedError(f"bf16 storage: {cout} output channels (multiples of 8, or a 1-channel head without BatchNorm)")
        w8 = torch.cat([w, w.new_zeros((7,) + tuple(w.shape[1:]))], 0)
        b8 = None if bias is None else torch.cat([bias, bias.new_zeros(7)], 0)
        return _ChannelToF32.apply(_DenseH.apply(x, w8, b8, g, False), 0)
    return _DenseH.apply(x, w, bias, g, want_stats)


def _stem_form(x, w, g: Geom):
    """Geometry of _StemS2D: odd square kernel, stride 2, "same" padding, no dilation, <= 4 input channels, GEMM-sized
    Cout, even padded image, and no gradient wanted for the input."""
    cin, cout = x.shape[-1], w.shape[0]
    return (USE_STEM_S2D and not x.requires_grad and g.kh == g.kw and g.kh % 2 == 1 and g.kh >= 3 and g.sh == g.sw == 2 and
            g.ph == g.pw == (g.kh - 1) // 2 and g.dh == g.dw == 1 and cin <= 4 and cout % 4 == 0 and cout >= 16 and
            (x.shape[1] + 2 * g.ph) % 2 == 0 and (x.shape[2] + 2 * g.pw) % 2 == 0)


def _bf16_stem_form(x, w, g: Geom):
    """Geometry of _StemS2DH, the data layer of bf16 activation storage: as _stem_form, but independent of the fp32 path's
    A/B switch and with the bf16 kernels' channel rule (Cout a multiple of 8)."""
    cin, cout = x.shape[-1], w.shape[0]
    return (not x.requires_grad and g.kh == g.kw and g.kh % 2 == 1 and g.kh >= 3 and g.sh == g.sw == 2 and
            g.ph == g.pw == (g.kh - 1) // 2 and g.dh == g.dw == 1 and cin <= 4 and cout % 8 == 0 and cout >= 16 and
            (x.shape[1] + 2 * g.ph) % 2 == 0 and (x.shape[2] + 2 * g.pw) % 2 == 0)


def pconv_dense(x, w, bias, mfull, r0, split, r1, denom, keep, inv, g: Geom, want_stats=False):
    """With ``want_stats`` returns (y, stat_part or None) -- None when the geometry is not on the implicit-GEMM path."""
    data_layer = (_ACT_STORAGE == BF16 and x.dtype == torch.float32 and x.shape[-1] <= 4 and
                  mfull is None and r0 is None and denom is None)
    if data_layer and not _bf16_stem_form(x, w, g):
        # bf16 storage was asked for and this mask-free image-sized input is where it would begin: running the whole net in
        # fp32 instead would be a silent fallback
        raise NotImplementedError("bf16 activation storage: the data layer cannot enter it (needs an odd-k stride-2 'same' stem over "
                                  f"<= 4 channels with Cout % 8 == 0, even padded size and no input gradient; got k={g.kh}x{g.kw} "
                                  f"stride={g.sh} cin={x.shape[-1]} cout={w.shape[0]} requires_grad={x.requires_grad})")
    if _h(x) or data_layer:
        _no_masks("dense convolution", mfull, r0, r1, denom, keep)
        out = _dense_bf16(x, w, bias, g, want_stats)
        return (out, None) if (want_stats and not isinstance(out, tuple)) else out
    if _stem_form(x, w, g):
        out = _StemS2D.apply(x, w, bias, mfull, r0, r1, denom, keep, inv, split, g, want_stats)
    else:
        out = _Dense.apply(x, w, bias, mfull, r0, r1, denom, keep, inv, split, g, want_stats)
    if want_stats and not isinstance(out, tuple):
        return out, None
    return out


# ---------------------------------------------------------------------------------------
# K6 BatchNorm (+activation, +residual)
# ---------------------------------------------------------------------------------------
class _BNAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, residual, training, momentum, eps, act, slope, part=None):
        _lib.check_device(y)
        y = y.contiguous()
        c = y.shape[-1]
        m = y.numel() // c
        st = _lib.stream()
        if training:
            mean = torch.empty(c, dtype=torch.float32, device=y.device)
            var = torch.empty(c, dtype=torch.float32, device=y.device)
            if part is not None:      # statistics from the partials the producing conv left behind (K6b)
                rows = part.shape[0]
                nbytes = _lib.lib().tsii_bn_finalize_ws_bytes(rows, c)
                ws = _ws(nbytes, y)
                call("tsii_bn_finalize", ptr(part), rows, c, m, ptr(mean), ptr(var), ptr(running_mean), ptr(running_var),
                     float(momentum), None, None, float(eps), None, None, ptr(ws), nbytes, st)
            else:
                nbytes = _lib.lib().tsii_bn_ws_bytes(m, c)
                ws = _ws(nbytes, y)
                call("tsii_bn_stats", ptr(y), m, c, ptr(mean), ptr(var), ptr(running_mean), ptr(running_var),
                     float(momentum), ptr(ws), nbytes, st)
        else:
            mean, var = running_mean, running_var
        residual = _c(residual)
        out = torch.empty_like(y)
        call("tsii_bn_act_fwd", ptr(y), m, c, ptr(mean), ptr(var), ptr(gamma), ptr(beta), float(eps), int(act),
             float(slope), ptr(residual), ptr(out), st)
        ctx.save_for_backward(y, mean, var, gamma, beta)
        ctx.cfg = (bool(training), float(eps), int(act), float(slope), residual is not None)
        return out

    @staticmethod
    def backward(ctx, gout):
        y, mean, var, gamma, beta = ctx.saved_tensors
        training, eps, act, slope, has_res = ctx.cfg
        gout = gout.contiguous()
        c = y.shape[-1]
        m = y.numel() // c
        dy = torch.empty_like(y)
        dgamma = torch.empty(c, dtype=torch.float32, device=y.device)
        dbeta = torch.empty(c, dtype=torch.float32, device=y.device)
        nbytes = _lib.lib().tsii_bn_ws_bytes(m, c)
        ws = _ws(nbytes, y)
        call("tsii_bn_act_bwd", ptr(gout), ptr(y), m, c, ptr(mean), ptr(var), ptr(gamma), ptr(beta), eps, act,
             slope, int(training), ptr(dy), ptr(dgamma), ptr(dbeta), ptr(ws), nbytes, _lib.stream())
        return dy, dgamma, dbeta, None, None, (gout if has_res else None), None, None, None, None, None, None


def bn_act(y, gamma, beta, running_mean, running_var, training, momentum=0.1, eps=1e-5,
           act=ACT_NONE, slope=0.0, residual=None, part=None):
    if _h(y):          # bf16 storage: statistics -> (scale, shift) -> one apply pass, the lazy form written out
        return bn_lazy(y, gamma, beta, running_mean, running_var, training, momentum, eps, act, slope, part).materialize(residual)
    return _BNAct.apply(y, gamma, beta, running_mean, running_var, residual, training, momentum, eps, act, slope, part)


def load_time_act(act, slope) -> bool:
    """Activations the conv kernels can apply while loading (include/tsii_hip.h, K6b)."""
    return act in (ACT_NONE, ACT_RELU, ACT_RELU6) or (act == ACT_LEAKY and 0.0 <= slope <= 1.0)


class _BwdSlot:
    """Hand-over between the backward of a LazyBN's consumer and _BNLazy.backward (K6c): a consumer whose dX kernel
    also took the BatchNorm-backward reductions leaves them here; they are used only if it was the sole consumer."""

    __slots__ = ("consumers", "part")

    def __init__(self):
        self.consumers, self.part = 0, None


class LazyBN:
    """BatchNorm(+activation) output that has not been written to memory (K6b): ``token`` aliases the RAW conv
    output y and carries the autograd edge of the normalised activation a = act(scale*y + shift); consumers that
    can apply (scale, shift, act) while loading take it as is, everything else calls ``materialize()``."""

    __slots__ = ("token", "scale", "shift", "act", "slope", "mean", "var", "gamma", "beta", "eps", "slot")

    def __init__(self, token, scale, shift, act, slope, mean, var, gamma, beta, eps, slot=None):
        self.token, self.scale, self.shift, self.act, self.slope = token, scale, shift, int(act), float(slope)
        self.mean, self.var, self.gamma, self.beta, self.eps = mean, var, gamma, beta, float(eps)
        self.slot = slot

    @property
    def shape(self):
        return self.token.shape

    def consumed(self):
        if self.slot is not None:
            self.slot.consumers += 1

    def materialize(self, residual=None):
        self.consumed()
        return _LazyApply.apply(self.token, self.mean, self.var, self.gamma, self.beta, self.eps, self.act, self.slope,
                                residual, self.scale, self.shift)


class _BNLazy(torch.autograd.Function):
    """Statistics (from the producer's partial sums when given) + (scale, shift); the output token aliases y.
    backward receives the gradient w.r.t. the normalised activation and is the full BatchNorm(+act) backward."""

    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, part, training, momentum, eps, act, slope, slot, pool=None, fold=None):
        _lib.check_device(y, bf16_ok=True)
        y = y.contiguous()
        c = y.shape[-1]
        m = y.numel() // c
        st = _lib.stream()
        dev = y.device
        ctx.slot = slot
        ctx.pool = pool     # _PoolHandOver of the conv that produced y (K7b) or None
        ctx.fold = fold     # _FoldHandOver of the depth-wise conv that produced y (K6e) or None
        scale = torch.empty(c, dtype=torch.float32, device=dev)
        shift = torch.empty(c, dtype=torch.float32, device=dev)
        if training:
            mean = torch.empty(c, dtype=torch.float32, device=dev)
            var = torch.empty(c, dtype=torch.float32, device=dev)
            if part is None and _h(y):       # bf16 storage: the separate statistics pass emits partials in the fused layout
                part = torch.empty((int(_lib.lib().tsii_bf16_bn_stat_rows(m, c)), 4, c), dtype=torch.float32, device=dev)
                call("tsii_bf16_bn_stats", ptr(y), m, c, ptr(part), st)
            if part is not None:
                rows = part.shape[0]
                nbytes = _lib.lib().tsii_bn_finalize_ws_bytes(rows, c)
                ws = _ws(nbytes, y)
                call("tsii_bn_finalize", ptr(part), rows, c, m, ptr(mean), ptr(var), ptr(running_mean),
                     ptr(running_var), float(momentum), ptr(gamma), ptr(beta), float(eps), ptr(scale), ptr(shift),
                     ptr(ws), nbytes, st)
            else:
                nbytes = _lib.lib().tsii_bn_ws_bytes(m, c)
                ws = _ws(nbytes, y)
                call("tsii_bn_stats", ptr(y), m, c, ptr(mean), ptr(var), ptr(running_mean), ptr(running_var),
                     float(momentum), ptr(ws), nbytes, st)
                call("tsii_bn_scale_shift", ptr(mean), ptr(var), ptr(gamma), ptr(beta), float(eps), c, ptr(scale),
                     ptr(shift), st)
        else:
            mean, var = running_mean, running_var
            call("tsii_bn_scale_shift", ptr(mean), ptr(var), ptr(gamma), ptr(beta), float(eps), c, ptr(scale), ptr(shift), st)
        ctx.save_for_backward(y, mean, var, gamma, beta)
        ctx.cfg = (bool(training), float(eps), int(act), float(slope))
        token = y.detach()          # same storage, fresh autograd identity
        ctx.mark_non_differentiable(scale, shift, mean, var)
        ctx.set_materialize_grads(False)
        return token, scale, shift, mean, var

    @staticmethod
    def backward(ctx, ga, *_):
        if ga is None:
            return (None,) * 14
        y, mean, var, gamma, beta = ctx.saved_tensors
        training, eps, act, slope = ctx.cfg
        ga = ga.contiguous()
        c = y.shape[-1]
        m = y.numel() // c
        dgamma = torch.empty(c, dtype=torch.float32, device=y.device)
        dbeta = torch.empty(c, dtype=torch.float32, device=y.device)
        slot = ctx.slot
        part = slot.part if (slot is not None and slot.consumers == 1) else None
        if slot is not None:
            slot.part = None
        if _h(y):
            if ga.dtype != BF16:
                raise RuntimeError("bf16 storage: the incoming gradient of a bf16 BatchNorm must be bf16")
            dy = torch.empty_like(y)
            nbytes = _lib.lib().tsii_bf16_bn_ws_bytes(m, c)
            ws = _ws(nbytes, y)
            call("tsii_bf16_bn_act_bwd", ptr(ga), ptr(y), m, c, ptr(mean), ptr(var), ptr(gamma), ptr(beta), eps, act, slope, int(training),
                 ptr(part), 0 if part is None else part.shape[0], ptr(dy), ptr(dgamma), ptr(dbeta), ptr(ws), nbytes, _lib.stream())
            return (dy, dgamma, dbeta) + (None,) * 11
        pool = ctx.pool
        fold = ctx.fold
        if (fold is not None and part is not None and FUSE_DW_BN2_FOLD and load_time_act(act, slope) and c % 4 == 0 and _al16(ga, y)
                and not (pool is not None and pool.want)):
            # K6e: only the reduction of the K6c partial rows runs here (dgamma, dbeta, the constants' table); the apply pass rides in the
            # producing depth-wise layer's dX + dW kernel, which receives ga itself
            L = _lib.lib()
            coef = torch.empty((6, c), dtype=torch.float32, device=y.device)
            rbytes = L.tsii_bn_bwd_reduce_ws_bytes(part.shape[0], c)
            wsr = _ws(rbytes, y)
            call("tsii_bn_bwd_reduce", ptr(mean), ptr(var), ptr(gamma), ptr(beta), eps, int(training), ptr(part), part.shape[0], m, c,
                 ptr(dgamma), ptr(dbeta), ptr(coef), ptr(wsr), rbytes, _lib.stream())
            fold.pending = (ga, y, coef, act, slope)
            return (ga, dgamma, dbeta) + (None,) * 11
        dy = torch.empty_like(y)
        nbytes = _lib.lib().tsii_bn_ws_bytes(m, c)
        ws = _ws(nbytes, y)
        if part is not None and pool is not None and pool.want and y.dim() == 4 and (y.shape[1], y.shape[2]) == (pool.h, pool.w):
            dz = torch.empty((y.shape[0], pool.h // 2, pool.w // 2, c), dtype=torch.float32, device=y.device)
            call("tsii_bn_act_bwd_pre_pool", ptr(ga), ptr(y), m, c, ptr(mean), ptr(var), ptr(gamma), ptr(beta), eps, act,
                 slope, int(training), ptr(part), part.shape[0], pool.h, pool.w, ptr(pool.inv), ptr(dy), ptr(dz), ptr(dgamma),
                 ptr(dbeta), ptr(ws), nbytes, _lib.stream())
            pool.dy, pool.dz = dy, dz
        elif part is not None:     # K6c: the sole consumer's dX kernel already took sum(dz), sum(dz*xhat)
            call("tsii_bn_act_bwd_pre", ptr(ga), ptr(y), m, c, ptr(mean), ptr(var), ptr(gamma), ptr(beta), eps, act,
                 slope, int(training), ptr(part), part.shape[0], ptr(dy), ptr(dgamma), ptr(dbeta), ptr(ws), nbytes,
                 _lib.stream())
        else:
            call("tsii_bn_act_bwd", ptr(ga), ptr(y), m, c, ptr(mean), ptr(var), ptr(gamma), ptr(beta), eps, act,
                 slope, int(training), ptr(dy), ptr(dgamma), ptr(dbeta), ptr(ws), nbytes, _lib.stream())
        return (dy, dgamma, dbeta) + (None,) * 11


class _LazyApply(torch.autograd.Function):
    """Writes a = act(bn(y)) (+ residual) out; gradient-wise the identity on the token (the BatchNorm backward
    lives in _BNLazy)."""

    @staticmethod
    def forward(ctx, token, mean, var, gamma, beta, eps, act, slope, residual, scale=None, shift=None):
        c = token.shape[-1]
        m = token.numel() // c
        residual = _c(residual)
        out = torch.empty_like(token)
        if _h(token):      # bf16 storage: the (scale, shift) form -- exactly what a load-time consumer of this BatchNorm computes
            if residual is not None and residual.dtype != BF16:
                raise RuntimeError("bf16 storage: the residual of a bf16 BatchNorm must be bf16")
            call("tsii_bf16_bn_act_fwd", ptr(token), m, c, ptr(scale), ptr(shift), int(act), float(slope), ptr(residual), ptr(out), _lib.stream())
        else:
            call("tsii_bn_act_fwd", ptr(token), m, c, ptr(mean), ptr(var), ptr(gamma), ptr(beta), float(eps), int(act),
                 float(slope), ptr(residual), ptr(out), _lib.stream())
        ctx.has_res = residual is not None
        return out

    @staticmethod
    def backward(ctx, gout):
        return (gout,) + (None,) * 7 + ((gout if ctx.has_res else None),) + (None, None)


def bn_lazy(y, gamma, beta, running_mean, running_var, training, momentum=0.1, eps=1e-5, act=ACT_NONE, slope=0.0,
            part=None) -> LazyBN:
    """BatchNorm(+act) of a conv output as a LazyBN; ``part``: the statistics partials the conv left behind."""
    slot = _BwdSlot()
    token, scale, shift, mean, var = _BNLazy.apply(y, gamma, beta, running_mean, running_var, part, training,
                                                   momentum, eps, act, slope, slot, getattr(y, "_tsii_pool", None),
                                                   getattr(y, "_tsii_fold", None))
    return LazyBN(token, scale, shift, act, slope, mean, var, gamma.detach(), beta.detach(), eps, slot)


class _Act(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act, slope):
        _lib.check_device(x)
        x = x.contiguous()
        out = torch.empty_like(x)
        call("tsii_act_fwd", ptr(x), x.numel(), int(act), float(slope), ptr(out), _lib.stream())
        ctx.save_for_backward(x)
        ctx.cfg = (int(act), float(slope))
        return out

    @staticmethod
    def backward(ctx, gout):
        (x,) = ctx.saved_tensors
        gout = gout.contiguous()
        dx = torch.empty_like(x)
        call("tsii_act_bwd", ptr(gout), ptr(x), x.numel(), ctx.cfg[0], ctx.cfg[1], ptr(dx), _lib.stream())
        return dx, None, None


def activation(x, act, slope=0.0):
    return _Act.apply(x, act, slope)


# ---------------------------------------------------------------------------------------
# K7 nearest x2 up-sampling + concat
# ---------------------------------------------------------------------------------------
class _UpCat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, low, skip):
        _lib.check_device(low, skip)
        low, skip = low.contiguous(), skip.contiguous()
        n, h, w, c1 = low.shape
        assert skip.shape[0] == n and skip.shape[1] == 2 * h and skip.shape[2] == 2 * w, "upcat: shape mismatch"
        c2 = skip.shape[3]
        out = torch.empty((n, 2 * h, 2 * w, c1 + c2), dtype=torch.float32, device=low.device)
        call("tsii_upcat_fwd", ptr(low), ptr(skip), n, h, w, c1, c2, ptr(out), _lib.stream())
        ctx.dims = (n, h, w, c1, c2)
        return out

    @staticmethod
    def backward(ctx, gout):
        n, h, w, c1, c2 = ctx.dims
        gout = gout.contiguous()
        dlow = torch.empty((n, h, w, c1), dtype=torch.float32, device=gout.device) if ctx.needs_input_grad[0] else None
        dskip = torch.empty((n, 2 * h, 2 * w, c2), dtype=torch.float32, device=gout.device) if ctx.needs_input_grad[1] else None
        if dlow is not None or dskip is not None:
            call("tsii_upcat_bwd", ptr(gout), n, h, w, c1, c2, ptr(dlow), ptr(dskip), _lib.stream())
        return dlow, dskip


def upcat(low, skip):
    return _UpCat.apply(low, skip)


class VirtualCat:
    """cat(nearest-x2(low), skip) along channels, NOT written: the decoder's DoubleUpSample + torch.cat.  Consumers that read
    the two tensors directly: the few-output-channel head kernels (K4c) and 1x1 partial convolutions, which run their low half
    at low resolution (K7b, include/tsii_hip.h).  Anything else calls ``materialize()`` (= upcat)."""

    __slots__ = ("low", "skip", "low_plane")

    def __init__(self, low, skip, low_plane=None):
        n, h, w, _ = low.shape
        assert skip.shape[0] == n and skip.shape[1] == 2 * h and skip.shape[2] == 2 * w, "upcat: shape mismatch"
        self.low, self.skip = low, skip
        self.low_plane = low_plane      # [n, h, w] mask plane of ``low`` BEFORE up-sampling (K7b: the low half runs at low resolution)

    @property
    def shape(self):
        n, h, w, c2 = self.skip.shape
        return torch.Size((n, h, w, self.low.shape[3] + c2))

    def materialize(self):
        return upcat(self.low, self.skip)


def head_cat_ok(vc: "VirtualCat", cout: int, g) -> bool:
    """Does the fused head (3x3, stride 1, pad 1, <= 4 output channels) exist for this virtual concatenation?"""
    n, h, w, c = vc.shape
    c1 = vc.low.shape[3]
    return tuple(g) == (3, 3, 1, 1, 1, 1, 1, 1) and bool(_lib.lib().tsii_head_cat_ok(n, h, w, c1, c - c1, int(cout)))


class _HeadCat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, low, skip, w, bias, r0, r1, denom, keep, inv, r0_low=None):
        _lib.check_device(low, skip)
        ctx.r0_low = r0_low      # the low tensor's own mask plane (r0 = its nearest-x2 up-sampling), or None when r0 is None
        low, skip, w = low.contiguous(), skip.contiguous(), w.contiguous()
        n, h, wd, c2 = skip.shape
        c1, cout = low.shape[3], w.shape[0]
        y = torch.empty((n, h, wd, cout), dtype=torch.float32, device=low.device)
        if USE_HEAD_MFMA and (r0_low is not None or r0 is None) and _lib.lib().tsii_head_cat_fwd_low_ok(n, h, wd, c1, c2, cout):
            call("tsii_head_cat_fwd_low", ptr(low), ptr(skip), c1, c2, ptr(r0_low), ptr(r1), ptr(w), ptr(bias), ptr(denom), ptr(keep),
                 n, h, wd, cout, ptr(y), _lib.stream())
        else:
            nbytes = _lib.lib().tsii_dense_ws_bytes(c1 + c2, cout, 3, 3)
            ws = _ws(nbytes, low)
            call("tsii_head_cat_fwd", ptr(low), ptr(skip), c1, c2, ptr(r0), ptr(r1), ptr(w), ptr(bias), ptr(denom), ptr(keep),
                 n, h, wd, cout, ptr(y), ptr(ws), nbytes, _lib.stream())
        ctx.save_for_backward(low, skip, w, r0, r1, inv, keep)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        low, skip, w, r0, r1, inv, keep = ctx.saved_tensors
        gy = gy.contiguous()
        n, h, wd, c2 = skip.shape
        c1, cout = low.shape[3], w.shape[0]
        L, st = _lib.lib(), _lib.stream()
        dlow = dskip = dw = db = None
        want_dw = ctx.needs_input_grad[2] or (ctx.has_bias and ctx.needs_input_grad[3])
        low_mask_ok = USE_HEAD_MFMA and (ctx.r0_low is not None or r0 is None)
        if (low_mask_ok and FUSE_HEAD_DLOW and want_dw and ctx.needs_input_grad[0] and not ctx.needs_input_grad[1]
                and L.tsii_head_cat_bwd_low_ok(n, h, wd, c1, c2, cout)):
            # K4d: weight gradient and d low in one pass over dy (the skip is a data tensor: no d skip)
            dw = torch.empty_like(w)
            db = torch.empty(cout, dtype=torch.float32, device=low.device) if ctx.has_bias else None
            dlow = torch.empty_like(low)
            nbytes = L.tsii_dense_bwd_dw_ws_bytes(n, h, wd, c1 + c2, cout, 3, 3)
            ws = _ws(nbytes, low)
            call("tsii_head_cat_bwd_low", ptr(gy), ptr(inv), ptr(keep), ptr(low), ptr(skip), c1, c2, ptr(ctx.r0_low), ptr(r1), ptr(w),
                 n, h, wd, cout, ptr(dw), ptr(db), ptr(dlow), ptr(ws), nbytes, st)
            return dlow, None, dw, db, None, None, None, None, None, None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            dlow = torch.empty_like(low)
            dskip = torch.empty_like(skip) if ctx.needs_input_grad[1] else None
            nbytes = L.tsii_dense_ws_bytes(c1 + c2, cout, 3, 3)
            ws = _ws(nbytes, low)
            call("tsii_head_cat_bwd_dx", ptr(gy), ptr(inv), ptr(w), c1, c2, ptr(r0), ptr(r1), n, h, wd, cout,
                 ptr(dlow), ptr(dskip), ptr(ws), nbytes, st)
            if not ctx.needs_input_grad[0]:
                dlow = None
        if want_dw:
            dw = torch.empty_like(w)
            db = torch.empty(cout, dtype=torch.float32, device=low.device) if ctx.has_bias else None
            nbytes = L.tsii_dense_bwd_dw_ws_bytes(n, h, wd, c1 + c2, cout, 3, 3)
            ws = _ws(nbytes, low)
            if low_mask_ok and L.tsii_head_cat_low_ok(n, h, wd, c1, c2, cout):
                call("tsii_head_cat_bwd_dw_low", ptr(gy), ptr(inv), ptr(keep), ptr(low), ptr(skip), c1, c2, ptr(ctx.r0_low), ptr(r1),
                     n, h, wd, cout, ptr(dw), ptr(db), ptr(ws), nbytes, st)
            else:
                call("tsii_head_cat_bwd_dw", ptr(gy), ptr(inv), ptr(keep), ptr(low), ptr(skip), c1, c2, ptr(r0), ptr(r1),
                     n, h, wd, cout, ptr(dw), ptr(db), ptr(ws), nbytes, st)
        return dlow, dskip, dw, db, None, None, None, None, None, None


def pconv_head_cat(vc: VirtualCat, w, bias, r0, r1, denom, keep, inv):
    """PartialConv 3x3 head over a VirtualCat (K4c): y = keep ? conv(cat * mask) / denom + bias : 0."""
    return _HeadCat.apply(vc.low, vc.skip, w, bias, r0, r1, denom, keep, inv, vc.low_plane if r0 is not None else None)


class _Up2x(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _lib.check_device(x)
        x = x.contiguous()
        n, h, w, c = x.shape
        out = torch.empty((n, 2 * h, 2 * w, c), dtype=torch.float32, device=x.device)
        call("tsii_upcat_fwd", ptr(x), None, n, h, w, c, 0, ptr(out), _lib.stream())
        ctx.dims = (n, h, w, c)
        return out

    @staticmethod
    def backward(ctx, gout):
        n, h, w, c = ctx.dims
        gout = gout.contiguous()
        dlow = torch.empty((n, h, w, c), dtype=torch.float32, device=gout.device)
        call("tsii_upcat_bwd", ptr(gout), n, h, w, c, 0, ptr(dlow), None, _lib.stream())
        return dlow


def upsample2x(x):
    """nearest x2 of an NHWC tensor (DoubleUpSample on x, models/partial_convolution.py:231)."""
    return _Up2x.apply(x)


# ---------------------------------------------------------------------------------------
# general per-channel mask multiply and mean-L1 loss
# ---------------------------------------------------------------------------------------
class _MulMask(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mask):
        _lib.check_device(x, mask)
        x, mask = x.contiguous(), mask.contiguous()
        assert x.shape == mask.shape
        out = torch.empty_like(x)
        call("tsii_mul_mask", ptr(x), ptr(mask), x.numel(), ptr(out), _lib.stream())
        ctx.save_for_backward(mask)
        return out

    @staticmethod
    def backward(ctx, gout):
        (mask,) = ctx.saved_tensors
        gout = gout.contiguous()
        dx = torch.empty_like(gout)
        call("tsii_mul_mask", ptr(gout), ptr(mask), gout.numel(), ptr(dx), _lib.stream())
        return dx, None


def mul_mask(x, mask_full):
    return _MulMask.apply(x, mask_full)


class _L1Mean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        _lib.check_device(a, b)
        a, b = a.contiguous(), b.contiguous()
        assert a.shape == b.shape
        loss = torch.empty(1, dtype=torch.float32, device=a.device)
        nbytes = _lib.lib().tsii_l1_ws_bytes(a.numel())
        ws = _ws(nbytes, a)
        call("tsii_l1_mean_fwd", ptr(a), ptr(b), a.numel(), ptr(loss), ptr(ws), nbytes, _lib.stream())
        ctx.save_for_backward(a, b)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.reshape(1).contiguous()
        da = torch.empty_like(a)
        call("tsii_l1_mean_bwd", ptr(a), ptr(b), a.numel(), ptr(g), ptr(da), _lib.stream())
        return da, None


def l1_mean(a, b):
    """mean(|a - b|); both NHWC-contiguous with identical shape."""
    return _L1Mean.apply(a, b)


def sgd_nesterov_(p, grad, buf, lr, momentum, weight_decay):
    _lib.check_device(p)
    assert p.is_contiguous() and grad.is_contiguous() and buf.is_contiguous()
    call("tsii_sgd_nesterov", ptr(p), ptr(grad), ptr(buf), p.numel(), float(lr), float(momentum),
         float(weight_decay), _lib.stream())


# ---------------------------------------------------------------------------------------
# segmentation path (models/common.py, models/text_segmentation.py, loss.py:58-83)
# ---------------------------------------------------------------------------------------
def conv2d(x, w, b, g: Geom, groups: int, want_stats=False):
    """Plain nn.Conv2d on NHWC data through the partial-conv kernels with no mask planes.  With ``want_stats``
    returns (y, stat_part or None): the BatchNorm partials of y when the kernel taken can emit them (K6b)."""
    cin, cout = x.shape[-1], w.shape[0]
    if groups == 1:
        if tuple(g)[:6] == (1, 1, 1, 1, 0, 0):
            return pconv_pointwise(x, w, b, want_stats=want_stats)
        if isinstance(x, LazyBN):
            x = x.materialize()       # the gather loaders have no load-time BatchNorm
        return pconv_dense(x, w, b, None, None, 0, None, None, None, None, g, want_stats=want_stats)
    if groups == cin == cout:
        return pconv_depthwise(x, w, b, None, None, None, None, g, want_stats=want_stats)
    raise NotImplementedError(f"conv2d groups={groups} (neither 1 nor depth-wise) has no HIP kernel")


_POOL_W = {}


def avg_pool(x, k: int, stride: int, padding: int):
    """nn.AvgPool2d(k, stride, padding), count_include_pad=True: a depth-wise conv with weights 1/k^2."""
    if _h(x):
        if stride != 1 or padding != (k - 1) // 2 or k not in (3, 5, 9):
            raise NotImplementedError(f"bf16 storage: AvgPool2d({k}, {stride}, {padding}) has no kernel (k in 3/5/9, stride 1, same padding)")
        return _AvgPoolH.apply(x, k)
    c = x.shape[-1]
    key = (c, k, x.device)
    w = _POOL_W.get(key)
    if w is None:
        w = _POOL_W[key] = torch.full((c, 1, k, k), 1.0 / (k * k), dtype=torch.float32, device=x.device)
    return pconv_depthwise(x, w, None, None, None, None, None, make_geom(k, stride, padding, 1))


class _AvgPoolH(torch.autograd.Function):
    """bf16 storage: k x k / stride 1 / same-padding average pool (count_include_pad); the operator is its own adjoint."""

    @staticmethod
    def forward(ctx, x, k):
        _lib.check_device(x, bf16_ok=True)
        x = x.contiguous()
        n, h, w, c = x.shape
        y = torch.empty_like(x)
        call("tsii_bf16_avgpool", ptr(x), n, h, w, c, int(k), ptr(y), _lib.stream())
        ctx.k = int(k)
        return y

    @staticmethod
    def backward(ctx, gy):
        gy = gy.contiguous()
        n, h, w, c = gy.shape
        dx = torch.empty_like(gy)
        call("tsii_bf16_avgpool", ptr(gy), n, h, w, c, ctx.k, ptr(dx), _lib.stream())
        return dx, None


class _AddAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, act, slope):
        _lib.check_device(a, b, bf16_ok=True)
        a, b = a.contiguous(), b.contiguous()
        assert a.shape == b.shape and a.dtype == b.dtype, "add: operands must agree in shape and storage type"
        out = torch.empty_like(a)
        call("tsii_bf16_add_act_fwd" if _h(a) else "tsii_add_act_fwd", ptr(a), ptr(b), a.numel(), int(act), float(slope), ptr(out), _lib.stream())
        ctx.cfg = (int(act), float(slope))
        if act != ACT_NONE:
            ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, gout):
        act, slope = ctx.cfg
        gout = gout.contiguous()
        if act == ACT_NONE:
            return gout, gout, None, None
        if act == ACT_SIGMOID:
            raise NotImplementedError("add+sigmoid backward is not used by the reference networks")
        (out,) = ctx.saved_tensors   # sign / range of the output decides the ReLU-family derivative
        g = torch.empty_like(gout)
        call("tsii_bf16_act_bwd" if _h(out) else "tsii_act_bwd", ptr(gout), ptr(out), out.numel(), act, slope, ptr(g), _lib.stream())
        return g, g, None, None


def add_act(a, b, act=ACT_NONE, slope=0.0):
    return _AddAct.apply(a, b, act, slope)


class _Concat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *xs):
        _lib.check_device(*xs, bf16_ok=True)
        xs = [x.contiguous() for x in xs]
        lead = xs[0].shape[:-1]
        chans = [x.shape[-1] for x in xs]
        m = xs[0].numel() // chans[0]
        dt = xs[0].dtype
        assert all(x.dtype == dt for x in xs), "concat: mixed storage types"
        fn = "tsii_bf16_copy_channels" if dt == BF16 else "tsii_copy_channels"
        out = torch.empty(lead + (sum(chans),), dtype=dt, device=xs[0].device)
        off = 0
        for x, c in zip(xs, chans):
            assert x.shape[:-1] == lead
            call(fn, ptr(out), m, sum(chans), off, ptr(x), c, 1, _lib.stream())
            off += c
        ctx.chans, ctx.lead = chans, lead
        return out

    @staticmethod
    def backward(ctx, gout):
        gout = gout.contiguous()
        chans, lead = ctx.chans, ctx.lead
        m = gout.numel() // sum(chans)
        grads, off = [], 0
        for i, c in enumerate(chans):
            if ctx.needs_input_grad[i]:
                g = torch.empty(lead + (c,), dtype=gout.dtype, device=gout.device)
                call("tsii_bf16_copy_channels" if _h(gout) else "tsii_copy_channels", ptr(gout), m, sum(chans), off, ptr(g), c, 0, _lib.stream())
                grads.append(g)
            else:
                grads.append(None)
            off += c
        return tuple(grads)


def concat(xs):
    """torch.cat(dim=1) of NCHW == concat of the last (channel) axis in NHWC."""
    return _Concat.apply(*xs)


class _BilinearUp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale):
        _lib.check_device(x, bf16_ok=True)
        x = x.contiguous()
        n, h, w, c = x.shape
        y = torch.empty((n, h * scale, w * scale, c), dtype=x.dtype, device=x.device)
        call("tsii_bf16_bilinear_up_fwd" if _h(x) else "tsii_bilinear_up_fwd", ptr(x), n, h, w, c, int(scale), ptr(y), _lib.stream())
        ctx.dims = (n, h, w, c, int(scale))
        return y

    @staticmethod
    def backward(ctx, gy):
        n, h, w, c, scale = ctx.dims
        gy = gy.contiguous()
        dx = torch.empty((n, h, w, c), dtype=gy.dtype, device=gy.device)
        call("tsii_bf16_bilinear_up_bwd" if _h(gy) else "tsii_bilinear_up_bwd", ptr(gy), n, h, w, c, scale, ptr(dx), _lib.stream())
        return dx, None


def bilinear_up(x, scale: int):
    """F.interpolate(scale_factor=scale, mode='bilinear', align_corners=False) on NHWC."""
    return _BilinearUp.apply(x, scale)


class _GAP(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _lib.check_device(x)
        x = x.contiguous()
        n, h, w, c = x.shape
        out = torch.empty((n, c), dtype=torch.float32, device=x.device)
        nbytes = _lib.lib().tsii_gap_ws_bytes(n, h * w, c)
        ws = _ws(nbytes, x)
        call("tsii_gap_fwd", ptr(x), n, h * w, c, ptr(out), ptr(ws), nbytes, _lib.stream())
        ctx.dims = (n, h, w, c)
        return out

    @staticmethod
    def backward(ctx, g):
        n, h, w, c = ctx.dims
        g = g.contiguous()
        dx = torch.empty((n, h, w, c), dtype=torch.float32, device=g.device)
        call("tsii_gap_bwd", ptr(g), n, h * w, c, ptr(dx), _lib.stream())
        return dx


def global_avg_pool(x):
    return _GAP.apply(x)


class _SCSE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, cse, sse):
        _lib.check_device(x, cse, sse)
        x, cse, sse = x.contiguous(), cse.contiguous(), sse.contiguous()
        n, h, w, c = x.shape
        out = torch.empty_like(x)
        call("tsii_scse_fwd", ptr(x), ptr(cse), ptr(sse), n, h * w, c, ptr(out), _lib.stream())
        ctx.save_for_backward(x, cse, sse)
        return out

    @staticmethod
    def backward(ctx, g):
        x, cse, sse = ctx.saved_tensors
        g = g.contiguous()
        n, h, w, c = x.shape
        dx, dcse, dsse = torch.empty_like(x), torch.empty_like(cse), torch.empty_like(sse)
        nbytes = _lib.lib().tsii_gap_ws_bytes(n, h * w, c)
        ws = _ws(nbytes, x)
        call("tsii_scse_bwd", ptr(g), ptr(x), ptr(cse), ptr(sse), n, h * w, c, ptr(dx), ptr(dcse), ptr(dsse),
             ptr(ws), nbytes, _lib.stream())
        return dx, dcse, dsse


def scse_combine(x, cse, sse):
    """x*cse[n,c] + x*sse[n,h,w]  (models/common.py:38-43)."""
    return _SCSE.apply(x, cse, sse)


class _BCEFocal(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, t, gamma, bw, ww):
        _lib.check_device(x, t)
        x, t = x.contiguous(), t.contiguous()
        assert x.numel() == t.numel()
        loss = torch.empty(1, dtype=torch.float32, device=x.device)
        nbytes = _lib.lib().tsii_l1_ws_bytes(x.numel())
        ws = _ws(nbytes, x)
        call("tsii_bce_focal_fwd", ptr(x), ptr(t), x.numel(), float(gamma), float(bw), float(ww), ptr(loss), ptr(ws),
             nbytes, _lib.stream())
        ctx.save_for_backward(x, t)
        ctx.cfg = (float(gamma), float(bw), float(ww))
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        x, t = ctx.saved_tensors
        g = g.reshape(1).contiguous()
        dx = torch.empty_like(x)
        call("tsii_bce_focal_bwd", ptr(x), ptr(t), x.numel(), *ctx.cfg, ptr(g), ptr(dx), _lib.stream())
        return dx, None, None, None, None


def bce_focal(x, t, gamma=0.0, background_w=1.0, words_w=2.0):
    return _BCEFocal.apply(x, t, gamma, background_w, words_w)


# ---------------------------------------------------------------------------------------
# InpaintingLoss pieces (loss.py:185-225,294-307)
# ---------------------------------------------------------------------------------------
class _Compose(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw, mask, out):
        _lib.check_device(raw, mask, out)
        raw, mask, out = raw.contiguous(), mask.contiguous(), out.contiguous()
        assert raw.shape == mask.shape == out.shape
        comp = torch.empty_like(out)
        call("tsii_compose_fwd", ptr(raw), ptr(mask), ptr(out), out.numel(), ptr(comp), _lib.stream())
        ctx.save_for_backward(mask)
        return comp

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        g = g.contiguous()
        dout = torch.empty_like(g)
        call("tsii_compose_bwd", ptr(g), ptr(mask), g.numel(), ptr(dout), _lib.stream())
        return None, None, dout


def compose(raw, mask, out):
    """mask*raw + (1-mask)*out; gradient flows to ``out`` only (raw / mask are data)."""
    return _Compose.apply(raw, mask, out)


class _MaskedL1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, out, gt, mask, wv, wh):
        _lib.check_device(out, gt, mask)
        out, gt, mask = out.contiguous(), gt.contiguous(), mask.contiguous()
        assert out.shape == gt.shape == mask.shape
        loss = torch.empty(1, dtype=torch.float32, device=out.device)
        nbytes = _lib.lib().tsii_l1_ws_bytes(out.numel())
        ws = _ws(nbytes, out)
        call("tsii_masked_l1_fwd", ptr(out), ptr(gt), ptr(mask), out.numel(), float(wv), float(wh), ptr(loss), ptr(ws),
             nbytes, _lib.stream())
        ctx.save_for_backward(out, gt, mask)
        ctx.w = (float(wv), float(wh))
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        out, gt, mask = ctx.saved_tensors
        g = g.reshape(1).contiguous()
        dout = torch.empty_like(out)
        call("tsii_masked_l1_bwd", ptr(out), ptr(gt), ptr(mask), out.numel(), *ctx.w, ptr(g), ptr(dout), _lib.stream())
        return dout, None, None, None, None


def masked_l1(out, gt, mask, w_valid=1.0, w_hole=6.0):
    """w_valid*L1(m*out, m*gt) + w_hole*L1((1-m)*out, (1-m)*gt)  (loss.py:199-200,223)."""
    return _MaskedL1.apply(out, gt, mask, w_valid, w_hole)


class _TV(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _lib.check_device(x)
        x = x.contiguous()
        n, h, w, c = x.shape
        loss = torch.empty(1, dtype=torch.float32, device=x.device)
        nbytes = 2 * _lib.lib().tsii_l1_ws_bytes(x.numel())
        ws = _ws(nbytes, x)
        call("tsii_tv_fwd", ptr(x), n, h, w, c, ptr(loss), ptr(ws), nbytes, _lib.stream())
        ctx.save_for_backward(x)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        n, h, w, c = x.shape
        g = g.reshape(1).contiguous()
        dx = torch.empty_like(x)
        call("tsii_tv_bwd", ptr(x), n, h, w, c, ptr(g), ptr(dx), _lib.stream())
        return dx


def total_variation(x):
    return _TV.apply(x)


class _Gram(torch.autograd.Function):
    """gram[n] = F_n^T F_n / (C*H*W) for NHWC features: the TN GEMM (dW form) per sample; backward is the NT GEMM
    dF_n = F_n (dG + dG^T) / (C*H*W)."""

    @staticmethod
    def forward(ctx, f):
        _lib.check_device(f)
        f = f.contiguous()
        n, h, w, c = f.shape
        hw = h * w
        gram = torch.empty((n, c, c), dtype=torch.float32, device=f.device)
        L, st = _lib.lib(), _lib.stream()
        nbytes = L.tsii_pw_bwd_dw_ws_bytes(hw, c, c)
        ws = _ws(nbytes, f)
        for i in range(n):
            fi = f[i]
            call("tsii_pw_bwd_dw", ptr(fi), ptr(fi), hw, c, c, None, None, None, 0, None, ptr(gram[i]), None, ptr(ws), nbytes, st)
        ctx.save_for_backward(f)
        ctx.scale = 1.0 / (c * hw)
        return gram * ctx.scale

    @staticmethod
    def backward(ctx, dg):
        (f,) = ctx.saved_tensors
        n, h, w, c = f.shape
        hw = h * w
        sym = ((dg + dg.transpose(1, 2)) * ctx.scale).contiguous()   # tiny [n,c,c]
        df = torch.empty_like(f)
        st = _lib.stream()
        for i in range(n):
            call("tsii_pw_fwd", ptr(f[i]), hw, c, ptr(sym[i]), c, None, None, 0, None, None, None, ptr(df[i]), None, 0, st)
        return df


def gram_matrix(f_nhwc):
    return _Gram.apply(f_nhwc)


class _PixelShuffle(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, r):
        _lib.check_device(x)
        x = x.contiguous()
        n, h, w, crr = x.shape
        assert crr % (r * r) == 0
        c = crr // (r * r)
        y = torch.empty((n, h * r, w * r, c), dtype=torch.float32, device=x.device)
        call("tsii_pixel_shuffle", ptr(x), n, h, w, c, int(r), 0, ptr(y), _lib.stream())
        ctx.dims = (n, h, w, c, int(r))
        return y

    @staticmethod
    def backward(ctx, gy):
        n, h, w, c, r = ctx.dims
        gy = gy.contiguous()
        dx = torch.empty((n, h, w, c * r * r), dtype=torch.float32, device=gy.device)
        call("tsii_pixel_shuffle", ptr(gy), n, h, w, c, r, 1, ptr(dx), _lib.stream())
        return dx, None


def pixel_shuffle(x, r: int):
    """torch.nn.PixelShuffle(r) semantics on NHWC data (K12; perf-only variant of the decoder head)."""
    return _PixelShuffle.apply(x, r)
