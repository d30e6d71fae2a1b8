"""Thin tensor-level wrappers over the C ABI (no autograd here; see functional.py).

Torch is used only as the owner of device memory and of the current HIP stream: every function
checks its operands, allocates outputs with torch.empty and passes raw device pointers to
libhvc_hip.so.  Non-HIP tensors are rejected -- there is no fallback path.
"""
import math
import os

import torch

from . import _lib
from ._lib import HVC_BF16, HVC_F32, check

_DT = {torch.float32: HVC_F32, torch.bfloat16: HVC_BF16}

# Optional per-call timing (bench.py): when PROFILE is a list, the wrappers bracket their launches
# with HIP events recorded on the launch stream and append (name, algorithmic_work, start, end).
# PROFILE_ONLY (a set of names, or None = all) restricts the bracketing, so the timed region of the
# benchmark can carry events for the dominant kernels only (a few dozen per step).
PROFILE = None
PROFILE_ONLY = None


class _Timed:
    __slots__ = ("name", "work", "start")

    def __init__(self, name, work):
        self.name, self.work, self.start = name, work, None

    def __enter__(self):
        if PROFILE is not None and (PROFILE_ONLY is None or self.name in PROFILE_ONLY):
            self.start = torch.cuda.Event(enable_timing=True)
            self.start.record()
        return self

    def __exit__(self, *exc):
        if self.start is not None:
            end = torch.cuda.Event(enable_timing=True)
            end.record()
            PROFILE.append((self.name, self.work, self.start, end))
        return False


def set_option(name: str, value: int) -> None:
    """Run-time switch of the library (include/hvc_hip.h: hvc_set_option), e.g. set_option("HVC_ATTN_FWD_ROWS", 64)."""
    check(_lib.load().hvc_set_option(name.encode(), int(value)), "hvc_set_option")


def get_option(name: str) -> int:
    import ctypes as C
    v = C.c_int()
    check(_lib.load().hvc_get_option(name.encode(), C.byref(v)), "hvc_get_option")
    return v.value


class options:
    """`with ops.options(HVC_ATTN_FWD_ROWS=64, HVC_ATTN_BWD_WAVES=4): ...` - sets the switches, restores the old values on exit."""

    def __init__(self, **kv):
        self.kv, self.old = kv, {}

    def __enter__(self):
        for k, v in self.kv.items():
            self.old[k] = get_option(k)
            set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            set_option(k, v)
        return False


def _code(dtype):
    try:
        return _DT[dtype]
    except KeyError:
        raise TypeError(f"HVC ops support float32 and bfloat16 tensors, got {dtype}") from None


def _dev(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("HVC ops run on the MI355X HIP path only: got a CPU tensor "
                               "(there is no CPU fallback; use oracle/ for CPU checks)")
    return tensors[0].device


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _f32c(t, name):
    if t is None:
        return None
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise ValueError(f"{name} must be a contiguous float32 tensor")
    return t


# --------------------------------------------------------------------------------------------
# attention
# --------------------------------------------------------------------------------------------
def _bnhd_strides(t):
    """t is a (B, N, H, D) view with unit stride in D -> (sb, sn, sh)."""
    if t.dim() != 4 or t.stride(3) != 1:
        raise ValueError("attention operands must be (B, N, H, D) views with contiguous head dim")
    return t.stride(0), t.stride(1), t.stride(2)


def attention_fwd(q, k, v, scale, p_drop=0.0, seed=0, fp8=False):
    """q: (B,Nq,H,D), k/v: (B,Nk,H,D) views (any batch/token/head strides). Returns o (B,Nq,H,D), lse (B,H,Nq).
    fp8=True runs the Q K^T and P V products as fp8 (e4m3) MFMAs (bf16 operands only; hvc_attention_fwd_fp8)."""
    _dev(q, k, v)
    B, Nq, H, D = q.shape
    Nk = k.shape[1]
    if k.shape != (B, Nk, H, D) or v.shape != (B, Nk, H, D) or not (q.dtype == k.dtype == v.dtype):
        raise ValueError("attention: q/k/v shape or dtype mismatch")
    o = torch.empty((B, Nq, H, D), dtype=q.dtype, device=q.device)
    lse = torch.empty((B, H, Nq), dtype=torch.float32, device=q.device)
    if fp8:
        if q.dtype != torch.bfloat16:
            raise TypeError("fp8 attention takes bfloat16 operands (run it under torch.autocast)")
        lib = _lib.load()
        nbytes = lib.hvc_attention_fwd_fp8_workspace(B, H, Nk, D)
        if nbytes < 0:
            raise ValueError(f"fp8 attention supports head dims 32 and 64 (got D = {D})")
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=q.device)
        with _Timed("attn_fwd_fp8_kernel", 4.0 * B * H * Nq * Nk * D):
            check(lib.hvc_attention_fwd_fp8(
                q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), ws.data_ptr(), B, H, Nq, Nk, D,
                *_bnhd_strides(q), *_bnhd_strides(k), *_bnhd_strides(v), *_bnhd_strides(o),
                float(scale), float(p_drop), int(seed), _stream()), "hvc_attention_fwd_fp8")
        return o, lse
    # the library runs the 64-rows-per-wave kernel from 512 workgroups up (csrc/attention.hip launch_fwd): label the timing so
    label = "attn_fwd_kernel"
    if PROFILE is not None:
        rows_pin = get_option("HVC_ATTN_FWD_ROWS")
        if rows_pin == 64 or (rows_pin != 32 and q.dtype == torch.bfloat16 and ((Nq + 255) // 256) * B * H >= 512):
            label = "attn_fwd2_kernel"
    with _Timed(label, 4.0 * B * H * Nq * Nk * D):
      check(_lib.load().hvc_attention_fwd(
        q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), B, H, Nq, Nk, D,
        *_bnhd_strides(q), *_bnhd_strides(k), *_bnhd_strides(v), *_bnhd_strides(o),
        float(scale), float(p_drop), int(seed), _code(q.dtype), _stream()), "hvc_attention_fwd")
    return o, lse


def attention_bwd(q, k, v, o, dout, lse, scale, p_drop=0.0, seed=0, dq=None, dk=None, dv=None):
    """Gradients w.r.t. q, k, v.  dq/dk/dv may be preallocated views with the strides of q/k/v."""
    _dev(q, k, v, o, dout, lse)
    B, Nq, H, D = q.shape
    Nk = k.shape[1]
    if not o.is_contiguous():
        raise ValueError("attention_bwd: o must be the contiguous (B,Nq,H,D) tensor returned by attention_fwd")
    if dout.shape != o.shape or dout.dtype != o.dtype:
        raise ValueError("attention_bwd: dout must match o")
    dout = dout.contiguous()
    # gradients default to fresh buffers laid out like their primals (a strided view of a packed projection gets a
    # buffer with the same gaps): the ABI gives dq/dk/dv the strides of q/k/v
    if dq is None:
        dq = torch.empty_strided(q.shape, q.stride(), dtype=q.dtype, device=q.device)
    if dk is None:
        dk = torch.empty_strided(k.shape, k.stride(), dtype=k.dtype, device=k.device)
    if dv is None:
        dv = torch.empty_strided(v.shape, v.stride(), dtype=v.dtype, device=v.device)
    for g, x, n in ((dq, q, "dq"), (dk, k, "dk"), (dv, v, "dv")):
        if g.shape != x.shape or any(gs != xs for gs, xs, sz in zip(g.stride(), x.stride(), x.shape) if sz > 1):
            raise ValueError(f"attention_bwd: {n} must have the shape and strides of its primal")
    delta = torch.empty((_lib.load().hvc_attention_bwd_workspace(B, H, Nq, Nk, D),), dtype=torch.float32, device=q.device)
    args = (q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), dout.data_ptr(), lse.data_ptr(),
            delta.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), B, H, Nq, Nk, D,
            *_bnhd_strides(q), *_bnhd_strides(k), *_bnhd_strides(v), *_bnhd_strides(o),
            float(scale), float(p_drop), int(seed))
    fn = _lib.load().hvc_attention_bwd
    base = 2.0 * B * H * Nq * Nk * D          # flops of one Nq x Nk x D product
    if PROFILE is None or (PROFILE_ONLY is not None and "attn_bwd_dkv_kernel" not in PROFILE_ONLY):
        check(fn(*args, 0, _code(q.dtype), _stream()), "hvc_attention_bwd")
    else:
        # time the three launches separately; algorithmic flops: dK/dV kernel = S, dP, dV, dK (4 products),
        # dQ kernel = dQ (its S / dP recompute is overhead of the split, not algorithmic work)
        with _Timed("attn_delta_kernel", 0.0):
            check(fn(*args, 1, _code(q.dtype), _stream()), "hvc_attention_bwd")
        with _Timed("attn_bwd_dkv_kernel", 4 * base):
            check(fn(*args, 2, _code(q.dtype), _stream()), "hvc_attention_bwd")
        with _Timed("attn_bwd_dq_kernel", 1 * base):
            check(fn(*args, 4, _code(q.dtype), _stream()), "hvc_attention_bwd")
    return dq, dk, dv


# --------------------------------------------------------------------------------------------
# gemm
# --------------------------------------------------------------------------------------------
ACT_NONE, ACT_GELU, ACT_GELU_GRAD = 0, 1, 2


def _unit_inner(t):
    return t.shape[1] == 1 or t.stride(1) == 1


def _ld(t):
    """Leading dimension of a 2-D tensor (torch reports arbitrary strides for size-1 dims)."""
    return t.stride(0) if t.shape[0] > 1 else t.shape[1]


def gemm(a, b, *, a_kmajor=False, b_kmajor=False, alpha=1.0, bias=None, act=ACT_NONE, aux=None, zsave=None,
         gate=None, residual=None, residual_rows=0, rows_per_batch=0, p_drop=0.0, seed=0, out_dtype=None, out=None):
    """C[i][j] = sum_k A(i,k) B(j,k) with the fused epilogue of hvc_gemm.

    a: (M,K) if not a_kmajor else (K,M);  b: (N,K) if not b_kmajor else (K,N); both 2-D with unit
    inner stride.  Returns C (M,N)."""
    _dev(a, b, bias, aux, zsave, gate, residual, out)
    if a.dim() != 2 or b.dim() != 2 or not (_unit_inner(a) and _unit_inner(b)):
        raise ValueError("gemm operands must be 2-D with unit inner stride")
    if a.dtype != b.dtype:
        raise ValueError("gemm operands must share a dtype")
    M, K = (a.shape[1], a.shape[0]) if a_kmajor else a.shape
    N, Kb = (b.shape[1], b.shape[0]) if b_kmajor else b.shape
    if K != Kb:
        raise ValueError(f"gemm: contraction mismatch {K} vs {Kb}")
    out_dtype = out_dtype or a.dtype
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=a.device)
    elif out.shape != (M, N) or not _unit_inner(out) or out.dtype != out_dtype:
        raise ValueError("gemm: bad out tensor")
    if aux is not None and (aux.shape != (M, N) or aux.dtype != out_dtype or _ld(aux) != _ld(out) or not _unit_inner(aux)):
        raise ValueError("gemm: aux must match the output tensor")
    _f32c(bias, "bias"), _f32c(gate, "gate")
    if zsave is not None and (zsave.shape != (M, N) or zsave.dtype != a.dtype or not _unit_inner(zsave)):
        raise ValueError("gemm: zsave must be (M,N) in the operand dtype")
    if residual is not None and (residual.dtype != torch.float32 or residual.shape != (residual_rows or M, N) or not _unit_inner(residual)):
        raise ValueError("gemm: residual must be fp32 (M,N) (or (residual_rows,N))")
    lib = _lib.load()
    ws, ws_n = None, 0
    if bias is None and act == ACT_NONE and gate is None and residual is None and zsave is None and p_drop == 0.0:
        ws_n = lib.hvc_gemm_workspace(M, N, K)
        if ws_n > 0:
            ws = torch.empty((ws_n,), dtype=torch.float32, device=a.device)
    with _Timed("gemm_kernel", 2.0 * M * N * K):
      check(_lib.load().hvc_gemm(
        a.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, _ld(a), _ld(b), _ld(out),
        int(a_kmajor), int(b_kmajor), float(alpha), _ptr(bias), int(act), _ptr(aux), _ptr(zsave),
        _ld(zsave) if zsave is not None else 0, _ptr(gate),
        _ptr(residual), _ld(residual) if residual is not None else 0, int(residual_rows), int(rows_per_batch),
        float(p_drop), int(seed), _ptr(ws), int(ws_n), _code(a.dtype), _code(out_dtype), _stream()), "hvc_gemm")
    return out


# --------------------------------------------------------------------------------------------
# layernorm
# --------------------------------------------------------------------------------------------
def layernorm_fwd(x, gamma, beta, scale=None, shift=None, *, rows_per_batch=None, eps=1e-5, out_dtype=torch.float32):
    """x: (rows, C) fp32.  Returns y (rows, C) out_dtype, mean (rows,), rstd (rows,)."""
    _dev(x, gamma, beta, scale, shift)
    _f32c(x, "x"), _f32c(gamma, "gamma"), _f32c(beta, "beta"), _f32c(scale, "scale"), _f32c(shift, "shift")
    rows, Cn = x.shape
    rpb = rows_per_batch or rows
    y = torch.empty((rows, Cn), dtype=out_dtype, device=x.device)
    mean = torch.empty((rows,), dtype=torch.float32, device=x.device)
    rstd = torch.empty((rows,), dtype=torch.float32, device=x.device)
    check(_lib.load().hvc_layernorm_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), _ptr(scale), _ptr(shift),
                                        y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), rows, Cn, rpb, float(eps),
                                        _code(out_dtype), _stream()), "hvc_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy, x, gamma, beta, scale, mean, rstd, *, dres=None, rows_per_batch=None):
    """Returns dx (fp32, includes dres), dgamma, dbeta, dscale, dshift (the last two None without scale)."""
    _dev(dy, x, gamma, beta, scale, mean, rstd, dres)
    rows, Cn = x.shape
    rpb = rows_per_batch or rows
    if not dy.is_contiguous() or dy.shape != x.shape:
        raise ValueError("layernorm_bwd: dy must be contiguous (rows, C)")
    _f32c(dres, "dres")
    lib = _lib.load()
    ws = torch.empty((lib.hvc_layernorm_bwd_workspace(rows, Cn, rpb),), dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x)
    dgamma = torch.empty_like(gamma)
    dbeta = torch.empty_like(beta)
    dscale = torch.empty_like(scale) if scale is not None else None
    dshift = torch.empty_like(scale) if scale is not None else None
    check(lib.hvc_layernorm_bwd(dy.data_ptr(), x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), _ptr(scale),
                                mean.data_ptr(), rstd.data_ptr(), _ptr(dres), dx.data_ptr(), dgamma.data_ptr(),
                                dbeta.data_ptr(), _ptr(dscale), _ptr(dshift), ws.data_ptr(), rows, Cn, rpb,
                                _code(dy.dtype), _stream()), "hvc_layernorm_bwd")
    return dx, dgamma, dbeta, dscale, dshift


# --------------------------------------------------------------------------------------------
# residual-branch backward, column sums, casts
# --------------------------------------------------------------------------------------------
def branch_bwd(dy, z=None, gate=None, *, rows_per_batch=None, out_dtype=torch.float32, want_bias=True,
               p_drop=0.0, seed=0):
    """dy: (rows,N) fp32.  Returns dz (rows,N) out_dtype, dgate (nbatch,N) or None, dbias (N,) or None."""
    _dev(dy, z, gate)
    _f32c(dy, "dy"), _f32c(gate, "gate")
    rows, N = dy.shape
    rpb = rows_per_batch or rows
    if z is not None and (z.shape != dy.shape or not z.is_contiguous() or z.dtype != out_dtype):
        raise ValueError("branch_bwd: z must be contiguous (rows,N) in out_dtype")
    lib = _lib.load()
    ws = torch.empty((lib.hvc_branch_bwd_workspace(rows, N, rpb),), dtype=torch.float32, device=dy.device)
    dz = torch.empty((rows, N), dtype=out_dtype, device=dy.device)
    dgate = torch.empty((rows // rpb, N), dtype=torch.float32, device=dy.device) if (gate is not None and z is not None) else None
    dbias = torch.empty((N,), dtype=torch.float32, device=dy.device) if want_bias else None
    check(lib.hvc_branch_bwd(dy.data_ptr(), _ptr(z), _ptr(gate), dz.data_ptr(), _ptr(dgate), _ptr(dbias), ws.data_ptr(),
                             rows, N, rpb, float(p_drop), int(seed), _code(out_dtype), _stream()), "hvc_branch_bwd")
    return dz, dgate, dbias


def colsum(x):
    _dev(x)
    if x.dim() != 2 or not x.is_contiguous():
        raise ValueError("colsum: contiguous 2-D tensor expected")
    M, N = x.shape
    lib = _lib.load()
    if N % 8 and N < 128 and M >= 4096:
        # narrow ragged matrices (the bias gradient of a 1-channel convolution output is a (16.7 M, 1) column at 256^3): r rows
        # are folded into one row of r N columns, a multiple of 8, so that the 16-byte vector path sums them (column j of the
        # folded matrix = column j % N of x); the r partial sums per column are added at the end.  33 MB: 2 ms -> ~20 us.
        r = 8 // math.gcd(8, N)
        while 2 * r * N <= 256 and M % (2 * r) == 0:
            r *= 2
        if M % r == 0:
            return colsum(x.view(M // r, r * N)).view(r, N).sum(dim=0)
    ws = torch.empty((lib.hvc_colsum_workspace(M, N),), dtype=torch.float32, device=x.device)
    out = torch.empty((N,), dtype=torch.float32, device=x.device)
    check(lib.hvc_colsum(x.data_ptr(), out.data_ptr(), ws.data_ptr(), M, N, _code(x.dtype), _stream()), "hvc_colsum")
    return out


def cast(x, dtype):
    _dev(x)
    if x.dtype == dtype:
        return x
    xc = x.contiguous()
    y = torch.empty(xc.shape, dtype=dtype, device=x.device)
    if xc.numel():
        check(_lib.load().hvc_cast(xc.data_ptr(), y.data_ptr(), xc.numel(), _code(xc.dtype), _code(dtype), _stream()), "hvc_cast")
    return y


# --------------------------------------------------------------------------------------------
# DRR
# --------------------------------------------------------------------------------------------
def drr_fwd(vol, axis, *, exp_mode, mu=0.3, out_scale=1.0, clamp_min=-math.inf, transpose_out=False):
    """vol: (B,D,H,W) contiguous.  axis 0 -> (B,H,W); axis 2 -> (B,D,H) or (B,H,D) with transpose_out."""
    _dev(vol)
    if vol.dim() != 4 or not vol.is_contiguous():
        raise ValueError("drr: contiguous (B,D,H,W) volume expected")
    B, D, H, W = vol.shape
    if axis == 0:
        shape = (B, H, W)
    elif axis == 2:
        shape = (B, H, D) if transpose_out else (B, D, H)
    else:
        raise ValueError("drr: axis must be 0 or 2")
    out = torch.empty(shape, dtype=vol.dtype, device=vol.device)
    check(_lib.load().hvc_drr_fwd(vol.data_ptr(), out.data_ptr(), B, D, H, W, axis, int(exp_mode), float(mu),
                                  float(out_scale), float(clamp_min), int(transpose_out), _code(vol.dtype), _stream()),
          "hvc_drr_fwd")
    return out


def drr_bwd(vol, out, dout, axis, *, exp_mode, mu=0.3, out_scale=1.0, clamp_min=-math.inf, transpose_out=False):
    _dev(vol, out, dout)
    B, D, H, W = vol.shape
    dout = dout.contiguous()
    if dout.shape != out.shape or dout.dtype != vol.dtype:
        raise ValueError("drr_bwd: dout must match the forward output")
    dvol = torch.empty_like(vol)
    check(_lib.load().hvc_drr_bwd(vol.data_ptr(), out.data_ptr(), dout.data_ptr(), dvol.data_ptr(), B, D, H, W, axis,
                                  int(exp_mode), float(mu), float(out_scale), float(clamp_min), int(transpose_out),
                                  _code(vol.dtype), _stream()), "hvc_drr_bwd")
    return dvol


# --------------------------------------------------------------------------------------------
# convolution stems (channels-last), normalisation, resize, loss
# --------------------------------------------------------------------------------------------
class ConvGeometry:
    """Geometry of one convolution on channels-last activations (2-D convs use D = KD = 1, PD = 0)."""

    def __init__(self, B, C, src, kernel, stride, pad, out_depth=None):
        """pad = (front D pad, H pad, W pad); out_depth overrides the symmetric-padding formula for depth slabs."""
        self.B, self.C = int(B), int(C)
        self.src, self.kernel, self.stride, self.pad = tuple(src), tuple(kernel), int(stride), tuple(pad)
        self.out = tuple((s + 2 * p - k) // self.stride + 1 for s, k, p in zip(self.src, self.kernel, self.pad))
        self.out_depth = int(out_depth) if out_depth else 0
        if self.out_depth:
            self.out = (self.out_depth, self.out[1], self.out[2])
        self.taps = self.kernel[0] * self.kernel[1] * self.kernel[2]
        k = self.taps * self.C
        self.Kp = k if self.C % 8 == 0 else (k + 7) // 8 * 8
        self.M = self.B * self.out[0] * self.out[1] * self.out[2]

    def args(self):
        return (self.B, self.C, *self.src, *self.kernel, self.stride, *self.pad, self.out_depth, self.Kp)


def im2col(x, geom):
    """x: channels-last (B, D, H, W, C) contiguous -> patch matrix (M, Kp)."""
    _dev(x)
    if not x.is_contiguous() or x.shape != (geom.B, *geom.src, geom.C):
        raise ValueError("im2col: contiguous channels-last (B,D,H,W,C) tensor expected")
    col = torch.empty((geom.M, geom.Kp), dtype=x.dtype, device=x.device)
    check(_lib.load().hvc_im2col(x.data_ptr(), col.data_ptr(), *geom.args(), _code(x.dtype), _stream()), "hvc_im2col")
    return col


def col2im(dcol, geom):
    _dev(dcol)
    if not dcol.is_contiguous() or dcol.shape != (geom.M, geom.Kp):
        raise ValueError("col2im: contiguous (M, Kp) matrix expected")
    dx = torch.empty((geom.B, *geom.src, geom.C), dtype=dcol.dtype, device=dcol.device)
    check(_lib.load().hvc_col2im(dcol.data_ptr(), dx.data_ptr(), *geom.args(), _code(dcol.dtype), _stream()), "hvc_col2im")
    return dx


def conv_c1_supported(geom, cout, dtype):
    """Conv3d(1 -> 32 | 64, k3, p1, stride 1 | 2) on bf16 activations: the layers hvc_conv_c1_* cover."""
    return (dtype == torch.bfloat16 and geom.C == 1 and geom.kernel == (3, 3, 3) and geom.pad == (1, 1, 1) and geom.stride in (1, 2)
            and cout in (32, 64) and not geom.out_depth)


def conv_c1_fwd(x, w2d, bias, geom):
    """x: (B, D, H, W[, 1]) bf16 contiguous, w2d: (Cout, 32) bf16 (tap-major, columns 27.. ignored) -> (B, OD, OH, OW, Cout) bf16."""
    _dev(x, w2d, bias)
    cout = w2d.shape[0]
    if not conv_c1_supported(geom, cout, x.dtype) or x.numel() != geom.B * geom.src[0] * geom.src[1] * geom.src[2] or not x.is_contiguous():
        raise ValueError("conv_c1_fwd: contiguous bf16 one-channel volume and a k3 p1 geometry with 32 / 64 output channels expected")
    if w2d.shape != (cout, 32) or w2d.dtype != torch.bfloat16 or not w2d.is_contiguous():
        raise ValueError("conv_c1_fwd: weights must be a contiguous (Cout, 32) bf16 matrix")
    y = torch.empty((geom.B, *geom.out, cout), dtype=torch.bfloat16, device=x.device)
    with _Timed("conv_c1_fwd_kernel", 2.0 * geom.M * cout * 27):
        check(_lib.load().hvc_conv_c1_fwd(x.data_ptr(), w2d.data_ptr(), _ptr(_f32c(bias, "bias")), y.data_ptr(), geom.B, *geom.src, cout,
                                          geom.stride, _stream()), "hvc_conv_c1_fwd")
    return y


def conv_c1_dw(x, dy, geom):
    """Weight and bias gradient of conv_c1_fwd: x as there, dy (B, OD, OH, OW, Cout) bf16 contiguous -> (dw (Cout, 27) fp32, db (Cout,) fp32)."""
    _dev(x, dy)
    cout = dy.shape[-1]
    if not conv_c1_supported(geom, cout, x.dtype) or dy.dtype != torch.bfloat16 or not (x.is_contiguous() and dy.is_contiguous()) \
            or dy.numel() != geom.M * cout or x.numel() != geom.B * geom.src[0] * geom.src[1] * geom.src[2]:
        raise ValueError("conv_c1_dw: contiguous bf16 x (one channel) and dy (channels-last, 32 / 64 channels) of a k3 p1 geometry expected")
    lib = _lib.load()
    ws = torch.empty(lib.hvc_conv_c1_dw_workspace(geom.B, *geom.src, cout, geom.stride), dtype=torch.float32, device=x.device)
    dw = torch.empty((cout, 32), dtype=torch.float32, device=x.device)
    with _Timed("conv_c1_dw_kernel", 2.0 * geom.M * cout * 28):
        check(lib.hvc_conv_c1_dw(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), geom.B, *geom.src, cout, geom.stride, _stream()),
              "hvc_conv_c1_dw")
    return dw[:, :27], dw[:, 27]


def conv_c1_dx(dy, w2d, geom):
    """Input gradient of conv_c1_fwd at stride 1: dy (B, D, H, W, Cout) bf16, w2d (Cout, 32) bf16 -> dx (B, D, H, W, 1) bf16."""
    _dev(dy, w2d)
    cout = dy.shape[-1]
    if not conv_c1_supported(geom, cout, dy.dtype) or geom.stride != 1 or not dy.is_contiguous() or dy.numel() != geom.M * cout \
            or w2d.shape != (cout, 32) or w2d.dtype != torch.bfloat16:
        raise ValueError("conv_c1_dx: contiguous bf16 dy of a stride-1 k3 p1 one-channel layer with 32 / 64 output channels expected")
    wt = w2d.t().contiguous()
    wt[27:].zero_()
    dx = torch.empty((geom.B, *geom.src, 1), dtype=torch.bfloat16, device=dy.device)
    check(_lib.load().hvc_conv_c1_dx(dy.data_ptr(), wt.data_ptr(), dx.data_ptr(), geom.B, *geom.src, cout, _stream()), "hvc_conv_c1_dx")
    return dx


def conv3_halo_supported(geom, cout, dtype):
    """Conv3d(32 | 64 -> 32 | 64, k3, s1, p1) on bf16 activations: the layers hvc_conv3_halo covers."""
    return (dtype == torch.bfloat16 and geom.C in (32, 64) and cout in (32, 64) and geom.kernel == (3, 3, 3) and geom.pad == (1, 1, 1)
            and geom.stride == 1 and not geom.out_depth)


def conv3_halo(x, wfrag, bias, cout):
    """x: (B, D, H, W, CI) bf16 contiguous; wfrag: weight fragments (functional.conv_weight_frags) -> (B, D, H, W, cout) bf16."""
    _dev(x, wfrag, bias)
    B, D, H, W, CI = x.shape
    if x.dtype != torch.bfloat16 or not x.is_contiguous() or CI not in (32, 64) or cout not in (32, 64):
        raise ValueError("conv3_halo: contiguous bf16 channels-last activations with 32 / 64 channels expected")
    if wfrag.dtype != torch.bfloat16 or not wfrag.is_contiguous() or wfrag.numel() != 27 * CI * cout:
        raise ValueError("conv3_halo: wfrag must hold 27 * CI * CO bf16 weights in fragment order")
    y = torch.empty((B, D, H, W, cout), dtype=torch.bfloat16, device=x.device)
    with _Timed("conv3_halo_kernel", 2.0 * B * D * H * W * cout * 27 * CI):
        check(_lib.load().hvc_conv3_halo(x.data_ptr(), wfrag.data_ptr(), _ptr(_f32c(bias, "bias")), y.data_ptr(), B, D, H, W, CI, cout, _stream()),
              "hvc_conv3_halo")
    return y


def conv_o1_supported(C, dtype):
    return dtype == torch.bfloat16 and C in (8, 16, 32, 64, 128)


def conv_o1_fwd(x2d, w, bias):
    """y[m] = bias + x2d[m] . w : the Conv3d(C, 1, 1) of the cascade's detail enhancer on the (M, C) view.  bf16 -> (M,) bf16."""
    _dev(x2d, w, bias)
    M, C = x2d.shape
    if not conv_o1_supported(C, x2d.dtype) or not x2d.is_contiguous() or w.shape != (C,) or w.dtype != torch.bfloat16 or not w.is_contiguous():
        raise ValueError("conv_o1_fwd: contiguous bf16 (M, C) activations and (C,) weights expected, C in {8,16,32,64,128}")
    y = torch.empty((M,), dtype=torch.bfloat16, device=x2d.device)
    check(_lib.load().hvc_conv_o1_fwd(x2d.data_ptr(), w.data_ptr(), _ptr(_f32c(bias, "bias")), y.data_ptr(), M, C, _stream()), "hvc_conv_o1_fwd")
    return y


def conv_o1_bwd(x2d, dy, w, need_dx=True):
    """-> (dx (M, C) bf16 or None, dw (C,) fp32, db () fp32)."""
    _dev(x2d, dy, w)
    M, C = x2d.shape
    if not conv_o1_supported(C, x2d.dtype) or not (x2d.is_contiguous() and dy.is_contiguous()) or dy.numel() != M or dy.dtype != torch.bfloat16 \
            or w.shape != (C,) or w.dtype != torch.bfloat16 or not w.is_contiguous():
        raise ValueError("conv_o1_bwd: contiguous bf16 (M, C) activations, (M,) output gradient and (C,) weights expected")
    lib = _lib.load()
    ws = torch.empty(lib.hvc_conv_o1_bwd_workspace(M, C), dtype=torch.float32, device=x2d.device)
    dwb = torch.empty((C + 1,), dtype=torch.float32, device=x2d.device)
    dx = torch.empty_like(x2d) if need_dx else None
    check(lib.hvc_conv_o1_bwd(x2d.data_ptr(), dy.data_ptr(), w.data_ptr(), _ptr(dx), dwb.data_ptr(), ws.data_ptr(), M, C, _stream()), "hvc_conv_o1_bwd")
    return dx, dwb[:C], dwb[C]


def _conv_gemm_call(mode, src, other, out, geom, flip, n, bias, residual, residual_rows, ws, ws_n, flops):
    with _Timed("gemm_kernel", flops):
        check(_lib.load().hvc_conv_gemm(
            mode, src.data_ptr(), other.data_ptr(), out.data_ptr(), geom.B, geom.C, *geom.src, *geom.kernel, geom.stride, *geom.pad,
            int(flip), int(n), _ld(other), _ld(out), _ptr(bias), _ptr(residual), _ld(residual) if residual is not None else 0,
            int(residual_rows), _ptr(ws), int(ws_n), _code(src.dtype), _code(out.dtype), _stream()), "hvc_conv_gemm")
    return out


def conv_gemm(x, w2d, geom, *, flip=False, bias=None, residual=None, residual_rows=0, out_dtype=None, out=None):
    """Implicit-GEMM convolution: y[M][N] = patches(x)[M][taps*C] . w2d[N][taps*C]^T (+ bias, + residual rows) without a
    patch matrix in HBM.  x: channels-last (B, D, H, W, C) contiguous, C % 8 == 0.  With flip=True the taps are
    mirrored (the stride-1 input gradient: x = dy, w2d = W^T[Cin][taps*Cout], geom built on dy with pads K-1-P)."""
    _dev(x, w2d, bias, residual, out)
    if geom.out_depth:
        raise ValueError("conv_gemm: depth-slab geometries are not needed here (no patch matrix to bound)")
    if not x.is_contiguous() or x.shape != (geom.B, *geom.src, geom.C) or geom.C % 8:
        raise ValueError("conv_gemm: contiguous channels-last (B,D,H,W,C) tensor with C % 8 == 0 expected")
    K = geom.taps * geom.C
    if w2d.dim() != 2 or w2d.shape[1] != K or not _unit_inner(w2d) or w2d.dtype != x.dtype:
        raise ValueError("conv_gemm: weights must be (N, taps*C) in the activation dtype")
    N = w2d.shape[0]
    out_dtype = out_dtype or x.dtype
    if out is None:
        out = torch.empty((geom.M, N), dtype=out_dtype, device=x.device)
    elif out.shape != (geom.M, N) or not _unit_inner(out) or out.dtype != out_dtype:
        raise ValueError("conv_gemm: bad out tensor")
    _f32c(bias, "bias")
    if residual is not None and (residual.dtype != torch.float32 or residual.shape != (residual_rows or geom.M, N) or not _unit_inner(residual)):
        raise ValueError("conv_gemm: residual must be fp32 (M,N) (or (residual_rows,N))")
    return _conv_gemm_call(0, x, w2d, out, geom, flip, N, bias, residual, residual_rows, None, 0, 2.0 * geom.M * N * K)


def conv_gemm_dw(x, dy2d, geom):
    """Weight gradient of the implicit-GEMM convolution: (Cout, taps*C) fp32 = dy2d[M][Cout]^T . patches(x)[M][taps*C]."""
    _dev(x, dy2d)
    if geom.out_depth:
        raise ValueError("conv_gemm_dw: depth-slab geometries are not supported")
    if not x.is_contiguous() or x.shape != (geom.B, *geom.src, geom.C) or geom.C % 8:
        raise ValueError("conv_gemm_dw: contiguous channels-last (B,D,H,W,C) tensor with C % 8 == 0 expected")
    if dy2d.dim() != 2 or dy2d.shape[0] != geom.M or not _unit_inner(dy2d) or dy2d.dtype != x.dtype:
        raise ValueError("conv_gemm_dw: dy must be (M, Cout) in the activation dtype")
    K = geom.taps * geom.C
    N = dy2d.shape[1]
    out = torch.empty((N, K), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    ws, ws_n = None, 2 * lib.hvc_gemm_workspace(N, K, geom.M)     # the flat 256-column tiles run up to twice the slices of 128-column ones
    if ws_n > 0:
        ws = torch.empty((ws_n,), dtype=torch.float32, device=x.device)
    return _conv_gemm_call(1, x, dy2d, out, geom, False, N, None, None, 0, ws, max(ws_n, 0), 2.0 * geom.M * N * K)


def conv_dx_class_taps(geom):
    """Class-major tap order of the strided input gradient (hvc_conv_dx_class): a list, over the stride^3 parity classes in
    (cd, ch, cw) lexicographic order, of (class, [kernel tap indices (kd*KH + kh)*KW + kw in the class's column order])."""
    s = geom.stride

    def axis(K, P, p):
        kmin = (p + P) % s
        n = (K - 1 - kmin) // s + 1 if kmin < K else 0
        return [kmin + (n - 1 - t) * s for t in range(n)]           # t = 0 <-> largest kernel index (smallest source offset)
    out = []
    for cd in range(s if geom.src[0] > 1 or geom.kernel[0] > 1 else 1):
        for ch in range(s):
            for cw in range(s):
                kd, kh, kw = axis(geom.kernel[0], geom.pad[0], cd), axis(geom.kernel[1], geom.pad[1], ch), axis(geom.kernel[2], geom.pad[2], cw)
                out.append(((cd, ch, cw), [(a * geom.kernel[1] + b) * geom.kernel[2] + c for a in kd for b in kh for c in kw]))
    return out


def conv_dx_classes(dy5, wclass, geom):
    """Input gradient of a strided convolution as stride^3 implicit GEMMs over dy (one per parity class of the input position),
    each written to its interleaved positions of dx.  dy5: (B, OD, OH, OW, Cout) contiguous; wclass: (Cin, taps*Cout) in the
    class-major tap order of conv_dx_class_taps.  Returns dx (B, D, H, W, Cin)."""
    _dev(dy5, wclass)
    cout = dy5.shape[-1]
    if not dy5.is_contiguous() or dy5.shape != (geom.B, *geom.out, cout) or wclass.dtype != dy5.dtype or not wclass.is_contiguous():
        raise ValueError("conv_dx_classes: contiguous (B,OD,OH,OW,Cout) gradient and a contiguous class-major weight matrix expected")
    lib = _lib.load()
    classes = conv_dx_class_taps(geom)
    empty = any(len(t) == 0 for _, t in classes)
    dx = (torch.zeros if empty else torch.empty)((geom.B, *geom.src, geom.C), dtype=dy5.dtype, device=dy5.device)
    col = 0
    for (cd, ch, cw), taps in classes:
        n = len(taps)
        if n and cd < geom.src[0] and ch < geom.src[1] and cw < geom.src[2]:
            wc = wclass[:, col * cout:(col + n) * cout]
            with _Timed("gemm_kernel", 2.0 * geom.B * geom.C * n * cout * (geom.src[0] // geom.stride + 1) * (geom.src[1] // geom.stride + 1) * (geom.src[2] // geom.stride + 1)):
                check(lib.hvc_conv_dx_class(dy5.data_ptr(), wc.data_ptr(), dx.data_ptr(), geom.B, cout, *geom.out, geom.C, *geom.src,
                                            *geom.kernel, geom.stride, *geom.pad, cd, ch, cw, _ld(wclass), _code(dy5.dtype), _stream()),
                      "hvc_conv_dx_class")
        col += n
    return dx


def trilinear_fwd(x, size, align_corners=True):
    """x: (B, d, h, w) fp32 contiguous -> (B, D, H, W)."""
    _dev(x)
    _f32c(x, "x")
    B, d, h, w = x.shape
    y = torch.empty((B, *size), dtype=torch.float32, device=x.device)
    check(_lib.load().hvc_trilinear_fwd(x.data_ptr(), y.data_ptr(), B, d, h, w, *size, int(align_corners), _stream()), "hvc_trilinear_fwd")
    return y


def trilinear_bwd(dy, in_size, align_corners=True, separable=True):
    """Adjoint of trilinear_fwd.  separable=True: three 1-D passes through a workspace; False: single-pass 3-D gather."""
    _dev(dy)
    _f32c(dy, "dy")
    B, D, H, W = dy.shape
    dx = torch.empty((B, *in_size), dtype=torch.float32, device=dy.device)
    lib = _lib.load()
    ws = None
    if separable:
        ws = torch.empty((lib.hvc_trilinear_bwd_workspace(B, *in_size, D, H, W),), dtype=torch.float32, device=dy.device)
    check(lib.hvc_trilinear_bwd(dy.data_ptr(), dx.data_ptr(), _ptr(ws), B, *in_size, D, H, W, int(align_corners), _stream()),
          "hvc_trilinear_bwd")
    return dx


def _norm_ws(B, P, C, G, device):
    n = _lib.load().hvc_norm_workspace(B, P, C, G)
    return torch.empty((n,), dtype=torch.float32, device=device)


ACT_SILU, ACT_GELU_ERF = 0, 1


def groupnorm_silu_fwd(x, gamma, beta, G, eps=1e-5, act=ACT_SILU):
    """x: (B, P, C) channels-last contiguous.  Returns y, stats (B, G, 2)."""
    _dev(x, gamma, beta)
    B, P, Cn = x.shape
    if not x.is_contiguous():
        raise ValueError("groupnorm: contiguous (B,P,C) expected")
    y = torch.empty_like(x)
    stats = torch.empty((B, G, 2), dtype=torch.float32, device=x.device)
    ws = _norm_ws(B, P, Cn, G, x.device)
    check(_lib.load().hvc_groupnorm_act_fwd(x.data_ptr(), y.data_ptr(), _f32c(gamma, "gamma").data_ptr(), _f32c(beta, "beta").data_ptr(),
                                            stats.data_ptr(), ws.data_ptr(), B, P, Cn, G, float(eps), int(act), _code(x.dtype), _stream()),
          "hvc_groupnorm_act_fwd")
    return y, stats


def groupnorm_silu_bwd(x, dy, gamma, beta, stats, G, act=ACT_SILU):
    _dev(x, dy, gamma, beta, stats)
    B, P, Cn = x.shape
    dy = dy.contiguous()
    dx = torch.empty_like(x)
    dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(beta)
    ws = _norm_ws(B, P, Cn, G, x.device)
    check(_lib.load().hvc_groupnorm_act_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), gamma.data_ptr(), beta.data_ptr(), stats.data_ptr(),
                                            dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr(), B, P, Cn, G, int(act), _code(x.dtype), _stream()),
          "hvc_groupnorm_act_bwd")
    return dx, dgamma, dbeta


def bn_relu_pool_fwd(x, gamma, beta, running_mean, running_var, pool, training, eps=1e-5, momentum=0.1):
    """x: (N, H, W, C) channels-last.  pool = (k, s, p) or None.  Returns y (N,HP,WP,C), amax or None, stats (C,2).
    Updates running_mean / running_var in place when training."""
    _dev(x, gamma, beta, running_mean, running_var)
    N, H, W, Cn = x.shape
    if not x.is_contiguous():
        raise ValueError("bn_relu_pool: contiguous (N,H,W,C) expected")
    k, s, p = pool if pool else (1, 1, 0)
    HP, WP = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    y = torch.empty((N, HP, WP, Cn), dtype=x.dtype, device=x.device)
    amax = torch.empty((N, HP, WP, Cn), dtype=torch.uint8, device=x.device) if k > 1 else None
    stats = torch.empty((Cn, 2), dtype=torch.float32, device=x.device)
    ws = _norm_ws(N, H * W, Cn, Cn, x.device)
    check(_lib.load().hvc_bn_relu_pool_fwd(x.data_ptr(), y.data_ptr(), _ptr(amax), _f32c(gamma, "gamma").data_ptr(), _f32c(beta, "beta").data_ptr(),
                                           _ptr(_f32c(running_mean, "running_mean")), _ptr(_f32c(running_var, "running_var")), stats.data_ptr(),
                                           ws.data_ptr(), N, H, W, Cn, k, s, p, int(training), float(eps), float(momentum), _code(x.dtype), _stream()),
          "hvc_bn_relu_pool_fwd")
    return y, amax, stats


def bn_relu_pool_bwd(x, dy, amax, gamma, beta, stats, pool, training):
    _dev(x, dy, amax, gamma, beta, stats)
    N, H, W, Cn = x.shape
    k, s, p = pool if pool else (1, 1, 0)
    dy = dy.contiguous()
    dx = torch.empty_like(x)
    dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(beta)
    ws = _norm_ws(N, H * W, Cn, Cn, x.device)
    check(_lib.load().hvc_bn_relu_pool_bwd(x.data_ptr(), dy.data_ptr(), _ptr(amax), dx.data_ptr(), gamma.data_ptr(), beta.data_ptr(), stats.data_ptr(),
                                           dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr(), N, H, W, Cn, k, s, p, int(training), _code(x.dtype),
                                           _stream()), "hvc_bn_relu_pool_bwd")
    return dx, dgamma, dbeta


def ssim_l1_fwd(pred, target, window=11, l1_w=1.0, ssim_w=0.5):
    """pred/target: (B, D, H, W) fp32 contiguous.  Returns out3 (total, l1, ssim_loss) and the saved derivative maps."""
    _dev(pred, target)
    _f32c(pred, "pred"), _f32c(target, "target")
    B, D, H, W = pred.shape
    lib = _lib.load()
    ws = torch.empty((lib.hvc_ssim_l1_workspace(B, D, H, W),), dtype=torch.float32, device=pred.device)
    gmaps = torch.empty((3, B, D, H, W), dtype=torch.float32, device=pred.device)
    out = torch.empty((3,), dtype=torch.float32, device=pred.device)
    check(lib.hvc_ssim_l1_fwd(pred.data_ptr(), target.data_ptr(), out.data_ptr(), gmaps.data_ptr(), ws.data_ptr(), B, D, H, W, int(window),
                              float(l1_w), float(ssim_w), _stream()), "hvc_ssim_l1_fwd")
    return out, gmaps


def ssim_l1_bwd(pred, target, gmaps, gscale, window=11, l1_w=1.0, ssim_w=0.5):
    _dev(pred, target, gmaps, gscale)
    B, D, H, W = pred.shape
    ws = torch.empty((6 * pred.numel(),), dtype=torch.float32, device=pred.device)
    dpred = torch.empty_like(pred)
    check(_lib.load().hvc_ssim_l1_bwd(pred.data_ptr(), target.data_ptr(), gmaps.data_ptr(), _ptr(_f32c(gscale, "gscale")), dpred.data_ptr(),
                                      ws.data_ptr(), B, D, H, W, int(window), float(l1_w), float(ssim_w), _stream()), "hvc_ssim_l1_bwd")
    return dpred


def tv3d_fwd(vol, eps=1e-8):
    """vol: (B, D, H, W) fp32 contiguous -> means3: mean sqrt(diff^2 + eps) of the forward differences along D, H, W."""
    _dev(vol)
    _f32c(vol, "vol")
    B, D, H, W = vol.shape
    lib = _lib.load()
    ws = torch.empty((lib.hvc_tv3d_workspace(B, D, H, W),), dtype=torch.float32, device=vol.device)
    out = torch.empty((3,), dtype=torch.float32, device=vol.device)
    check(lib.hvc_tv3d_fwd(vol.data_ptr(), out.data_ptr(), ws.data_ptr(), B, D, H, W, float(eps), _stream()), "hvc_tv3d_fwd")
    return out


def tv3d_bwd(vol, gscale, eps=1e-8):
    _dev(vol, gscale)
    B, D, H, W = vol.shape
    dvol = torch.empty_like(vol)
    check(_lib.load().hvc_tv3d_bwd(vol.data_ptr(), _ptr(_f32c(gscale, "gscale")), dvol.data_ptr(), B, D, H, W, float(eps), _stream()), "hvc_tv3d_bwd")
    return dvol


def spectral_l1_fwd(pred_spec, target_spec):
    """pred_spec / target_spec: (B, D, H, W, 2) fp32 contiguous (torch.view_as_real of the 3-D FFTs) -> out2 (low, high)."""
    _dev(pred_spec, target_spec)
    _f32c(pred_spec, "pred_spec"), _f32c(target_spec, "target_spec")
    if pred_spec.dim() != 5 or pred_spec.shape[-1] != 2 or pred_spec.shape != target_spec.shape:
        raise ValueError("spectral_l1: (B, D, H, W, 2) spectra of equal shape expected")
    B, D, H, W, _ = pred_spec.shape
    lib = _lib.load()
    ws = torch.empty((lib.hvc_spectral_l1_workspace(B, D, H, W),), dtype=torch.float32, device=pred_spec.device)
    out = torch.empty((2,), dtype=torch.float32, device=pred_spec.device)
    check(lib.hvc_spectral_l1_fwd(pred_spec.data_ptr(), target_spec.data_ptr(), out.data_ptr(), ws.data_ptr(), B, D, H, W, _stream()),
          "hvc_spectral_l1_fwd")
    return out


def spectral_l1_bwd(pred_spec, target_spec, gscale):
    _dev(pred_spec, target_spec, gscale)
    B, D, H, W, _ = pred_spec.shape
    d = torch.empty_like(pred_spec)
    check(_lib.load().hvc_spectral_l1_bwd(pred_spec.data_ptr(), target_spec.data_ptr(), _ptr(_f32c(gscale, "gscale")), d.data_ptr(),
                                          B, D, H, W, _stream()), "hvc_spectral_l1_bwd")
    return d


def resize_loss_fwd(proj, target, align_corners, mode):
    """proj (B, h, w) fp32 contiguous; target (B, S1, S2) fp32 view with contiguous rows (batch stride free) -> scalar (1,)
    mean |resize(proj) - target| (mode 0) or mean squared difference (mode 1)."""
    _dev(proj, target)
    _f32c(proj, "proj")
    if target.dtype != torch.float32 or target.dim() != 3 or target.shape[0] != proj.shape[0] or target.stride(2) != 1 \
            or (target.shape[1] > 1 and target.stride(1) != target.shape[2]):
        raise ValueError("resize_loss: target must be a float32 (B, S1, S2) view with contiguous rows")
    B, h, w = proj.shape
    S1, S2 = target.shape[1], target.shape[2]
    tb = target.stride(0) if B > 1 else S1 * S2
    lib = _lib.load()
    ws = torch.empty((lib.hvc_resize_loss_workspace(B, S1, S2),), dtype=torch.float32, device=proj.device)
    out = torch.empty((1,), dtype=torch.float32, device=proj.device)
    check(lib.hvc_resize_loss_fwd(proj.data_ptr(), target.data_ptr(), out.data_ptr(), ws.data_ptr(), B, h, w, S1, S2, tb,
                                  int(bool(align_corners)), int(mode), _stream()), "hvc_resize_loss_fwd")
    return out


def resize_loss_grad(proj, target, gscale, align_corners, mode):
    """d loss / d resized image, (B, S1, S2); the adjoint of the resize (trilinear_bwd with depth 1) takes it from there."""
    _dev(proj, target, gscale)
    B, h, w = proj.shape
    S1, S2 = target.shape[1], target.shape[2]
    tb = target.stride(0) if B > 1 else S1 * S2
    d = torch.empty((B, S1, S2), dtype=torch.float32, device=proj.device)
    check(_lib.load().hvc_resize_loss_grad(proj.data_ptr(), target.data_ptr(), _ptr(_f32c(gscale, "gscale")), d.data_ptr(), B, h, w, S1, S2, tb,
                                           int(bool(align_corners)), int(mode), _stream()), "hvc_resize_loss_grad")
    return d


def view_mean_gap_fwd(feats, V):
    """feats (B*V, P, E) channels-last feature maps (fp32 / bf16, contiguous) -> mean over views (B, P, E) fp32, pooled (B, E) fp32."""
    _dev(feats)
    if feats.dim() != 3 or not feats.is_contiguous() or feats.shape[0] % V:
        raise ValueError("view_mean_gap: contiguous (B*V, P, E) expected")
    BV, P, E = feats.shape
    B = BV // V
    lib = _lib.load()
    ws = torch.empty((lib.hvc_view_mean_gap_workspace(B, P, E),), dtype=torch.float32, device=feats.device)
    mean = torch.empty((B, P, E), dtype=torch.float32, device=feats.device)
    pooled = torch.empty((B, E), dtype=torch.float32, device=feats.device)
    check(lib.hvc_view_mean_gap_fwd(feats.data_ptr(), mean.data_ptr(), pooled.data_ptr(), ws.data_ptr(), B, V, P, E, _code(feats.dtype), _stream()),
          "hvc_view_mean_gap_fwd")
    return mean, pooled


def view_mean_gap_bwd(dmean, dpooled, V, dtype):
    if dmean is None:
        raise ValueError("view_mean_gap_bwd: dmean (B, P, E) is required, it gives the geometry")
    _dev(dmean, dpooled)
    B, P, E = dmean.shape
    df = torch.empty((B * V, P, E), dtype=dtype, device=dmean.device)
    check(_lib.load().hvc_view_mean_gap_bwd(_ptr(_f32c(dmean, "dmean")), _ptr(_f32c(dpooled, "dpooled")), df.data_ptr(), B, V, P, E, _code(dtype),
                                            _stream()), "hvc_view_mean_gap_bwd")
    return df
