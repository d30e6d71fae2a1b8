"""Autograd glue over the C-ABI kernels: one torch.autograd.Function per residual branch of the
reference's HybridViTBlock3D (models/hybrid_vit_backbone.py:117-139), each running a fused HIP
forward chain and a hand-ordered HIP backward chain, plus small general-purpose Functions used by
the stand-alone module forwards.

Precision: the residual stream and all parameters / gradients are fp32; branch interiors run in
`cdt` (compute dtype): bfloat16 under torch.autocast (the reference trainers' mode,
direct_regression/train_direct_4gpu.py:65) or float32 otherwise (split-bf16 MFMA, ~1e-5 rel).
"""
import warnings

import torch

from . import ops

_warned_fp16 = False

# fp8 (e4m3) MFMA products in the attention forward (BASELINE configs[4]); off by default, switched by the trainers from the
# "mi355x": {"fp8_attention": true} section of the JSON config, or by hand.  Applies to bf16 (autocast) runs only.
ATTN_FP8 = False


def set_fp8_attention(on: bool) -> None:
    global ATTN_FP8
    ATTN_FP8 = bool(on)


# Gradient checkpointing of the cascade's stage-3 ViT (reference model_progressive.py:296-305 recomputes the refiner's forward in
# the backward pass to fit 40-80 GB cards).  On MI355X the saved activations of that ViT (~4 GB at 256^3, batch 1) are noise
# against 288 GB of HBM and the recomputation costs a sixth of the step, so the default policy honours the model's
# use_gradient_checkpointing flag only when the activations would not fit comfortably ("auto"); "on" = whenever the model
# asks (the reference's behaviour), "off" = never.  Results are identical either way (same dropout seeds are replayed).
# Set from the "mi355x": {"gradient_checkpointing": ...} section of the JSON config, or by hand.
CHECKPOINT_POLICY = "auto"


def set_checkpoint_policy(policy) -> None:
    """'auto' | 'on' | 'off'; JSON booleans are accepted too (true = 'on', the reference's behaviour; false = 'off')."""
    global CHECKPOINT_POLICY
    if isinstance(policy, bool):
        policy = "on" if policy else "off"
    if policy not in ("auto", "on", "off"):
        raise ValueError("checkpoint policy must be 'auto', 'on' or 'off' (or a boolean)")
    CHECKPOINT_POLICY = policy
    _CHECKPOINT_DECISIONS.clear()


_CHECKPOINT_DECISIONS = {}      # (device, saved_bytes) -> bool: 'auto' decides once per module shape, not once per forward


def use_checkpoint(requested: bool, saved_bytes: int, device) -> bool:
    """Should a module that was asked to checkpoint (requested) really recompute?  saved_bytes = what it would keep otherwise.
    'auto' compares with the memory this process can still get: the driver's free figure PLUS what torch's caching allocator
    holds but has not handed out (mem_get_info alone under-reports after the first steps and could flip the answer mid-run or
    differ between DDP ranks); the answer is cached per (device, size), so there is one driver query per shape, not per step."""
    if not requested or CHECKPOINT_POLICY == "off":
        return False
    if CHECKPOINT_POLICY == "on" or not torch.cuda.is_available():
        return True
    key = (str(device), int(saved_bytes))
    hit = _CHECKPOINT_DECISIONS.get(key)
    if hit is None:
        free, _total = torch.cuda.mem_get_info(device)
        free += torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)
        hit = _CHECKPOINT_DECISIONS[key] = 4 * saved_bytes > free      # recompute only if four times the saved activations would not fit
    return hit


def _fp8(t) -> bool:
    """fp8 forward for this operand?  Only bf16 operands with a head dim the fp8 kernel is built for (32 / 64: the cascade's and
    the direct model's heads); anything else keeps the bf16 kernel."""
    return ATTN_FP8 and t.dtype == torch.bfloat16 and t.shape[-1] in (32, 64)


def compute_dtype(ref: torch.Tensor) -> torch.dtype:
    """bf16 inside torch.autocast('cuda'), else fp32."""
    global _warned_fp16
    if torch.is_autocast_enabled():
        dt = torch.get_autocast_dtype("cuda")
        if dt == torch.float16 and not _warned_fp16:
            warnings.warn("HVC kernels compute in bfloat16 on MI355X; autocast(float16) is mapped to bfloat16")
            _warned_fp16 = True
        return torch.bfloat16
    return torch.float32


# ---- weight casts are cached ON the parameter object, keyed by a global epoch, the tensor's version counter, device and
# storage address: parameters change once per optimizer step, so the bf16 copies are refreshed once per step, not per use.
#   * every torch optimizer's step() advances the epoch and refreshes every registered copy IN PLACE with one multi-tensor
#     cast (global step post-hook below).  This is REQUIRED, not a nicety: the fused / foreach optimizer kernels
#     (AdamW(fused=True), what the trainers and bench.py use) update parameters without bumping their version counters, so a
#     version-only key would keep serving the pre-step weights for ever (found by the two-step train_step fixture: step 2
#     ran on step-0 conv weights).  One batched refresh instead of ~60 lazy per-tensor casts matters where the step is
#     launch-bound (64^3: 12 ms of ~300 launches);
#   * the version counter covers load_state_dict, copy_ and other autograd-visible in-place writes;
#   * device + storage address cover `p.data = ...` and module.to(device).
# The one write nothing here can see is an IN-PLACE write through `.data` outside an optimizer (p.data.mul_(..), hand-written
# EMA updates): same storage, `.data` has its own version counter -- call invalidate_param_casts() after such a write.
import os
import weakref

_CAST_EPOCH = 0
_CAST_REGISTRY = {}          # id(param) -> (weakref(param), attribute name); entries die with their parameter


class trace_range:
    """roctx range around a phase of the step (forward / loss / backward / optimizer) when HVC_TRACE_RANGES=1, so that a
    `rocprofv3 --marker-trace` timeline shows the phases above the kernels; a no-op otherwise."""
    ON = os.environ.get("HVC_TRACE_RANGES") == "1"

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if trace_range.ON:
            torch.cuda.nvtx.range_push(self.name)      # nvtx maps to roctx on ROCm builds of torch

    def __exit__(self, *exc):
        if trace_range.ON:
            torch.cuda.nvtx.range_pop()
        return False


def invalidate_param_casts():
    """Drop every cached low-precision weight copy (call after writing parameters in place through `.data`,
    which does not bump the tensor version counter)."""
    global _CAST_EPOCH
    _CAST_EPOCH += 1
    _CAST_REGISTRY.clear()


def _cache_key(p, *extra):
    return (p._version, _CAST_EPOCH, p.device, p.data_ptr(), *extra)


def _register(p, attr):
    pid = id(p)
    _CAST_REGISTRY[(pid, attr)] = (weakref.ref(p, lambda _r, k=(pid, attr): _CAST_REGISTRY.pop(k, None)), attr)


def _conv_w2d_fill(dst, w):
    """dst (Cout, Kp) <- w (Cout, Cin, *k) with column = tap * Cin + c (zero pad columns stay untouched)."""
    cout, cin = w.shape[0], w.shape[1]
    taps = w[0, 0].numel()
    dst[:, :taps * cin].view(cout, taps, cin).copy_(w.detach().reshape(cout, cin, taps).permute(0, 2, 1))


def _conv_w2dT_fill(dst, w):
    """dst (Cin, taps*Cout) <- w (Cout, Cin, *k) with column = tap * Cout + co: the operand of the stride-1 input gradient
    taken as an implicit GEMM over dy (ops.conv_gemm with flip=True mirrors the taps in the gather, not here)."""
    cout, cin = w.shape[0], w.shape[1]
    taps = w[0, 0].numel()
    dst.view(cin, taps, cout).copy_(w.detach().reshape(cout, cin, taps).permute(1, 2, 0))


def _conv_w2dC_fill(dst, w):
    """dst (Cin, taps*Cout) <- w (Cout, Cin, *k) with the taps in the class-major order of the strided input gradient
    (ops.conv_dx_class_taps); the permutation was stored on the parameter when the copy was first built."""
    cout, cin = w.shape[0], w.shape[1]
    taps = w[0, 0].numel()
    dst.view(cin, taps, cout).copy_(w.detach().reshape(cout, cin, taps).permute(1, 2, 0).index_select(1, w._hvc_w2dC_perm))


def _frag_order(w3):
    """(CO, 27, CI) -> MFMA fragment order [tap][ck][nt][h][r][8] of hvc_conv3_halo (lane = 32 h + r, 16-channel slices ck)."""
    co, taps, ci = w3.shape
    return w3.reshape(co // 32, 32, taps, ci // 16, 2, 8).permute(2, 3, 0, 4, 1, 5)


def _conv_wfrag_fill(dst, w):
    """forward operand: W[co][tap][ci]"""
    co, ci = w.shape[0], w.shape[1]
    dst.view(27, ci // 16, co // 32, 2, 32, 8).copy_(_frag_order(w.detach().reshape(co, ci, 27).permute(0, 2, 1)))


def _conv_wfragT_fill(dst, w):
    """input-gradient operand: the convolution of dy with the mirrored kernel, W'[ci][tap'][co] = W[co][ci][26 - tap']"""
    co, ci = w.shape[0], w.shape[1]
    dst.view(27, co // 16, ci // 32, 2, 32, 8).copy_(_frag_order(w.detach().reshape(co, ci, 27).flip(2).permute(1, 2, 0)))


_CONV_FILL = {"_hvc_w2d": _conv_w2d_fill, "_hvc_w2dT": _conv_w2dT_fill, "_hvc_w2dC": _conv_w2dC_fill,
              "_hvc_wfrag": _conv_wfrag_fill, "_hvc_wfragT": _conv_wfragT_fill}


def _after_optimizer_step(optimizer, args, kwargs):
    """Refresh, in place, the cached copies of the parameters THIS optimizer just updated (fused AdamW does not bump their
    version counters): plain casts in ONE multi-tensor copy, conv layouts by one strided cast each.  Copies of parameters that
    belong to no group of this optimizer (frozen cascade stages, a second model, an EMA) keep their contents and are only
    re-keyed to the new epoch; entries whose parameter moved (device / storage) or vanished are dropped and rebuilt lazily."""
    global _CAST_EPOCH
    before = _CAST_EPOCH
    _CAST_EPOCH += 1          # any copy this hook does not know about is stale from here on
    mine = {id(q) for group in optimizer.param_groups for q in group["params"] if q.requires_grad}
    srcs, dsts, rekey = [], [], []
    for key, (ref, attr) in list(_CAST_REGISTRY.items()):
        p = ref()
        hit = getattr(p, attr, None) if p is not None else None
        if hit is None or hit[0][2] != p.device or hit[0][3] != p.data_ptr() or not p.is_cuda:
            _CAST_REGISTRY.pop(key, None)
            continue
        old_key, dst = hit
        if key[0] not in mine:                 # not updated by this optimizer: a copy that WAS current stays right, its key moves on
            if old_key[:4] == (p._version, before, p.device, p.data_ptr()):
                rekey.append((p, attr, old_key, dst))
            else:                              # already stale (e.g. load_state_dict wrote the parameter): rebuilt on next use
                _CAST_REGISTRY.pop(key, None)
            continue
        if attr == "_hvc_cast":
            srcs.append(p.detach())
            dsts.append(dst)
        else:
            with torch.no_grad():
                _CONV_FILL[attr](dst, p)
        rekey.append((p, attr, old_key, dst))
    if dsts:
        with torch.no_grad():
            torch._foreach_copy_(dsts, srcs)
    for p, attr, old_key, dst in rekey:
        setattr(p, attr, (_cache_key(p, *old_key[4:]), dst))


from torch.optim.optimizer import register_optimizer_step_post_hook as _register_step_hook  # noqa: E402

_register_step_hook(_after_optimizer_step)


def cast_param(p: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    if p.dtype == dtype:
        d = p.detach()
        return d if d.is_contiguous() else d.contiguous()
    hit = getattr(p, "_hvc_cast", None)
    key = _cache_key(p, dtype)
    if hit is not None and hit[0] == key:
        return hit[1]
    out = ops.cast(p.detach(), dtype)
    try:
        p._hvc_cast = (key, out)
        _register(p, "_hvc_cast")
    except AttributeError:
        pass
    return out


def new_seed() -> int:
    """Dropout seed drawn from torch's CPU generator (so torch.manual_seed / checkpoint replay apply)."""
    return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())


def _f32(t):
    return None if t is None else t.detach().float().contiguous()


def _as_cdt(t, cdt):
    t = t.detach()
    if t.dtype != cdt:
        return ops.cast(t, cdt)
    return t.contiguous()


# --------------------------------------------------------------------------------------------
# generic pieces (stand-alone module forwards, AdaLN linear, context projection ...)
# --------------------------------------------------------------------------------------------
class LinearFn(torch.autograd.Function):
    """y = x W^T + b on (M,K) x (N,K); x/W in cdt, y in out_dtype; grads for W, b in fp32."""

    @staticmethod
    def forward(ctx, x, weight, bias, cdt, out_dtype, p_drop, seed):
        xc = _as_cdt(x, cdt)
        wc = cast_param(weight, cdt)
        y = ops.gemm(xc, wc, bias=_f32(bias), out_dtype=out_dtype, p_drop=p_drop, seed=seed)
        ctx.save_for_backward(xc, weight)
        ctx.cfg = (cdt, x.dtype, bias is not None, p_drop, seed)
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, weight = ctx.saved_tensors
        cdt, xdt, has_bias, p_drop, seed = ctx.cfg
        # branch_bwd = cast to cdt (+ output-dropout mask of the forward epilogue) + bias column sum
        dyc, _, db = ops.branch_bwd(_f32(dy), None, None, out_dtype=cdt, want_bias=has_bias and ctx.needs_input_grad[2],
                                    p_drop=p_drop, seed=seed)
        wc = cast_param(weight, cdt)
        dx = ops.gemm(dyc, wc, b_kmajor=True, out_dtype=torch.float32 if xdt == torch.float32 else cdt) if ctx.needs_input_grad[0] else None
        dw = ops.gemm(dyc, xc, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32) if ctx.needs_input_grad[1] else None
        if dx is not None and dx.dtype != xdt:
            dx = dx.to(xdt)
        return dx, dw, db, None, None, None, None


def linear(x, weight, bias=None, cdt=None, out_dtype=None, p_drop=0.0, seed=0):
    """x: (..., K) -> (..., N); optional output dropout fused into the GEMM epilogue."""
    cdt = cdt or compute_dtype(x)
    out_dtype = out_dtype or cdt
    shp = x.shape
    y = LinearFn.apply(x.reshape(-1, shp[-1]), weight, bias, cdt, out_dtype, p_drop, seed)
    return y.view(*shp[:-1], weight.shape[0])


class LayerNormFn(torch.autograd.Function):
    """LayerNorm (+ AdaLN modulate) of the fp32 residual stream, rows = B * N."""

    @staticmethod
    def forward(ctx, x, gamma, beta, scale, shift, rows_per_batch, out_dtype):
        xf = _f32(x)
        y, mean, rstd = ops.layernorm_fwd(xf, _f32(gamma), _f32(beta), _f32(scale), _f32(shift),
                                          rows_per_batch=rows_per_batch, out_dtype=out_dtype)
        ctx.save_for_backward(xf, gamma, beta, scale, mean, rstd)
        ctx.rpb = rows_per_batch
        return y

    @staticmethod
    def backward(ctx, dy):
        xf, gamma, beta, scale, mean, rstd = ctx.saved_tensors
        dx, dg, db, dsc, dsh = ops.layernorm_bwd(dy.contiguous(), xf, _f32(gamma), _f32(beta), _f32(scale), mean, rstd,
                                                 rows_per_batch=ctx.rpb)
        return dx, dg, db, dsc, dsh, None, None


def layer_norm(x, gamma, beta, scale=None, shift=None, out_dtype=torch.float32):
    """x: (B,N,C) fp32; scale/shift: (B,1,C) or (B,C) or None."""
    B, N, Cn = x.shape
    if scale is not None:
        scale, shift = scale.reshape(B, Cn), shift.reshape(B, Cn)
    y = LayerNormFn.apply(x.reshape(B * N, Cn), gamma, beta, scale, shift, N, out_dtype)
    return y.view(B, N, Cn)


class PackedSelfAttnFn(torch.autograd.Function):
    """Attention core on the packed projection qkv: (B, N, 3, H, D) -> o: (B, N, H*D)."""

    @staticmethod
    def forward(ctx, qkv, scale, p_drop, seed):
        q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
        o, lse = ops.attention_fwd(q, k, v, scale, p_drop, seed, fp8=_fp8(q))
        ctx.save_for_backward(qkv, o, lse)
        ctx.cfg = (scale, p_drop, seed)
        B, N, H, D = o.shape
        return o.view(B, N, H * D)

    @staticmethod
    def backward(ctx, do):
        qkv, o, lse = ctx.saved_tensors
        scale, p_drop, seed = ctx.cfg
        dqkv = torch.empty_like(qkv)
        ops.attention_bwd(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], o, do.reshape(o.shape), lse, scale, p_drop, seed,
                          dq=dqkv[:, :, 0], dk=dqkv[:, :, 1], dv=dqkv[:, :, 2])
        return dqkv, None, None, None


class PackedCrossAttnFn(torch.autograd.Function):
    """q: (B, N, H, D), kv: (B, M, 2, H, D) -> o: (B, N, H*D)."""

    @staticmethod
    def forward(ctx, q, kv, scale, p_drop, seed):
        o, lse = ops.attention_fwd(q, kv[:, :, 0], kv[:, :, 1], scale, p_drop, seed, fp8=_fp8(q))
        ctx.save_for_backward(q, kv, o, lse)
        ctx.cfg = (scale, p_drop, seed)
        B, N, H, D = o.shape
        return o.view(B, N, H * D)

    @staticmethod
    def backward(ctx, do):
        q, kv, o, lse = ctx.saved_tensors
        scale, p_drop, seed = ctx.cfg
        dq = torch.empty_like(q)
        dkv = torch.empty_like(kv)
        ops.attention_bwd(q, kv[:, :, 0], kv[:, :, 1], o, do.reshape(o.shape), lse, scale, p_drop, seed,
                          dq=dq, dk=dkv[:, :, 0], dv=dkv[:, :, 1])
        return dq, dkv, None, None, None


# --------------------------------------------------------------------------------------------
# fused residual branches of HybridViTBlock3D
# --------------------------------------------------------------------------------------------
def _mod2d(t, B, Cn):
    return None if t is None else _f32(t.reshape(B, Cn))


class SelfAttnBranchFn(torch.autograd.Function):
    """x + gate * proj(attn(qkv((1 + scale) * LN(x) + shift)))   -- models/hybrid_vit_backbone.py:120-123."""

    @staticmethod
    def forward(ctx, x, gamma, beta, scale, shift, gate, w_qkv, w_proj, b_proj, heads, cdt, p_drop, seeds):
        B, N, Cn = x.shape
        D = Cn // heads
        x2 = _f32(x).view(B * N, Cn)
        sc, sh, gt = _mod2d(scale, B, Cn), _mod2d(shift, B, Cn), _mod2d(gate, B, Cn)
        h, mean, rstd = ops.layernorm_fwd(x2, _f32(gamma), _f32(beta), sc, sh, rows_per_batch=N, out_dtype=cdt)
        qkv = ops.gemm(h, cast_param(w_qkv, cdt)).view(B, N, 3, heads, D)
        o, lse = ops.attention_fwd(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], D ** -0.5, p_drop, seeds[0], fp8=_fp8(qkv))
        o2 = o.view(B * N, Cn)
        z = torch.empty((B * N, Cn), dtype=cdt, device=x.device) if gt is not None else None
        out = ops.gemm(o2, cast_param(w_proj, cdt), bias=_f32(b_proj), zsave=z, gate=gt, residual=x2, rows_per_batch=N,
                       p_drop=p_drop, seed=seeds[1], out_dtype=torch.float32)
        ctx.save_for_backward(x2, gamma, beta, sc, gt, w_qkv, w_proj, h, mean, rstd, qkv, o, lse, z)
        ctx.cfg = (B, N, Cn, heads, cdt, p_drop, seeds, scale is not None, gate is not None)
        return out.view(B, N, Cn)

    @staticmethod
    def backward(ctx, dout):
        x2, gamma, beta, sc, gt, w_qkv, w_proj, h, mean, rstd, qkv, o, lse, z = ctx.saved_tensors
        B, N, Cn, heads, cdt, p_drop, seeds, has_mod, has_gate = ctx.cfg
        D = Cn // heads
        dy = _f32(dout).view(B * N, Cn)
        dz, dgate, dbp = ops.branch_bwd(dy, z, gt, rows_per_batch=N, out_dtype=cdt, p_drop=p_drop, seed=seeds[1])
        o2 = o.view(B * N, Cn)
        dwp = ops.gemm(dz, o2, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32)
        do = ops.gemm(dz, cast_param(w_proj, cdt), b_kmajor=True)
        dqkv = torch.empty_like(qkv)
        ops.attention_bwd(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], o, do.view(o.shape), lse, D ** -0.5, p_drop, seeds[0],
                          dq=dqkv[:, :, 0], dk=dqkv[:, :, 1], dv=dqkv[:, :, 2])
        dqkv2 = dqkv.view(B * N, 3 * Cn)
        dwqkv = ops.gemm(dqkv2, h, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32)
        dh = ops.gemm(dqkv2, cast_param(w_qkv, cdt), b_kmajor=True)
        dx, dg, db, dsc, dsh = ops.layernorm_bwd(dh, x2, _f32(gamma), _f32(beta), sc, mean, rstd, dres=dy, rows_per_batch=N)
        v3 = lambda t: None if t is None else t.view(B, 1, Cn)
        return (dx.view(B, N, Cn), dg, db, v3(dsc) if has_mod else None, v3(dsh) if has_mod else None,
                v3(dgate) if has_gate else None, dwqkv, dwp, dbp, None, None, None, None)


class CrossAttnBranchFn(torch.autograd.Function):
    """x + proj(attn(q(LN(x)), kv(context)))   -- models/hybrid_vit_backbone.py:126-128."""

    @staticmethod
    def forward(ctx, x, context, gamma, beta, w_q, w_kv, w_proj, b_proj, heads, cdt, p_drop, seeds):
        B, N, Cn = x.shape
        M, Cc = context.shape[1], context.shape[2]
        D = Cn // heads
        x2 = _f32(x).view(B * N, Cn)
        c2 = _as_cdt(context, cdt).view(B * M, Cc)
        h, mean, rstd = ops.layernorm_fwd(x2, _f32(gamma), _f32(beta), rows_per_batch=N, out_dtype=cdt)
        q = ops.gemm(h, cast_param(w_q, cdt)).view(B, N, heads, D)
        kv = ops.gemm(c2, cast_param(w_kv, cdt)).view(B, M, 2, heads, D)
        o, lse = ops.attention_fwd(q, kv[:, :, 0], kv[:, :, 1], D ** -0.5, p_drop, seeds[0], fp8=_fp8(q))
        out = ops.gemm(o.view(B * N, Cn), cast_param(w_proj, cdt), bias=_f32(b_proj), residual=x2,
                       p_drop=p_drop, seed=seeds[1], out_dtype=torch.float32)
        ctx.save_for_backward(x2, c2, gamma, beta, w_q, w_kv, w_proj, h, mean, rstd, q, kv, o, lse)
        ctx.cfg = (B, N, M, Cn, Cc, heads, cdt, p_drop, seeds, context.dtype)
        return out.view(B, N, Cn)

    @staticmethod
    def backward(ctx, dout):
        x2, c2, gamma, beta, w_q, w_kv, w_proj, h, mean, rstd, q, kv, o, lse = ctx.saved_tensors
        B, N, M, Cn, Cc, heads, cdt, p_drop, seeds, ctx_dtype = ctx.cfg
        D = Cn // heads
        dy = _f32(dout).view(B * N, Cn)
        dz, _, dbp = ops.branch_bwd(dy, None, None, rows_per_batch=N, out_dtype=cdt, p_drop=p_drop, seed=seeds[1])
        dwp = ops.gemm(dz, o.view(B * N, Cn), a_kmajor=True, b_kmajor=True, out_dtype=torch.float32)
        do = ops.gemm(dz, cast_param(w_proj, cdt), b_kmajor=True)
        dq = torch.empty_like(q)
        dkv = torch.empty_like(kv)
        ops.attention_bwd(q, kv[:, :, 0], kv[:, :, 1], o, do.view(o.shape), lse, D ** -0.5, p_drop, seeds[0],
                          dq=dq, dk=dkv[:, :, 0], dv=dkv[:, :, 1])
        dq2, dkv2 = dq.view(B * N, Cn), dkv.view(B * M, 2 * Cn)
        dwq = ops.gemm(dq2, h, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32)
        dh = ops.gemm(dq2, cast_param(w_q, cdt), b_kmajor=True)
        dwkv = ops.gemm(dkv2, c2, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32)
        dctx = None
        if ctx.needs_input_grad[1]:
            dctx = ops.gemm(dkv2, cast_param(w_kv, cdt), b_kmajor=True, out_dtype=torch.float32).view(B, M, Cc)
            if dctx.dtype != ctx_dtype:
                dctx = dctx.to(ctx_dtype)
        dx, dg, db, _, _ = ops.layernorm_bwd(dh, x2, _f32(gamma), _f32(beta), None, mean, rstd, dres=dy, rows_per_batch=N)
        return dx.view(B, N, Cn), dctx, dg, db, dwq, dwkv, dwp, dbp, None, None, None, None


class MlpBranchFn(torch.autograd.Function):
    """x + gate * fc2(drop(gelu(fc1((1 + scale) * LN(x) + shift))))   -- models/hybrid_vit_backbone.py:136-139, :75-81."""

    @staticmethod
    def forward(ctx, x, gamma, beta, scale, shift, gate, w1, b1, w2, b2, cdt, p_drop, seeds):
        B, N, Cn = x.shape
        Hd = w1.shape[0]
        x2 = _f32(x).view(B * N, Cn)
        sc, sh, gt = _mod2d(scale, B, Cn), _mod2d(shift, B, Cn), _mod2d(gate, B, Cn)
        h, mean, rstd = ops.layernorm_fwd(x2, _f32(gamma), _f32(beta), sc, sh, rows_per_batch=N, out_dtype=cdt)
        pre = torch.empty((B * N, Hd), dtype=cdt, device=x.device)
        a = ops.gemm(h, cast_param(w1, cdt), bias=_f32(b1), act=ops.ACT_GELU, aux=pre, p_drop=p_drop, seed=seeds[0])
        z = torch.empty((B * N, Cn), dtype=cdt, device=x.device) if gt is not None else None
        out = ops.gemm(a, cast_param(w2, cdt), bias=_f32(b2), zsave=z, gate=gt, residual=x2, rows_per_batch=N,
                       p_drop=p_drop, seed=seeds[1], out_dtype=torch.float32)
        ctx.save_for_backward(x2, gamma, beta, sc, gt, w1, w2, h, mean, rstd, pre, a, z)
        ctx.cfg = (B, N, Cn, cdt, p_drop, seeds, scale is not None, gate is not None)
        return out.view(B, N, Cn)

    @staticmethod
    def backward(ctx, dout):
        x2, gamma, beta, sc, gt, w1, w2, h, mean, rstd, pre, a, z = ctx.saved_tensors
        B, N, Cn, cdt, p_drop, seeds, has_mod, has_gate = ctx.cfg
        dy = _f32(dout).view(B * N, Cn)
        dz, dgate, db2 = ops.branch_bwd(dy, z, gt, rows_per_batch=N, out_dtype=cdt, p_drop=p_drop, seed=seeds[1])
        dw2 = ops.gemm(dz, a, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32)
        # d(pre) = (dz W2) * dropmask(fc1) * gelu'(pre): fused into the GEMM epilogue
        dpre = ops.gemm(dz, cast_param(w2, cdt), b_kmajor=True, act=ops.ACT_GELU_GRAD, aux=pre, p_drop=p_drop, seed=seeds[0])
        dw1 = ops.gemm(dpre, h, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32)
        db1 = ops.colsum(dpre)
        dh = ops.gemm(dpre, cast_param(w1, cdt), b_kmajor=True)
        dx, dg, db, dsc, dsh = ops.layernorm_bwd(dh, x2, _f32(gamma), _f32(beta), sc, mean, rstd, dres=dy, rows_per_batch=N)
        v3 = lambda t: None if t is None else t.view(B, 1, Cn)
        return (dx.view(B, N, Cn), dg, db, v3(dsc) if has_mod else None, v3(dsh) if has_mod else None,
                v3(dgate) if has_gate else None, dw1, db1, dw2, db2, None, None, None)


# --------------------------------------------------------------------------------------------
# DRR ray sums
# --------------------------------------------------------------------------------------------
class DrrFn(torch.autograd.Function):
    """Ray-sum projection of a (B,D,H,W) volume along D (axis 0) or W (axis 2)."""

    @staticmethod
    def forward(ctx, vol, axis, exp_mode, mu, out_scale, clamp_min, transpose_out):
        v = vol.detach()
        if v.dtype not in (torch.float32, torch.bfloat16):
            v = v.float()
        v = v.contiguous()
        out = ops.drr_fwd(v, axis, exp_mode=exp_mode, mu=mu, out_scale=out_scale, clamp_min=clamp_min,
                          transpose_out=transpose_out)
        ctx.save_for_backward(v, out)
        ctx.cfg = (axis, exp_mode, mu, out_scale, clamp_min, transpose_out, vol.dtype)
        return out if out.dtype == vol.dtype else out.to(vol.dtype)

    @staticmethod
    def backward(ctx, dout):
        v, out = ctx.saved_tensors
        axis, exp_mode, mu, out_scale, clamp_min, transpose_out, in_dtype = ctx.cfg
        dv = ops.drr_bwd(v, out, dout.to(v.dtype), axis, exp_mode=exp_mode, mu=mu, out_scale=out_scale,
                         clamp_min=clamp_min, transpose_out=transpose_out)
        return (dv if dv.dtype == in_dtype else dv.to(in_dtype)), None, None, None, None, None, None


def drr_project(vol, axis, *, exp_mode, mu=0.3, out_scale=1.0, clamp_min=float("-inf"), transpose_out=False):
    return DrrFn.apply(vol, axis, exp_mode, mu, out_scale, clamp_min, transpose_out)


# --------------------------------------------------------------------------------------------
# conv stems on channels-last activations, normalisation, resize, loss
# --------------------------------------------------------------------------------------------
def conv_weight_2d(weight: torch.Tensor, dtype: torch.dtype, Kp: int) -> torch.Tensor:
    """(Cout, Cin, *k) parameter -> GEMM operand (Cout, Kp), column = tap * Cin + c; cached per version / optimizer step."""
    key = _cache_key(weight, dtype, Kp)
    hit = getattr(weight, "_hvc_w2d", None)
    if hit is not None and hit[0] == key:
        return hit[1]
    w2 = torch.zeros((weight.shape[0], Kp), dtype=dtype, device=weight.device)
    with torch.no_grad():
        _conv_w2d_fill(w2, weight)
    try:
        weight._hvc_w2d = (key, w2)
        _register(weight, "_hvc_w2d")
    except AttributeError:
        pass
    return w2


def conv_weight_frags(weight: torch.Tensor, dtype: torch.dtype, transposed: bool = False) -> torch.Tensor:
    """(Cout, Cin, 3, 3, 3) parameter -> the fragment-ordered operand of ops.conv3_halo (forward, or with transposed=True the input
    gradient: mirrored taps, channels swapped); cached like conv_weight_2d."""
    attr = "_hvc_wfragT" if transposed else "_hvc_wfrag"
    key = _cache_key(weight, dtype)
    hit = getattr(weight, attr, None)
    if hit is not None and hit[0] == key:
        return hit[1]
    wf = torch.empty((weight.numel(),), dtype=dtype, device=weight.device)
    with torch.no_grad():
        _CONV_FILL[attr](wf, weight)
    try:
        setattr(weight, attr, (key, wf))
        _register(weight, attr)
    except AttributeError:
        pass
    return wf


def conv_weight_2d_t(weight: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """(Cout, Cin, *k) parameter -> (Cin, taps*Cout) operand of the implicit-GEMM input gradient; cached like conv_weight_2d."""
    key = _cache_key(weight, dtype)
    hit = getattr(weight, "_hvc_w2dT", None)
    if hit is not None and hit[0] == key:
        return hit[1]
    wt = torch.empty((weight.shape[1], weight[0, 0].numel() * weight.shape[0]), dtype=dtype, device=weight.device)
    with torch.no_grad():
        _conv_w2dT_fill(wt, weight)
    try:
        weight._hvc_w2dT = (key, wt)
        _register(weight, "_hvc_w2dT")
    except AttributeError:
        pass
    return wt


def conv_weight_classes(weight: torch.Tensor, dtype: torch.dtype, geom) -> torch.Tensor:
    """(Cout, Cin, *k) parameter -> (Cin, taps*Cout) with class-major taps: operand of ops.conv_dx_classes; cached like conv_weight_2d."""
    sig = (geom.kernel, geom.stride, geom.pad, geom.src[0] > 1 or geom.kernel[0] > 1)
    key = _cache_key(weight, dtype, sig)
    hit = getattr(weight, "_hvc_w2dC", None)
    if hit is not None and hit[0] == key:
        return hit[1]
    order = [t for _, taps in ops.conv_dx_class_taps(geom) for t in taps]
    if sorted(order) != list(range(geom.taps)):
        raise RuntimeError("conv_weight_classes: every tap must reach exactly one parity class")
    wc = torch.empty((weight.shape[1], geom.taps * weight.shape[0]), dtype=dtype, device=weight.device)
    try:
        weight._hvc_w2dC_perm = torch.tensor(order, dtype=torch.long, device=weight.device)
        with torch.no_grad():
            _conv_w2dC_fill(wc, weight)
        weight._hvc_w2dC = (key, wc)
        _register(weight, "_hvc_w2dC")
    except AttributeError:
        perm = torch.tensor(order, dtype=torch.long, device=weight.device)
        with torch.no_grad():
            wc.view(weight.shape[1], geom.taps, weight.shape[0]).copy_(
                weight.detach().reshape(weight.shape[0], weight.shape[1], geom.taps).permute(1, 2, 0).index_select(1, perm))
    return wc


CONV_IMPLICIT = True          # C % 8 == 0 convolutions gather their patches inside the GEMM (no im2col / col2im round trip)
CONV_DIRECT = True            # Conv3d(1 -> 32 | 64, k3, p1) and Conv3d(C -> 1, k1) in bf16 run as streaming kernels (ops.conv_c1_* / conv_o1_*)
CONV_SLAB_BYTES = 2 << 30     # patch matrices larger than this are built (and rebuilt in backward) slab by slab along D


def _slab_plan(geom, esize, cap):
    """Output-depth slab size so that one sample's slab patch matrix stays under `cap` bytes."""
    per_od = geom.out[1] * geom.out[2] * geom.Kp * esize
    return max(1, min(geom.out[0], cap // max(per_od, 1)))


def _out_slab_geom(geom, od0, od1):
    """Source window + slab geometry producing output depths [od0, od1) of one sample."""
    s, p, K = geom.stride, geom.pad[0], geom.kernel[0]
    lo_raw = od0 * s - p
    lo, hi = max(0, lo_raw), min(geom.src[0], (od1 - 1) * s - p + K)
    g = ops.ConvGeometry(1, geom.C, (hi - lo, geom.src[1], geom.src[2]), geom.kernel, s, (lo - lo_raw, geom.pad[1], geom.pad[2]),
                         out_depth=od1 - od0)
    return lo, hi, g


def _conv_dx_slabs(dy5, w2d, geom, cdt):
    """Input gradient as dcol = dy W (GEMM) -> col2im, in depth slabs when the dcol matrix would exceed CONV_SLAB_BYTES."""
    cout = dy5.shape[-1]
    esize = 2 if cdt == torch.bfloat16 else 4
    if geom.M * geom.Kp * esize <= CONV_SLAB_BYTES and not geom.out_depth:
        return ops.col2im(ops.gemm(dy5.view(geom.M, cout), w2d, b_kmajor=True), geom)
    s, p, K = geom.stride, geom.pad[0], geom.kernel[0]
    dx = torch.empty((geom.B, *geom.src, geom.C), dtype=cdt, device=dy5.device)
    # input slab [d0, d1) gathers from output depths [od_lo, od_hi); slab size chosen on the dcol matrix
    step_in = max(1, _slab_plan(geom, esize, CONV_SLAB_BYTES) * s)
    for b in range(geom.B):
        for d0 in range(0, geom.src[0], step_in):
            d1 = min(geom.src[0], d0 + step_in)
            od_lo = max(0, -((-(d0 + p - K + 1)) // s))
            od_hi = min(geom.out[0], (d1 - 1 + p) // s + 1)
            if od_hi <= od_lo:
                dx[b, d0:d1].zero_()
                continue
            g = ops.ConvGeometry(1, geom.C, (d1 - d0, geom.src[1], geom.src[2]), geom.kernel, s,
                                 (p + d0 - od_lo * s, geom.pad[1], geom.pad[2]), out_depth=od_hi - od_lo)
            dcol = ops.gemm(dy5[b, od_lo:od_hi].view(g.M, cout), w2d, b_kmajor=True)
            dx[b, d0:d1] = ops.col2im(dcol, g)[0]
            del dcol
    return dx


class ConvFn(torch.autograd.Function):
    """Convolution on channels-last x (B, D, H, W, Cin) as im2col + MFMA GEMM (+ bias, + broadcast add of
    `addvec` (N_tok, Cout), i.e. pos_embed, on the last stem layer).  Output (B, OD, OH, OW, Cout).
    Volumes whose patch matrix exceeds CONV_SLAB_BYTES run slab by slab along D (forward, dW and dx), the patch
    matrix being rebuilt in the backward instead of saved."""

    @staticmethod
    def forward(ctx, x, weight, bias, addvec, geom, cdt, out_dtype):
        xc = _as_cdt(x, cdt)
        w2d = conv_weight_2d(weight, cdt, geom.Kp)
        cout = weight.shape[0]
        esize = 2 if cdt == torch.bfloat16 else 4
        implicit = CONV_IMPLICIT and geom.C % 8 == 0 and not geom.out_depth
        slabbed = not implicit and geom.M * geom.Kp * esize > CONV_SLAB_BYTES
        # single-channel layers as streaming kernels (csrc/conv_direct.hip): no patch matrix, the 1-channel side is what is saved
        direct = None
        if CONV_DIRECT and addvec is None and out_dtype == cdt:
            if ops.conv_c1_supported(geom, cout, cdt):
                direct = "c1"
            elif cout == 1 and geom.taps == 1 and geom.stride == 1 and not any(geom.pad) and ops.conv_o1_supported(geom.C, cdt) and not geom.out_depth:
                direct = "o1"
        if direct == "c1":
            xc = xc.contiguous()
            y = ops.conv_c1_fwd(xc, w2d, _f32(bias), geom)
            ctx.save_for_backward(xc, weight)
            ctx.cfg = (geom, cdt, x.dtype, bias is not None, None, False, direct)
            return y
        if direct == "o1":
            xc = xc.contiguous()
            y = ops.conv_o1_fwd(xc.view(geom.M, geom.C), w2d.view(-1), _f32(bias)).view(geom.B, *geom.out, 1)
            ctx.save_for_backward(xc, weight)
            ctx.cfg = (geom, cdt, x.dtype, bias is not None, None, False, direct)
            return y
        add = None
        if addvec is not None:
            if slabbed:
                raise RuntimeError("ConvFn: the fused pos_embed add is only supported for un-slabbed (token-sized) outputs")
            add = _f32(addvec).reshape(-1, cout)
        halo = implicit and CONV_DIRECT and add is None and out_dtype == cdt and ops.conv3_halo_supported(geom, cout, cdt)
        if halo:
            xc = xc.contiguous()
            y = ops.conv3_halo(xc, conv_weight_frags(weight, cdt), _f32(bias), cout)
            ctx.save_for_backward(xc, weight)
        elif implicit:
            xc = xc.contiguous()
            y = ops.conv_gemm(xc, w2d, geom, bias=_f32(bias), residual=add, residual_rows=add.shape[0] if add is not None else 0,
                              out_dtype=out_dtype)
            ctx.save_for_backward(xc, weight)
            y = y.view(geom.B, *geom.out, cout)
        elif not slabbed:
            col = ops.im2col(xc, geom)
            y = ops.gemm(col, w2d, bias=_f32(bias), residual=add, residual_rows=add.shape[0] if add is not None else 0,
                         out_dtype=out_dtype)
            ctx.save_for_backward(col, weight)
            y = y.view(geom.B, *geom.out, cout)
        else:
            y = torch.empty((geom.B, *geom.out, cout), dtype=out_dtype, device=x.device)
            step = _slab_plan(geom, esize, CONV_SLAB_BYTES)
            bf = _f32(bias)
            for b in range(geom.B):
                for od0 in range(0, geom.out[0], step):
                    od1 = min(geom.out[0], od0 + step)
                    lo, hi, g = _out_slab_geom(geom, od0, od1)
                    col = ops.im2col(xc[b:b + 1, lo:hi], g)
                    ops.gemm(col, w2d, bias=bf, out_dtype=out_dtype, out=y[b, od0:od1].view(g.M, cout))
                    del col
            ctx.save_for_backward(xc, weight)
        ctx.cfg = (geom, cdt, x.dtype, bias is not None, None if addvec is None else tuple(addvec.shape), slabbed, implicit)
        return y

    @staticmethod
    def backward(ctx, dy):
        saved, weight = ctx.saved_tensors
        geom, cdt, xdt, has_bias, add_shape, slabbed, implicit = ctx.cfg
        cout = weight.shape[0]
        taps = geom.taps
        dyc = _as_cdt(dy.reshape(geom.M, cout), cdt)
        if implicit == "c1":
            xc = saved
            dwt, db = ops.conv_c1_dw(xc, dyc.contiguous(), geom) if (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]) else (None, None)
            dw = dwt.reshape(weight.shape) if ctx.needs_input_grad[1] else None
            dx = None
            if ctx.needs_input_grad[0]:
                if geom.stride == 1:
                    dx = ops.conv_c1_dx(dyc.contiguous().view(geom.B, *geom.out, cout), conv_weight_2d(weight, cdt, geom.Kp), geom)
                else:
                    dx = _conv_dx_slabs(dyc.view(geom.B, *geom.out, cout), conv_weight_2d(weight, cdt, geom.Kp), geom, cdt)
                if dx.dtype != xdt:
                    dx = dx.to(xdt)
            return dx, dw, (db if has_bias and ctx.needs_input_grad[2] else None), None, None, None, None
        if implicit == "o1":
            xc = saved
            dx, dwv, db = ops.conv_o1_bwd(xc.view(geom.M, geom.C), dyc.contiguous().view(-1), conv_weight_2d(weight, cdt, geom.Kp).view(-1),
                                          need_dx=ctx.needs_input_grad[0])
            if dx is not None:
                dx = dx.view(geom.B, *geom.src, geom.C)
                if dx.dtype != xdt:
                    dx = dx.to(xdt)
            return (dx, dwv.reshape(weight.shape) if ctx.needs_input_grad[1] else None, db.reshape(1) if has_bias and ctx.needs_input_grad[2] else None,
                    None, None, None, None)
        db = ops.colsum(dyc) if has_bias and ctx.needs_input_grad[2] else None
        w2d = conv_weight_2d(weight, cdt, geom.Kp)
        dw = dx = None
        if implicit:
            xc = saved
            if ctx.needs_input_grad[1]:
                dw2d = ops.conv_gemm_dw(xc, dyc, geom)                                                   # (Cout, taps*C)
            if ctx.needs_input_grad[0]:
                if CONV_DIRECT and ops.conv3_halo_supported(geom, cout, cdt):
                    # the mirrored-kernel convolution of dy through the LDS halo tile
                    dx = ops.conv3_halo(dyc.view(geom.B, *geom.out, cout), conv_weight_frags(weight, cdt, transposed=True), None, geom.C)
                elif geom.stride == 1 and cout % 8 == 0:
                    # dx as an implicit GEMM over dy with the mirrored kernel: no dcol matrix, no col2im
                    gd = ops.ConvGeometry(geom.B, cout, geom.out, geom.kernel, 1, tuple(k - 1 - p for k, p in zip(geom.kernel, geom.pad)))
                    assert gd.out == geom.src
                    dx = ops.conv_gemm(dyc.view(geom.B, *geom.out, cout), conv_weight_2d_t(weight, cdt), gd, flip=True)
                    dx = dx.view(geom.B, *geom.src, geom.C)
                elif cout % 8 == 0:
                    # strided layer: one implicit GEMM over dy per parity class of the input position, written in place
                    dx = ops.conv_dx_classes(dyc.view(geom.B, *geom.out, cout), conv_weight_classes(weight, cdt, geom), geom)
                else:
                    dx = _conv_dx_slabs(dyc.view(geom.B, *geom.out, cout), w2d, geom, cdt)
        elif not slabbed:
            col = saved
            if ctx.needs_input_grad[1]:
                dw2d = ops.gemm(dyc, col, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32)        # (Cout, Kp)
            if ctx.needs_input_grad[0]:
                dx = ops.col2im(ops.gemm(dyc, w2d, b_kmajor=True), geom)
        else:
            xc = saved
            esize = 2 if cdt == torch.bfloat16 else 4
            dy5 = dyc.view(geom.B, *geom.out, cout)
            if ctx.needs_input_grad[1]:
                dw2d = torch.zeros((cout, geom.Kp), dtype=torch.float32, device=dy.device)
                step = _slab_plan(geom, esize, CONV_SLAB_BYTES)
                for b in range(geom.B):
                    for od0 in range(0, geom.out[0], step):
                        od1 = min(geom.out[0], od0 + step)
                        lo, hi, g = _out_slab_geom(geom, od0, od1)
                        col = ops.im2col(xc[b:b + 1, lo:hi], g)
                        dw2d += ops.gemm(dy5[b, od0:od1].view(g.M, cout), col, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32)
                        del col
            if ctx.needs_input_grad[0]:
                dx = _conv_dx_slabs(dy5, w2d, geom, cdt)
        if ctx.needs_input_grad[1]:
            dw = dw2d[:, :taps * geom.C].reshape(cout, taps, geom.C).permute(0, 2, 1).reshape(weight.shape)
        if dx is not None and dx.dtype != xdt:
            dx = dx.to(xdt)
        dadd = None
        if add_shape is not None and ctx.needs_input_grad[3]:
            dadd = dy.reshape(geom.B, -1, cout).float().sum(dim=0).reshape(add_shape)
        return dx, dw, db, dadd, None, None, None


class GroupNormSiluFn(torch.autograd.Function):
    """GroupNorm(G) + activation (ops.ACT_SILU / ops.ACT_GELU_ERF) on channels-last (B, ..., C)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, G, eps, act=0):
        shp = x.shape
        x3 = x.detach().contiguous().view(shp[0], -1, shp[-1])
        y, stats = ops.groupnorm_silu_fwd(x3, _f32(gamma), _f32(beta), G, eps, act)
        ctx.save_for_backward(x3, gamma, beta, stats)
        ctx.G, ctx.act = G, act
        return y.view(shp)

    @staticmethod
    def backward(ctx, dy):
        x3, gamma, beta, stats = ctx.saved_tensors
        dyc = dy.reshape(x3.shape)
        if dyc.dtype != x3.dtype:
            dyc = ops.cast(dyc, x3.dtype)
        dx, dg, db = ops.groupnorm_silu_bwd(x3, dyc, _f32(gamma), _f32(beta), stats, ctx.G, ctx.act)
        return dx.view(dy.shape), dg, db, None, None, None


class BnReluPoolFn(torch.autograd.Function):
    """BatchNorm2d + ReLU (+ MaxPool2d) on channels-last (N, H, W, C).  running stats are updated in place."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, pool, training, eps, momentum):
        xc = x.detach().contiguous()
        y, amax, stats = ops.bn_relu_pool_fwd(xc, _f32(gamma), _f32(beta), running_mean, running_var, pool, training, eps, momentum)
        ctx.save_for_backward(xc, gamma, beta, amax, stats)
        ctx.cfg = (pool, training)
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, gamma, beta, amax, stats = ctx.saved_tensors
        pool, training = ctx.cfg
        dyc = dy if dy.dtype == xc.dtype else ops.cast(dy.contiguous(), xc.dtype)
        dx, dg, db = ops.bn_relu_pool_bwd(xc, dyc, amax, _f32(gamma), _f32(beta), stats, pool, training)
        return dx, dg, db, None, None, None, None, None, None


class TrilinearFn(torch.autograd.Function):
    """(B, 1, d, h, w) fp32 -> (B, 1, D, H, W)."""

    @staticmethod
    def forward(ctx, x, size, align_corners=True):
        B = x.shape[0]
        ctx.in_size, ctx.ac = tuple(x.shape[2:]), bool(align_corners)
        return ops.trilinear_fwd(_f32(x).view(B, *ctx.in_size), tuple(size), ctx.ac).view(B, 1, *size)

    @staticmethod
    def backward(ctx, dy):
        B = dy.shape[0]
        dx = ops.trilinear_bwd(_f32(dy).view(B, *dy.shape[2:]), ctx.in_size, ctx.ac)
        return dx.view(B, 1, *ctx.in_size), None, None


class SsimL1LossFn(torch.autograd.Function):
    """(total, l1, ssim_loss) of direct_regression/model_direct.py:118-131 in one fused pass."""

    @staticmethod
    def forward(ctx, pred, target, l1_w, ssim_w, window):
        B = pred.shape[0]
        p = _f32(pred).view(B, *pred.shape[-3:])
        t = _f32(target).view(B, *target.shape[-3:])
        out, gmaps = ops.ssim_l1_fwd(p, t, window, l1_w, ssim_w)
        ctx.save_for_backward(p, t, gmaps)
        ctx.cfg = (l1_w, ssim_w, window, pred.shape, pred.dtype)
        return out

    @staticmethod
    def backward(ctx, dout):
        p, t, gmaps = ctx.saved_tensors
        l1_w, ssim_w, window, shape, dtype = ctx.cfg
        dp = ops.ssim_l1_bwd(p, t, gmaps, _f32(dout), window, l1_w, ssim_w).view(shape)
        return (dp if dp.dtype == dtype else dp.to(dtype)), None, None, None, None


class TotalVariationFn(torch.autograd.Function):
    """The three per-axis means of TotalVariationLoss (loss_multiscale.py:162-170) in one pass; backward gathers."""

    @staticmethod
    def forward(ctx, vol, eps):
        v = _f32(vol).view(-1, *vol.shape[-3:])            # (B, C, D, H, W): channels are further samples
        ctx.save_for_backward(v)
        ctx.cfg = (eps, vol.shape, vol.dtype)
        return ops.tv3d_fwd(v, eps)

    @staticmethod
    def backward(ctx, dout):
        (v,) = ctx.saved_tensors
        eps, shape, dtype = ctx.cfg
        dv = ops.tv3d_bwd(v, _f32(dout), eps).view(shape)
        return (dv if dv.dtype == dtype else dv.to(dtype)), None


class SpectralL1Fn(torch.autograd.Function):
    """(low, high) magnitude-spectrum L1 terms of FrequencyLoss (loss_multiscale.py:203-236) from the (B,D,H,W,2) real views
    of the two spectra; the gradient flows to the prediction's spectrum only (the target carries none in the trainers)."""

    @staticmethod
    def forward(ctx, pred_spec, target_spec):
        p, t = pred_spec.detach().contiguous(), target_spec.detach().contiguous()
        ctx.save_for_backward(p, t)
        return ops.spectral_l1_fwd(p, t)

    @staticmethod
    def backward(ctx, dout):
        p, t = ctx.saved_tensors
        return ops.spectral_l1_bwd(p, t, _f32(dout)), None


class ResizeLossFn(torch.autograd.Function):
    """mean |bilinear(proj) - target| (mode 0) or mean squared difference (mode 1) without materialising the resized image
    (tails of DRRReprojectionLoss, loss_multiscale.py:269-293, and ProjectionLoss, models/diagnostic_losses.py:161-169)."""

    @staticmethod
    def forward(ctx, proj, target, align_corners, mode):
        p = _f32(proj)
        t = target.detach()
        if t.dtype != torch.float32:
            t = t.float()
        ctx.save_for_backward(p, t)
        ctx.cfg = (bool(align_corners), int(mode), proj.dtype)
        return ops.resize_loss_fwd(p, t, align_corners, mode)[0]

    @staticmethod
    def backward(ctx, dout):
        p, t = ctx.saved_tensors
        ac, mode, dtype = ctx.cfg
        dres = ops.resize_loss_grad(p, t, _f32(dout).reshape(1), ac, mode)
        B, h, w = p.shape
        dp = ops.trilinear_bwd(dres.view(B, 1, *dres.shape[1:]), (1, h, w), ac).view(B, h, w)      # adjoint of the resize, depth 1
        return (dp if dp.dtype == dtype else dp.to(dtype)), None, None, None


class ViewMeanGapFn(torch.autograd.Function):
    """Mean over the V views of the channels-last X-ray feature maps + global average pool, one pass
    (models/diagnostic_losses.py:126, :131).  feats (B*V, P, E) -> (B, P, E) fp32, (B, E) fp32."""

    @staticmethod
    def forward(ctx, feats, V):
        f = feats.detach().contiguous()
        ctx.cfg = (V, f.dtype)
        return ops.view_mean_gap_fwd(f, V)

    @staticmethod
    def backward(ctx, dmean, dpooled):
        V, dtype = ctx.cfg
        if not ctx.needs_input_grad[0]:          # frozen encoder: no feature-map gradient to form
            return None, None
        # both outputs are always used downstream (context tokens and the pooled conditioning), and autograd materialises
        # the gradient of an unused one as zeros
        df = ops.view_mean_gap_bwd(_f32(dmean), _f32(dpooled), V, dtype)
        return df, None
