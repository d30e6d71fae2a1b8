"""MI355X-native counterparts of the reference's models/vit_components.py.

Parameter containers (nn.Linear / nn.Dropout children, created in the reference's order so that a
seeded construction reproduces the reference's initial weights and state_dict keys) whose forwards
call the HIP kernels through hvc.functional.  There is no eager fallback: CPU tensors raise.
"""
import math

import torch
import torch.nn as nn

from hvc import functional as HF


def _drop(module_training, p):
    return p if (module_training and p > 0) else 0.0


class MultiHeadSelfAttention(nn.Module):
    """Reference: models/vit_components.py:13-57.  State: qkv.weight (3C,C), proj.{weight,bias}.
    qkv output columns are [q(h,d) | k(h,d) | v(h,d)]; the fused kernel reads them in place."""

    def __init__(self, embed_dim, num_heads=8, dropout=0.1):
        super().__init__()
        assert embed_dim % num_heads == 0
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.head_dim = embed_dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.qkv = nn.Linear(embed_dim, embed_dim * 3, bias=False)
        self.attn_drop = nn.Dropout(dropout)
        self.proj = nn.Linear(embed_dim, embed_dim)
        self.proj_drop = nn.Dropout(dropout)

    def forward(self, x):
        B, N, Cn = x.shape
        cdt = HF.compute_dtype(x)
        pa, pp = _drop(self.training, self.attn_drop.p), _drop(self.training, self.proj_drop.p)
        qkv = HF.linear(x, self.qkv.weight, None, cdt).view(B, N, 3, self.num_heads, self.head_dim)
        o = HF.PackedSelfAttnFn.apply(qkv, self.scale, pa, HF.new_seed() if pa else 0)
        return HF.linear(o, self.proj.weight, self.proj.bias, cdt, p_drop=pp, seed=HF.new_seed() if pp else 0)


class MultiHeadCrossAttention(nn.Module):
    """Reference: models/vit_components.py:60-119.  State: q.weight (C,C), kv.weight (2C,Cc),
    proj.{weight,bias}.  `attention_weights` is only materialised when store_attention=True
    (diagnostic side output, reference :107-108)."""

    def __init__(self, embed_dim, context_dim, num_heads=8, dropout=0.1, store_attention=False):
        super().__init__()
        assert embed_dim % num_heads == 0
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.head_dim = embed_dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.store_attention = store_attention
        self.q = nn.Linear(embed_dim, embed_dim, bias=False)
        self.kv = nn.Linear(context_dim, embed_dim * 2, bias=False)
        self.attn_drop = nn.Dropout(dropout)
        self.proj = nn.Linear(embed_dim, embed_dim)
        self.proj_drop = nn.Dropout(dropout)
        self.attention_weights = None

    def _store_probs(self, q, kv):
        # (B,h,N,M) probabilities for diagnostics only; not on the training hot path.
        with torch.no_grad():
            qq = q.permute(0, 2, 1, 3).float()
            kk = kv[:, :, 0].permute(0, 2, 1, 3).float()
            self.attention_weights = ((qq @ kk.transpose(-2, -1)) * self.scale).softmax(dim=-1)

    def forward(self, x, context):
        B, N, Cn = x.shape
        M = context.shape[1]
        cdt = HF.compute_dtype(x)
        pa, pp = _drop(self.training, self.attn_drop.p), _drop(self.training, self.proj_drop.p)
        q = HF.linear(x, self.q.weight, None, cdt).view(B, N, self.num_heads, self.head_dim)
        kv = HF.linear(context, self.kv.weight, None, cdt).view(B, M, 2, self.num_heads, self.head_dim)
        if self.store_attention:
            self._store_probs(q, kv)
        o = HF.PackedCrossAttnFn.apply(q, kv, self.scale, pa, HF.new_seed() if pa else 0)
        return HF.linear(o, self.proj.weight, self.proj.bias, cdt, p_drop=pp, seed=HF.new_seed() if pp else 0)


class AdaLNModulation(nn.Module):
    """Reference: models/vit_components.py:122-149.  Linear(cond_dim, 6C) zero-initialised;
    returns (shift_sa, scale_sa, gate_sa, shift_mlp, scale_mlp, gate_mlp), each (B,1,C), fp32."""

    def __init__(self, embed_dim, cond_dim):
        super().__init__()
        self.linear = nn.Linear(cond_dim, embed_dim * 6, bias=True)
        nn.init.zeros_(self.linear.weight)
        nn.init.zeros_(self.linear.bias)

    def forward(self, x, cond):
        params = HF.linear(cond.float(), self.linear.weight, self.linear.bias, torch.float32, torch.float32).unsqueeze(1)
        return tuple(params.chunk(6, dim=-1))


class SinusoidalTimeEmbedding(nn.Module):
    """Reference: models/vit_components.py:152-174 (kept for the import surface; no in-scope model calls it)."""

    def __init__(self, embed_dim):
        super().__init__()
        self.embed_dim = embed_dim

    def forward(self, t):
        half = self.embed_dim // 2
        freq = torch.exp(torch.arange(half, device=t.device) * -(math.log(10000) / (half - 1)))
        ang = t[:, None] * freq[None, :]
        return torch.cat([ang.sin(), ang.cos()], dim=-1)
