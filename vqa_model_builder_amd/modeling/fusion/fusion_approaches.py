"""Stand-alone fusion modules of the reference (``src/modeling/fusion/fusion_approaches.py``; used by its examples, not by
``VietnameseVQAModel``: SURVEY section 8f rank 4) on the HIP path.

``CrossAttentionFusion`` (:59-191) with its bidirectional ``CrossAttentionBlock`` (:194-281) is built: same names, constructor
arguments, attribute names and ``state_dict`` keys; every forward is a chain of the per-op HIP autograd nodes (hip/ops.py).
``QFormerFusion`` / ``SingleStreamFusion`` (:284-683) are declared and raise at construction.
"""

from abc import ABC, abstractmethod
from typing import Optional, Tuple

import torch
import torch.nn as nn

from ...hip import kernels as K
from ...hip import ops
from ...hip.kernels import ACT_GELU, Drop
from ..meta_arch.vqa_model import _MHAParams


class BaseFusion(ABC, nn.Module):
    """Reference fusion_approaches.py:13-56."""

    def __init__(self, vision_dim: int, text_dim: int, output_dim: int):
        super().__init__()
        self.vision_dim, self.text_dim, self.output_dim = vision_dim, text_dim, output_dim

    @abstractmethod
    def forward(self, vision_features, text_features, vision_mask=None, text_mask=None):
        pass

    @abstractmethod
    def get_output_dim(self) -> int:
        pass


def _ffn(seq, x, p, training, stream0):
    seed = ops.new_seed() if (training and p > 0) else 0
    d = (lambda st: Drop(p, seed, st)) if (training and p > 0) else (lambda st: Drop())
    h = ops.linear(x, seq[0].weight, seq[0].bias, act=ACT_GELU, drop=d(stream0))
    return ops.linear(h, seq[3].weight, seq[3].bias, drop=d(stream0 + 1))


def _ln(m, x):
    return ops.layer_norm(x, m.weight, m.bias, m.eps)


class CrossAttentionBlock(nn.Module):
    """Reference :194-281: text attends to vision, then vision attends to the UPDATED text; each side post-LN with a GELU FFN."""

    def __init__(self, dim: int, num_heads: int, intermediate_dim: int, dropout: float = 0.1):
        super().__init__()
        mk_ffn = lambda: nn.Sequential(nn.Linear(dim, intermediate_dim), nn.GELU(), nn.Dropout(dropout), nn.Linear(intermediate_dim, dim),
                                       nn.Dropout(dropout))
        self.v2t_attention = _MHAParams(dim, num_heads, dropout)
        self.v2t_norm1, self.v2t_norm2 = nn.LayerNorm(dim), nn.LayerNorm(dim)
        self.v2t_ffn = mk_ffn()
        self.t2v_attention = _MHAParams(dim, num_heads, dropout)
        self.t2v_norm1, self.t2v_norm2 = nn.LayerNorm(dim), nn.LayerNorm(dim)
        self.t2v_ffn = mk_ffn()
        self._p = dropout

    def forward(self, vision_features, text_features, vision_mask=None, text_mask=None) -> Tuple[torch.Tensor, torch.Tensor]:
        p, tr = self._p, self.training
        a, _ = self.v2t_attention(text_features, vision_features, vision_features,
                                  key_padding_mask=~vision_mask if vision_mask is not None else None)
        text_features = _ln(self.v2t_norm1, ops.add(text_features, a))
        text_features = _ln(self.v2t_norm2, ops.add(text_features, _ffn(self.v2t_ffn, text_features, p, tr, 191)))
        a, _ = self.t2v_attention(vision_features, text_features, text_features,
                                  key_padding_mask=~text_mask if text_mask is not None else None)
        vision_features = _ln(self.t2v_norm1, ops.add(vision_features, a))
        vision_features = _ln(self.t2v_norm2, ops.add(vision_features, _ffn(self.t2v_ffn, vision_features, p, tr, 193)))
        return vision_features, text_features


class _MeanTokensFn(torch.autograd.Function):
    """torch.mean(x, dim=1) over the token axis (fusion_approaches.py:169-170; padded tokens included, as there)."""

    @staticmethod
    def forward(ctx, x):
        B, S, D = x.shape
        out = torch.empty((B, D), dtype=torch.float32, device=x.device)
        K.rows_mean(x.contiguous().float(), S, B, D, out=out)
        ctx.shape = (B, S, D)
        return out

    @staticmethod
    def backward(ctx, dy):
        B, S, D = ctx.shape
        dx, _ = K.repeat_rows(dy.contiguous().float(), B * S, D, S, 0, alpha=1.0 / S)
        return dx.view(B, S, D)


class CrossAttentionFusion(BaseFusion):
    """Reference :59-191."""

    def __init__(self, vision_dim: int = 768, text_dim: int = 768, output_dim: int = 768, num_attention_heads: int = 8, num_layers: int = 4,
                 intermediate_dim: int = 3072, dropout: float = 0.1, fusion_method: str = 'concat'):
        super().__init__(vision_dim, text_dim, output_dim)
        self.num_attention_heads, self.num_layers, self.fusion_method = num_attention_heads, num_layers, fusion_method
        self.vision_projection = nn.Linear(vision_dim, output_dim) if vision_dim != output_dim else nn.Identity()
        self.text_projection = nn.Linear(text_dim, output_dim) if text_dim != output_dim else nn.Identity()
        self.cross_attention_layers = nn.ModuleList([CrossAttentionBlock(dim=output_dim, num_heads=num_attention_heads,
                                                                         intermediate_dim=intermediate_dim, dropout=dropout)
                                                     for _ in range(num_layers)])
        fusion_input_dim = output_dim * 2 if fusion_method == 'concat' else output_dim
        self.fusion_layer = nn.Sequential(nn.Linear(fusion_input_dim, output_dim), nn.LayerNorm(output_dim), nn.GELU(), nn.Dropout(dropout),
                                          nn.Linear(output_dim, output_dim), nn.LayerNorm(output_dim))
        self.pooling = nn.AdaptiveAvgPool1d(1)              # declared by the reference, never called (:140)
        self._p = dropout

    def forward(self, vision_features, text_features, vision_mask: Optional[torch.Tensor] = None, text_mask: Optional[torch.Tensor] = None):
        if not vision_features.is_cuda:
            raise RuntimeError('CrossAttentionFusion: HIP path needs GPU tensors; no CPU fallback on the product path')
        v, t = vision_features, text_features
        if isinstance(self.vision_projection, nn.Linear):
            v = ops.linear(v, self.vision_projection.weight, self.vision_projection.bias)
        if isinstance(self.text_projection, nn.Linear):
            t = ops.linear(t, self.text_projection.weight, self.text_projection.bias)
        for layer in self.cross_attention_layers:
            v, t = layer(v, t, vision_mask, text_mask)
        vp, tp = _MeanTokensFn.apply(v), _MeanTokensFn.apply(t)
        if self.fusion_method == 'concat':
            fused = torch.cat([vp, tp], dim=-1)                      # [B, 2D]: plumbing
        elif self.fusion_method == 'add':
            fused = ops.add(vp, tp)
        elif self.fusion_method == 'multiply':
            fused = vp * tp                                          # [B, D] element-wise product of two pooled vectors
        else:
            raise ValueError(f"Unknown fusion method: {self.fusion_method}")
        f = self.fusion_layer
        h = _ln(f[1], ops.linear(fused, f[0].weight, f[0].bias))
        h = ops.activation(h, ACT_GELU, self._p, self.training)
        return _ln(f[5], ops.linear(h, f[4].weight, f[4].bias))

    def get_output_dim(self) -> int:
        return self.output_dim


class QFormerFusion(BaseFusion):
    """Reference :284-400 (examples only)."""

    def __init__(self, *a, **kw):
        raise NotImplementedError('QFormerFusion is examples-only in the reference and not built on the HIP path')

    def forward(self, *a, **kw):
        raise NotImplementedError

    def get_output_dim(self):
        raise NotImplementedError


class SingleStreamFusion(BaseFusion):
    """Reference :516-677 (examples only)."""

    def __init__(self, *a, **kw):
        raise NotImplementedError('SingleStreamFusion is examples-only in the reference and not built on the HIP path')

    def forward(self, *a, **kw):
        raise NotImplementedError

    def get_output_dim(self):
        raise NotImplementedError


def create_fusion_model(fusion_type: str = 'cross_attention', **kwargs) -> BaseFusion:
    """Reference :680-737."""
    registry = {'cross_attention': CrossAttentionFusion, 'qformer': QFormerFusion, 'q_former': QFormerFusion,
                'single_stream': SingleStreamFusion, 'vilt': SingleStreamFusion}
    if fusion_type not in registry:
        raise ValueError(f"Unknown fusion type: {fusion_type}. Available types: {', '.join(registry.keys())}")
    return registry[fusion_type](**kwargs)
