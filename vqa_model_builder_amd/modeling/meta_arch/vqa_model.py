"""Vietnamese VQA classification model on the MI355X HIP path.

Same class names, constructor signatures, attribute names, ``state_dict`` keys and ``VQAOutput`` as the reference's
``src/modeling/meta_arch/vqa_model.py`` (cited per class below), so the reference's training loops
(``training_pipeline.py:440-534``, ``vqa_trainer.py:746-823``) drive it unchanged.  ``forward`` of every module here
launches HIP kernels (hip/blocks.py, hip/ops.py); a CPU tensor raises -- there is no CPU fallback on this path.
"""

import warnings
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from ...hip import kernels as K
from ...hip import ops
from ...hip.blocks import CrossModalAttentionRunner, TailRunner
from ...hip.kernels import ACT_NONE, ACT_RELU, Drop
from ...hip.kernels import skip_weight_grads as K_skip
from .backbones import ClipVisionBackbone, RobertaBackbone, _BlockFn, _Weights, _flatten_param_keys, _require_cuda, _split_grads
from .vqa_config import (AnswerHeadConfig, FusionConfig, KnowledgeConfig, MOEConfig, TextEncoderConfig,  # noqa: F401
                         VisualEncoderConfig, VQAModelConfig)


@dataclass
class VQAOutput:
    """Reference vqa_model.py:24-48 (same nine fields, same order)."""
    logits: torch.Tensor
    loss: Optional[torch.Tensor] = None
    predictions: Optional[torch.Tensor] = None
    visual_features: Optional[torch.Tensor] = None
    text_features: Optional[torch.Tensor] = None
    fused_features: Optional[torch.Tensor] = None
    knowledge_features: Optional[torch.Tensor] = None
    moe_info: Optional[Dict[str, Any]] = None
    auxiliary_outputs: Optional[Dict[str, Any]] = None


# architecture hyper-parameters of the two hub names the reference defaults to (no network needed to build them)
_KNOWN_VISION = {
    'openai/clip-vit-base-patch32': dict(hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12,
                                         image_size=224, patch_size=32),
    'openai/clip-vit-base-patch16': dict(hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12,
                                         image_size=224, patch_size=16),
}
_KNOWN_TEXT = {
    'vinai/phobert-base': dict(vocab_size=64001, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                               intermediate_size=3072, max_position_embeddings=258, type_vocab_size=1, pad_token_id=1),
    'vinai/phobert-base-v2': dict(vocab_size=64001, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                                  intermediate_size=3072, max_position_embeddings=258, type_vocab_size=1, pad_token_id=1),
}


def _try_load_pretrained(native: nn.Module, hf_loader, name: str):
    """Copies hub weights into the native containers when transformers can reach them; random init otherwise."""
    try:
        hf = hf_loader(name)
        missing, unexpected = native.load_state_dict(hf.state_dict(), strict=False)
        del hf
    except Exception as e:  # offline / not cached: keep the architecture, warn loudly
        warnings.warn(f'could not load pretrained weights for {name!r} ({type(e).__name__}: {e}); using random init')


class VisualEncoder(nn.Module):
    """Reference vqa_model.py:51-131.  ``backbone`` is the native CLIP ViT tower (HIP)."""

    def __init__(self, config: VisualEncoderConfig):
        super().__init__()
        self.config = config
        self._init_backbone()
        if hasattr(self, 'backbone_dim') and self.backbone_dim != config.output_dim:
            self.projection = nn.Linear(self.backbone_dim, config.output_dim)
        else:
            self.projection = None

    def _init_backbone(self):
        name = self.config.model_name
        arch = getattr(self.config, 'arch', None) or _KNOWN_VISION.get(name)
        if arch is None:
            try:
                from transformers import CLIPVisionConfig
                hc = CLIPVisionConfig.from_pretrained(name)
                arch = dict(hidden_size=hc.hidden_size, intermediate_size=hc.intermediate_size, num_hidden_layers=hc.num_hidden_layers,
                            num_attention_heads=hc.num_attention_heads, image_size=hc.image_size, patch_size=hc.patch_size)
            except ImportError:
                raise ImportError("transformers required for visual encoder")
        if 'clip' not in name.lower() and getattr(self.config, 'arch', None) is None and name not in _KNOWN_VISION:
            raise NotImplementedError(f'HIP visual backbone implemented for CLIP ViT towers only (got {name!r}); Swin is a '
                                      'README/enum name in the reference, never constructed by it (SURVEY F10)')
        self.processor = None
        self.backbone = ClipVisionBackbone(**arch)
        self.backbone_dim = arch['hidden_size']
        if self.config.pretrained and getattr(self.config, 'arch', None) is None:
            def loader(n):
                from transformers import CLIPVisionModel
                return CLIPVisionModel.from_pretrained(n)
            _try_load_pretrained(self.backbone, loader, name)
        if self.config.freeze_backbone:
            for p in self.backbone.parameters():
                p.requires_grad = False

    def forward(self, pixel_values: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        outputs = self.backbone(pixel_values=pixel_values)
        spatial = outputs.last_hidden_state
        pooled = spatial[:, 0, :]
        if self.projection is not None:
            pooled = ops.linear(pooled, self.projection.weight, self.projection.bias)
            spatial = ops.linear(spatial, self.projection.weight, self.projection.bias)
        return pooled, spatial


class TextEncoder(nn.Module):
    """Reference vqa_model.py:134-234.  ``encoder`` is the native RoBERTa/PhoBERT stack (HIP)."""

    def __init__(self, config: TextEncoderConfig):
        super().__init__()
        self.config = config
        self._init_encoder()
        if hasattr(self, 'encoder_dim') and self.encoder_dim != config.output_dim:
            self.projection = nn.Linear(self.encoder_dim, config.output_dim)
        else:
            self.projection = None

    def _init_encoder(self):
        name = self.config.model_name
        arch = getattr(self.config, 'arch', None) or _KNOWN_TEXT.get(name)
        if arch is None:
            try:
                from transformers import AutoConfig
                hc = AutoConfig.from_pretrained(name)
                arch = dict(vocab_size=hc.vocab_size, hidden_size=hc.hidden_size, num_hidden_layers=hc.num_hidden_layers,
                            num_attention_heads=hc.num_attention_heads, intermediate_size=hc.intermediate_size,
                            max_position_embeddings=hc.max_position_embeddings, type_vocab_size=hc.type_vocab_size,
                            pad_token_id=hc.pad_token_id)
            except ImportError:
                raise ImportError("transformers required for text encoder")
        self.tokenizer = None
        self.encoder = RobertaBackbone(**arch)
        self.encoder_dim = arch['hidden_size']
        if self.config.pretrained and getattr(self.config, 'arch', None) is None:
            def loader(n):
                from transformers import AutoModel
                return AutoModel.from_pretrained(n)
            _try_load_pretrained(self.encoder, loader, name)
        if self.config.freeze_encoder:
            for p in self.encoder.parameters():
                p.requires_grad = False

    def _pool_features(self, hidden_states, attention_mask):
        """Reference :179-204.  'cls' is a row view; 'mean'/'max' are masked reductions over <= 64 tokens."""
        strat = self.config.pooling_strategy
        if strat == 'cls':
            return hidden_states[:, 0, :]
        mask = attention_mask.unsqueeze(-1).expand(hidden_states.size())
        if strat == 'max':
            return hidden_states.masked_fill(mask == 0, -1e9).max(dim=1)[0]
        mask = mask.float()
        return (hidden_states * mask).sum(dim=1) / mask.sum(dim=1).clamp(min=1e-9)

    def forward(self, input_ids, attention_mask):
        outputs = self.encoder(input_ids=input_ids, attention_mask=attention_mask)
        seq = outputs.last_hidden_state
        pooled = self._pool_features(seq, attention_mask)
        if self.projection is not None:
            pooled = ops.linear(pooled, self.projection.weight, self.projection.bias)
            seq = ops.linear(seq, self.projection.weight, self.projection.bias)
        return pooled, seq


class _MHAParams(nn.Module):
    """Parameter layout of ``nn.MultiheadAttention`` (packed in_proj + out_proj), nothing else."""

    def __init__(self, embed_dim, num_heads, dropout=0.0):
        super().__init__()
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.constant_(self.out_proj.bias, 0.0)

    def forward(self, query, key, value, key_padding_mask=None, need_weights=False):
        out = ops.multi_head_attention(query, key, value, self.in_proj_weight, self.in_proj_bias, self.out_proj.weight,
                                       self.out_proj.bias, self.num_heads, key_padding_mask, self.dropout, self.training)
        return out, None


class CrossModalAttention(nn.Module):
    """Reference vqa_model.py:237-311: post-LN block of self-MHA, cross-MHA and a 4x GELU FFN.
    Runs as ONE hand-scheduled forward/backward (hip.blocks.CrossModalAttentionRunner)."""

    def __init__(self, embed_dim: int, num_heads: int = 8, dropout: float = 0.1):
        super().__init__()
        self.self_attn = _MHAParams(embed_dim, num_heads, dropout)
        self.cross_attn = _MHAParams(embed_dim, num_heads, dropout)
        self.ffn = nn.Sequential(nn.Linear(embed_dim, embed_dim * 4), nn.GELU(), nn.Dropout(dropout),
                                 nn.Linear(embed_dim * 4, embed_dim), nn.Dropout(dropout))
        self.norm1, self.norm2, self.norm3 = nn.LayerNorm(embed_dim), nn.LayerNorm(embed_dim), nn.LayerNorm(embed_dim)
        self.dropout = nn.Dropout(dropout)
        W = _Weights()
        P, S = W.params, W.shadows
        D = embed_dim
        S.add('sa_in_w', self.self_attn.in_proj_weight, (3 * D, D))
        S.add('sa_out_w', self.self_attn.out_proj.weight, (D, D))
        S.add('ca_in_w', self.cross_attn.in_proj_weight, (3 * D, D))
        S.add('ca_out_w', self.cross_attn.out_proj.weight, (D, D))
        S.add('ffn0_w', self.ffn[0].weight, (4 * D, D))
        S.add('ffn3_w', self.ffn[3].weight, (D, 4 * D))
        P['sa_in_w'], P['sa_in_b'] = self.self_attn.in_proj_weight, self.self_attn.in_proj_bias
        P['sa_out_w'], P['sa_out_b'] = self.self_attn.out_proj.weight, self.self_attn.out_proj.bias
        P['ca_in_w'], P['ca_in_b'] = self.cross_attn.in_proj_weight, self.cross_attn.in_proj_bias
        P['ca_out_w'], P['ca_out_b'] = self.cross_attn.out_proj.weight, self.cross_attn.out_proj.bias
        P['ffn0_w'], P['ffn0_b'] = self.ffn[0].weight, self.ffn[0].bias
        P['ffn3_w'], P['ffn3_b'] = self.ffn[3].weight, self.ffn[3].bias
        for i, n in enumerate((self.norm1, self.norm2, self.norm3), 1):
            P[f'n{i}.w'], P[f'n{i}.b'] = n.weight, n.bias
        self._W = W
        self._runner = CrossModalAttentionRunner(W, D, num_heads, dropout)
        self._flat = _flatten_param_keys(W.params)
        self._masks = (None, None)
        self._first_only = False

    def _hip_forward(self, query, key_value):
        self._W.shadows.refresh(query.device)
        qm, km = self._masks
        return self._runner.forward(query, key_value, qm, km, self.training, first_only=self._first_only)

    def _hip_backward(self, saved, dout, needs):
        with K_skip(not any(p.requires_grad for _, p in self._flat)):     # frozen fusion layer: dX only, no weight-gradient GEMMs
            G, dq, dkv = self._runner.backward(saved, dout, need_dkv=needs[1])
        return [dq if needs[0] else None, dkv], _split_grads(self._flat, G)

    def forward(self, query, key_value, query_mask=None, kv_mask=None, first_token_only=False):
        """``first_token_only`` (not in the reference signature): returns [B, 1, D] = row 0 of the block's output, computing only
        what that row depends on (MultimodalFusion sets it for its last layer, whose other rows nobody reads)."""
        _require_cuda(query, 'CrossModalAttention')
        self._masks = (query_mask, kv_mask)
        self._first_only = bool(first_token_only)
        try:
            return _BlockFn.apply(self, 2, query, key_value, *[p for _, p in self._flat])
        finally:
            self._masks = (None, None)
            self._first_only = False


class MultimodalFusion(nn.Module):
    """Reference vqa_model.py:314-433.  Unknown ``fusion_type`` (e.g. 'mcan') takes the add branch, as there (F3)."""

    def __init__(self, config: FusionConfig):
        super().__init__()
        self.config = config
        if config.fusion_type == 'cross_attention':
            self.fusion_layers = nn.ModuleList([CrossModalAttention(config.hidden_dim, config.num_heads, config.dropout)
                                                for _ in range(config.num_layers)])
            self.output_proj = nn.Linear(config.hidden_dim, config.output_dim)
        elif config.fusion_type == 'concat':
            self.fusion_layer = nn.Sequential(nn.Linear(config.hidden_dim * 2, config.hidden_dim), nn.ReLU(),
                                              nn.Dropout(config.dropout), nn.Linear(config.hidden_dim, config.output_dim))
        elif config.fusion_type == 'bilinear':
            self.bilinear = nn.Bilinear(config.hidden_dim, config.hidden_dim, config.output_dim)
        else:
            self.fusion_layer = nn.Linear(config.hidden_dim, config.output_dim)
        self.layer_norm = nn.LayerNorm(config.output_dim) if config.use_layer_norm else None

    @staticmethod
    def _cls(x):
        return x[:, 0, :] if x.dim() == 3 else x

    def forward(self, visual_features, text_features, visual_mask=None, text_mask=None, project=True):
        """``project=False`` (not in the reference signature; cross_attention only): returns token 0 of the last fusion layer [B, D]
        BEFORE output_proj / layer_norm -- the model's tail node (_Tail) runs those together with the dropout and the answer head."""
        ft = self.config.fusion_type
        if ft == 'cross_attention':
            last = len(self.fusion_layers) - 1
            for i, layer in enumerate(self.fusion_layers):
                # only token 0 of the last layer's output is read below: that layer computes just what token 0 depends on
                text_features = layer(text_features, visual_features, text_mask, visual_mask, first_token_only=(i == last))
            if not project:
                return text_features[:, 0, :]
            fused = ops.linear(text_features[:, 0, :], self.output_proj.weight, self.output_proj.bias)
        elif ft == 'concat':
            combined = torch.cat([self._cls(visual_features), self._cls(text_features)], dim=-1)
            l0, l3 = self.fusion_layer[0], self.fusion_layer[3]
            p = self.config.dropout if self.training else 0.0
            h = ops.linear(combined, l0.weight, l0.bias, act=ACT_RELU, drop=Drop(p, ops.new_seed() if p > 0 else 0, 11))
            fused = ops.linear(h, l3.weight, l3.bias)
        elif ft == 'bilinear':
            fused = ops.bilinear(self._cls(visual_features), self._cls(text_features), self.bilinear.weight, self.bilinear.bias)
        else:
            s = ops.add(self._cls(visual_features), self._cls(text_features))
            fused = ops.linear(s, self.fusion_layer.weight, self.fusion_layer.bias)
        if self.layer_norm is not None:
            fused = ops.layer_norm(fused, self.layer_norm.weight, self.layer_norm.bias, self.layer_norm.eps)
        return fused


# True: the chain behind the last fusion layer runs as ONE autograd node (hip.blocks.TailRunner) where its shapes allow.  False: the
# op-by-op chain (what the runner is tested against).
TAIL_RUNNER = True


class _Tail:
    """Owner of the tail node: [fusion.output_proj -> fusion.layer_norm ->] model.dropout -> answer_head.classifier.  Not a Module:
    it only REFERENCES the parameters (names / state_dict stay the reference's)."""

    def __init__(self, model, with_projection, with_dropout):
        head, fusion = model.answer_head, model.fusion
        lins = [m for m in head.classifier if isinstance(m, nn.Linear)]
        self.model, self.dims = model, [lins[0].in_features] + [l.out_features for l in lins]
        W = _Weights()
        pre, eps = None, 1e-5
        if with_projection:
            W.params['proj_w'], W.params['proj_b'] = fusion.output_proj.weight, fusion.output_proj.bias
            W.shadows.add('proj_w', fusion.output_proj.weight, tuple(fusion.output_proj.weight.shape))
            if fusion.layer_norm is not None:
                W.params['ln.w'], W.params['ln.b'] = fusion.layer_norm.weight, fusion.layer_norm.bias
                eps = fusion.layer_norm.eps
            pre = (fusion.output_proj.in_features, fusion.layer_norm is not None)
        for i, lin in enumerate(lins):
            W.params[f'l{i}_w'], W.params[f'l{i}_b'] = lin.weight, lin.bias
            W.shadows.add(f'l{i}_w', lin.weight, tuple(lin.weight.shape))
        self._W, self.pre = W, pre
        self._flat = _flatten_param_keys(W.params)
        self._runner = TailRunner(W, self.dims, pre, model.dropout.p if with_dropout else 0.0, head.config.dropout, eps)

        self._with_dropout = with_dropout
        self._sources = lins + ([fusion.output_proj, fusion.layer_norm] if with_projection else [])

    def covers(self):
        return TailRunner.covers(self.dims, self.pre)

    def current(self):
        """Still describes the model: nobody swapped a module or re-assigned a parameter since this tail was built."""
        model = self.model
        lins = [m for m in model.answer_head.classifier if isinstance(m, nn.Linear)]
        now = lins + ([model.fusion.output_proj, model.fusion.layer_norm] if self.pre is not None else [])
        if len(now) != len(self._sources) or any(a is not b for a, b in zip(now, self._sources)):
            return False
        live = {id(p) for m in now if m is not None for p in m.parameters(recurse=False)}
        return all(id(p) in live for _, p in self._flat)

    def _hip_forward(self, x):
        self._W.shadows.refresh(x.device)
        self._runner.p_in = self.model.dropout.p if self._with_dropout else 0.0
        self._runner.p_h = self.model.answer_head.config.dropout
        return self._runner.forward(x, self.model.training)

    def _hip_backward(self, saved, dout, needs):
        with K_skip(not any(p.requires_grad for _, p in self._flat)):
            G, dx = self._runner.backward(saved, dout, need_dx=needs[0])
        return [dx], _split_grads(self._flat, G)

    def __call__(self, x):
        return _BlockFn.apply(self, 1, x, *[p for _, p in self._flat])


class AnswerHead(nn.Module):
    """Reference vqa_model.py:436-477: (Linear, ReLU, Dropout)* + Linear; ``classifier`` keeps the Sequential indices."""

    def __init__(self, config: AnswerHeadConfig, input_dim: int):
        super().__init__()
        self.config = config
        layers, prev = [], input_dim
        for h in config.hidden_dims:
            layers.extend([nn.Linear(prev, h), nn.ReLU(), nn.Dropout(config.dropout)])
            prev = h
        layers.append(nn.Linear(prev, config.num_answers))
        self.classifier = nn.Sequential(*layers)

    def forward(self, features):
        lins = [m for m in self.classifier if isinstance(m, nn.Linear)]
        p = self.config.dropout if self.training else 0.0
        x = features
        for i, lin in enumerate(lins[:-1]):
            x = ops.linear(x, lin.weight, lin.bias, act=ACT_RELU, drop=Drop(p, ops.new_seed() if p > 0 else 0, 20 + i))
        return ops.linear(x, lins[-1].weight, lins[-1].bias)


class VietnameseVQAModel(nn.Module):
    """Reference vqa_model.py:480-727.  Attribute names visual_encoder / text_encoder / fusion / moe_layer /
    answer_head / dropout are the contract the training strategies and the ablation harness rely on."""

    def __init__(self, config: VQAModelConfig):
        super().__init__()
        self.config = config
        self.visual_encoder = VisualEncoder(config.visual_encoder)
        self.text_encoder = TextEncoder(config.text_encoder)
        self.fusion = MultimodalFusion(config.fusion)
        if config.moe.use_moe:
            self._init_moe(config.moe)
        else:
            self.moe_layer = None
        if config.knowledge.use_knowledge:
            self._init_knowledge(config.knowledge)
        else:
            self.knowledge_module = None
        self.answer_head = AnswerHead(config.answer_head, input_dim=config.fusion.output_dim)
        self.dropout = nn.Dropout(config.dropout)

    def _init_moe(self, config: MOEConfig):
        """Expert split of reference :531-546 (router_type / expert_type / load_balance_weight ignored there too)."""
        try:
            from ..moe import VQAMOELayer
            per, rem = max(1, config.num_experts // 4), config.num_experts % 4
            self.moe_layer = VQAMOELayer(
                input_dim=self.config.fusion.output_dim, hidden_dim=config.hidden_dim, output_dim=self.config.fusion.output_dim,
                num_vision_experts=per + (1 if rem > 0 else 0), num_text_experts=per + (1 if rem > 1 else 0),
                num_multimodal_experts=per + (1 if rem > 2 else 0), num_specialized_experts=per, top_k=config.top_k,
                dropout=self.config.dropout)
        except ImportError:
            warnings.warn("MOE module not available, disabling MOE")
            self.moe_layer = None

    def _init_knowledge(self, config: KnowledgeConfig):
        # RAG is out of the hot-path scope; same degradation as the reference when its
        # knowledge_base package is unavailable (vqa_model.py:572-576)
        warnings.warn("Knowledge base module not available")
        self.knowledge_module = None
        self.knowledge_encoder = None

    def set_knowledge_base(self, retriever, context_encoder=None):
        pass

    def encode_visual(self, pixel_values):
        return self.visual_encoder(pixel_values)

    def encode_text(self, input_ids, attention_mask):
        return self.text_encoder(input_ids, attention_mask)

    def forward(self, pixel_values, input_ids, attention_mask, questions: Optional[List[str]] = None,
                labels: Optional[torch.Tensor] = None, return_features: bool = False) -> VQAOutput:
        visual_pooled, visual_spatial, text_pooled, text_sequence = self.encode_both(pixel_values, input_ids, attention_mask)
        return self.forward_from_features(visual_pooled, visual_spatial, text_pooled, text_sequence, attention_mask, labels, return_features)

    def encode_both(self, pixel_values, input_ids, attention_mask):
        """(visual_pooled, visual_spatial, text_pooled, text_sequence): the two encoders, on parallel HIP streams when
        ``parallel_towers`` is set."""
        if pixel_values.is_cuda:
            K.set_training_numerics(self.training)       # train(): per-XCD k rotation in the ring GEMMs; eval(): batch-position-independent sums
        if getattr(self, 'parallel_towers', False) and pixel_values.is_cuda:
            # The two encoders share nothing until the fusion: run the vision tower on a side HIP stream (its backward
            # follows it there -- autograd replays every node on its forward stream).  A single short GEMM leaves most CUs
            # idle during its cold start and its C-tile stores; a second independent launch chain fills them.  Captured
            # into a HIP graph (graph.GraphedTrainStep) the two chains become parallel branches at no host cost.
            main = torch.cuda.current_stream()
            if getattr(self, '_tower_stream', None) is None:
                self._tower_stream = torch.cuda.Stream()
            side = self._tower_stream
            side.wait_stream(main)
            with torch.cuda.stream(side):
                visual_pooled, visual_spatial = self.encode_visual(pixel_values)
            text_pooled, text_sequence = self.encode_text(input_ids, attention_mask)
            main.wait_stream(side)
            for t in (visual_pooled, visual_spatial):
                t.record_stream(main)
        else:
            visual_pooled, visual_spatial = self.encode_visual(pixel_values)
            text_pooled, text_sequence = self.encode_text(input_ids, attention_mask)
        return visual_pooled, visual_spatial, text_pooled, text_sequence

    def forward_from_features(self, visual_pooled, visual_spatial, text_pooled, text_sequence, attention_mask,
                              labels: Optional[torch.Tensor] = None, return_features: bool = False) -> VQAOutput:
        """Everything behind the two encoders (reference vqa_model.py:662-727): fusion, MoE, dropout, answer head, loss, argmax.
        A method of its own so that the data-parallel captured step (graph.GraphedTrainStep) can cut the autograd graph at the
        encoder outputs: head / fusion backward, text backward and vision backward become separate HIP graphs, and each block's
        gradient all-reduce travels while the next block's backward computes."""
        text_mask = ~attention_mask.bool()
        whole = (TAIL_RUNNER and not return_features and self.moe_layer is None and self.config.fusion.fusion_type == 'cross_attention'
                 and text_sequence.is_cuda)
        if whole:
            tail = self._tail(True, True)
            if tail.covers():
                logits = tail(self.fusion(visual_spatial, text_sequence, text_mask=text_mask, project=False))
                return self._finish(logits, labels, None, None, None, None)
        fused = self.fusion(visual_spatial, text_sequence, text_mask=text_mask)
        moe_info = None
        if self.moe_layer is not None:
            if fused.dim() == 2:
                moe_output = self.moe_layer(fused.unsqueeze(1))
                if isinstance(moe_output, tuple):
                    fused, moe_info = moe_output
                    fused = fused.squeeze(1)
                else:
                    fused = moe_output.squeeze(1)
            else:
                moe_output = self.moe_layer(fused)
                fused, moe_info = moe_output if isinstance(moe_output, tuple) else (moe_output, None)
        tail = self._tail(False, not return_features) if (TAIL_RUNNER and fused.is_cuda and fused.dim() == 2) else None
        if tail is not None and tail.covers():
            if return_features:                 # the dropped features are an output: that dropout stays an op of its own
                fused = ops.dropout(fused, self.dropout.p, self.training)
            logits = tail(fused)
        else:
            fused = ops.dropout(fused, self.dropout.p, self.training)
            logits = self.answer_head(fused)
        if return_features:
            return self._finish(logits, labels, visual_pooled, text_pooled, fused, moe_info)
        return self._finish(logits, labels, None, None, None, moe_info)

    def _tail(self, with_projection, with_dropout):
        tails = self.__dict__.setdefault('_tails', {})
        key = (with_projection, with_dropout)
        if key not in tails or not tails[key].current():
            tails[key] = _Tail(self, with_projection, with_dropout)
        return tails[key]

    @staticmethod
    def _finish(logits, labels, visual, text, fused, moe_info):
        loss = None
        if labels is not None:
            loss, predictions = ops.cross_entropy_argmax(logits, labels)
        else:
            predictions = ops.argmax(logits)
        return VQAOutput(logits=logits, loss=loss, predictions=predictions, visual_features=visual, text_features=text,
                         fused_features=fused, knowledge_features=None, moe_info=moe_info)


def create_vqa_model(config: Optional[VQAModelConfig] = None, **kwargs) -> VietnameseVQAModel:
    """Reference vqa_model.py:730-756."""
    if config is None:
        from .vqa_config import get_default_vietnamese_vqa_config
        config = get_default_vietnamese_vqa_config()
    if kwargs:
        d = config.to_dict()
        for k, v in kwargs.items():
            if k in d:
                d[k] = v
        config = VQAModelConfig.from_dict(d)
    return VietnameseVQAModel(config)


__all__ = ['VQAOutput', 'VisualEncoder', 'TextEncoder', 'CrossModalAttention', 'MultimodalFusion', 'AnswerHead',
           'VietnameseVQAModel', 'create_vqa_model']
