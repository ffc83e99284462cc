"""Generative Vietnamese VQA model on the MI355X HIP path (reference ``src/modeling/meta_arch/generative_vqa_model.py``;
SURVEY section 8f rank 3 -- started in round 2: correct against reference-run fixtures, first tuning pass done).

Same class names, constructor signatures, attribute names and ``state_dict`` keys as the reference: ``GenerativeVQAConfig``
(:28-105), ``GenerativeVQAOutput`` (:108-116), ``VisualEncoder`` (:119-151), ``QuestionEncoder`` (:154-190), ``CrossModalFusion``
(:193-339: visual and question tokens concatenated into ONE sequence, pre-LN ``nn.TransformerEncoderLayer`` x 2 with the question's
padding mask, optional MoE, LayerNorm), ``TransformerDecoder`` (:342-451: tied token embedding + sinusoidal positions, pre-LN
``nn.TransformerDecoderLayer`` x 6 with causal + padding masks over the fused memory, LayerNorm, tied 64 000-way output projection),
``GenerativeVQAModel`` (:479-703: teacher-forced forward with label-smoothed cross entropy, ``generate``).

What runs where: the two encoders are the block runners of the classification path (hip/blocks.py); every fusion-encoder and decoder
layer is one hand-scheduled autograd node (hip/gen_blocks.py; the op-by-op chains of hip/ops.py stay as ``_forward_ops``, the form the
runners are tested against); embedding, final LayerNorms and the label-smoothed CE are per-op nodes; the 64 000 x 768 output projection
is one MFMA GEMM over all B x A rows (201 GFLOP at B = 32, A = 64: the first GEMM of this code base large enough to run near the matrix cores' rate).  ``moe_type`` 'standard' / 'vqa' / 'sparse' all build (the reference's own 'sparse' call site raises a TypeError: SURVEY F11).  Not
built: ``moe_position`` 'decoder' / 'both' (the reference never constructs a decoder MoE either: ``CrossModalFusion`` is its only MoE
site), beam search (the reference's ``generate`` ignores ``num_beams`` too).
"""

import math
from dataclasses import dataclass
from typing import Optional, Tuple

import torch
import torch.nn as nn

from ...hip import kernels as K
from ...hip import ops
from ...hip.gen_blocks import PreLNDecoderLayerRunner, PreLNEncoderLayerRunner
from ...hip.kernels import ACT_GELU, Drop
from ...hip.kernels import skip_weight_grads as K_skip
from .backbones import ClipVisionBackbone, RobertaBackbone, _BlockFn, _Weights, _flatten_param_keys, _require_cuda, _split_grads
from .vqa_model import _KNOWN_TEXT, _KNOWN_VISION, _MHAParams, _try_load_pretrained


@dataclass
class GenerativeVQAConfig:
    """Reference generative_vqa_model.py:28-105 (same fields, defaults and alias syncing).  ``visual_arch`` / ``text_arch``
    (not in the reference): architecture dicts for offline construction of non-hub encoders (tests)."""
    visual_backbone: str = 'openai/clip-vit-base-patch32'
    visual_output_dim: int = 768
    freeze_visual_encoder: bool = False
    freeze_visual: bool = False
    text_encoder: str = 'vinai/phobert-base'
    text_output_dim: int = 768
    freeze_question_encoder: bool = False
    freeze_text_encoder: bool = False
    max_question_length: int = 64
    decoder_type: str = 'transformer'
    hidden_size: int = 768
    decoder_hidden_dim: int = 768
    num_decoder_layers: int = 6
    decoder_num_layers: int = 6
    num_attention_heads: int = 8
    decoder_num_heads: int = 8
    decoder_ff_dim: int = 2048
    decoder_dropout: float = 0.1
    max_answer_length: int = 64
    fusion_dim: int = 768
    fusion_num_heads: int = 8
    fusion_num_layers: int = 2
    fusion_dropout: float = 0.1
    use_moe: bool = False
    moe_type: str = 'standard'
    num_experts: int = 4
    num_experts_per_token: int = 2
    expert_capacity_factor: float = 1.25
    moe_loss_weight: float = 0.01
    moe_position: str = 'fusion'
    num_vision_experts: int = 1
    num_text_experts: int = 1
    num_multimodal_experts: int = 1
    num_specialized_experts: int = 1
    vietnamese_optimized: bool = True
    vocab_size: int = 64000
    pad_token_id: int = 1
    bos_token_id: int = 0
    eos_token_id: int = 2
    label_smoothing: float = 0.1
    tie_word_embeddings: bool = True
    visual_arch: Optional[dict] = None
    text_arch: Optional[dict] = None

    def __post_init__(self):
        if self.freeze_visual_encoder:
            self.freeze_visual = True
        if self.freeze_visual:
            self.freeze_visual_encoder = True
        if self.freeze_question_encoder:
            self.freeze_text_encoder = True
        if self.freeze_text_encoder:
            self.freeze_question_encoder = True
        self.decoder_hidden_dim = self.hidden_size
        self.decoder_num_layers = self.num_decoder_layers
        self.decoder_num_heads = self.num_attention_heads


@dataclass
class GenerativeVQAOutput:
    """Reference generative_vqa_model.py:108-116."""
    logits: torch.Tensor
    loss: Optional[torch.Tensor] = None
    generated_ids: Optional[torch.Tensor] = None
    encoder_hidden_states: Optional[torch.Tensor] = None
    decoder_hidden_states: Optional[torch.Tensor] = None
    cross_attention_weights: Optional[torch.Tensor] = None


def _drop(p, training, seed, stream):
    return Drop(p, seed, stream) if (training and p > 0) else Drop()


class VisualEncoder(nn.Module):
    """Reference :119-151: CLIP vision tower, all tokens, optional projection to ``fusion_dim``."""

    def __init__(self, config: GenerativeVQAConfig):
        super().__init__()
        self.config = config
        arch = config.visual_arch or _KNOWN_VISION.get(config.visual_backbone)
        if arch is None:
            raise NotImplementedError(f'HIP visual backbone: CLIP ViT towers only (got {config.visual_backbone!r})')
        self.vision_model = ClipVisionBackbone(**arch)
        if config.visual_arch is None:
            def loader(n):
                from transformers import CLIPVisionModel
                return CLIPVisionModel.from_pretrained(n)
            _try_load_pretrained(self.vision_model, loader, config.visual_backbone)
        if config.freeze_visual:
            for p in self.vision_model.parameters():
                p.requires_grad = False
        hs = arch['hidden_size']
        self.projection = nn.Linear(hs, config.fusion_dim) if hs != config.fusion_dim else nn.Identity()

    def forward(self, pixel_values):
        h = self.vision_model(pixel_values=pixel_values).last_hidden_state
        return ops.linear(h, self.projection.weight, self.projection.bias) if isinstance(self.projection, nn.Linear) else h


class QuestionEncoder(nn.Module):
    """Reference :154-190: PhoBERT, all tokens, optional projection."""

    def __init__(self, config: GenerativeVQAConfig):
        super().__init__()
        self.config = config
        arch = config.text_arch or _KNOWN_TEXT.get(config.text_encoder)
        if arch is None:
            raise NotImplementedError(f'HIP text encoder: RoBERTa / PhoBERT only (got {config.text_encoder!r})')
        self.encoder = RobertaBackbone(**arch)
        if config.text_arch is None:
            def loader(n):
                from transformers import AutoModel
                return AutoModel.from_pretrained(n)
            _try_load_pretrained(self.encoder, loader, config.text_encoder)
        if config.freeze_text_encoder:
            for p in self.encoder.parameters():
                p.requires_grad = False
        hs = arch['hidden_size']
        self.projection = nn.Linear(hs, config.fusion_dim) if hs != config.fusion_dim else nn.Identity()

    def forward(self, input_ids, attention_mask):
        h = self.encoder(input_ids=input_ids, attention_mask=attention_mask).last_hidden_state
        if isinstance(self.projection, nn.Linear):
            h = ops.linear(h, self.projection.weight, self.projection.bias)
        return h, attention_mask


# True: every fusion-encoder / decoder layer is ONE autograd node (hip/gen_blocks.py).  False: the op-by-op chains (_forward_ops) the
# runners are tested against.
LAYER_RUNNERS = True
# True: output projection + label-smoothed cross entropy as one node when labels are given (logits stay an output, without a gradient)
FUSED_HEAD_LOSS = True


def _bind_layer(layer, cross):
    """Key -> parameter map + bf16 weight shadows of one Transformer layer (torch's parameter names stay the state_dict's)."""
    W = _Weights()
    P, S = W.params, W.shadows
    pairs = [('sa_in', layer.self_attn.in_proj_weight, layer.self_attn.in_proj_bias),
             ('sa_out', layer.self_attn.out_proj.weight, layer.self_attn.out_proj.bias)]
    if cross:
        pairs += [('ca_in', layer.multihead_attn.in_proj_weight, layer.multihead_attn.in_proj_bias),
                  ('ca_out', layer.multihead_attn.out_proj.weight, layer.multihead_attn.out_proj.bias)]
    pairs += [('l1', layer.linear1.weight, layer.linear1.bias), ('l2', layer.linear2.weight, layer.linear2.bias)]
    for key, w, b in pairs:
        P[key + '_w'], P[key + '_b'] = w, b
        S.add(key + '_w', w, tuple(w.shape))
    norms = [layer.norm1, layer.norm2] + ([layer.norm3] if cross else [])
    for i, n in enumerate(norms, 1):
        P[f'n{i}.w'], P[f'n{i}.b'] = n.weight, n.bias
    return W


class _EncoderLayer(nn.Module):
    """Parameter layout + HIP forward of ``nn.TransformerEncoderLayer(activation='gelu', batch_first=True, norm_first=True)``."""

    def __init__(self, d_model, nhead, dim_feedforward, dropout):
        super().__init__()
        self.self_attn = _MHAParams(d_model, nhead, dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1, self.norm2 = nn.LayerNorm(d_model), nn.LayerNorm(d_model)
        self.dropout1, self.dropout2 = nn.Dropout(dropout), nn.Dropout(dropout)
        self._p = dropout
        self._W = _bind_layer(self, cross=False)
        self._flat = _flatten_param_keys(self._W.params)
        self._runner = PreLNEncoderLayerRunner(self._W, d_model, nhead, dim_feedforward, dropout, self.norm1.eps)
        self._mask = None

    def _hip_forward(self, src):
        self._W.shadows.refresh(src.device)
        return self._runner.forward(src, self._mask, self.training)

    def _hip_backward(self, saved, dout, needs):
        with K_skip(not any(p.requires_grad for _, p in self._flat)):
            G, dx = self._runner.backward(saved, dout)
        return [dx], _split_grads(self._flat, G)

    def forward(self, src, src_key_padding_mask=None):
        if LAYER_RUNNERS and src.is_cuda and src.dim() == 3:
            self._mask = src_key_padding_mask
            try:
                return _BlockFn.apply(self, 1, src, *[p for _, p in self._flat])
            finally:
                self._mask = None
        return self._forward_ops(src, src_key_padding_mask)

    def _forward_ops(self, src, src_key_padding_mask=None):
        p, tr = self._p, self.training
        h = ops.layer_norm(src, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        a, _ = self.self_attn(h, h, h, key_padding_mask=src_key_padding_mask)
        x = ops.add(src, ops.dropout(a, p, tr))
        h = ops.layer_norm(x, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        seed = ops.new_seed() if (tr and p > 0) else 0
        f = ops.linear(h, self.linear1.weight, self.linear1.bias, act=ACT_GELU, drop=_drop(p, tr, seed, 171))
        f = ops.linear(f, self.linear2.weight, self.linear2.bias, drop=_drop(p, tr, seed, 172))
        return ops.add(x, f)


class CrossModalFusion(nn.Module):
    """Reference :193-339."""

    def __init__(self, config: GenerativeVQAConfig):
        super().__init__()
        self.config = config
        self.use_moe = config.use_moe and config.moe_position in ['fusion', 'both']
        self.moe_type = config.moe_type if hasattr(config, 'moe_type') else 'standard'
        self.layers = nn.ModuleList([_EncoderLayer(config.fusion_dim, config.fusion_num_heads, config.decoder_ff_dim, config.fusion_dropout)
                                     for _ in range(config.fusion_num_layers)])
        self.moe_layer = None
        self.moe_aux_loss = 0.0
        if self.use_moe:
            self._create_moe_layer(config)
        self.layer_norm = nn.LayerNorm(config.fusion_dim)

    def _create_moe_layer(self, config):
        from ..moe import MOELayer, VQAMOELayer
        from ..moe.moe_config import MOEConfig, RouterConfig
        if self.moe_type == 'vqa':
            self.moe_layer = VQAMOELayer(input_dim=config.fusion_dim, hidden_dim=config.decoder_ff_dim, output_dim=config.fusion_dim,
                                         num_vision_experts=config.num_vision_experts, num_text_experts=config.num_text_experts,
                                         num_multimodal_experts=config.num_multimodal_experts,
                                         num_specialized_experts=config.num_specialized_experts, top_k=config.num_experts_per_token,
                                         dropout=config.fusion_dropout, vietnamese_optimized=config.vietnamese_optimized)
        elif self.moe_type == 'sparse':
            # the reference's call here is ``SparseMOELayer(config=moe_config)`` (:262), which its own constructor rejects with a TypeError
            # (SURVEY F11); built instead from the fields that MOEConfig was filled with
            from ..moe import SparseMOELayer
            self.moe_layer = SparseMOELayer(input_dim=config.fusion_dim, hidden_dim=config.decoder_ff_dim, output_dim=config.fusion_dim,
                                            num_experts=config.num_experts, top_k=config.num_experts_per_token,
                                            capacity_factor=config.expert_capacity_factor, dropout=config.fusion_dropout, use_aux_loss=True)
        else:
            rc = RouterConfig(router_type='topk', num_experts=config.num_experts, top_k=config.num_experts_per_token,
                              capacity_factor=config.expert_capacity_factor, load_balance_weight=config.moe_loss_weight, use_aux_loss=True)
            mc = MOEConfig(input_dim=config.fusion_dim, hidden_dim=config.decoder_ff_dim, output_dim=config.fusion_dim,
                           num_experts=config.num_experts, num_experts_per_token=config.num_experts_per_token, router_config=rc,
                           expert_dropout=config.fusion_dropout)
            self.moe_layer = MOELayer(config=mc)

    def forward(self, visual_features, question_features, question_mask=None) -> Tuple[torch.Tensor, float]:
        fused = torch.cat([visual_features, question_features], dim=1)            # [B, P+1+S, D]: plumbing
        kpm = None
        if question_mask is not None:
            B, nv = visual_features.size(0), visual_features.size(1)
            kpm = torch.cat([torch.zeros(B, nv, dtype=torch.bool, device=fused.device), ~question_mask.bool()], dim=1)
        for layer in self.layers:
            fused = layer(fused, src_key_padding_mask=kpm)
        moe_aux_loss = 0.0
        if self.moe_layer is not None:
            fused = self.moe_layer(fused)
            aux = self.moe_layer.get_aux_loss()
            if isinstance(aux, torch.Tensor):
                moe_aux_loss = aux.item() if aux.numel() == 1 else aux.mean().item()      # a python float, as in the reference (:331-335)
            else:
                moe_aux_loss = float(aux) if aux else 0.0
        return ops.layer_norm(fused, self.layer_norm.weight, self.layer_norm.bias, self.layer_norm.eps), moe_aux_loss


class PositionalEncoding(nn.Module):
    """Reference :454-476: sinusoidal table (buffer ``pe``) added to the embeddings, then dropout."""

    def __init__(self, d_model: int, dropout: float = 0.1, max_len: int = 512):
        super().__init__()
        self.dropout = nn.Dropout(p=dropout)
        position = torch.arange(max_len).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
        pe = torch.zeros(1, max_len, d_model)
        pe[0, :, 0::2] = torch.sin(position * div_term)
        pe[0, :, 1::2] = torch.cos(position * div_term)
        self.register_buffer('pe', pe)

    def forward(self, x):
        x = ops.add(x, self.pe[:, :x.size(1)].expand_as(x).contiguous())
        return ops.dropout(x, self.dropout.p, self.training)


class _DecoderLayer(nn.Module):
    """Parameter layout + HIP forward of ``nn.TransformerDecoderLayer(activation='gelu', batch_first=True, norm_first=True)``."""

    def __init__(self, d_model, nhead, dim_feedforward, dropout):
        super().__init__()
        self.self_attn = _MHAParams(d_model, nhead, dropout)
        self.multihead_attn = _MHAParams(d_model, nhead, dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1, self.norm2, self.norm3 = nn.LayerNorm(d_model), nn.LayerNorm(d_model), nn.LayerNorm(d_model)
        self.dropout1, self.dropout2, self.dropout3 = nn.Dropout(dropout), nn.Dropout(dropout), nn.Dropout(dropout)
        self._p = dropout
        self._W = _bind_layer(self, cross=True)
        self._flat = _flatten_param_keys(self._W.params)
        self._runner = PreLNDecoderLayerRunner(self._W, d_model, nhead, dim_feedforward, dropout, self.norm1.eps)
        self._masks = (None, None)

    def _hip_forward(self, tgt, memory):
        self._W.shadows.refresh(tgt.device)
        return self._runner.forward(tgt, memory, self._masks[0], self._masks[1], self.training)

    def _hip_backward(self, saved, dout, needs):
        with K_skip(not any(p.requires_grad for _, p in self._flat)):
            G, dx, dmem = self._runner.backward(saved, dout, need_dmem=needs[1])
        return [dx, dmem], _split_grads(self._flat, G)

    def forward(self, tgt, memory, tgt_key_padding_mask=None, memory_key_padding_mask=None):
        if LAYER_RUNNERS and tgt.is_cuda and tgt.dim() == 3:
            self._masks = (tgt_key_padding_mask, memory_key_padding_mask)
            try:
                return _BlockFn.apply(self, 2, tgt, memory, *[p for _, p in self._flat])
            finally:
                self._masks = (None, None)
        return self._forward_ops(tgt, memory, tgt_key_padding_mask, memory_key_padding_mask)

    def _forward_ops(self, tgt, memory, tgt_key_padding_mask=None, memory_key_padding_mask=None):
        p, tr = self._p, self.training
        sa = self.self_attn
        h = ops.layer_norm(tgt, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        a = ops.multi_head_attention(h, h, h, sa.in_proj_weight, sa.in_proj_bias, sa.out_proj.weight, sa.out_proj.bias, sa.num_heads,
                                     tgt_key_padding_mask, sa.dropout, tr, causal=True)
        x = ops.add(tgt, ops.dropout(a, p, tr))
        h = ops.layer_norm(x, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        a, _ = self.multihead_attn(h, memory, memory, key_padding_mask=memory_key_padding_mask)
        x = ops.add(x, ops.dropout(a, p, tr))
        h = ops.layer_norm(x, self.norm3.weight, self.norm3.bias, self.norm3.eps)
        seed = ops.new_seed() if (tr and p > 0) else 0
        f = ops.linear(h, self.linear1.weight, self.linear1.bias, act=ACT_GELU, drop=_drop(p, tr, seed, 181))
        f = ops.linear(f, self.linear2.weight, self.linear2.bias, drop=_drop(p, tr, seed, 182))
        return ops.add(x, f)


class _DecoderStack(nn.Module):
    """``nn.TransformerDecoder``'s parameter layout (``layers.{i}.*``)."""

    def __init__(self, d_model, nhead, dim_feedforward, dropout, num_layers):
        super().__init__()
        self.layers = nn.ModuleList(_DecoderLayer(d_model, nhead, dim_feedforward, dropout) for _ in range(num_layers))


class _EmbeddingFn(torch.autograd.Function):
    """nn.Embedding without padding_idx: row gather forward, dense scatter-add backward (fp32)."""

    @staticmethod
    def forward(ctx, weight, ids32):
        n, D = ids32.numel(), weight.shape[1]
        out, _ = K.gather_rows(weight.detach(), ids32, n, D, want_f32=True, want_bf16=False)
        ctx.save_for_backward(ids32)
        ctx.shape = tuple(weight.shape)
        return out

    @staticmethod
    def backward(ctx, dy):
        (ids32,) = ctx.saved_tensors
        V, D = ctx.shape
        dw = torch.zeros((V, D), dtype=torch.float32, device=dy.device)
        dy = dy.contiguous().float()
        K._chk(K.L().vqa_embedding_rows_bwd(dy.data_ptr(), ids32.data_ptr(), dw.data_ptr(), ids32.numel(), D, V, K._stream()), 'vqa_embedding_rows_bwd')
        return dw, None


class TransformerDecoder(nn.Module):
    """Reference :342-451."""

    def __init__(self, config: GenerativeVQAConfig, embedding: Optional[nn.Embedding] = None):
        super().__init__()
        self.config = config
        self.embedding = embedding if embedding is not None else nn.Embedding(config.vocab_size, config.decoder_hidden_dim)
        self.pos_encoding = PositionalEncoding(config.decoder_hidden_dim, config.decoder_dropout, max_len=config.max_answer_length)
        self.decoder = _DecoderStack(config.decoder_hidden_dim, config.decoder_num_heads, config.decoder_ff_dim, config.decoder_dropout,
                                     config.decoder_num_layers)
        self.layer_norm = nn.LayerNorm(config.decoder_hidden_dim)
        self.output_projection = nn.Linear(config.decoder_hidden_dim, config.vocab_size, bias=False)
        if config.tie_word_embeddings:
            self.output_projection.weight = self.embedding.weight

    def forward(self, encoder_hidden_states, decoder_input_ids, encoder_attention_mask=None, decoder_attention_mask=None, return_hidden=False):
        """``return_hidden`` (not in the reference signature): the normalised decoder states [B, A, D] in front of the output projection
        (GenerativeVQAModel fuses that projection with the loss: ops.linear_cross_entropy)."""
        _require_cuda(encoder_hidden_states, 'TransformerDecoder')
        B, A = decoder_input_ids.shape
        V, D = self.embedding.weight.shape
        ids = decoder_input_ids.reshape(-1)
        # nn.Embedding device-asserts on an id outside [0, V); here such an id is never dereferenced (clamped) and clears the device
        # status word that ``hip.kernels.check_device_status`` reads -- no host sync on the path, so the step can be captured
        ok = ((ids >= 0) & (ids < V)).all()
        K.status_word(ids.device).mul_(ok.to(torch.int32))
        x = _EmbeddingFn.apply(self.embedding.weight, ids.clamp(0, V - 1).to(torch.int32).contiguous()).view(B, A, D)
        x = self.pos_encoding(x)
        mem_kpm = (encoder_attention_mask == 0) if encoder_attention_mask is not None else None
        tgt_kpm = (decoder_attention_mask == 0) if decoder_attention_mask is not None else None
        for layer in self.decoder.layers:
            x = layer(x, encoder_hidden_states, tgt_key_padding_mask=tgt_kpm, memory_key_padding_mask=mem_kpm)
        x = ops.layer_norm(x, self.layer_norm.weight, self.layer_norm.bias, self.layer_norm.eps)
        if return_hidden:
            return x
        return ops.linear(x, self.output_projection.weight, None)                 # [B, A, V]: one MFMA GEMM over all rows

    def _generate_causal_mask(self, seq_len, device):
        mask = torch.triu(torch.ones(seq_len, seq_len, device=device), diagonal=1)
        return mask.masked_fill(mask == 1, float('-inf'))


class GenerativeVQAModel(nn.Module):
    """Reference :479-703."""

    def __init__(self, config: GenerativeVQAConfig):
        super().__init__()
        self.config = config
        # moe_position 'decoder' / 'both': the reference accepts the value and builds no decoder MoE either (CrossModalFusion is its only
        # MoE site: 'both' == 'fusion', 'decoder' == none)
        self.visual_encoder = VisualEncoder(config)
        self.question_encoder = QuestionEncoder(config)
        self.fusion = CrossModalFusion(config)
        self.answer_embedding = nn.Embedding(config.vocab_size, config.decoder_hidden_dim)
        self.decoder = TransformerDecoder(config, embedding=self.answer_embedding)
        self.loss_fn = nn.CrossEntropyLoss(ignore_index=-100, label_smoothing=config.label_smoothing)      # kept for attribute parity
        self._init_weights()

    def _init_weights(self):
        for module in [self.fusion, self.decoder]:
            for p in module.parameters():
                if p.dim() > 1:
                    nn.init.xavier_uniform_(p)

    def encode_both(self, pixel_values, input_ids, attention_mask):
        """The two encoders; with ``parallel_towers`` set (graph.GraphedTrainStep does) the vision tower runs on a side HIP stream -- its
        backward follows it there -- so the two towers are parallel branches of the captured step (vqa_model.VietnameseVQAModel.encode_both)."""
        if pixel_values.is_cuda:
            K.set_training_numerics(self.training)
        if getattr(self, 'parallel_towers', False) and pixel_values.is_cuda:
            main = torch.cuda.current_stream()
            if getattr(self, '_tower_stream', None) is None:
                self._tower_stream = torch.cuda.Stream()
            side = self._tower_stream
            side.wait_stream(main)
            with torch.cuda.stream(side):
                visual_features = self.visual_encoder(pixel_values)
            question_features, question_mask = self.question_encoder(input_ids, attention_mask)
            main.wait_stream(side)
            visual_features.record_stream(main)
        else:
            visual_features = self.visual_encoder(pixel_values)
            question_features, question_mask = self.question_encoder(input_ids, attention_mask)
        return visual_features, question_features, question_mask

    def forward(self, pixel_values, input_ids, attention_mask, decoder_input_ids=None, decoder_attention_mask=None, labels=None,
                return_dict: bool = True) -> GenerativeVQAOutput:
        _require_cuda(pixel_values, 'GenerativeVQAModel')
        visual_features, question_features, question_mask = self.encode_both(pixel_values, input_ids, attention_mask)
        encoder_hidden_states, moe_aux_loss = self.fusion(visual_features, question_features, question_mask)
        B, nv = pixel_values.size(0), visual_features.size(1)
        encoder_attention_mask = torch.cat([torch.ones(B, nv, device=pixel_values.device), attention_mask.float()], dim=1)
        if decoder_input_ids is None:
            decoder_input_ids = torch.full((B, 1), self.config.bos_token_id, dtype=torch.long, device=pixel_values.device)
        loss = None
        if labels is not None and FUSED_HEAD_LOSS:
            hidden = self.decoder(encoder_hidden_states=encoder_hidden_states, decoder_input_ids=decoder_input_ids,
                                  encoder_attention_mask=encoder_attention_mask, decoder_attention_mask=decoder_attention_mask, return_hidden=True)
            Bd, A, D = hidden.shape
            loss, logits, _ = ops.linear_cross_entropy(hidden.reshape(Bd * A, D), self.decoder.output_projection.weight, labels.reshape(-1),
                                                       label_smoothing=self.config.label_smoothing)
            logits = logits.view(Bd, A, -1)
        else:
            logits = self.decoder(encoder_hidden_states=encoder_hidden_states, decoder_input_ids=decoder_input_ids,
                                  encoder_attention_mask=encoder_attention_mask, decoder_attention_mask=decoder_attention_mask)
            if labels is not None:
                V = self.config.vocab_size
                loss, _ = ops.cross_entropy_argmax(logits.reshape(-1, V), labels.reshape(-1), label_smoothing=self.config.label_smoothing)
        if labels is not None:
            if self.config.use_moe and moe_aux_loss > 0:
                loss = loss + self.config.moe_loss_weight * moe_aux_loss
        return GenerativeVQAOutput(logits=logits, loss=loss, encoder_hidden_states=encoder_hidden_states)

    @torch.no_grad()
    def generate(self, pixel_values, input_ids, attention_mask, max_length: int = 64, min_length: int = 1, num_beams: int = 1,
                 temperature: float = 1.0, top_k: int = 50, top_p: float = 0.95, do_sample: bool = False, early_stopping: bool = True):
        """Reference :600-703 (token-by-token re-decode of the whole prefix, top-k / top-p filtering, greedy or sampled; ``num_beams``
        and ``min_length`` are accepted and unused there as well).  The filtering is host-side torch on [B, V] logits."""
        import torch.nn.functional as F
        B, device = pixel_values.size(0), pixel_values.device
        visual_features = self.visual_encoder(pixel_values)
        question_features, question_mask = self.question_encoder(input_ids, attention_mask)
        encoder_hidden_states, _ = self.fusion(visual_features, question_features, question_mask)
        nv = visual_features.size(1)
        encoder_attention_mask = torch.cat([torch.ones(B, nv, device=device), attention_mask.float()], dim=1)
        generated = torch.full((B, 1), self.config.bos_token_id, dtype=torch.long, device=device)
        finished = torch.zeros(B, dtype=torch.bool, device=device)
        for _ in range(max_length - 1):
            logits = self.decoder(encoder_hidden_states=encoder_hidden_states, decoder_input_ids=generated,
                                  encoder_attention_mask=encoder_attention_mask)
            nxt = logits[:, -1, :].float() / temperature
            if top_k > 0:
                nxt[nxt < torch.topk(nxt, top_k)[0][..., -1, None]] = float('-inf')
            if top_p < 1.0:
                sorted_logits, sorted_indices = torch.sort(nxt, descending=True)
                cum = torch.cumsum(F.softmax(sorted_logits, dim=-1), dim=-1)
                rem = cum > top_p
                rem[..., 1:] = rem[..., :-1].clone()
                rem[..., 0] = 0
                nxt[rem.scatter(1, sorted_indices, rem)] = float('-inf')
            if do_sample:
                tok = torch.multinomial(F.softmax(nxt, dim=-1), num_samples=1).squeeze(-1)
            else:
                tok = nxt.argmax(dim=-1)
            tok = tok.masked_fill(finished, self.config.pad_token_id)
            generated = torch.cat([generated, tok.unsqueeze(-1)], dim=1)
            finished = finished | (tok == self.config.eos_token_id)
            if early_stopping and finished.all():
                break
        return generated


def create_generative_vqa_model(config: Optional[GenerativeVQAConfig] = None, **kwargs) -> GenerativeVQAModel:
    """Reference :706-727."""
    if config is None:
        config = GenerativeVQAConfig()
    for key, value in kwargs.items():
        if hasattr(config, key):
            setattr(config, key, value)
    return GenerativeVQAModel(config)


def get_default_generative_vqa_config(visual_backbone: str = 'openai/clip-vit-base-patch32', text_encoder: str = 'vinai/phobert-base',
                                      vocab_size: int = 64000, bos_token_id: int = 0, eos_token_id: int = 2, pad_token_id: int = 1,
                                      **kwargs) -> GenerativeVQAConfig:
    """Reference :730-823 (same defaults: VQA-MoE counts default to 2 each here, 1 each in the dataclass)."""
    config = GenerativeVQAConfig(
        visual_backbone=visual_backbone, visual_output_dim=768, freeze_visual_encoder=kwargs.get('freeze_visual_encoder', False),
        text_encoder=text_encoder, text_output_dim=768, freeze_question_encoder=kwargs.get('freeze_question_encoder', False),
        max_question_length=64, decoder_type='transformer', hidden_size=768, num_decoder_layers=6, num_attention_heads=8,
        decoder_ff_dim=2048, decoder_dropout=0.1, max_answer_length=kwargs.get('max_answer_length', 64), fusion_dim=768,
        fusion_num_heads=8, fusion_num_layers=2, fusion_dropout=0.1, use_moe=kwargs.get('use_moe', False),
        moe_type=kwargs.get('moe_type', 'standard'), num_experts=kwargs.get('num_experts', 4),
        num_experts_per_token=kwargs.get('num_experts_per_token', 2), expert_capacity_factor=kwargs.get('expert_capacity_factor', 1.25),
        moe_loss_weight=kwargs.get('moe_loss_weight', 0.01), moe_position=kwargs.get('moe_position', 'fusion'),
        num_vision_experts=kwargs.get('num_vision_experts', 2), num_text_experts=kwargs.get('num_text_experts', 2),
        num_multimodal_experts=kwargs.get('num_multimodal_experts', 2), num_specialized_experts=kwargs.get('num_specialized_experts', 2),
        vietnamese_optimized=kwargs.get('vietnamese_optimized', True), vocab_size=vocab_size, pad_token_id=pad_token_id,
        bos_token_id=bos_token_id, eos_token_id=eos_token_id, label_smoothing=0.1, tie_word_embeddings=True)
    handled = ['freeze_visual_encoder', 'freeze_question_encoder', 'max_answer_length', 'use_moe', 'moe_type', 'num_experts',
               'num_experts_per_token', 'expert_capacity_factor', 'moe_loss_weight', 'moe_position', 'num_vision_experts',
               'num_text_experts', 'num_multimodal_experts', 'num_specialized_experts', 'vietnamese_optimized']
    for key, value in kwargs.items():
        if hasattr(config, key) and key not in handled:
            setattr(config, key, value)
    config.__post_init__()
    return config
