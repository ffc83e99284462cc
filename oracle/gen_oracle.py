"""CPU fp32 restatement of the reference's GENERATIVE model (test infrastructure: imported by tests/ only).

Follows ``src/modeling/meta_arch/generative_vqa_model.py``: VisualEncoder :141-151 (all CLIP tokens), QuestionEncoder :175-190,
CrossModalFusion :286-339 (concatenated sequence through pre-LN ``nn.TransformerEncoderLayer`` x N with the question's padding
mask, LayerNorm; MoE off), TransformerDecoder :383-451 (tied embedding + sinusoidal positions, pre-LN ``nn.TransformerDecoderLayer``
x N with causal + padding masks, LayerNorm, tied output projection), GenerativeVQAModel.forward :523-598 (label-smoothed cross
entropy, ignore_index -100).  ``torch.nn`` layer algorithms (norm_first encoder / decoder layers, eval mode) restated from their
published definitions; pinned by tests/golden/generative_*.npz, produced by the reference itself (oracle/gen_golden.py --only
generative).
"""

import math

import torch
import torch.nn.functional as F

from . import vqa_oracle as vo


def _sdpa(q, k, v, num_heads, key_padding_mask=None, causal=False):
    qh, kh, vh = vo._heads(q, num_heads), vo._heads(k, num_heads), vo._heads(v, num_heads)
    scores = torch.matmul(qh, kh.transpose(-1, -2)) * (qh.shape[-1] ** -0.5)
    if causal:
        Sq, Sk = scores.shape[-2:]
        scores = scores + torch.triu(torch.full((Sq, Sk), float('-inf')), diagonal=1)
    if key_padding_mask is not None:
        scores = scores.masked_fill(key_padding_mask[:, None, None, :].bool(), float('-inf'))
    out = torch.matmul(torch.softmax(scores, dim=-1), vh)
    b, h, s, dh = out.shape
    return out.transpose(1, 2).reshape(b, s, h * dh)


def _mha(sd, name, query, key, value, num_heads, key_padding_mask=None, causal=False):
    w, b = sd[name + '.in_proj_weight'], sd[name + '.in_proj_bias']
    d = w.shape[1]
    q = F.linear(query, w[:d], b[:d])
    k = F.linear(key, w[d:2 * d], b[d:2 * d])
    v = F.linear(value, w[2 * d:], b[2 * d:])
    return vo.linear(sd, name + '.out_proj', _sdpa(q, k, v, num_heads, key_padding_mask, causal))


def encoder_layer(sd, p, x, num_heads, kpm):
    """nn.TransformerEncoderLayer(norm_first=True, activation='gelu'), eval."""
    h = vo.layer_norm(sd, p + 'norm1', x)
    x = x + _mha(sd, p + 'self_attn', h, h, h, num_heads, kpm)
    h = vo.layer_norm(sd, p + 'norm2', x)
    return x + vo.linear(sd, p + 'linear2', F.gelu(vo.linear(sd, p + 'linear1', h)))


def decoder_layer(sd, p, x, memory, num_heads, tgt_kpm, mem_kpm):
    """nn.TransformerDecoderLayer(norm_first=True, activation='gelu'), eval; causal self-attention."""
    h = vo.layer_norm(sd, p + 'norm1', x)
    x = x + _mha(sd, p + 'self_attn', h, h, h, num_heads, tgt_kpm, causal=True)
    h = vo.layer_norm(sd, p + 'norm2', x)
    x = x + _mha(sd, p + 'multihead_attn', h, memory, memory, num_heads, mem_kpm)
    h = vo.layer_norm(sd, p + 'norm3', x)
    return x + vo.linear(sd, p + 'linear2', F.gelu(vo.linear(sd, p + 'linear1', h)))


def _count(sd, fmt):
    n = 0
    while fmt.format(n) in sd:
        n += 1
    return n


def generative_forward(sd, pixel_values, input_ids, attention_mask, decoder_input_ids, decoder_attention_mask=None, labels=None, *,
                       vit_heads, text_heads, fusion_heads, decoder_heads, label_smoothing=0.1):
    """Returns (logits [B, A, V], loss | None, encoder_hidden_states)."""
    vis = vo.clip_vision_forward(sd, 'visual_encoder.vision_model.', pixel_values, vit_heads)
    txt = vo.roberta_forward(sd, 'question_encoder.encoder.', input_ids, attention_mask, text_heads)
    fused = torch.cat([vis, txt], dim=1)
    B, nv = vis.shape[0], vis.shape[1]
    kpm = torch.cat([torch.zeros(B, nv, dtype=torch.bool), ~attention_mask.bool()], dim=1)
    for i in range(_count(sd, 'fusion.layers.{}.norm1.weight')):
        fused = encoder_layer(sd, f'fusion.layers.{i}.', fused, fusion_heads, kpm)
    memory = vo.layer_norm(sd, 'fusion.layer_norm', fused)
    emb = sd['answer_embedding.weight']
    x = F.embedding(decoder_input_ids, emb) + sd['decoder.pos_encoding.pe'][:, :decoder_input_ids.shape[1]]
    tgt_kpm = (decoder_attention_mask == 0) if decoder_attention_mask is not None else None
    for i in range(_count(sd, 'decoder.decoder.layers.{}.norm1.weight')):
        x = decoder_layer(sd, f'decoder.decoder.layers.{i}.', x, memory, decoder_heads, tgt_kpm, kpm)
    x = vo.layer_norm(sd, 'decoder.layer_norm', x)
    logits = F.linear(x, emb)
    loss = None
    if labels is not None:
        loss = F.cross_entropy(logits.reshape(-1, logits.shape[-1]), labels.reshape(-1), ignore_index=-100, label_smoothing=label_smoothing)
    return logits, loss, memory


def sinusoidal_pe(max_len, d_model):
    position = torch.arange(max_len).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
    pe = torch.zeros(1, max_len, d_model)
    pe[0, :, 0::2] = torch.sin(position * div_term)
    pe[0, :, 1::2] = torch.cos(position * div_term)
    return pe


def tie(sd):
    """One tensor under three names (answer_embedding / decoder.embedding / decoder.output_projection): ``load_state_dict`` copies them
    in key order into the same parameter, so the LAST one is what the model holds."""
    sd = dict(sd)
    w = sd['decoder.output_projection.weight']
    sd['answer_embedding.weight'] = sd['decoder.embedding.weight'] = w
    return sd


def fixture_inputs(meta):
    """The seeded inputs of a generative fixture (same construction as oracle/gen_golden.py: run_generative_case)."""
    from . import det_weights as dw
    d, seed = meta['dims'], meta['seed']
    px, ids, mask, _ = dw.make_inputs(d['batch'], d['seq'], d['image'], vocab_hi=min(30000, d['vocab']), num_answers=8, seed=seed)
    B, A, V = d['batch'], d['answer_len'], d['gen_vocab']
    toks = dw.randint('answer_tokens', (B, A + 1), 3, V, seed)
    toks[:, 0] = 0
    dec_in, labels = toks[:, :-1].clone(), toks[:, 1:].clone()
    dmask = torch.ones(B, A, dtype=torch.int64)
    if B > 1:
        cut = max(2, (A * 5) // 8)
        dmask[1, cut:] = 0
        dec_in[1, cut:] = 1
        labels[1, cut - 1:] = -100
    return px, ids, mask, dec_in, dmask, labels
