"""CPU fp32 restatement of the reference's GENERATIVE model (test infrastructure: imported by tests/ only).

Follows ``src/modeling/meta_arch/generative_vqa_model.py``: VisualEncoder :141-151 (all CLIP tokens), QuestionEncoder :175-190,
CrossModalFusion :286-339 (concatenated sequence through pre-LN ``nn.TransformerEncoderLayer`` x N with the question's padding
mask, optional MoE over all tokens :224-284,326-337 -- moe_type 'vqa' = VQAMOELayer with the default one-of-each expert mix, 'standard' =
MOELayer of FeedForward experts behind a TopK router; its load-balance statistic enters the loss as a python float :590-591 --, LayerNorm),
TransformerDecoder :383-451 (tied embedding + sinusoidal positions, pre-LN ``nn.TransformerDecoderLayer``
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


def fusion_moe(sd, fused, top_k=2, moe_loss_weight=0.01, router_out=None):
    """The fusion MoE when the state dict holds one (``fusion.moe_layer.*``): returns (output, aux).  Expert kinds from the parameter
    names: VQAMOELayer's mix (vqa_oracle.vqa_moe_expert_kinds; generative_vqa_model.py:232-245: one Vision / Text / Multimodal / Segmentation
    expert by default, NoisyTopK router = clean logits in eval) or FeedForward experts (``experts.N.fc1``: MOELayer(config=...), TopK router
    whose load-balance weight is config.moe_loss_weight :249-255,266-283).  Every expert sees ALL tokens of a sample (moe_layer.py:151-168)."""
    p = 'fusion.moe_layer.'
    n = _count(sd, p + 'experts.{}.output_norm.weight') or _count(sd, p + 'experts.{}.layer_norm.weight')
    if (p + 'experts.0.fc1.weight') in sd:
        kinds, lbw = ['feedforward'] * n, moe_loss_weight
    else:
        kinds, lbw = [], 0.01
        for e in range(n):
            q = f'{p}experts.{e}.'
            kinds.append('vision' if q + 'spatial_norm.weight' in sd else 'text' if q + 'attention_norm.weight' in sd else
                         'segmentation' if q + 'mask_tokens' in sd else 'detection' if q + 'object_queries' in sd else 'multimodal')
    if router_out is None:
        router_out = vo.noisy_topk_router(sd, p + 'router.', fused, top_k, None, 1.0, lbw)
    return vo.moe_layer(sd, p, fused, kinds, top_k, router_out=router_out)


def generative_forward(sd, pixel_values, input_ids, attention_mask, decoder_input_ids, decoder_attention_mask=None, labels=None, *,
                       vit_heads, text_heads, fusion_heads, decoder_heads, label_smoothing=0.1, top_k=2, moe_loss_weight=0.01, aux_out=None):
    """Returns (logits [B, A, V], loss | None, encoder_hidden_states).  ``aux_out``: optional dict that receives the MoE router's aux outputs."""
    vis = vo.clip_vision_forward(sd, 'visual_encoder.vision_model.', pixel_values, vit_heads)
    txt = vo.roberta_forward(sd, 'question_encoder.encoder.', input_ids, attention_mask, text_heads)
    fused = torch.cat([vis, txt], dim=1)
    B, nv = vis.shape[0], vis.shape[1]
    kpm = torch.cat([torch.zeros(B, nv, dtype=torch.bool), ~attention_mask.bool()], dim=1)
    for i in range(_count(sd, 'fusion.layers.{}.norm1.weight')):
        fused = encoder_layer(sd, f'fusion.layers.{i}.', fused, fusion_heads, kpm)
    moe_aux = 0.0
    if 'fusion.moe_layer.router.gate.weight' in sd:
        fused, aux = fusion_moe(sd, fused, top_k, moe_loss_weight)
        moe_aux = float(aux['load_balance_loss'])            # .item()-ed in the reference (:333-337): a constant, no gradient
        if aux_out is not None:
            aux_out.update(aux)
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
        if moe_aux > 0:
            loss = loss + moe_loss_weight * moe_aux          # generative_vqa_model.py:590-591
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


# ---------------------------------------------------------------------------------------------------------------------------------------
# stand-alone CrossAttentionFusion (reference src/modeling/fusion/fusion_approaches.py:59-281)
# ---------------------------------------------------------------------------------------------------------------------------------------

def cross_attention_block(sd, p, v, t, vmask, tmask, num_heads):
    """fusion_approaches.py:243-281: text attends to vision, vision attends to the updated text; post-LN, GELU FFN."""
    a = vo.mha(sd, p + 'v2t_attention', t, v, v, num_heads, (~vmask) if vmask is not None else None)
    t = vo.layer_norm(sd, p + 'v2t_norm1', t + a)
    t = vo.layer_norm(sd, p + 'v2t_norm2', t + vo.linear(sd, p + 'v2t_ffn.3', F.gelu(vo.linear(sd, p + 'v2t_ffn.0', t))))
    a = vo.mha(sd, p + 't2v_attention', v, t, t, num_heads, (~tmask) if tmask is not None else None)
    v = vo.layer_norm(sd, p + 't2v_norm1', v + a)
    v = vo.layer_norm(sd, p + 't2v_norm2', v + vo.linear(sd, p + 't2v_ffn.3', F.gelu(vo.linear(sd, p + 't2v_ffn.0', v))))
    return v, t


def cross_attention_fusion(sd, v, t, vmask=None, tmask=None, *, num_heads, fusion_method='concat'):
    """fusion_approaches.py:143-188."""
    if 'vision_projection.weight' in sd:
        v = vo.linear(sd, 'vision_projection', v)
    if 'text_projection.weight' in sd:
        t = vo.linear(sd, 'text_projection', t)
    for i in range(_count(sd, 'cross_attention_layers.{}.v2t_norm1.weight')):
        v, t = cross_attention_block(sd, f'cross_attention_layers.{i}.', v, t, vmask, tmask, num_heads)
    vp, tp = v.mean(dim=1), t.mean(dim=1)
    fused = torch.cat([vp, tp], dim=-1) if fusion_method == 'concat' else (vp + tp if fusion_method == 'add' else vp * tp)
    h = F.gelu(vo.layer_norm(sd, 'fusion_layer.1', vo.linear(sd, 'fusion_layer.0', fused)))
    return vo.layer_norm(sd, 'fusion_layer.5', vo.linear(sd, 'fusion_layer.4', h))


def fusion_fixture_inputs(meta):
    from . import det_weights as dw
    c, seed = meta['case'], meta['seed']
    v = dw.normal('fusion.vision', (c['B'], c['V'], c['vision_dim']), seed)
    t = dw.normal('fusion.text', (c['B'], c['T'], c['text_dim']), seed)
    vmask = torch.ones(c['B'], c['V'], dtype=torch.bool)
    tmask = torch.ones(c['B'], c['T'], dtype=torch.bool)
    tmask[1, (c['T'] * 5) // 8:] = False
    vmask[0, -2:] = False
    gy = dw.normal('fusion.gy', (c['B'], c['output_dim']), seed)
    return v, t, vmask, tmask, gy
