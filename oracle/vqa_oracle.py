"""CPU oracle for the AutoViVQA forward/backward hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, with plain fp32 ``torch`` ops on the CPU, the arithmetic of the reference's
hot path.  It is the checker for the HIP path; it is never the thing
shipped or measured: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.  It imports neither ``transformers`` nor anything under
``/root/reference`` and is driven purely by a flat ``state_dict`` using the reference's key names
(SURVEY.md Appendix A), so it travels to the GPU box.

Pinning: the reference holds no golden vectors for this path.  This restatement is
pinned against fixtures under ``tests/golden/`` that were produced by importing the reference's
own modules in the build container (``oracle/gen_golden.py``); ``tests/test_oracle_golden.py``
checks every function here against them.

Third-party arithmetic (absent from /root/reference): ``transformers`` (pinned 4.57.2 by the
reference's poetry.lock, 5.15.0 in this image) for ``CLIPVisionModel`` / ``RobertaModel`` and
``torch.nn`` for ``MultiheadAttention`` / ``TransformerDecoderLayer``; their published algorithms
are restated below and anchored on the reference's call sites:
``src/modeling/meta_arch/vqa_model.py:103-131,206-234,279-311,361-433,467-477,632-727``,
``src/modeling/moe/router.py:287-366``, ``src/modeling/moe/moe_layer.py:122-173,551-692``,
``src/modeling/moe/expert_types.py:159-199,270-312,390-445``,
``src/modeling/moe/specialized_experts.py:120-173,256-308``.
"""

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]
LN_EPS = 1e-5


# --------------------------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------------------------

def linear(sd: SD, name: str, x: torch.Tensor, bias: bool = True) -> torch.Tensor:
    b = sd.get(name + '.bias') if bias else None
    return F.linear(x, sd[name + '.weight'], b)


def layer_norm(sd: SD, name: str, x: torch.Tensor) -> torch.Tensor:
    w = sd[name + '.weight']
    return F.layer_norm(x, (w.shape[0],), w, sd[name + '.bias'], LN_EPS)


def quick_gelu(x: torch.Tensor) -> torch.Tensor:
    return x * torch.sigmoid(1.702 * x)


def _heads(x: torch.Tensor, h: int) -> torch.Tensor:
    b, s, d = x.shape
    return x.view(b, s, h, d // h).transpose(1, 2)          # [B,H,S,Dh]


def sdpa(q, k, v, num_heads: int, key_padding_mask: Optional[torch.Tensor] = None):
    """softmax(q k^T / sqrt(Dh) + mask) v on [B,S,D] tensors; ``key_padding_mask`` True = ignore key."""
    qh, kh, vh = _heads(q, num_heads), _heads(k, num_heads), _heads(v, num_heads)
    scores = torch.matmul(qh, kh.transpose(-1, -2)) * (qh.shape[-1] ** -0.5)
    if key_padding_mask is not None:
        scores = scores.masked_fill(key_padding_mask[:, None, None, :].bool(), float('-inf'))
    probs = torch.softmax(scores.float(), dim=-1).to(qh.dtype)
    out = torch.matmul(probs, vh)
    b, h, s, dh = out.shape
    return out.transpose(1, 2).reshape(b, s, h * dh)


def mha(sd: SD, name: str, query, key, value, num_heads: int, key_padding_mask=None):
    """``torch.nn.MultiheadAttention`` (batch_first, packed in_proj, eval mode, averaged weights
    discarded) as called at vqa_model.py:300,304 and expert_types.py:187,297."""
    w, b = sd[name + '.in_proj_weight'], sd[name + '.in_proj_bias']
    d = w.shape[1]
    q = F.linear(query, w[:d], b[:d])
    k = F.linear(key, w[d:2 * d], b[d:2 * d])
    v = F.linear(value, w[2 * d:], b[2 * d:])
    ctx = sdpa(q, k, v, num_heads, key_padding_mask)
    return linear(sd, name + '.out_proj', ctx)


# --------------------------------------------------------------------------------------------
# a1: CLIP ViT vision tower (HF CLIPVisionModel, eager path)
# --------------------------------------------------------------------------------------------

def clip_vision_forward(sd: SD, prefix: str, pixel_values: torch.Tensor, num_heads: int = 12) -> torch.Tensor:
    """Returns the UN-normalised ``last_hidden_state`` [B, 1+P, D] (vqa_model.py:116-121)."""
    p = prefix
    wpe = sd[p + 'embeddings.patch_embedding.weight']                  # [D,3,ps,ps], no bias
    patch = wpe.shape[-1]
    x = F.conv2d(pixel_values, wpe, stride=patch).flatten(2).transpose(1, 2)   # [B,P,D]
    cls = sd[p + 'embeddings.class_embedding'].expand(x.shape[0], 1, -1)
    x = torch.cat([cls, x], dim=1) + sd[p + 'embeddings.position_embedding.weight'][None]
    x = layer_norm(sd, p + 'pre_layrnorm', x)
    n_layers = 0
    while f'{p}encoder.layers.{n_layers}.layer_norm1.weight' in sd:
        n_layers += 1
    for i in range(n_layers):
        lp = f'{p}encoder.layers.{i}.'
        h = layer_norm(sd, lp + 'layer_norm1', x)
        q = linear(sd, lp + 'self_attn.q_proj', h)
        k = linear(sd, lp + 'self_attn.k_proj', h)
        v = linear(sd, lp + 'self_attn.v_proj', h)
        x = x + linear(sd, lp + 'self_attn.out_proj', sdpa(q, k, v, num_heads))
        h = layer_norm(sd, lp + 'layer_norm2', x)
        h = linear(sd, lp + 'mlp.fc2', quick_gelu(linear(sd, lp + 'mlp.fc1', h)))
        x = x + h
    return x


# --------------------------------------------------------------------------------------------
# a2: RoBERTa / PhoBERT encoder (HF RobertaModel, eval mode)
# --------------------------------------------------------------------------------------------

def roberta_position_ids(input_ids: torch.Tensor, pad_id: int = 1) -> torch.Tensor:
    m = input_ids.ne(pad_id).int()
    return (torch.cumsum(m, dim=1) * m).long() + pad_id


def roberta_forward(sd: SD, prefix: str, input_ids: torch.Tensor, attention_mask: torch.Tensor,
                    num_heads: int = 12, pad_id: int = 1) -> torch.Tensor:
    """Returns ``last_hidden_state`` [B,S,D]; the pooler is computed by HF but unused (vqa_model.py:226)."""
    p = prefix
    pos = roberta_position_ids(input_ids, pad_id)
    # both tables are nn.Embedding(padding_idx=pad_id): row ``pad_id`` receives no gradient
    x = (F.embedding(input_ids, sd[p + 'embeddings.word_embeddings.weight'], padding_idx=pad_id)
         + sd[p + 'embeddings.token_type_embeddings.weight'][0]
         + F.embedding(pos, sd[p + 'embeddings.position_embeddings.weight'], padding_idx=pad_id))
    x = layer_norm(sd, p + 'embeddings.LayerNorm', x)
    kpm = attention_mask == 0
    n_layers = 0
    while f'{p}encoder.layer.{n_layers}.attention.self.query.weight' in sd:
        n_layers += 1
    for i in range(n_layers):
        lp = f'{p}encoder.layer.{i}.'
        q = linear(sd, lp + 'attention.self.query', x)
        k = linear(sd, lp + 'attention.self.key', x)
        v = linear(sd, lp + 'attention.self.value', x)
        a = linear(sd, lp + 'attention.output.dense', sdpa(q, k, v, num_heads, kpm))
        x = layer_norm(sd, lp + 'attention.output.LayerNorm', a + x)
        h = F.gelu(linear(sd, lp + 'intermediate.dense', x))
        x = layer_norm(sd, lp + 'output.LayerNorm', linear(sd, lp + 'output.dense', h) + x)
    return x


def pool_text(seq: torch.Tensor, attention_mask: torch.Tensor, strategy: str) -> torch.Tensor:
    """vqa_model.py:179-204."""
    if strategy == 'cls':
        return seq[:, 0, :]
    m = attention_mask.unsqueeze(-1).expand(seq.size())
    if strategy == 'max':
        return seq.masked_fill(m == 0, -1e9).max(dim=1)[0]
    m = m.float()
    return (seq * m).sum(dim=1) / m.sum(dim=1).clamp(min=1e-9)


# --------------------------------------------------------------------------------------------
# a3/a4: fusion
# --------------------------------------------------------------------------------------------

def cross_modal_attention(sd: SD, prefix: str, query, key_value, num_heads: int,
                          query_mask=None, kv_mask=None) -> torch.Tensor:
    """vqa_model.py:279-311 (post-LN: self-MHA, cross-MHA, FFN), eval mode."""
    p = prefix
    x = query
    x = layer_norm(sd, p + 'norm1', x + mha(sd, p + 'self_attn', x, x, x, num_heads, query_mask))
    x = layer_norm(sd, p + 'norm2', x + mha(sd, p + 'cross_attn', x, key_value, key_value, num_heads, kv_mask))
    f = linear(sd, p + 'ffn.3', F.gelu(linear(sd, p + 'ffn.0', x)))
    return layer_norm(sd, p + 'norm3', x + f)


def multimodal_fusion(sd: SD, prefix: str, fusion_type: str, num_heads: int, visual, text,
                      visual_mask=None, text_mask=None, use_layer_norm: bool = True) -> torch.Tensor:
    """vqa_model.py:361-433; unknown ``fusion_type`` (e.g. 'mcan') takes the add branch (F3)."""
    p = prefix
    if fusion_type == 'cross_attention':
        i = 0
        while f'{p}fusion_layers.{i}.norm1.weight' in sd:
            text = cross_modal_attention(sd, f'{p}fusion_layers.{i}.', text, visual, num_heads, text_mask, visual_mask)
            i += 1
        fused = linear(sd, p + 'output_proj', text[:, 0, :])
    else:
        v = visual[:, 0, :] if visual.dim() == 3 else visual
        t = text[:, 0, :] if text.dim() == 3 else text
        if fusion_type == 'concat':
            h = F.relu(linear(sd, p + 'fusion_layer.0', torch.cat([v, t], dim=-1)))
            fused = linear(sd, p + 'fusion_layer.3', h)
        elif fusion_type == 'bilinear':
            fused = F.bilinear(v, t, sd[p + 'bilinear.weight'], sd[p + 'bilinear.bias'])
        else:
            fused = linear(sd, p + 'fusion_layer', v + t)
    if use_layer_norm:
        fused = layer_norm(sd, p + 'layer_norm', fused)
    return fused


# --------------------------------------------------------------------------------------------
# a5: routers
# --------------------------------------------------------------------------------------------

def load_balance_loss(logits: torch.Tensor, expert_indices: torch.Tensor, weight: float = 0.01) -> torch.Tensor:
    """router.py:333-366 / :144-178."""
    e = logits.shape[-1]
    n = logits.shape[0] * logits.shape[1]
    onehot = F.one_hot(expert_indices, e).float().sum(dim=2)
    frac = onehot.sum(dim=[0, 1]) / n
    mean_prob = torch.softmax(logits, dim=-1).mean(dim=[0, 1])
    return weight * e * torch.sum(frac * mean_prob)


def noisy_topk_router(sd: SD, prefix: str, x: torch.Tensor, top_k: int, noise: Optional[torch.Tensor] = None,
                      noise_std: float = 1.0, lb_weight: float = 0.01):
    """router.py:287-331.  ``noise`` = the injected ``randn_like`` tensor of train mode (None = eval)."""
    clean = F.linear(x, sd[prefix + 'gate.weight'])
    logits = clean
    if noise is not None:
        logits = clean + noise * F.softplus(F.linear(x, sd[prefix + 'w_noise.weight'])) * noise_std
    w, idx = torch.topk(torch.softmax(logits, dim=-1), top_k, dim=-1)
    w = w / w.sum(dim=-1, keepdim=True)
    aux = {'load_balance_loss': load_balance_loss(clean, idx, lb_weight),
           'router_probs': torch.softmax(clean, dim=-1)}
    return w, idx, aux


def topk_router(sd: SD, prefix: str, x: torch.Tensor, top_k: int, lb_weight: float = 0.01):
    """router.py:105-142."""
    return noisy_topk_router(sd, prefix, x, top_k, None, 1.0, lb_weight)


def soft_router(sd: SD, prefix: str, x: torch.Tensor, temperature: float = 1.0):
    """router.py:205-235."""
    w = torch.softmax(F.linear(x, sd[prefix + 'gate.weight']) / temperature, dim=-1)
    e = w.shape[-1]
    idx = torch.arange(e).expand(x.size(0), x.size(1), -1)
    return w, idx, {'router_probs': w, 'entropy': (-(w * torch.log(w + 1e-10)).sum(-1)).mean()}


# --------------------------------------------------------------------------------------------
# a7-a11: experts (eval mode: every Dropout is the identity)
# --------------------------------------------------------------------------------------------

def vision_expert(sd: SD, p: str, x, num_heads: int = 8):
    """expert_types.py:159-199."""
    h = linear(sd, p + 'input_proj', x)
    h = layer_norm(sd, p + 'spatial_norm', h + mha(sd, p + 'spatial_attention', h, h, h, num_heads))
    h = h + linear(sd, p + 'transform.3', F.gelu(linear(sd, p + 'transform.0', h)))
    return layer_norm(sd, p + 'output_norm', linear(sd, p + 'output_proj', h))


def text_expert(sd: SD, p: str, x, num_heads: int = 8):
    """expert_types.py:270-312 (mask=None as MOELayer passes it)."""
    h = linear(sd, p + 'input_proj', x)
    h = layer_norm(sd, p + 'attention_norm', h + mha(sd, p + 'self_attention', h, h, h, num_heads))
    h = layer_norm(sd, p + 'ffn_norm', h + linear(sd, p + 'ffn.3', F.gelu(linear(sd, p + 'ffn.0', h))))
    return layer_norm(sd, p + 'output_norm', linear(sd, p + 'output_proj', h))


def multimodal_expert(sd: SD, p: str, x):
    """expert_types.py:390-445 with ``context=None`` (never passed: cross_attention/gate are dead, F9)."""
    h = linear(sd, p + 'input_proj', x)
    h = layer_norm(sd, p + 'transform_norm', h + linear(sd, p + 'transform.3', F.gelu(linear(sd, p + 'transform.0', h))))
    return layer_norm(sd, p + 'output_norm', linear(sd, p + 'output_proj', h))


def feedforward_expert(sd: SD, p: str, x):
    """expert_types.py:59-92."""
    h = linear(sd, p + 'fc2', F.gelu(linear(sd, p + 'fc1', x)))
    if x.size(-1) == h.size(-1):
        h = h + x
    return layer_norm(sd, p + 'layer_norm', h)


def transformer_decoder_layer(sd: SD, p: str, tgt, memory, num_heads: int = 8):
    """torch ``nn.TransformerDecoderLayer`` defaults (post-LN, gelu, no masks), eval mode."""
    x = layer_norm(sd, p + 'norm1', tgt + mha(sd, p + 'self_attn', tgt, tgt, tgt, num_heads))
    x = layer_norm(sd, p + 'norm2', x + mha(sd, p + 'multihead_attn', x, memory, memory, num_heads))
    f = linear(sd, p + 'linear2', F.gelu(linear(sd, p + 'linear1', x)))
    return layer_norm(sd, p + 'norm3', x + f)


def _decoder(sd: SD, p: str, tgt, memory, num_heads: int = 8):
    i = 0
    while f'{p}layers.{i}.norm1.weight' in sd:
        tgt = transformer_decoder_layer(sd, f'{p}layers.{i}.', tgt, memory, num_heads)
        i += 1
    return tgt


def segmentation_expert(sd: SD, p: str, x):
    """specialized_experts.py:120-173."""
    b, s, _ = x.shape
    h = linear(sd, p + 'input_proj', x)
    mask_feat = _decoder(sd, p + 'mask_transformer.', sd[p + 'mask_tokens'].expand(b, -1, -1), h)
    ht = h.transpose(1, 2)
    bf = F.gelu(F.conv1d(ht, sd[p + 'boundary_conv.0.weight'], sd[p + 'boundary_conv.0.bias'], padding=1))
    bf = F.gelu(F.conv1d(bf, sd[p + 'boundary_conv.2.weight'], sd[p + 'boundary_conv.2.bias'], padding=1))
    bf = bf.transpose(1, 2)
    pooled = mask_feat.mean(dim=1, keepdim=True).expand(-1, s, -1)
    sp = linear(sd, p + 'spatial_mlp.3', F.gelu(linear(sd, p + 'spatial_mlp.0', torch.cat([bf, pooled], dim=-1))))
    h = h + bf + sp
    return layer_norm(sd, p + 'output_norm', linear(sd, p + 'output_proj', h))


def object_detection_expert(sd: SD, p: str, x):
    """specialized_experts.py:256-308."""
    b = x.shape[0]
    h = linear(sd, p + 'input_proj', x)
    obj = _decoder(sd, p + 'decoder.', sd[p + 'object_queries'].expand(b, -1, -1), h)
    obj = F.gelu(linear(sd, p + 'object_aggregation.0', obj))
    h = h + mha(sd, p + 'query_feature_attention', h, obj, obj, 8)
    return layer_norm(sd, p + 'output_norm', linear(sd, p + 'output_proj', h))


EXPERT_FNS = {
    'vision': vision_expert, 'text': text_expert, 'multimodal': multimodal_expert,
    'segmentation': segmentation_expert, 'detection': object_detection_expert,
    'feedforward': feedforward_expert,
}


def vqa_moe_expert_kinds(num_vision: int, num_text: int, num_multimodal: int, num_specialized: int):
    """Expert order of ``VQAMOELayer.__init__`` (moe_layer.py:616-689); only the first two
    specialised kinds are restated (Seg, Det: the ones reachable at <= 8 experts)."""
    spec = ['segmentation', 'detection', 'ocr', 'scene']
    return (['vision'] * num_vision + ['text'] * num_text + ['multimodal'] * num_multimodal
            + [spec[i % 4] for i in range(num_specialized)])


def expert_split(num_experts: int):
    """vqa_model.py:533-543."""
    per, rem = max(1, num_experts // 4), num_experts % 4
    return per + (1 if rem > 0 else 0), per + (1 if rem > 1 else 0), per + (1 if rem > 2 else 0), per


# --------------------------------------------------------------------------------------------
# a6: MOELayer.forward (dense masked combine)
# --------------------------------------------------------------------------------------------

def moe_combine(expert_outputs, routing_weights, expert_indices):
    """moe_layer.py:146-168: sum_e expert_e(x) * (sum_k w_k [idx_k == e]); skips unrouted experts.
    ``expert_outputs`` is a callable e -> [B,S,D] so skipped experts are never evaluated."""
    out = None
    for e in range(expert_outputs.n):
        sel = (expert_indices == e)
        if not sel.any():
            continue
        w = (routing_weights * sel.float()).sum(dim=-1)
        y = expert_outputs(e) * w.unsqueeze(-1)
        out = y if out is None else out + y
    return out


class _ExpertBank:
    def __init__(self, sd, prefix, kinds, x):
        self.sd, self.prefix, self.kinds, self.x, self.n = sd, prefix, kinds, x, len(kinds)

    def __call__(self, e):
        return EXPERT_FNS[self.kinds[e]](self.sd, f'{self.prefix}experts.{e}.', self.x)


def moe_layer(sd: SD, prefix: str, x: torch.Tensor, kinds, top_k: int, router_out=None,
              noise: Optional[torch.Tensor] = None):
    """``VQAMOELayer`` forward = ``MOELayer.forward`` (moe_layer.py:122-173)."""
    if router_out is None:
        router_out = noisy_topk_router(sd, prefix + 'router.', x, top_k, noise)
    w, idx, aux = router_out
    out = moe_combine(_ExpertBank(sd, prefix, kinds, x), w, idx)
    if out is None:
        out = torch.zeros(x.shape[0], x.shape[1], sd[prefix + 'output_norm.weight'].shape[0])
    return layer_norm(sd, prefix + 'output_norm', out), aux


# --------------------------------------------------------------------------------------------
# a12/a13: answer head + whole model
# --------------------------------------------------------------------------------------------

def answer_head(sd: SD, prefix: str, x: torch.Tensor) -> torch.Tensor:
    """vqa_model.py:452-477: (Linear, ReLU, Dropout)* + Linear; Sequential indices 0,3,6,..."""
    idx = sorted(int(k[len(prefix + 'classifier.'):].split('.')[0]) for k in sd
                 if k.startswith(prefix + 'classifier.') and k.endswith('.weight'))
    for j, i in enumerate(idx):
        x = linear(sd, f'{prefix}classifier.{i}', x)
        if j + 1 < len(idx):
            x = F.relu(x)
    return x


def vqa_forward(sd: SD, cfg, pixel_values, input_ids, attention_mask, labels=None,
                vit_heads: int = 12, text_heads: int = 12):
    """``VietnameseVQAModel.forward`` (vqa_model.py:632-727) in eval mode, knowledge off.
    ``cfg`` is any object with the VQAModelConfig attribute tree.  Returns (logits, loss, predictions)."""
    vis = clip_vision_forward(sd, 'visual_encoder.backbone.', pixel_values, vit_heads)
    txt = roberta_forward(sd, 'text_encoder.encoder.', input_ids, attention_mask, text_heads)
    if 'visual_encoder.projection.weight' in sd:
        vis = linear(sd, 'visual_encoder.projection', vis)
    if 'text_encoder.projection.weight' in sd:
        txt = linear(sd, 'text_encoder.projection', txt)
    fused = multimodal_fusion(sd, 'fusion.', cfg.fusion.fusion_type, cfg.fusion.num_heads, vis, txt,
                              text_mask=~attention_mask.bool(), use_layer_norm=cfg.fusion.use_layer_norm)
    if cfg.moe.use_moe:
        kinds = vqa_moe_expert_kinds(*expert_split(cfg.moe.num_experts))
        fused, _ = moe_layer(sd, 'moe_layer.', fused.unsqueeze(1), kinds, cfg.moe.top_k)
        fused = fused.squeeze(1)
    logits = answer_head(sd, 'answer_head.', fused)
    loss = F.cross_entropy(logits, labels) if labels is not None else None
    return logits, loss, logits.argmax(dim=-1)


def forward_backward(sd: SD, cfg, pixel_values, input_ids, attention_mask, labels, **kw):
    """eval-mode forward + backward: returns (logits, loss, predictions, {name: grad})."""
    leaves = {k: v.detach().clone().requires_grad_(v.is_floating_point() and v.dim() > 0) for k, v in sd.items()}
    logits, loss, pred = vqa_forward(leaves, cfg, pixel_values, input_ids, attention_mask, labels, **kw)
    loss.backward()
    grads = {k: v.grad for k, v in leaves.items() if v.requires_grad and v.grad is not None}
    return logits.detach(), loss.detach(), pred, grads
