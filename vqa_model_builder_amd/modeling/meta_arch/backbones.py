"""Native parameter containers + HIP runners for the two third-party backbones of the path.

The reference obtains them from ``transformers`` (``CLIPVisionModel.from_pretrained`` /
``AutoModel.from_pretrained`` -> ``RobertaModel``; reference vqa_model.py:75-101,158-177).  Here the same
``state_dict`` layout (transformers-5.x spelling, 4.x ``vision_model.`` spelling accepted on
load) is held by plain ``nn.Linear`` / ``nn.LayerNorm`` / ``nn.Embedding`` containers whose ``forward`` is never
called: the arithmetic runs in ``hip.blocks.ClipRunner`` / ``RobertaRunner``.
"""

from types import SimpleNamespace

import torch
import torch.nn as nn

from ...hip.blocks import ClipRunner, RobertaRunner
from ...hip.shadow import ShadowSet


class _Weights:
    """Key -> tensor accessor handed to a runner: .p(key) fp32 parameter, .s(key) shadow view."""

    def __init__(self):
        self.params = {}          # key -> nn.Parameter (or list of parameters for packed grads)
        self.shadows = ShadowSet()

    def p(self, key):
        return self.params[key]

    def s(self, key):
        return self.shadows.get(key)


class _BlockFn(torch.autograd.Function):
    """One autograd node per block.  ``fwd(*inputs)`` -> (output, saved); ``bwd(saved, grad)`` -> (input_grads, G)."""

    @staticmethod
    def forward(ctx, owner, n_inputs, *tensors):
        inputs, params = tensors[:n_inputs], tensors[n_inputs:]
        out, saved = owner._hip_forward(*inputs)
        ctx.owner, ctx.saved, ctx.n_inputs = owner, saved, n_inputs
        return out

    @staticmethod
    def backward(ctx, dout):
        in_grads, pgrads = ctx.owner._hip_backward(ctx.saved, dout.contiguous(), ctx.needs_input_grad[2:2 + ctx.n_inputs])
        ctx.saved = None
        return (None, None) + tuple(in_grads) + tuple(pgrads)


class _ResumableBackward:
    """Data-parallel captured step: ``split_backward_after = l`` (or several such indices, descending) makes the block's autograd node run only layers L-1 .. l and hand
    out the gradient views (the arena slots of the lower layers are filled by ``resume_backward()``, in the next graph of the step,
    so the upper half's gradients travel while the lower half computes: graph.GraphedTrainStep)."""
    split_backward_after = None
    _pending_backward = None

    def _run_backward(self, saved, dout):
        if self.split_backward_after is None:
            return self._runner.backward(saved, dout)
        at = self.split_backward_after
        gen = self._runner.backward_steps(saved, dout, int(at) if isinstance(at, int) else tuple(int(a) for a in at))
        try:
            G = next(gen)
            self._pending_backward = gen
        except StopIteration as done:               # nothing to split (split point outside the layer range)
            G = done.value
        # the views handed to autograd now are FILLED LATER (resume_backward): correct only while ``p.grad`` IS that view
        # (AccumulateGrad steals it); graph.GraphedTrainStep checks every parameter's gradient storage against this one
        self._split_arena_ptr = next(iter(G.values())).untyped_storage().data_ptr()
        return G

    def resume_backward(self, to_end: bool = True):
        """Continues the pending backward: to its end (``to_end``, the two-piece split) or to the next split point."""
        gen = self._pending_backward
        if gen is None:
            return
        if to_end:
            self._pending_backward = None
            for _ in gen:
                pass
            return
        try:
            next(gen)
        except StopIteration:
            self._pending_backward = None


def _require_cuda(t, what):
    if not t.is_cuda:
        raise RuntimeError(f'{what}: the MI355X HIP path needs tensors on the GPU (got {t.device}); '
                           'there is no CPU fallback on the product path')


# ---------------------------------------------------------------------------------------------------------------------
# CLIP ViT
# ---------------------------------------------------------------------------------------------------------------------

class _ClipAttention(nn.Module):
    def __init__(self, D):
        super().__init__()
        self.k_proj, self.v_proj, self.q_proj, self.out_proj = (nn.Linear(D, D) for _ in range(4))


class _ClipMLP(nn.Module):
    def __init__(self, D, inter):
        super().__init__()
        self.fc1, self.fc2 = nn.Linear(D, inter), nn.Linear(inter, D)


class _ClipLayer(nn.Module):
    def __init__(self, D, inter, eps):
        super().__init__()
        self.self_attn = _ClipAttention(D)
        self.layer_norm1 = nn.LayerNorm(D, eps=eps)
        self.mlp = _ClipMLP(D, inter)
        self.layer_norm2 = nn.LayerNorm(D, eps=eps)


class _ClipEncoder(nn.Module):
    def __init__(self, L, D, inter, eps):
        super().__init__()
        self.layers = nn.ModuleList(_ClipLayer(D, inter, eps) for _ in range(L))


class _ClipEmbeddings(nn.Module):
    def __init__(self, D, image_size, patch):
        super().__init__()
        self.class_embedding = nn.Parameter(torch.randn(D))
        self.patch_embedding = nn.Conv2d(3, D, kernel_size=patch, stride=patch, bias=False)
        self.position_embedding = nn.Embedding((image_size // patch) ** 2 + 1, D)


class ClipVisionBackbone(_ResumableBackward, nn.Module):
    """CLIP vision tower.  ``forward(pixel_values)`` returns an object with ``.last_hidden_state`` [B,1+P,D]
    (un-normalised, exactly what the reference consumes at vqa_model.py:119-121)."""

    def __init__(self, hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12,
                 image_size=224, patch_size=32, layer_norm_eps=1e-5):
        super().__init__()
        D = hidden_size
        self.config = SimpleNamespace(hidden_size=D, intermediate_size=intermediate_size, num_hidden_layers=num_hidden_layers,
                                      num_attention_heads=num_attention_heads, image_size=image_size, patch_size=patch_size,
                                      hidden_act='quick_gelu', layer_norm_eps=layer_norm_eps)
        self.embeddings = _ClipEmbeddings(D, image_size, patch_size)
        self.pre_layrnorm = nn.LayerNorm(D, eps=layer_norm_eps)            # (sic) HF spelling
        self.encoder = _ClipEncoder(num_hidden_layers, D, intermediate_size, layer_norm_eps)
        self.post_layernorm = nn.LayerNorm(D, eps=layer_norm_eps)          # never on the path
        self._register_load_state_dict_pre_hook(self._accept_v4_keys)
        self._build_runner()

    @staticmethod
    def _accept_v4_keys(state_dict, prefix, *args):
        old = prefix + 'vision_model.'
        for k in [k for k in state_dict if k.startswith(old)]:
            state_dict[prefix + k[len(old):]] = state_dict.pop(k)

    def _build_runner(self):
        c = self.config
        D = c.hidden_size
        W = _Weights()
        P, S = W.params, W.shadows
        emb = self.embeddings
        P['cls'], P['pos'], P['patch_w'] = emb.class_embedding, emb.position_embedding.weight, emb.patch_embedding.weight
        P['pre_ln.w'], P['pre_ln.b'] = self.pre_layrnorm.weight, self.pre_layrnorm.bias
        S.add('patch_w', emb.patch_embedding.weight, (D, 3 * c.patch_size * c.patch_size))
        for l, layer in enumerate(self.encoder.layers):
            k, a, m = f'l{l}.', layer.self_attn, layer.mlp
            S.add(k + 'qkv_w', [a.q_proj.weight, a.k_proj.weight, a.v_proj.weight], (3 * D, D))
            S.add(k + 'qkv_b', [a.q_proj.bias, a.k_proj.bias, a.v_proj.bias], (3 * D,), arena='f32')
            S.add(k + 'out_w', a.out_proj.weight, (D, D))
            S.add(k + 'fc1_w', m.fc1.weight, (c.intermediate_size, D))
            S.add(k + 'fc2_w', m.fc2.weight, (D, c.intermediate_size))
            P[k + 'qkv_w'] = [a.q_proj.weight, a.k_proj.weight, a.v_proj.weight]
            P[k + 'qkv_b'] = [a.q_proj.bias, a.k_proj.bias, a.v_proj.bias]
            P[k + 'out_w'], P[k + 'out_b'] = a.out_proj.weight, a.out_proj.bias
            P[k + 'fc1_w'], P[k + 'fc1_b'] = m.fc1.weight, m.fc1.bias
            P[k + 'fc2_w'], P[k + 'fc2_b'] = m.fc2.weight, m.fc2.bias
            P[k + 'ln1.w'], P[k + 'ln1.b'] = layer.layer_norm1.weight, layer.layer_norm1.bias
            P[k + 'ln2.w'], P[k + 'ln2.b'] = layer.layer_norm2.weight, layer.layer_norm2.bias
        self._W = W
        self._runner = ClipRunner(W, c.num_hidden_layers, D, c.num_attention_heads, c.intermediate_size, c.patch_size,
                                  c.layer_norm_eps)
        self._flat = _flatten_param_keys(W.params)

    # -- autograd glue ---------------------------------------------------------------------------------------------
    def _hip_forward(self, px):
        self._W.shadows.refresh(px.device)
        return self._runner.forward(px.float())

    def _hip_backward(self, saved, dout, needs):
        return [None], _split_grads(self._flat, self._run_backward(saved, dout))

    def forward(self, pixel_values):
        _require_cuda(pixel_values, 'ClipVisionBackbone')
        c = self.config
        if pixel_values.shape[-1] != c.image_size or pixel_values.shape[-2] != c.image_size:
            raise ValueError(f"Input image size ({pixel_values.shape[-2]}*{pixel_values.shape[-1]}) doesn't match model "
                             f"({c.image_size}*{c.image_size}).")
        out = _BlockFn.apply(self, 1, pixel_values, *[p for _, p in self._flat])
        return SimpleNamespace(last_hidden_state=out)


def _flatten_param_keys(params):
    """[(key-or-(key,i,n0,n1), parameter)] in a fixed order; packed lists become row ranges of the packed grad."""
    flat = []
    for key, p in params.items():
        if isinstance(p, (list, tuple)):
            off = 0
            for q in p:
                flat.append(((key, off, off + q.shape[0]), q))
                off += q.shape[0]
        else:
            flat.append((key, p))
    return flat


def _split_grads(flat, G):
    out = []
    for key, p in flat:
        if not p.requires_grad:
            out.append(None)
        elif isinstance(key, tuple):
            k, a, b = key
            out.append(G[k][a:b].view(p.shape))
        else:
            out.append(G[key].view(p.shape))
    return out


# ---------------------------------------------------------------------------------------------------------------------
# RoBERTa / PhoBERT
# ---------------------------------------------------------------------------------------------------------------------

class _RobertaEmbeddings(nn.Module):
    def __init__(self, V, D, max_pos, type_vocab, pad, eps):
        super().__init__()
        self.word_embeddings = nn.Embedding(V, D, padding_idx=pad)
        self.word_embeddings.weight._vqa_sparse_rows = True      # FusedAdamW: a step's gradient is non-zero in <= batch x seq of the V rows (optim._touched_map)
        self.token_type_embeddings = nn.Embedding(type_vocab, D)
        self.LayerNorm = nn.LayerNorm(D, eps=eps)
        self.position_embeddings = nn.Embedding(max_pos, D, padding_idx=pad)   # registered last, as HF does (parameter order)


class _RobertaSelfAttention(nn.Module):
    def __init__(self, D):
        super().__init__()
        self.query, self.key, self.value = nn.Linear(D, D), nn.Linear(D, D), nn.Linear(D, D)


class _DenseLN(nn.Module):
    def __init__(self, din, dout, eps):
        super().__init__()
        self.dense = nn.Linear(din, dout)
        self.LayerNorm = nn.LayerNorm(dout, eps=eps)


class _Dense(nn.Module):
    def __init__(self, din, dout):
        super().__init__()
        self.dense = nn.Linear(din, dout)


class _RobertaAttention(nn.Module):
    def __init__(self, D, eps):
        super().__init__()
        self.self = _RobertaSelfAttention(D)
        self.output = _DenseLN(D, D, eps)


class _RobertaLayer(nn.Module):
    def __init__(self, D, inter, eps):
        super().__init__()
        self.attention = _RobertaAttention(D, eps)
        self.intermediate = _Dense(D, inter)
        self.output = _DenseLN(inter, D, eps)


class _RobertaEncoder(nn.Module):
    def __init__(self, L, D, inter, eps):
        super().__init__()
        self.layer = nn.ModuleList(_RobertaLayer(D, inter, eps) for _ in range(L))


class RobertaBackbone(_ResumableBackward, nn.Module):
    """RoBERTa encoder as PhoBERT uses it.  ``forward(input_ids, attention_mask)`` -> ``.last_hidden_state`` [B,S,D].
    The pooler exists for state_dict compatibility only (computed by HF, unused by the reference: F9)."""

    def __init__(self, vocab_size=64001, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                 intermediate_size=3072, max_position_embeddings=258, type_vocab_size=1, pad_token_id=1,
                 layer_norm_eps=1e-5, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1):
        super().__init__()
        D = hidden_size
        self.config = SimpleNamespace(vocab_size=vocab_size, hidden_size=D, num_hidden_layers=num_hidden_layers,
                                      num_attention_heads=num_attention_heads, intermediate_size=intermediate_size,
                                      max_position_embeddings=max_position_embeddings, type_vocab_size=type_vocab_size,
                                      pad_token_id=pad_token_id, layer_norm_eps=layer_norm_eps,
                                      hidden_dropout_prob=hidden_dropout_prob,
                                      attention_probs_dropout_prob=attention_probs_dropout_prob)
        self.embeddings = _RobertaEmbeddings(vocab_size, D, max_position_embeddings, type_vocab_size, pad_token_id, layer_norm_eps)
        self.encoder = _RobertaEncoder(num_hidden_layers, D, intermediate_size, layer_norm_eps)
        self.pooler = _Dense(D, D)
        self._build_runner()

    def _build_runner(self):
        c = self.config
        D = c.hidden_size
        W = _Weights()
        P, S = W.params, W.shadows
        e = self.embeddings
        P['word'], P['pos'], P['type'] = e.word_embeddings.weight, e.position_embeddings.weight, e.token_type_embeddings.weight
        P['emb_ln.w'], P['emb_ln.b'] = e.LayerNorm.weight, e.LayerNorm.bias
        for l, layer in enumerate(self.encoder.layer):
            k, a = f'l{l}.', layer.attention
            qkv = [a.self.query, a.self.key, a.self.value]
            S.add(k + 'qkv_w', [m.weight for m in qkv], (3 * D, D))
            S.add(k + 'qkv_b', [m.bias for m in qkv], (3 * D,), arena='f32')
            S.add(k + 'ao_w', a.output.dense.weight, (D, D))
            S.add(k + 'i_w', layer.intermediate.dense.weight, (c.intermediate_size, D))
            S.add(k + 'o_w', layer.output.dense.weight, (D, c.intermediate_size))
            P[k + 'qkv_w'], P[k + 'qkv_b'] = [m.weight for m in qkv], [m.bias for m in qkv]
            P[k + 'ao_w'], P[k + 'ao_b'] = a.output.dense.weight, a.output.dense.bias
            P[k + 'ao_ln.w'], P[k + 'ao_ln.b'] = a.output.LayerNorm.weight, a.output.LayerNorm.bias
            P[k + 'i_w'], P[k + 'i_b'] = layer.intermediate.dense.weight, layer.intermediate.dense.bias
            P[k + 'o_w'], P[k + 'o_b'] = layer.output.dense.weight, layer.output.dense.bias
            P[k + 'o_ln.w'], P[k + 'o_ln.b'] = layer.output.LayerNorm.weight, layer.output.LayerNorm.bias
        self._W = W
        self._runner = RobertaRunner(W, c.num_hidden_layers, D, c.num_attention_heads, c.intermediate_size, c.pad_token_id,
                                     c.hidden_dropout_prob, c.attention_probs_dropout_prob, c.layer_norm_eps)
        self._flat = _flatten_param_keys(W.params)

    def _hip_forward(self, ids, mask):
        self._W.shadows.refresh(ids.device)
        return self._runner.forward(ids, mask, self.training)

    def sparse_grad_rows(self):
        """(word-embedding parameter, token ids int64 [M], gradient rows fp32 [M, D], pad id) of the LAST backward, or None: the
        word table's gradient is M <= batch x seq rows scattered into a [64 001, 768] table -- the data-parallel exchange gathers
        the rows instead of all-reducing 196 MB of mostly zeros (dp.GradReducer.prepare_static)."""
        rows = self._runner.last_embed_rows
        w = self.embeddings.word_embeddings.weight
        if rows is None or not w.requires_grad:
            return None
        return w, rows[0].reshape(-1), rows[1], self.config.pad_token_id

    def _hip_backward(self, saved, dout, needs):
        return [None, None], _split_grads(self._flat, self._run_backward(saved, dout))

    def forward(self, input_ids, attention_mask=None):
        _require_cuda(input_ids, 'RobertaBackbone')
        c = self.config
        if input_ids.dim() != 2 or input_ids.dtype.is_floating_point or input_ids.dtype == torch.bool:
            raise ValueError(f'RobertaBackbone: input_ids must be an integer [batch, seq] tensor (got {input_ids.dtype} {tuple(input_ids.shape)})')
        if input_ids.shape[1] + c.pad_token_id + 1 > c.max_position_embeddings:      # position ids run up to seq + pad_id (HF raises an IndexError here)
            raise ValueError(f'RobertaBackbone: sequence length {input_ids.shape[1]} needs position ids up to {input_ids.shape[1] + c.pad_token_id}, '
                             f'max_position_embeddings is {c.max_position_embeddings}')
        input_ids = input_ids.long()                           # int32 ids from a collator are widened, never reinterpreted
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        if attention_mask.shape != input_ids.shape or attention_mask.device != input_ids.device:
            raise ValueError(f'RobertaBackbone: attention_mask {tuple(attention_mask.shape)} on {attention_mask.device} does not match '
                             f'input_ids {tuple(input_ids.shape)} on {input_ids.device}')
        out = _BlockFn.apply(self, 2, input_ids, attention_mask, *[p for _, p in self._flat])
        return SimpleNamespace(last_hidden_state=out)
