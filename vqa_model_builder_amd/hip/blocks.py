"""Block runners: hand-scheduled forward AND backward of the three heavy structures of the path -- the CLIP ViT layer
stack, the RoBERTa/PhoBERT layer stack and the CrossModalAttention fusion layer -- as straight sequences of kernel
launches with every fusion the kernels' epilogues offer (bias, activation + saved pre-activation, residual add,
dropout masks, activation derivative in dX, bf16 operand copies written by the producer).

Conventions: the residual stream is fp32 (as under torch autocast, where LayerNorm/adds run in fp32); every GEMM
operand is bf16; weight gradients are fp32.  Each runner's ``forward`` returns (output, saved) and ``backward`` takes
(saved, grad_output) and returns a dict {param_key: fp32 grad}.  Host code is Python on PyTorch-ROCm (north star);
torch supplies memory and the stream only.
"""

import torch

from . import kernels as K
from .kernels import Drop, NO_DROP

F32 = torch.float32


_rng_epoch = None       # device int64 word behind INDIRECT seeds (HIP-graph mode), see csrc/common.h resolve_seed
_rng_salt = 0


def enable_indirect_seeds(device) -> torch.Tensor:
    """Switches ``new_seed`` to INDIRECT seeds: bit 63 | 15-bit call-site salt << 48 | address of a device epoch word.
    The kernels derive the key from the epoch word's CURRENT value, so a captured HIP graph (whose kernel arguments are
    frozen) still draws fresh dropout masks every replay once ``advance_rng_epoch()`` is part of the graph."""
    global _rng_epoch
    if _rng_epoch is None or _rng_epoch.device != torch.device(device):
        start = int(torch.empty((), dtype=torch.int64).random_().item()) & 0x7FFFFFFFFFFF      # honours torch.manual_seed
        _rng_epoch = torch.full((1,), start, dtype=torch.int64, device=device)
        assert _rng_epoch.data_ptr() < (1 << 48)
    return _rng_epoch


def disable_indirect_seeds():
    global _rng_epoch
    _rng_epoch = None


def advance_rng_epoch():
    """One device-side increment; call once per training step (inside the captured region in graph mode)."""
    if _rng_epoch is not None:
        _rng_epoch.add_(1)


def new_seed() -> int:
    """Per-forward dropout seed from torch's CPU generator (honours torch.manual_seed; no device sync); an INDIRECT
    seed when ``enable_indirect_seeds`` is active."""
    global _rng_salt
    if _rng_epoch is not None:
        _rng_salt = (_rng_salt + 1) & 0x7FFF
        return (1 << 63) | (_rng_salt << 48) | _rng_epoch.data_ptr()
    return int(torch.empty((), dtype=torch.int64).random_().item()) & 0x7FFFFFFFFFFFFFFF


class GradArena:
    """One flat fp32 buffer per block and backward pass holding every parameter gradient of the block (packed where the
    weights are packed); the block's gradients are contiguous for the DP exchange.  Two regions:
      * ACCUMULATED slots first (biases, LayerNorm affine, embedding tables, cls/pos): zero-filled by ONE fill per backward,
        then added to by GEMM / LayerNorm / attention epilogues (fp32 atomics) and the embedding scatter;
      * OVERWRITTEN slots behind them (every ``*_w`` Linear weight: each is written whole by its weight-gradient GEMM), which
        are left uninitialised -- three quarters of the arena's bytes need no fill."""

    def __init__(self, params):
        self.slots, self.total = {}, 0
        shapes = {}
        for key, p in params.items():
            ps = p if isinstance(p, (list, tuple)) else [p]
            shapes[key] = (sum(q.shape[0] for q in ps),) + tuple(ps[0].shape[1:])
        order = [k for k in shapes if not self.overwritten(k)] + [k for k in shapes if self.overwritten(k)]
        self.zero_total = 0
        for key in order:
            shape = shapes[key]
            n = 1
            for d in shape:
                n *= d
            self.slots[key] = (self.total, n, shape)
            self.total += (n + 3) // 4 * 4                      # 16-byte aligned slots
            if not self.overwritten(key):
                self.zero_total = self.total

    @staticmethod
    def overwritten(key):
        return key.endswith('_w')

    def alloc(self, device):
        flat = torch.empty(self.total, dtype=F32, device=device)
        flat[:self.zero_total].zero_()
        return flat, {k: flat[o:o + n].view(shape) for k, (o, n, shape) in self.slots.items()}


def _mask_u8(mask_bool):
    if mask_bool is None:
        return None
    return mask_bool.to(torch.uint8).contiguous()


def _qkv_attention(xb, w_in, b_in, B, H, S, D, mask_u8, drop, fused):
    """Self-attention up to the out-projection: (qkv bf16 [B*S, 3D] kept for backward, context bf16 [B*S, D]).  One launch
    (csrc/fused_attn.h) where the shape is covered, else packed in-projection GEMM + attention kernel."""
    M = B * S
    if fused and K.fused_attention_covers(D, H, S, S):
        qkv = torch.empty((M, 3 * D), dtype=K.HALF(), device=xb.device)
        ctx = K.fused_inproj_attention_fwd(xb, xb, w_in, b_in, B, H, S, S, D, mask_u8, drop, q=qkv[:, :D], k=qkv[:, D:2 * D], v=qkv[:, 2 * D:],
                                           ldq=3 * D, ldk=3 * D, ldv=3 * D)
        return qkv, ctx
    _, qkv, _ = K.linear_fwd(xb, w_in, b_in, M, 3 * D, D, want_bf16=True)
    ctx = K.attention_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], 3 * D, 3 * D, 3 * D, B, H, S, S, D // H, mask_u8, drop)
    return qkv, ctx


def _q_kv_attention(xqb, xkvb, w_in, b_in, B, H, Sq, Skv, D, mask_u8, drop):
    """Attention with queries projected from ``xqb`` [B*Sq, D] and keys / values from ``xkvb`` [B*Skv, D] through one packed
    in_proj (cross-attention; self-attention of a single query row): (q bf16 [B*Sq, D], kv bf16 [B*Skv, 2D], context)."""
    Mq, Mkv = B * Sq, B * Skv
    if K.FUSED_ATTENTION_FUSION and K.fused_attention_covers(D, H, Sq, Skv):
        q = torch.empty((Mq, D), dtype=K.HALF(), device=xqb.device)
        kv = torch.empty((Mkv, 2 * D), dtype=K.HALF(), device=xqb.device)
        ctx = K.fused_inproj_attention_fwd(xqb, xkvb, w_in, b_in, B, H, Sq, Skv, D, mask_u8, drop, q=q, k=kv[:, :D], v=kv[:, D:],
                                           ldq=D, ldk=2 * D, ldv=2 * D)
        return q, kv, ctx
    _, q, _ = K.linear_fwd(xqb, w_in[:D], b_in[:D], Mq, D, D, want_bf16=True)
    _, kv, _ = K.linear_fwd(xkvb, w_in[D:], b_in[D:], Mkv, 2 * D, D, want_bf16=True)
    ctx = K.attention_fwd(q, kv[:, :D], kv[:, D:], D, 2 * D, 2 * D, B, H, Sq, Skv, D // H, mask_u8, drop)
    return q, kv, ctx


# ==================================================================================================================
# CLIP ViT vision tower (pre-LN, quick-GELU)     reference: vqa_model.py:103-131 -> HF CLIPVisionModel
# ==================================================================================================================

class ClipRunner:
    """W: object with .p(key)->fp32 parameter tensor and .s(key)->bf16/fp32 shadow view (see modeling.backbones)."""
    WEIGHTS = ('qkv_w', 'out_w', 'fc1_w', 'fc2_w')          # a layer's 16-bit weights: one contiguous span of the block's shadow arena

    def __init__(self, W, num_layers, D, heads, inter, patch, eps=1e-5):
        self.W, self.L, self.D, self.H, self.I, self.ps, self.eps = W, num_layers, D, heads, inter, patch, eps
        self.arena = GradArena(W.params)

    def forward(self, px):
        W, D, H, I = self.W, self.D, self.H, self.I
        B, C, Hh, Ww = px.shape
        P = (Hh // self.ps) * (Ww // self.ps)
        T, M, Kp = P + 1, B * (P + 1), C * self.ps * self.ps
        xp = K.patchify(px.contiguous(), self.ps)
        E, _, _ = K.linear_fwd(xp, W.s('patch_w'), None, B * P, D, Kp, want_f32=True)
        u = K.clip_assemble(E, W.p('cls'), W.p('pos'), B, P, D)
        x, _, mean0, rstd0 = K.layernorm_fwd(u, W.p('pre_ln.w'), W.p('pre_ln.b'), M, D, eps=self.eps)
        saved = dict(B=B, P=P, xp=xp, u=u, mean0=mean0, rstd0=rstd0, layers=[])
        for l in range(self.L):
            k = f'l{l}.'
            if l + 1 < self.L:                             # the next layer's weights leave HBM while this layer computes (kernels.prefetch_weights)
                K.prefetch_weights([W.s(f'l{l + 1}.{n}') for n in self.WEIGHTS])
            _, h1, m1, r1 = K.layernorm_fwd(x, W.p(k + 'ln1.w'), W.p(k + 'ln1.b'), M, D, want_f32=False, want_bf16=True, eps=self.eps)
            qkv, ctx = _qkv_attention(h1, W.s(k + 'qkv_w'), W.s(k + 'qkv_b'), B, H, T, D, None, NO_DROP, K.FUSED_ATTENTION_ENCODERS)
            x1, _, _ = K.linear_fwd(ctx, W.s(k + 'out_w'), W.p(k + 'out_b'), M, D, D, want_f32=True, residual=x)
            _, h2, m2, r2 = K.layernorm_fwd(x1, W.p(k + 'ln2.w'), W.p(k + 'ln2.b'), M, D, want_f32=False, want_bf16=True, eps=self.eps)
            _, g, a = K.linear_fwd(h2, W.s(k + 'fc1_w'), W.p(k + 'fc1_b'), M, I, D, want_bf16=True, want_pre=True, act=K.ACT_QUICK_GELU)
            x2, _, _ = K.linear_fwd(g, W.s(k + 'fc2_w'), W.p(k + 'fc2_b'), M, D, I, want_f32=True, residual=x1)
            saved['layers'].append((x, h1, m1, r1, qkv, ctx, x1, h2, m2, r2, a, g))
            x = x2
        return x.view(B, T, D), saved

    def backward(self, saved, dout):
        gen = self.backward_steps(saved, dout, None)
        try:
            while True:
                next(gen)
        except StopIteration as done:
            return done.value

    def backward_steps(self, saved, dout, split_after):
        """The backward as a generator: with ``split_after = l`` (or a collection of such layer indices) it yields the gradient dict once layers L-1 .. l are done (their
        LayerNorm partials reduced; their weight-gradient GEMMs queued) and finishes the rest when resumed -- the data-parallel
        captured step puts the two halves into two graphs so that the upper half's gradients travel while the lower half computes
        (graph.GraphedTrainStep).  Returns the gradient dict."""
        W, D, H, I = self.W, self.D, self.H, self.I
        B, P = saved['B'], saved['P']
        T, M = P + 1, B * (P + 1)
        dx = dout.reshape(M, D).contiguous().float()
        _, G = self.arena.alloc(dx.device)
        dxb = K.cast_bf16(dx)
        K.colsum_bf16(dxb, M, D, out=G[f'l{self.L - 1}.fc2_b'])          # later layers get it fused into LN1-backward
        for l in reversed(range(self.L)):
            if split_after is not None and l >= 0 and (l + 1) in ((split_after,) if isinstance(split_after, int) else tuple(split_after)):
                K.ln_reduce_flush()
                yield G
            k = f'l{l}.'
            if l > 0:
                K.prefetch_weights([W.s(f'l{l - 1}.{n}') for n in self.WEIGHTS])
            x, h1, m1, r1, qkv, ctx, x1, h2, m2, r2, a, g = saved['layers'][l]
            K.linear_dw(dxb, g, M, D, I, out=G[k + 'fc2_w'], prezeroed=False)
            _, da = K.linear_dx(dxb, W.s(k + 'fc2_w'), M, D, I, want_bf16=True, act_grad_of=a, act_bwd=K.ACT_QUICK_GELU, colsum=G[k + 'fc1_b'])
            K.linear_dw(da, h2, M, I, D, out=G[k + 'fc1_w'], prezeroed=False)
            dh2, _ = K.linear_dx(da, W.s(k + 'fc1_w'), M, I, D, want_f32=True)
            dx1, dx1b, _, _ = K.layernorm_bwd(dh2, x1, m2, r2, W.p(k + 'ln2.w'), M, D, dres=dx, want_bf16=True,
                                              dgamma=G[k + 'ln2.w'], dbeta=G[k + 'ln2.b'], dx_colsum=G[k + 'out_b'], defer=True)
            K.linear_dw(dx1b, ctx, M, D, D, out=G[k + 'out_w'], prezeroed=False)
            _, dctx = K.linear_dx(dx1b, W.s(k + 'out_w'), M, D, D, want_bf16=True)
            dqkv = torch.empty((M, 3 * D), dtype=K.HALF(), device=dx.device)
            qb = G[k + 'qkv_b']
            K.attention_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], dctx, 3 * D, 3 * D, 3 * D, B, H, T, T, D // H,
                            dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:], 3 * D, 3 * D, 3 * D,
                            dq_colsum=qb[:D], dk_colsum=qb[D:2 * D], dv_colsum=qb[2 * D:])      # qkv bias gradient, fused
            K.linear_dw(dqkv, h1, M, 3 * D, D, out=G[k + 'qkv_w'], prezeroed=False)
            dh1, _ = K.linear_dx(dqkv, W.s(k + 'qkv_w'), M, 3 * D, D, want_f32=True)
            nxt = G[f'l{l - 1}.fc2_b'] if l > 0 else None                  # dx is the fc2 output gradient of the layer below
            dx, dxb, _, _ = K.layernorm_bwd(dh1, x, m1, r1, W.p(k + 'ln1.w'), M, D, dres=dx1, want_bf16=True,
                                            dgamma=G[k + 'ln1.w'], dbeta=G[k + 'ln1.b'], dx_colsum=nxt, defer=True)
        du, _, _, _ = K.layernorm_bwd(dx, saved['u'], saved['mean0'], saved['rstd0'], W.p('pre_ln.w'), M, D,
                                      dgamma=G['pre_ln.w'], dbeta=G['pre_ln.b'], defer=True)
        dE = K.clip_assemble_bwd(du, B, P, D, G['cls'], G['pos'])
        Kp = saved['xp'].shape[1]
        K.linear_dw(dE, saved['xp'], B * P, D, Kp, out=G['patch_w'].view(D, Kp), prezeroed=False)
        K.ln_reduce_flush()
        K.wgrad_join()
        return G


# ==================================================================================================================
# RoBERTa / PhoBERT encoder (post-LN, erf-GELU, pad-aware positions)   reference: vqa_model.py:206-234 -> HF RobertaModel
# ==================================================================================================================

class RobertaRunner:
    last_embed_rows = None
    WEIGHTS = ('qkv_w', 'ao_w', 'i_w', 'o_w')               # a layer's 16-bit weights: one contiguous span of the block's shadow arena

    def __init__(self, W, num_layers, D, heads, inter, pad_id=1, hidden_drop=0.1, attn_drop=0.1, eps=1e-5):
        self.W, self.L, self.D, self.H, self.I = W, num_layers, D, heads, inter
        self.pad, self.pd, self.pa, self.eps = pad_id, hidden_drop, attn_drop, eps
        self.arena = GradArena(W.params)

    def forward(self, ids, attention_mask, training):
        W, D, H, I = self.W, self.D, self.H, self.I
        B, S = ids.shape
        M = B * S
        seed = new_seed() if training else 0
        pd, pa = (self.pd, self.pa) if training else (0.0, 0.0)
        ids = ids.contiguous()
        kpm = _mask_u8(attention_mask == 0) if attention_mask is not None else None
        u, pos_ids = K.roberta_embed_fwd(ids, W.p('word'), W.p('pos'), W.p('type'), B, S, D, self.pad)
        x, xb, mean0, rstd0 = K.layernorm_fwd(u, W.p('emb_ln.w'), W.p('emb_ln.b'), M, D, want_bf16=True, eps=self.eps,
                                              drop=Drop(pd, seed, 1))
        saved = dict(B=B, S=S, ids=ids, pos_ids=pos_ids, u=u, mean0=mean0, rstd0=rstd0, kpm=kpm, seed=seed, pd=pd, pa=pa, layers=[])
        for l in range(self.L):
            k = f'l{l}.'
            st = 8 * (l + 1)
            if l + 1 < self.L:
                K.prefetch_weights([W.s(f'l{l + 1}.{n}') for n in self.WEIGHTS])
            qkv, ctx = _qkv_attention(xb, W.s(k + 'qkv_w'), W.s(k + 'qkv_b'), B, H, S, D, kpm, Drop(pa, seed, st), K.FUSED_ATTENTION_ENCODERS)
            s1, _, _ = K.linear_fwd(ctx, W.s(k + 'ao_w'), W.p(k + 'ao_b'), M, D, D, want_f32=True, residual=x, drop=Drop(pd, seed, st + 1))
            x1, x1b, m1, r1 = K.layernorm_fwd(s1, W.p(k + 'ao_ln.w'), W.p(k + 'ao_ln.b'), M, D, want_bf16=True, eps=self.eps)
            _, g, a = K.linear_fwd(x1b, W.s(k + 'i_w'), W.p(k + 'i_b'), M, I, D, want_bf16=True, want_pre=True, act=K.ACT_GELU)
            s2, _, _ = K.linear_fwd(g, W.s(k + 'o_w'), W.p(k + 'o_b'), M, D, I, want_f32=True, residual=x1, drop=Drop(pd, seed, st + 2))
            x2, x2b, m2, r2 = K.layernorm_fwd(s2, W.p(k + 'o_ln.w'), W.p(k + 'o_ln.b'), M, D, want_bf16=True, eps=self.eps)
            saved['layers'].append((xb, qkv, ctx, s1, m1, r1, x1b, a, g, s2, m2, r2))
            x, xb = x2, x2b
        return x.view(B, S, D), saved

    def backward(self, saved, dout):
        gen = self.backward_steps(saved, dout, None)
        try:
            while True:
                next(gen)
        except StopIteration as done:
            return done.value

    def backward_steps(self, saved, dout, split_after):
        """Generator form of the backward (see ClipRunner.backward_steps): yields once layers L-1 .. ``split_after`` are done."""
        W, D, H, I = self.W, self.D, self.H, self.I
        B, S, seed, pd, pa, kpm = saved['B'], saved['S'], saved['seed'], saved['pd'], saved['pa'], saved['kpm']
        M = B * S
        dx = dout.reshape(M, D).contiguous().float()
        dev = dx.device
        _, G = self.arena.alloc(dev)
        for l in reversed(range(self.L)):
            if split_after is not None and l >= 0 and (l + 1) in ((split_after,) if isinstance(split_after, int) else tuple(split_after)):
                K.ln_reduce_flush()
                yield G
            k = f'l{l}.'
            st = 8 * (l + 1)
            if l > 0:
                K.prefetch_weights([W.s(f'l{l - 1}.{n}') for n in self.WEIGHTS])
            xb, qkv, ctx, s1, m1, r1, x1b, a, g, s2, m2, r2 = saved['layers'][l]
            # output LayerNorm:  x2 = LN(s2),  s2 = x1 + drop(dense(g));  bias grad of `dense` = colsum of the masked ds2
            ds2, ds2b, _, _ = K.layernorm_bwd(dx, s2, m2, r2, W.p(k + 'o_ln.w'), M, D, want_bf16=True, drop=Drop(pd, seed, st + 2), drop_mode=1,
                                              dgamma=G[k + 'o_ln.w'], dbeta=G[k + 'o_ln.b'], dx_colsum=G[k + 'o_b'], defer=True)
            K.linear_dw(ds2b, g, M, D, I, out=G[k + 'o_w'], prezeroed=False)
            _, da = K.linear_dx(ds2b, W.s(k + 'o_w'), M, D, I, want_bf16=True, act_grad_of=a, act_bwd=K.ACT_GELU, colsum=G[k + 'i_b'])
            K.linear_dw(da, x1b, M, I, D, out=G[k + 'i_w'], prezeroed=False)
            dx1, _ = K.linear_dx(da, W.s(k + 'i_w'), M, I, D, want_f32=True, residual=ds2)
            # attention-output LayerNorm:  x1 = LN(s1),  s1 = x + drop(dense(ctx))
            ds1, ds1b, _, _ = K.layernorm_bwd(dx1, s1, m1, r1, W.p(k + 'ao_ln.w'), M, D, want_bf16=True, drop=Drop(pd, seed, st + 1), drop_mode=1,
                                              dgamma=G[k + 'ao_ln.w'], dbeta=G[k + 'ao_ln.b'], dx_colsum=G[k + 'ao_b'], defer=True)
            K.linear_dw(ds1b, ctx, M, D, D, out=G[k + 'ao_w'], prezeroed=False)
            _, dctx = K.linear_dx(ds1b, W.s(k + 'ao_w'), M, D, D, want_bf16=True)
            dqkv = torch.empty((M, 3 * D), dtype=K.HALF(), device=dev)
            qb = G[k + 'qkv_b']
            K.attention_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], dctx, 3 * D, 3 * D, 3 * D, B, H, S, S, D // H,
                            dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:], 3 * D, 3 * D, 3 * D, kpm, Drop(pa, seed, st),
                            dq_colsum=qb[:D], dk_colsum=qb[D:2 * D], dv_colsum=qb[2 * D:])      # qkv bias gradient, fused
            K.linear_dw(dqkv, xb, M, 3 * D, D, out=G[k + 'qkv_w'], prezeroed=False)
            dx, _ = K.linear_dx(dqkv, W.s(k + 'qkv_w'), M, 3 * D, D, want_f32=True, residual=ds1)
        du, _, _, _ = K.layernorm_bwd(dx, saved['u'], saved['mean0'], saved['rstd0'], W.p('emb_ln.w'), M, D,
                                      drop=Drop(pd, seed, 1), drop_mode=2, dgamma=G['emb_ln.w'], dbeta=G['emb_ln.b'], defer=True)
        K.roberta_embed_bwd(du, saved['ids'], saved['pos_ids'], G['word'], G['pos'], G['type'], B, S, D, self.pad)
        self.last_embed_rows = (saved['ids'], du)          # what was scattered into the word table: the data-parallel exchange sends THIS
        K.ln_reduce_flush()
        K.wgrad_join()
        return G


# ==================================================================================================================
# CrossModalAttention fusion layer (post-LN: self-MHA, cross-MHA, FFN)       reference: vqa_model.py:237-311
# ==================================================================================================================

class CrossModalAttentionRunner:
    def __init__(self, W, D, heads, dropout, eps=1e-5):
        self.W, self.D, self.H, self.pd, self.eps = W, D, heads, dropout, eps
        self.arena = GradArena(W.params)

    def forward(self, query, key_value, query_mask, kv_mask, training, first_only=False):
        """query [B,Sq,D] fp32, key_value [B,Skv,D] fp32; masks bool [B,S] (True = ignore) or None.
        ``first_only``: only output row 0 of every sample is wanted (the LAST fusion layer: MultimodalFusion reads token 0 and
        nothing else, reference vqa_model.py:396-399) -> see forward_first."""
        if first_only:
            return self.forward_first(query, key_value, query_mask, kv_mask, training)
        W, D, H = self.W, self.D, self.H
        B, Sq, _ = query.shape
        Skv = key_value.shape[1]
        M, Mv, Dh, I = B * Sq, B * Skv, D // H, 4 * D
        seed = new_seed() if training else 0
        pd = self.pd if training else 0.0
        x = query.reshape(M, D).contiguous().float()
        xb = K.cast_bf16(x)
        kvb = K.cast_bf16(key_value.reshape(Mv, D).contiguous().float())
        qm, km = _mask_u8(query_mask), _mask_u8(kv_mask)
        # --- self attention
        qkv, ctx = _qkv_attention(xb, W.s('sa_in_w'), W.p('sa_in_b'), B, H, Sq, D, qm, Drop(pd, seed, 1), K.FUSED_ATTENTION_FUSION)
        s1, _, _ = K.linear_fwd(ctx, W.s('sa_out_w'), W.p('sa_out_b'), M, D, D, want_f32=True, residual=x, drop=Drop(pd, seed, 2))
        x1, x1b, m1, r1 = K.layernorm_fwd(s1, W.p('n1.w'), W.p('n1.b'), M, D, want_bf16=True, eps=self.eps)
        # --- cross attention: q from text, k/v from vision (rows D: of the packed in_proj)
        w_in, b_in = W.s('ca_in_w'), W.p('ca_in_b')
        q2, kv2, ctx2 = _q_kv_attention(x1b, kvb, w_in, b_in, B, H, Sq, Skv, D, km, Drop(pd, seed, 3))
        s2, _, _ = K.linear_fwd(ctx2, W.s('ca_out_w'), W.p('ca_out_b'), M, D, D, want_f32=True, residual=x1, drop=Drop(pd, seed, 4))
        x2, x2b, m2, r2 = K.layernorm_fwd(s2, W.p('n2.w'), W.p('n2.b'), M, D, want_bf16=True, eps=self.eps)
        # --- FFN: Linear, GELU, Dropout, Linear, Dropout
        _, g, a = K.linear_fwd(x2b, W.s('ffn0_w'), W.p('ffn0_b'), M, I, D, want_bf16=True, want_pre=True, act=K.ACT_GELU, drop=Drop(pd, seed, 5))
        s3, _, _ = K.linear_fwd(g, W.s('ffn3_w'), W.p('ffn3_b'), M, D, I, want_f32=True, residual=x2, drop=Drop(pd, seed, 6))
        x3, _, m3, r3 = K.layernorm_fwd(s3, W.p('n3.w'), W.p('n3.b'), M, D, eps=self.eps)
        saved = dict(B=B, Sq=Sq, Skv=Skv, seed=seed, pd=pd, qm=qm, km=km, xb=xb, kvb=kvb, qkv=qkv, ctx=ctx, s1=s1, m1=m1, r1=r1,
                     x1b=x1b, q2=q2, kv2=kv2, ctx2=ctx2, s2=s2, m2=m2, r2=r2, x2b=x2b, a=a, g=g, s3=s3, m3=m3, r3=r3)
        return x3.view(B, Sq, D), saved

    def backward(self, saved, dout, need_dkv=True):
        """Returns (G, dquery [B,Sq,D] fp32, dkey_value [B,Skv,D] fp32 | None)."""
        if saved.get('first'):
            return self.backward_first(saved, dout, need_dkv)
        W, D, H = self.W, self.D, self.H
        S = saved
        B, Sq, Skv, seed, pd = S['B'], S['Sq'], S['Skv'], S['seed'], S['pd']
        M, Mv, Dh, I = B * Sq, B * Skv, D // H, 4 * D
        dx3 = dout.reshape(M, D).contiguous().float()
        dev = dx3.device
        _, G = self.arena.alloc(dev)
        ds3, ds3b, _, _ = K.layernorm_bwd(dx3, S['s3'], S['m3'], S['r3'], W.p('n3.w'), M, D, want_bf16=True, drop=Drop(pd, seed, 6), drop_mode=1,
                                          dgamma=G['n3.w'], dbeta=G['n3.b'], dx_colsum=G['ffn3_b'], defer=True)
        K.linear_dw(ds3b, S['g'], M, D, I, out=G['ffn3_w'], prezeroed=False)
        _, da = K.linear_dx(ds3b, W.s('ffn3_w'), M, D, I, want_bf16=True, act_grad_of=S['a'], act_bwd=K.ACT_GELU, drop=Drop(pd, seed, 5),
                            colsum=G['ffn0_b'])
        K.linear_dw(da, S['x2b'], M, I, D, out=G['ffn0_w'], prezeroed=False)
        dx2, _ = K.linear_dx(da, W.s('ffn0_w'), M, I, D, want_f32=True, residual=ds3)
        # --- cross attention
        ds2, ds2b, _, _ = K.layernorm_bwd(dx2, S['s2'], S['m2'], S['r2'], W.p('n2.w'), M, D, want_bf16=True, drop=Drop(pd, seed, 4), drop_mode=1,
                                          dgamma=G['n2.w'], dbeta=G['n2.b'], dx_colsum=G['ca_out_b'], defer=True)
        K.linear_dw(ds2b, S['ctx2'], M, D, D, out=G['ca_out_w'], prezeroed=False)
        _, dctx2 = K.linear_dx(ds2b, W.s('ca_out_w'), M, D, D, want_bf16=True)
        dq2 = torch.empty((M, D), dtype=K.HALF(), device=dev)
        dkv2 = torch.empty((Mv, 2 * D), dtype=K.HALF(), device=dev)
        kv2 = S['kv2']
        K.attention_bwd(S['q2'], kv2[:, :D], kv2[:, D:], dctx2, D, 2 * D, 2 * D, B, H, Sq, Skv, Dh, dq2, dkv2[:, :D], dkv2[:, D:],
                        D, 2 * D, 2 * D, S['km'], Drop(pd, seed, 3),
                        dq_colsum=G['ca_in_b'][:D], dk_colsum=G['ca_in_b'][D:2 * D], dv_colsum=G['ca_in_b'][2 * D:])
        w_in = W.s('ca_in_w')
        K.linear_dw(dq2, S['x1b'], M, D, D, out=G['ca_in_w'][:D], prezeroed=False)
        dx1, _ = K.linear_dx(dq2, w_in[:D], M, D, D, want_f32=True, residual=ds2)
        K.linear_dw(dkv2, S['kvb'], Mv, 2 * D, D, out=G['ca_in_w'][D:], prezeroed=False)
        dkv = None
        if need_dkv:
            dkv, _ = K.linear_dx(dkv2, w_in[D:], Mv, 2 * D, D, want_f32=True)
            dkv = dkv.view(B, Skv, D)
        # --- self attention
        ds1, ds1b, _, _ = K.layernorm_bwd(dx1, S['s1'], S['m1'], S['r1'], W.p('n1.w'), M, D, want_bf16=True, drop=Drop(pd, seed, 2), drop_mode=1,
                                          dgamma=G['n1.w'], dbeta=G['n1.b'], dx_colsum=G['sa_out_b'], defer=True)
        K.linear_dw(ds1b, S['ctx'], M, D, D, out=G['sa_out_w'], prezeroed=False)
        _, dctx = K.linear_dx(ds1b, W.s('sa_out_w'), M, D, D, want_bf16=True)
        qkv = S['qkv']
        dqkv = torch.empty((M, 3 * D), dtype=K.HALF(), device=dev)
        K.attention_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], dctx, 3 * D, 3 * D, 3 * D, B, H, Sq, Sq, Dh,
                        dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:], 3 * D, 3 * D, 3 * D, S['qm'], Drop(pd, seed, 1),
                        dq_colsum=G['sa_in_b'][:D], dk_colsum=G['sa_in_b'][D:2 * D], dv_colsum=G['sa_in_b'][2 * D:])
        K.linear_dw(dqkv, S['xb'], M, 3 * D, D, out=G['sa_in_w'], prezeroed=False)
        dx, _ = K.linear_dx(dqkv, W.s('sa_in_w'), M, 3 * D, D, want_f32=True, residual=ds1)
        K.ln_reduce_flush()
        K.wgrad_join()
        return G, dx.view(B, Sq, D), dkv

    # ---- the same block when only token 0 of the output is consumed -------------------------------------------------------
    # Every row of this post-LN block depends on the other rows of ITS input only through the self-attention keys / values:
    # with nothing but output row 0 read downstream, all other query rows are dead code, forward and backward (their upstream
    # gradient is exactly zero in the reference too).  What remains: K|V projections over all text / vision rows, and the
    # whole chain -- Q projections, both attentions (one query per sample), out-projections, LayerNorms, the FFN -- on B rows
    # instead of B*Sq.  Outputs and every gradient are those of the full block (tests: golden parity of the whole model).
    def forward_first(self, query, key_value, query_mask, kv_mask, training):
        W, D, H = self.W, self.D, self.H
        B, Sq, _ = query.shape
        Skv = key_value.shape[1]
        M, Mv, Dh, I = B * Sq, B * Skv, D // H, 4 * D
        seed = new_seed() if training else 0
        pd = self.pd if training else 0.0
        x = query.reshape(M, D).contiguous().float()
        xb = K.cast_bf16(x)
        x0 = query[:, 0, :].contiguous().float()                        # [B, D]
        x0b = K.cast_bf16(x0)
        kvb = K.cast_bf16(key_value.reshape(Mv, D).contiguous().float())
        qm, km = _mask_u8(query_mask), _mask_u8(kv_mask)
        # --- self attention: one query (token 0) per sample over all text tokens
        w_sa, b_sa = W.s('sa_in_w'), W.p('sa_in_b')
        q0, kvs, ctx = _q_kv_attention(x0b, xb, w_sa, b_sa, B, H, 1, Sq, D, qm, Drop(pd, seed, 1))
        s1, _, _ = K.linear_fwd(ctx, W.s('sa_out_w'), W.p('sa_out_b'), B, D, D, want_f32=True, residual=x0, drop=Drop(pd, seed, 2))
        x1, x1b, m1, r1 = K.layernorm_fwd(s1, W.p('n1.w'), W.p('n1.b'), B, D, want_bf16=True, eps=self.eps)
        # --- cross attention
        w_in, b_in = W.s('ca_in_w'), W.p('ca_in_b')
        q2, kv2, ctx2 = _q_kv_attention(x1b, kvb, w_in, b_in, B, H, 1, Skv, D, km, Drop(pd, seed, 3))
        s2, _, _ = K.linear_fwd(ctx2, W.s('ca_out_w'), W.p('ca_out_b'), B, D, D, want_f32=True, residual=x1, drop=Drop(pd, seed, 4))
        x2, x2b, m2, r2 = K.layernorm_fwd(s2, W.p('n2.w'), W.p('n2.b'), B, D, want_bf16=True, eps=self.eps)
        # --- FFN
        _, g, a = K.linear_fwd(x2b, W.s('ffn0_w'), W.p('ffn0_b'), B, I, D, want_bf16=True, want_pre=True, act=K.ACT_GELU, drop=Drop(pd, seed, 5))
        s3, _, _ = K.linear_fwd(g, W.s('ffn3_w'), W.p('ffn3_b'), B, D, I, want_f32=True, residual=x2, drop=Drop(pd, seed, 6))
        x3, _, m3, r3 = K.layernorm_fwd(s3, W.p('n3.w'), W.p('n3.b'), B, D, eps=self.eps)
        saved = dict(first=True, B=B, Sq=Sq, Skv=Skv, seed=seed, pd=pd, qm=qm, km=km, xb=xb, x0b=x0b, kvb=kvb, q0=q0, kvs=kvs, ctx=ctx,
                     s1=s1, m1=m1, r1=r1, x1b=x1b, q2=q2, kv2=kv2, ctx2=ctx2, s2=s2, m2=m2, r2=r2, x2b=x2b, a=a, g=g, s3=s3, m3=m3, r3=r3)
        return x3.view(B, 1, D), saved

    def backward_first(self, saved, dout, need_dkv=True):
        W, D, H = self.W, self.D, self.H
        S = saved
        B, Sq, Skv, seed, pd = S['B'], S['Sq'], S['Skv'], S['seed'], S['pd']
        M, Mv, Dh, I = B * Sq, B * Skv, D // H, 4 * D
        dx3 = dout.reshape(B, D).contiguous().float()
        dev = dx3.device
        _, G = self.arena.alloc(dev)
        ds3, ds3b, _, _ = K.layernorm_bwd(dx3, S['s3'], S['m3'], S['r3'], W.p('n3.w'), B, D, want_bf16=True, drop=Drop(pd, seed, 6), drop_mode=1,
                                          dgamma=G['n3.w'], dbeta=G['n3.b'], dx_colsum=G['ffn3_b'], defer=True)
        K.linear_dw(ds3b, S['g'], B, D, I, out=G['ffn3_w'], prezeroed=False)
        _, da = K.linear_dx(ds3b, W.s('ffn3_w'), B, D, I, want_bf16=True, act_grad_of=S['a'], act_bwd=K.ACT_GELU, drop=Drop(pd, seed, 5),
                            colsum=G['ffn0_b'])
        K.linear_dw(da, S['x2b'], B, I, D, out=G['ffn0_w'], prezeroed=False)
        dx2, _ = K.linear_dx(da, W.s('ffn0_w'), B, I, D, want_f32=True, residual=ds3)
        # --- cross attention
        ds2, ds2b, _, _ = K.layernorm_bwd(dx2, S['s2'], S['m2'], S['r2'], W.p('n2.w'), B, D, want_bf16=True, drop=Drop(pd, seed, 4), drop_mode=1,
                                          dgamma=G['n2.w'], dbeta=G['n2.b'], dx_colsum=G['ca_out_b'], defer=True)
        K.linear_dw(ds2b, S['ctx2'], B, D, D, out=G['ca_out_w'], prezeroed=False)
        _, dctx2 = K.linear_dx(ds2b, W.s('ca_out_w'), B, D, D, want_bf16=True)
        dq2 = torch.empty((B, D), dtype=K.HALF(), device=dev)
        dkv2 = torch.empty((Mv, 2 * D), dtype=K.HALF(), device=dev)
        kv2 = S['kv2']
        K.attention_bwd(S['q2'], kv2[:, :D], kv2[:, D:], dctx2, D, 2 * D, 2 * D, B, H, 1, Skv, Dh, dq2, dkv2[:, :D], dkv2[:, D:],
                        D, 2 * D, 2 * D, S['km'], Drop(pd, seed, 3),
                        dq_colsum=G['ca_in_b'][:D], dk_colsum=G['ca_in_b'][D:2 * D], dv_colsum=G['ca_in_b'][2 * D:])
        w_in = W.s('ca_in_w')
        K.linear_dw(dq2, S['x1b'], B, D, D, out=G['ca_in_w'][:D], prezeroed=False)
        dx1, _ = K.linear_dx(dq2, w_in[:D], B, D, D, want_f32=True, residual=ds2)
        K.linear_dw(dkv2, S['kvb'], Mv, 2 * D, D, out=G['ca_in_w'][D:], prezeroed=False)
        dkv = None
        if need_dkv:
            dkv, _ = K.linear_dx(dkv2, w_in[D:], Mv, 2 * D, D, want_f32=True)
            dkv = dkv.view(B, Skv, D)
        # --- self attention
        ds1, ds1b, _, _ = K.layernorm_bwd(dx1, S['s1'], S['m1'], S['r1'], W.p('n1.w'), B, D, want_bf16=True, drop=Drop(pd, seed, 2), drop_mode=1,
                                          dgamma=G['n1.w'], dbeta=G['n1.b'], dx_colsum=G['sa_out_b'], defer=True)
        K.linear_dw(ds1b, S['ctx'], B, D, D, out=G['sa_out_w'], prezeroed=False)
        _, dctx = K.linear_dx(ds1b, W.s('sa_out_w'), B, D, D, want_bf16=True)
        kvs = S['kvs']
        dq0 = torch.empty((B, D), dtype=K.HALF(), device=dev)
        dkvs = torch.empty((M, 2 * D), dtype=K.HALF(), device=dev)
        K.attention_bwd(S['q0'], kvs[:, :D], kvs[:, D:], dctx, D, 2 * D, 2 * D, B, H, 1, Sq, Dh, dq0, dkvs[:, :D], dkvs[:, D:],
                        D, 2 * D, 2 * D, S['qm'], Drop(pd, seed, 1),
                        dq_colsum=G['sa_in_b'][:D], dk_colsum=G['sa_in_b'][D:2 * D], dv_colsum=G['sa_in_b'][2 * D:])
        w_sa = W.s('sa_in_w')
        K.linear_dw(dq0, S['x0b'], B, D, D, out=G['sa_in_w'][:D], prezeroed=False)
        K.linear_dw(dkvs, S['xb'], M, 2 * D, D, out=G['sa_in_w'][D:], prezeroed=False)
        dx, _ = K.linear_dx(dkvs, w_sa[D:], M, 2 * D, D, want_f32=True)            # every text row: through the keys / values
        dx0, _ = K.linear_dx(dq0, w_sa[:D], B, D, D, want_f32=True, residual=ds1)     # token 0: query path + residual stream
        dx = dx.view(B, Sq, D)
        dx[:, 0, :] += dx0
        K.ln_reduce_flush()
        K.wgrad_join()
        return G, dx, dkv


# ==================================================================================================================
# Tail of the model: [fusion.output_proj -> LayerNorm ->] Dropout -> AnswerHead.classifier
# reference: vqa_model.py:399-405 (projection + norm), 677 (dropout), 436-477 (classifier)
# ==================================================================================================================

class TailRunner:
    """Issued op by op this chain is ~45 launches of 2 - 7 us (casts, one GEMM + one cast per Linear, separate bias-gradient column
    sums, three passes for the dropout) with the launch gaps of a single dependent chain between them: profiles/r02/tail.md.  Here:
    forward = cast, projection GEMM, LayerNorm (emitting the DROPPED bf16 operand of the classifier), one GEMM per Linear with
    bias + ReLU + dropout in the epilogue; backward = one cast of dlogits with its column sums, one dX GEMM per Linear (ReLU' +
    dropout mask + the next bias gradient fused), LayerNorm backward masking dy on load, all weight gradients in the grouped launch.

    ``dims`` = [D, h1, .., C] of the classifier; ``pre`` = (D0, has_ln): the projection D0 -> D (and LayerNorm) in front."""

    def __init__(self, W, dims, pre=None, p_in=0.0, p_hidden=0.0, eps=1e-5):
        self.W, self.dims, self.pre, self.p_in, self.p_h, self.eps = W, list(dims), pre, p_in, p_hidden, eps
        self.arena = GradArena(W.params)

    @staticmethod
    def covers(dims, pre=None):
        return all(d % 8 == 0 for d in list(dims) + ([pre[0]] if pre else []))

    def forward(self, x, training):
        W, dims = self.W, self.dims
        B = x.shape[0]
        seed = new_seed() if training else 0
        p0 = self.p_in if training else 0.0
        ph = self.p_h if training else 0.0
        S = dict(B=B, seed=seed, p0=p0, ph=ph)
        x = x.contiguous().float()
        if self.pre is not None:
            D0, has_ln = self.pre
            xb = K.cast_bf16(x)
            S['xb'] = xb
            if has_ln:
                f, _, _ = K.linear_fwd(xb, W.s('proj_w'), W.p('proj_b'), B, dims[0], D0, want_f32=True)
                _, hb, m, r = K.layernorm_fwd(f, W.p('ln.w'), W.p('ln.b'), B, dims[0], want_f32=False, want_bf16=True, eps=self.eps,
                                              drop=Drop(p0, seed, 211))
                S.update(f=f, m=m, r=r)
            else:
                _, hb, _ = K.linear_fwd(xb, W.s('proj_w'), W.p('proj_b'), B, dims[0], D0, want_bf16=True, drop=Drop(p0, seed, 211))
        elif p0 > 0:
            _, hb = K.dropout_f32(x, Drop(p0, seed, 211), want_f32=False, want_bf16=True)
        else:
            hb = K.cast_bf16(x)
        acts, pres = [hb], []
        n = len(dims) - 1
        for i in range(n - 1):
            _, hb, a = K.linear_fwd(hb, W.s(f'l{i}_w'), W.p(f'l{i}_b'), B, dims[i + 1], dims[i], want_bf16=True, want_pre=True,
                                    act=K.ACT_RELU, drop=Drop(ph, seed, 212 + i))
            acts.append(hb)
            pres.append(a)
        logits, _, _ = K.linear_fwd(hb, W.s(f'l{n - 1}_w'), W.p(f'l{n - 1}_b'), B, dims[n], dims[n - 1], want_f32=True)
        S.update(acts=acts, pres=pres)
        return logits, S

    def backward(self, S, dlogits, need_dx=True):
        W, dims, B, seed, p0, ph = self.W, self.dims, S['B'], S['seed'], S['p0'], S['ph']
        n = len(dims) - 1
        _, G = self.arena.alloc(dlogits.device)
        acts, pres = S['acts'], S['pres']
        d = K.rows_mask_cast(dlogits.contiguous().float(), B, dims[n], colsum=G[f'l{n - 1}_b'])
        for i in range(n - 1, 0, -1):
            K.linear_dw(d, acts[i], B, dims[i + 1], dims[i], out=G[f'l{i}_w'])
            _, d = K.linear_dx(d, W.s(f'l{i}_w'), B, dims[i + 1], dims[i], want_bf16=True, act_grad_of=pres[i - 1], act_bwd=K.ACT_RELU,
                               drop=Drop(ph, seed, 212 + i - 1), colsum=G[f'l{i - 1}_b'])
        K.linear_dw(d, acts[0], B, dims[1], dims[0], out=G['l0_w'])
        dx = None
        if self.pre is None:
            if need_dx:
                dx, _ = K.linear_dx(d, W.s('l0_w'), B, dims[1], dims[0], want_f32=True, drop=Drop(p0, seed, 211))
        else:
            D0, has_ln = self.pre
            if has_ln:
                dh, _ = K.linear_dx(d, W.s('l0_w'), B, dims[1], dims[0], want_f32=True)
                _, dfb, _, _ = K.layernorm_bwd(dh, S['f'], S['m'], S['r'], W.p('ln.w'), B, dims[0], want_f32=False, want_bf16=True,
                                               drop=Drop(p0, seed, 211), drop_mode=2, dgamma=G['ln.w'], dbeta=G['ln.b'],
                                               dx_colsum=G['proj_b'], defer=True)
            else:
                _, dfb = K.linear_dx(d, W.s('l0_w'), B, dims[1], dims[0], want_bf16=True, drop=Drop(p0, seed, 211), colsum=G['proj_b'])
            K.linear_dw(dfb, S['xb'], B, dims[0], D0, out=G['proj_w'])
            if need_dx:
                dx, _ = K.linear_dx(dfb, W.s('proj_w'), B, dims[0], D0, want_f32=True)
            K.ln_reduce_flush()
        K.wgrad_join()
        return G, dx
