"""Expert runners: hand-scheduled forward AND backward of the MoE experts at ONE token per row -- the shape the VQA model
feeds them (the fused vector, reference vqa_model.py:674 -> moe_layer.py:151-168 with S = 1).

Why runners.  Issued op by op (hip/ops.py: cast -> GEMM -> cast -> attention -> ... each with its own autograd node) the four
experts of the 4-expert configuration cost ~460 launches per training step, every one of them a 5 - 8 us latency-floor launch
(profiles/r02/skinny_gemm.md): the MoE's 4.5 ms were launch count, not bytes.  A runner is ONE autograd node per expert whose
forward / backward are straight launch sequences with everything the kernels can fuse: bias + GELU + dropout + residual in
the GEMM epilogues, bf16 operand copies written by the producing kernel (no cast launches), bias gradients as fused column
sums, all weight gradients of the step in one grouped launch, one zero-fill per expert for the accumulated gradient slots.

Attention over ONE key.  Every attention site of these experts except the Segmentation decoder's 4-token self-attention
sees a single key per sample at S = 1 (Vision / Text self-attention over the one token; the decoder's cross-attention over
the one memory row): softmax of one score is 1, so the attention output is out_proj(keep * V) with `keep` the dropout scale
torch applies to the (sample, head, query) probability.  Q and K are never computed; their rows of in_proj_weight /
in_proj_bias get the exact-zero gradient the reference gives them (softmax'(single score) == 0) from the arena's zero fill.

Same conventions as hip/blocks.py: fp32 residual stream, bf16 GEMM operands, fp32 gradients in a GradArena; ``forward`` returns
(output, saved), ``backward`` returns (G, dx).
"""

import torch

from . import kernels as K
from .blocks import GradArena, new_seed
from .kernels import ACT_GELU, Drop, NO_DROP

F32 = torch.float32


def _one_key_attention_fwd(W, pre, kvb, T, R, Hd, heads, pa, seed, st, residual, out_drop):
    """nn.MultiheadAttention(query rows: R per sample, key = value = the sample's one row ``kvb`` [T, Hd] bf16) + residual:
    s [T*R, Hd] fp32 = residual + drop(out_proj(keep * (kv Wv^T + bv))).  Returns (s, saved)."""
    w_in, b_in = W.s(pre + 'in_wz'), W.p(pre + 'in_bz')
    _, v, _ = K.linear_fwd(kvb, w_in[2 * Hd:], b_in[2 * Hd:], T, Hd, Hd, want_bf16=True)
    vd = K.head_keep_fwd(v, T, R, heads, Hd // heads, Drop(pa, seed, st)) if (pa > 0 or R > 1) else v
    s, _, _ = K.linear_fwd(vd, W.s(pre + 'out_w'), W.p(pre + 'out_b'), T * R, Hd, Hd, want_f32=True, residual=residual, drop=out_drop)
    return s, (kvb, vd)


def _one_key_attention_bwd(W, G, pre, saved, dsb, T, R, Hd, heads, pa, seed, st, residual):
    """``dsb`` [T*R, Hd] bf16: gradient of the out-projection's output (already masked by its dropout; the caller has its column
    sums in G[pre + 'out_b']).  Returns the fp32 gradient of the key/value row [T, Hd] + ``residual``."""
    kvb, vd = saved
    w_in = W.s(pre + 'in_wz')
    bz = G[pre + 'in_bz'][2 * Hd:]
    K.linear_dw(dsb, vd, T * R, Hd, Hd, out=G[pre + 'out_w'])
    if pa > 0 or R > 1:
        _, dvd = K.linear_dx(dsb, W.s(pre + 'out_w'), T * R, Hd, Hd, want_bf16=True)
        dv = K.head_keep_bwd(dvd, T, R, heads, Hd // heads, Drop(pa, seed, st))
        K.colsum_bf16(dv, T, Hd, out=bz)
    else:
        _, dv = K.linear_dx(dsb, W.s(pre + 'out_w'), T, Hd, Hd, want_bf16=True, colsum=bz)
    K.linear_dw(dv, kvb, T, Hd, Hd, out=G[pre + 'in_wz'][2 * Hd:], prezeroed=True)
    dkv, _ = K.linear_dx(dv, w_in[2 * Hd:], T, Hd, Hd, want_f32=True, residual=residual)
    return dkv


class _Runner:
    def __init__(self, W, din, hidden, dout, heads, dropout, eps=1e-5):
        self.W, self.Din, self.Hd, self.Dout, self.heads, self.pd, self.eps = W, din, hidden, dout, heads, dropout, eps
        self.arena = GradArena(W.params)

    # common tail: out = LN(h_final W_out^T + b_out)
    def _out_fwd(self, hfb, T, S):
        W = self.W
        o, _, _ = K.linear_fwd(hfb, W.s('out_w'), W.p('out_b'), T, self.Dout, self.Hd, want_f32=True)
        out, _, mo, ro = K.layernorm_fwd(o, W.p('on.w'), W.p('on.b'), T, self.Dout, eps=self.eps)
        S.update(hfb=hfb, o=o, mo=mo, ro=ro)
        return out

    def _out_bwd(self, S, G, dout, T):
        W = self.W
        _, dob, _, _ = K.layernorm_bwd(dout, S['o'], S['mo'], S['ro'], W.p('on.w'), T, self.Dout, want_f32=False, want_bf16=True,
                                       dgamma=G['on.w'], dbeta=G['on.b'], dx_colsum=G['out_b'], defer=True)
        K.linear_dw(dob, S['hfb'], T, self.Dout, self.Hd, out=G['out_w'])
        return dob

    # common head: h = x W_in^T + b_in
    def _in_fwd(self, x, S):
        W = self.W
        T = x.shape[0]
        xb = K.cast_bf16(x.contiguous().float())
        h, hb, _ = K.linear_fwd(xb, W.s('in_w'), W.p('in_b'), T, self.Hd, self.Din, want_f32=True, want_bf16=True)
        S.update(T=T, xb=xb)
        return h, hb

    def _in_bwd(self, S, G, dh, T, need_dx):
        W = self.W
        dhb = K.rows_mask_cast(dh, T, self.Hd, colsum=G['in_b'])
        K.linear_dw(dhb, S['xb'], T, self.Hd, self.Din, out=G['in_w'])
        dx = None
        if need_dx:
            dx, _ = K.linear_dx(dhb, W.s('in_w'), T, self.Hd, self.Din, want_f32=True)
        K.ln_reduce_flush()
        K.wgrad_join()
        return dx

    def _begin(self, training):
        seed = new_seed() if training else 0
        return seed, (self.pd if training else 0.0)

    def _dout(self, dout, T):
        return dout.reshape(T, self.Dout).contiguous().float()


class VisionExpertRunner(_Runner):
    """Reference expert_types.py:159-199 (VisionExpert.forward) at S = 1, no spatial positions, no mask."""

    def __init__(self, W, din, hidden, dout, heads, dropout, attention=True):
        super().__init__(W, din, hidden, dout, heads, dropout)
        self.attn = attention

    def forward(self, x, training):
        W, Hd = self.W, self.Hd
        seed, pd = self._begin(training)
        S = dict(seed=seed, pd=pd)
        h, hb = self._in_fwd(x, S)
        T = S['T']
        if self.attn:
            s1, S['att'] = _one_key_attention_fwd(W, 'sa_', hb, T, 1, Hd, self.heads, pd, seed, 1, h, NO_DROP)
            h1, h1b, m1, r1 = K.layernorm_fwd(s1, W.p('sn.w'), W.p('sn.b'), T, Hd, want_bf16=True, eps=self.eps)
            S.update(s1=s1, m1=m1, r1=r1)
        else:
            h1, h1b = h, hb
        _, g, a = K.linear_fwd(h1b, W.s('t0_w'), W.p('t0_b'), T, Hd, Hd, want_bf16=True, want_pre=True, act=ACT_GELU, drop=Drop(pd, seed, 111))
        _, h2b, _ = K.linear_fwd(g, W.s('t3_w'), W.p('t3_b'), T, Hd, Hd, want_bf16=True, residual=h1, drop=Drop(pd, seed, 112))
        S.update(h1b=h1b, g=g, a=a)
        return self._out_fwd(h2b, T, S), S

    def backward(self, S, dout, need_dx=True):
        W, Hd, T, seed, pd = self.W, self.Hd, S['T'], S['seed'], S['pd']
        dout = self._dout(dout, T)
        _, G = self.arena.alloc(dout.device)
        dob = self._out_bwd(S, G, dout, T)
        # h2 = h1 + drop(t3(g)):  dh2 goes to h1 unmasked, to t3 masked
        if pd > 0:
            dh2, _ = K.linear_dx(dob, W.s('out_w'), T, self.Dout, Hd, want_f32=True)
            dt3 = K.rows_mask_cast(dh2, T, Hd, drop=Drop(pd, seed, 112), colsum=G['t3_b'])
        else:
            dh2, dt3 = K.linear_dx(dob, W.s('out_w'), T, self.Dout, Hd, want_f32=True, want_bf16=True, colsum=G['t3_b'])
        K.linear_dw(dt3, S['g'], T, Hd, Hd, out=G['t3_w'])
        _, da = K.linear_dx(dt3, W.s('t3_w'), T, Hd, Hd, want_bf16=True, act_grad_of=S['a'], act_bwd=ACT_GELU, drop=Drop(pd, seed, 111), colsum=G['t0_b'])
        K.linear_dw(da, S['h1b'], T, Hd, Hd, out=G['t0_w'])
        dh1, _ = K.linear_dx(da, W.s('t0_w'), T, Hd, Hd, want_f32=True, residual=dh2)
        if self.attn:
            ds1, ds1b, _, _ = K.layernorm_bwd(dh1, S['s1'], S['m1'], S['r1'], W.p('sn.w'), T, Hd, want_bf16=True,
                                              dgamma=G['sn.w'], dbeta=G['sn.b'], dx_colsum=G['sa_out_b'], defer=True)
            dh = _one_key_attention_bwd(W, G, 'sa_', S['att'], ds1b, T, 1, Hd, self.heads, pd, seed, 1, ds1)
        else:
            dh = dh1
        return G, self._in_bwd(S, G, dh, T, need_dx)


class TextExpertRunner(_Runner):
    """Reference expert_types.py:270-312 (TextExpert.forward) at S = 1, no mask.  ``attention=False, mid='transform'``: the
    MultimodalExpert without context (expert_types.py:395-445: input_proj -> transform + LayerNorm -> output)."""

    def __init__(self, W, din, hidden, dout, heads, dropout, attention=True, stream0=121):
        super().__init__(W, din, hidden, dout, heads, dropout)
        self.attn, self.st0 = attention, stream0

    def forward(self, x, training):
        W, Hd = self.W, self.Hd
        seed, pd = self._begin(training)
        S = dict(seed=seed, pd=pd)
        h, hb = self._in_fwd(x, S)
        T, F = S['T'], 2 * Hd
        if self.attn:
            s1, S['att'] = _one_key_attention_fwd(W, 'sa_', hb, T, 1, Hd, self.heads, pd, seed, 1, h, NO_DROP)
            h1, h1b, m1, r1 = K.layernorm_fwd(s1, W.p('an.w'), W.p('an.b'), T, Hd, want_bf16=True, eps=self.eps)
            S.update(s1=s1, m1=m1, r1=r1)
        else:
            h1, h1b = h, hb
        _, g, a = K.linear_fwd(h1b, W.s('f0_w'), W.p('f0_b'), T, F, Hd, want_bf16=True, want_pre=True, act=ACT_GELU, drop=Drop(pd, seed, self.st0))
        s2, _, _ = K.linear_fwd(g, W.s('f3_w'), W.p('f3_b'), T, Hd, F, want_f32=True, residual=h1, drop=Drop(pd, seed, self.st0 + 1))
        _, h2b, m2, r2 = K.layernorm_fwd(s2, W.p('fn.w'), W.p('fn.b'), T, Hd, want_f32=False, want_bf16=True, eps=self.eps)
        S.update(h1b=h1b, g=g, a=a, s2=s2, m2=m2, r2=r2)
        return self._out_fwd(h2b, T, S), S

    def backward(self, S, dout, need_dx=True):
        W, Hd, T, seed, pd = self.W, self.Hd, S['T'], S['seed'], S['pd']
        F = 2 * Hd
        dout = self._dout(dout, T)
        _, G = self.arena.alloc(dout.device)
        dob = self._out_bwd(S, G, dout, T)
        dh2, _ = K.linear_dx(dob, W.s('out_w'), T, self.Dout, Hd, want_f32=True)
        ds2, ds2b, _, _ = K.layernorm_bwd(dh2, S['s2'], S['m2'], S['r2'], W.p('fn.w'), T, Hd, want_bf16=True, drop=Drop(pd, seed, self.st0 + 1),
                                          drop_mode=1, dgamma=G['fn.w'], dbeta=G['fn.b'], dx_colsum=G['f3_b'], defer=True)
        K.linear_dw(ds2b, S['g'], T, Hd, F, out=G['f3_w'])
        _, da = K.linear_dx(ds2b, W.s('f3_w'), T, Hd, F, want_bf16=True, act_grad_of=S['a'], act_bwd=ACT_GELU, drop=Drop(pd, seed, self.st0),
                            colsum=G['f0_b'])
        K.linear_dw(da, S['h1b'], T, F, Hd, out=G['f0_w'])
        dh1, _ = K.linear_dx(da, W.s('f0_w'), T, F, Hd, want_f32=True, residual=ds2)
        if self.attn:
            ds1, ds1b, _, _ = K.layernorm_bwd(dh1, S['s1'], S['m1'], S['r1'], W.p('an.w'), T, Hd, want_bf16=True,
                                              dgamma=G['an.w'], dbeta=G['an.b'], dx_colsum=G['sa_out_b'], defer=True)
            dh = _one_key_attention_bwd(W, G, 'sa_', S['att'], ds1b, T, 1, Hd, self.heads, pd, seed, 1, ds1)
        else:
            dh = dh1
        return G, self._in_bwd(S, G, dh, T, need_dx)


class SegmentationExpertRunner(_Runner):
    """Reference specialized_experts.py:119-173 (SegmentationExpert.forward) at S = 1: a 2-layer post-LN TransformerDecoder
    (4 mask tokens per sample attend to each other, then to the sample's one projected row), the Conv1d(k=3, pad=1) boundary
    stack -- whose outer taps only ever see the zero padding of a length-1 sequence, so only the centre tap multiplies -- and
    the spatial MLP over [boundary | mean mask feature]."""

    def __init__(self, W, din, hidden, dout, heads, dropout, num_mask_tokens, num_layers):
        super().__init__(W, din, hidden, dout, heads, dropout)
        self.R, self.L = num_mask_tokens, num_layers

    def _centre_tap(self, key):
        Hd = self.Hd
        return K.take_stride(self.W.s(key).reshape(-1), Hd * Hd, 3, 1).view(Hd, Hd)

    def forward(self, x, training):
        W, Hd, R, heads = self.W, self.Hd, self.R, self.heads
        seed, pd = self._begin(training)
        S = dict(seed=seed, pd=pd, layers=[])
        h, hb = self._in_fwd(x, S)
        T, F, Dh = S['T'], 2 * Hd, Hd // heads
        Mr = T * R
        tgt, tgtb = K.repeat_rows(W.p('mask_tokens').reshape(R, Hd), Mr, Hd, R, 1, want_bf16=True)
        for l in range(self.L):
            k, st = f'd{l}.', 16 * (l + 1)
            _, qkv, _ = K.linear_fwd(tgtb, W.s(k + 'sa_in_w'), W.p(k + 'sa_in_b'), Mr, 3 * Hd, Hd, want_bf16=True)
            ctx = K.attention_fwd(qkv[:, :Hd], qkv[:, Hd:2 * Hd], qkv[:, 2 * Hd:], 3 * Hd, 3 * Hd, 3 * Hd, T, heads, R, R, Dh, None, Drop(pd, seed, st))
            s1, _, _ = K.linear_fwd(ctx, W.s(k + 'sa_out_w'), W.p(k + 'sa_out_b'), Mr, Hd, Hd, want_f32=True, residual=tgt, drop=Drop(pd, seed, st + 1))
            x1, _, m1, r1 = K.layernorm_fwd(s1, W.p(k + 'n1.w'), W.p(k + 'n1.b'), Mr, Hd, eps=self.eps)
            s2, att = _one_key_attention_fwd(W, k + 'ca_', hb, T, R, Hd, heads, pd, seed, st + 2, x1, Drop(pd, seed, st + 3))
            x2, x2b, m2, r2 = K.layernorm_fwd(s2, W.p(k + 'n2.w'), W.p(k + 'n2.b'), Mr, Hd, want_bf16=True, eps=self.eps)
            _, g, a = K.linear_fwd(x2b, W.s(k + 'l1_w'), W.p(k + 'l1_b'), Mr, F, Hd, want_bf16=True, want_pre=True, act=ACT_GELU, drop=Drop(pd, seed, st + 4))
            s3, _, _ = K.linear_fwd(g, W.s(k + 'l2_w'), W.p(k + 'l2_b'), Mr, Hd, F, want_f32=True, residual=x2, drop=Drop(pd, seed, st + 5))
            last = l == self.L - 1
            x3, x3b, m3, r3 = K.layernorm_fwd(s3, W.p(k + 'n3.w'), W.p(k + 'n3.b'), Mr, Hd, want_bf16=not last, eps=self.eps)
            S['layers'].append(dict(tgtb=tgtb, qkv=qkv, ctx=ctx, s1=s1, m1=m1, r1=r1, att=att, s2=s2, m2=m2, r2=r2, x2b=x2b, g=g, a=a, s3=s3, m3=m3, r3=r3))
            tgt, tgtb = x3, x3b
        # boundary stack on the centre taps; spatial MLP over [boundary | mean of the mask features]
        c0, c2 = self._centre_tap('c0_k'), self._centre_tap('c2_k')
        _, bf1, a1 = K.linear_fwd(hb, c0, W.p('c0_b'), T, Hd, Hd, want_bf16=True, want_pre=True, act=ACT_GELU)
        sp_in = torch.empty((T, 2 * Hd), dtype=K.HALF(), device=x.device)
        bf2 = torch.empty((T, Hd), dtype=F32, device=x.device)
        a2 = torch.empty((T, Hd), dtype=K.HALF(), device=x.device)
        K.gemm(bf1, c2, T, Hd, Hd, Hd, Hd, True, True, out_f32=bf2, out_bf16=sp_in, ldc_bf16=2 * Hd, pre_bf16=a2, bias=W.p('c2_b'), act=ACT_GELU)
        K.rows_mean(tgt, R, T, Hd, out_bf16=sp_in[:, Hd:], ld_out=2 * Hd)
        _, u, au = K.linear_fwd(sp_in, W.s('m0_w'), W.p('m0_b'), T, Hd, 2 * Hd, want_bf16=True, want_pre=True, act=ACT_GELU, drop=Drop(pd, seed, 151))
        hs, _, _ = K.linear_fwd(u, W.s('m3_w'), W.p('m3_b'), T, Hd, Hd, want_f32=True, residual=h)
        _, hfb = K.add_f32(hs, bf2, want_f32=False, want_bf16=True)
        S.update(hb=hb, c0=c0, c2=c2, bf1=bf1, a1=a1, a2=a2, sp_in=sp_in, u=u, au=au)
        return self._out_fwd(hfb, T, S), S

    def backward(self, S, dout, need_dx=True):
        W, Hd, R, heads, T, seed, pd = self.W, self.Hd, self.R, self.heads, S['T'], S['seed'], S['pd']
        F, Dh, Mr = 2 * Hd, Hd // heads, T * R
        dout = self._dout(dout, T)
        dev = dout.device
        _, G = self.arena.alloc(dev)
        dob = self._out_bwd(S, G, dout, T)
        # hf = h + bf2 + m3(u):  one gradient for the three of them
        dhf, dhfb = K.linear_dx(dob, W.s('out_w'), T, self.Dout, Hd, want_f32=True, want_bf16=True, colsum=G['m3_b'])
        K.linear_dw(dhfb, S['u'], T, Hd, Hd, out=G['m3_w'])
        _, du = K.linear_dx(dhfb, W.s('m3_w'), T, Hd, Hd, want_bf16=True, act_grad_of=S['au'], act_bwd=ACT_GELU, drop=Drop(pd, seed, 151), colsum=G['m0_b'])
        K.linear_dw(du, S['sp_in'], T, Hd, 2 * Hd, out=G['m0_w'])
        m0 = W.s('m0_w')
        dbf2, _ = K.linear_dx(du, m0, T, Hd, Hd, want_f32=True, residual=dhf, ldw=2 * Hd)               # boundary half (+ its direct path)
        dpool, _ = K.linear_dx(du, m0[:, Hd:], T, Hd, Hd, want_f32=True, ldw=2 * Hd)                       # mean-mask-feature half
        d2 = K.rows_mask_cast(dbf2, T, Hd, pre=S['a2'], act=ACT_GELU, colsum=G['c2_b'])
        K.scatter_stride(K.linear_dw(d2, S['bf1'], T, Hd, Hd), G['c2_k'], Hd * Hd, 3, 1)
        _, d1 = K.linear_dx(d2, S['c2'], T, Hd, Hd, want_bf16=True, act_grad_of=S['a1'], act_bwd=ACT_GELU, colsum=G['c0_b'])
        K.scatter_stride(K.linear_dw(d1, S['hb'], T, Hd, Hd), G['c0_k'], Hd * Hd, 3, 1)
        dh, _ = K.linear_dx(d1, S['c0'], T, Hd, Hd, want_f32=True, residual=dhf)
        # mask-token path: mean over the R tokens, then the decoder layers in reverse
        dtgt, _ = K.repeat_rows(dpool, Mr, Hd, R, 0, alpha=1.0 / R)
        for l in reversed(range(self.L)):
            k, st, Y = f'd{l}.', 16 * (l + 1), S['layers'][l]
            ds3, ds3b, _, _ = K.layernorm_bwd(dtgt, Y['s3'], Y['m3'], Y['r3'], W.p(k + 'n3.w'), Mr, Hd, want_bf16=True, drop=Drop(pd, seed, st + 5), drop_mode=1,
                                              dgamma=G[k + 'n3.w'], dbeta=G[k + 'n3.b'], dx_colsum=G[k + 'l2_b'], defer=True)
            K.linear_dw(ds3b, Y['g'], Mr, Hd, F, out=G[k + 'l2_w'])
            _, da = K.linear_dx(ds3b, W.s(k + 'l2_w'), Mr, Hd, F, want_bf16=True, act_grad_of=Y['a'], act_bwd=ACT_GELU, drop=Drop(pd, seed, st + 4),
                                colsum=G[k + 'l1_b'])
            K.linear_dw(da, Y['x2b'], Mr, F, Hd, out=G[k + 'l1_w'])
            dx2, _ = K.linear_dx(da, W.s(k + 'l1_w'), Mr, F, Hd, want_f32=True, residual=ds3)
            ds2, ds2b, _, _ = K.layernorm_bwd(dx2, Y['s2'], Y['m2'], Y['r2'], W.p(k + 'n2.w'), Mr, Hd, want_bf16=True, drop=Drop(pd, seed, st + 3), drop_mode=1,
                                              dgamma=G[k + 'n2.w'], dbeta=G[k + 'n2.b'], dx_colsum=G[k + 'ca_out_b'], defer=True)
            dh = _one_key_attention_bwd(W, G, k + 'ca_', Y['att'], ds2b, T, R, Hd, heads, pd, seed, st + 2, dh)
            ds1, ds1b, _, _ = K.layernorm_bwd(ds2, Y['s1'], Y['m1'], Y['r1'], W.p(k + 'n1.w'), Mr, Hd, want_bf16=True, drop=Drop(pd, seed, st + 1), drop_mode=1,
                                              dgamma=G[k + 'n1.w'], dbeta=G[k + 'n1.b'], dx_colsum=G[k + 'sa_out_b'], defer=True)
            K.linear_dw(ds1b, Y['ctx'], Mr, Hd, Hd, out=G[k + 'sa_out_w'])
            _, dctx = K.linear_dx(ds1b, W.s(k + 'sa_out_w'), Mr, Hd, Hd, want_bf16=True)
            qkv = Y['qkv']
            dqkv = torch.empty((Mr, 3 * Hd), dtype=K.HALF(), device=dev)
            qb = G[k + 'sa_in_b']
            K.attention_bwd(qkv[:, :Hd], qkv[:, Hd:2 * Hd], qkv[:, 2 * Hd:], dctx, 3 * Hd, 3 * Hd, 3 * Hd, T, heads, R, R, Dh,
                            dqkv[:, :Hd], dqkv[:, Hd:2 * Hd], dqkv[:, 2 * Hd:], 3 * Hd, 3 * Hd, 3 * Hd, None, Drop(pd, seed, st),
                            dq_colsum=qb[:Hd], dk_colsum=qb[Hd:2 * Hd], dv_colsum=qb[2 * Hd:])
            K.linear_dw(dqkv, Y['tgtb'], Mr, 3 * Hd, Hd, out=G[k + 'sa_in_w'])
            dtgt, _ = K.linear_dx(dqkv, W.s(k + 'sa_in_w'), Mr, 3 * Hd, Hd, want_f32=True, residual=ds1)
        K.colsum_f32(dtgt, T, R * Hd, out=G['mask_tokens'].view(-1))           # the R learned tokens were tiled over the samples
        return G, self._in_bwd(S, G, dh, T, need_dx)
