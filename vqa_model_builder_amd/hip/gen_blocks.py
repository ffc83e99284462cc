"""Layer runners of the generative model: hand-scheduled forward AND backward of torch's pre-LN
``nn.TransformerEncoderLayer`` / ``nn.TransformerDecoderLayer`` (activation='gelu', batch_first, norm_first) as the reference's
CrossModalFusion and TransformerDecoder build them (generative_vqa_model.py:207-214, 372-381).

Issued op by op (hip/ops.py) one decoder layer was ~35 launches forward and ~75 backward -- a cast in front of every GEMM, dropout
and residual add as launches of their own, bias gradients as separate column sums, every weight gradient its own GEMM launch
(profiles/r02/generative.md).  Here: LayerNorm emits the bf16 GEMM operand, bias + GELU + dropout + residual ride in the GEMM
epilogues, bias gradients are fused column sums, weight gradients join the grouped launch.

Same conventions as hip/blocks.py (fp32 residual stream, bf16 operands, GradArena gradients, one seed per forward with a stream id
per dropout site)."""

import torch

from . import kernels as K
from .blocks import GradArena, _mask_u8, _q_kv_attention, new_seed
from .kernels import ACT_GELU, Drop

F32 = torch.float32


def _self_attention_fwd(W, hb, B, H, S, D, mask_u8, drop, causal):
    """(qkv bf16 [B*S, 3D], context bf16 [B*S, D]); one launch where the fused kernel covers the shape."""
    M = B * S
    if K.FUSED_ATTENTION_FUSION and K.fused_attention_covers(D, H, S, S):
        qkv = torch.empty((M, 3 * D), dtype=K.HALF(), device=hb.device)
        ctx = K.fused_inproj_attention_fwd(hb, hb, W.s('sa_in_w'), W.p('sa_in_b'), B, H, S, S, D, mask_u8, drop, q=qkv[:, :D], k=qkv[:, D:2 * D],
                                           v=qkv[:, 2 * D:], ldq=3 * D, ldk=3 * D, ldv=3 * D, causal=causal)
        return qkv, ctx
    _, qkv, _ = K.linear_fwd(hb, W.s('sa_in_w'), W.p('sa_in_b'), M, 3 * D, D, want_bf16=True)
    ctx = K.attention_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], 3 * D, 3 * D, 3 * D, B, H, S, S, D // H, mask_u8, drop, causal=causal)
    return qkv, ctx


def _self_attention_bwd(W, G, S_, dctx, B, H, S, D, mask_u8, drop, causal, residual):
    """Gradient of the LayerNorm output that fed the packed in-projection (+ ``residual`` fp32), weight gradient queued."""
    M, dev = B * S, dctx.device
    qkv = S_['qkv']
    dqkv = torch.empty((M, 3 * D), dtype=K.HALF(), device=dev)
    gb = G['sa_in_b']
    K.attention_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], dctx, 3 * D, 3 * D, 3 * D, B, H, S, S, D // H,
                    dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:], 3 * D, 3 * D, 3 * D, mask_u8, drop,
                    dq_colsum=gb[:D], dk_colsum=gb[D:2 * D], dv_colsum=gb[2 * D:], causal=causal)
    K.linear_dw(dqkv, S_['h1b'], M, 3 * D, D, out=G['sa_in_w'])
    dh, _ = K.linear_dx(dqkv, W.s('sa_in_w'), M, 3 * D, D, want_f32=True, residual=residual)
    return dh


def _ffn_fwd(W, hb, x, M, D, F, p, seed, st):
    _, g, a = K.linear_fwd(hb, W.s('l1_w'), W.p('l1_b'), M, F, D, want_bf16=True, want_pre=True, act=ACT_GELU, drop=Drop(p, seed, st))
    y, _, _ = K.linear_fwd(g, W.s('l2_w'), W.p('l2_b'), M, D, F, want_f32=True, residual=x, drop=Drop(p, seed, st + 1))
    return y, g, a


def _ffn_bwd(W, G, dy, hb, g, a, M, D, F, p, seed, st):
    """dy fp32 = gradient of x + drop(l2(gelu-drop(l1(h)))): returns the fp32 gradient of h (the LayerNorm output)."""
    dyb = K.rows_mask_cast(dy, M, D, drop=Drop(p, seed, st + 1), colsum=G['l2_b'])
    K.linear_dw(dyb, g, M, D, F, out=G['l2_w'])
    _, da = K.linear_dx(dyb, W.s('l2_w'), M, D, F, want_bf16=True, act_grad_of=a, act_bwd=ACT_GELU, drop=Drop(p, seed, st), colsum=G['l1_b'])
    K.linear_dw(da, hb, M, F, D, out=G['l1_w'])
    dh, _ = K.linear_dx(da, W.s('l1_w'), M, F, D, want_f32=True)
    return dh


class PreLNEncoderLayerRunner:
    """x1 = x + drop(SA(LN1 x));  y = x1 + drop(l2(drop(gelu(l1(LN2 x1)))))        torch: TransformerEncoderLayer._sa_block / _ff_block"""

    def __init__(self, W, D, heads, ffn, dropout, eps=1e-5):
        self.W, self.D, self.H, self.F, self.p, self.eps = W, D, heads, ffn, dropout, eps
        self.arena = GradArena(W.params)

    def forward(self, x, key_padding_mask, training):
        W, D, H, F = self.W, self.D, self.H, self.F
        B, S, _ = x.shape
        M = B * S
        seed = new_seed() if training else 0
        p = self.p if training else 0.0
        x = x.reshape(M, D).contiguous().float()
        km = _mask_u8(key_padding_mask)
        _, h1b, m1, r1 = K.layernorm_fwd(x, W.p('n1.w'), W.p('n1.b'), M, D, want_f32=False, want_bf16=True, eps=self.eps)
        qkv, ctx = _self_attention_fwd(W, h1b, B, H, S, D, km, Drop(p, seed, 1), False)
        x1, _, _ = K.linear_fwd(ctx, W.s('sa_out_w'), W.p('sa_out_b'), M, D, D, want_f32=True, residual=x, drop=Drop(p, seed, 2))
        _, h2b, m2, r2 = K.layernorm_fwd(x1, W.p('n2.w'), W.p('n2.b'), M, D, want_f32=False, want_bf16=True, eps=self.eps)
        y, g, a = _ffn_fwd(W, h2b, x1, M, D, F, p, seed, 3)
        saved = dict(B=B, S=S, seed=seed, p=p, km=km, x=x, m1=m1, r1=r1, h1b=h1b, qkv=qkv, ctx=ctx, x1=x1, m2=m2, r2=r2, h2b=h2b, g=g, a=a)
        return y.view(B, S, D), saved

    def backward(self, S_, dout):
        W, D, H, F = self.W, self.D, self.H, self.F
        B, S, seed, p = S_['B'], S_['S'], S_['seed'], S_['p']
        M = B * S
        dy = dout.reshape(M, D).contiguous().float()
        _, G = self.arena.alloc(dy.device)
        dh2 = _ffn_bwd(W, G, dy, S_['h2b'], S_['g'], S_['a'], M, D, F, p, seed, 3)
        dx1, _, _, _ = K.layernorm_bwd(dh2, S_['x1'], S_['m2'], S_['r2'], W.p('n2.w'), M, D, dres=dy, dgamma=G['n2.w'], dbeta=G['n2.b'], defer=True)
        dob = K.rows_mask_cast(dx1, M, D, drop=Drop(p, seed, 2), colsum=G['sa_out_b'])
        K.linear_dw(dob, S_['ctx'], M, D, D, out=G['sa_out_w'])
        _, dctx = K.linear_dx(dob, W.s('sa_out_w'), M, D, D, want_bf16=True)
        dh1 = _self_attention_bwd(W, G, S_, dctx, B, H, S, D, S_['km'], Drop(p, seed, 1), False, None)
        dx, _, _, _ = K.layernorm_bwd(dh1, S_['x'], S_['m1'], S_['r1'], W.p('n1.w'), M, D, dres=dx1, dgamma=G['n1.w'], dbeta=G['n1.b'], defer=True)
        K.ln_reduce_flush()
        K.wgrad_join()
        return G, dx.view(B, S, D)


class PreLNDecoderLayerRunner:
    """x1 = t + drop(causal SA(LN1 t));  x2 = x1 + drop(CA(LN2 x1, memory));  y = x2 + drop(FFN(LN3 x2))
    torch: TransformerDecoderLayer._sa_block / _mha_block / _ff_block with tgt_is_causal."""

    def __init__(self, W, D, heads, ffn, dropout, eps=1e-5):
        self.W, self.D, self.H, self.F, self.p, self.eps = W, D, heads, ffn, dropout, eps
        self.arena = GradArena(W.params)

    def forward(self, tgt, memory, tgt_mask, mem_mask, training):
        W, D, H, F = self.W, self.D, self.H, self.F
        B, A, _ = tgt.shape
        Sm = memory.shape[1]
        M, Mm = B * A, B * Sm
        seed = new_seed() if training else 0
        p = self.p if training else 0.0
        x = tgt.reshape(M, D).contiguous().float()
        memb = K.cast_bf16(memory.reshape(Mm, D).contiguous().float())
        tm, mm = _mask_u8(tgt_mask), _mask_u8(mem_mask)
        _, h1b, m1, r1 = K.layernorm_fwd(x, W.p('n1.w'), W.p('n1.b'), M, D, want_f32=False, want_bf16=True, eps=self.eps)
        qkv, ctx = _self_attention_fwd(W, h1b, B, H, A, D, tm, Drop(p, seed, 1), True)
        x1, _, _ = K.linear_fwd(ctx, W.s('sa_out_w'), W.p('sa_out_b'), M, D, D, want_f32=True, residual=x, drop=Drop(p, seed, 2))
        _, h2b, m2, r2 = K.layernorm_fwd(x1, W.p('n2.w'), W.p('n2.b'), M, D, want_f32=False, want_bf16=True, eps=self.eps)
        q2, kv2, ctx2 = _q_kv_attention(h2b, memb, W.s('ca_in_w'), W.p('ca_in_b'), B, H, A, Sm, D, mm, Drop(p, seed, 3))
        x2, _, _ = K.linear_fwd(ctx2, W.s('ca_out_w'), W.p('ca_out_b'), M, D, D, want_f32=True, residual=x1, drop=Drop(p, seed, 4))
        _, h3b, m3, r3 = K.layernorm_fwd(x2, W.p('n3.w'), W.p('n3.b'), M, D, want_f32=False, want_bf16=True, eps=self.eps)
        y, g, a = _ffn_fwd(W, h3b, x2, M, D, F, p, seed, 5)
        saved = dict(B=B, A=A, Sm=Sm, seed=seed, p=p, tm=tm, mm=mm, x=x, memb=memb, m1=m1, r1=r1, h1b=h1b, qkv=qkv, ctx=ctx, x1=x1, m2=m2, r2=r2,
                     h2b=h2b, q2=q2, kv2=kv2, ctx2=ctx2, x2=x2, m3=m3, r3=r3, h3b=h3b, g=g, a=a)
        return y.view(B, A, D), saved

    def backward(self, S_, dout, need_dmem=True):
        W, D, H, F = self.W, self.D, self.H, self.F
        B, A, Sm, seed, p = S_['B'], S_['A'], S_['Sm'], S_['seed'], S_['p']
        M, Mm, Dh = B * A, B * Sm, D // H
        dy = dout.reshape(M, D).contiguous().float()
        dev = dy.device
        _, G = self.arena.alloc(dev)
        dh3 = _ffn_bwd(W, G, dy, S_['h3b'], S_['g'], S_['a'], M, D, F, p, seed, 5)
        dx2, _, _, _ = K.layernorm_bwd(dh3, S_['x2'], S_['m3'], S_['r3'], W.p('n3.w'), M, D, dres=dy, dgamma=G['n3.w'], dbeta=G['n3.b'], defer=True)
        # --- cross attention over the encoder memory
        dob2 = K.rows_mask_cast(dx2, M, D, drop=Drop(p, seed, 4), colsum=G['ca_out_b'])
        K.linear_dw(dob2, S_['ctx2'], M, D, D, out=G['ca_out_w'])
        _, dctx2 = K.linear_dx(dob2, W.s('ca_out_w'), M, D, D, want_bf16=True)
        dq2 = torch.empty((M, D), dtype=K.HALF(), device=dev)
        dkv2 = torch.empty((Mm, 2 * D), dtype=K.HALF(), device=dev)
        kv2, gb = S_['kv2'], G['ca_in_b']
        K.attention_bwd(S_['q2'], kv2[:, :D], kv2[:, D:], dctx2, D, 2 * D, 2 * D, B, H, A, Sm, Dh, dq2, dkv2[:, :D], dkv2[:, D:], D, 2 * D, 2 * D,
                        S_['mm'], Drop(p, seed, 3), dq_colsum=gb[:D], dk_colsum=gb[D:2 * D], dv_colsum=gb[2 * D:])
        w_in = W.s('ca_in_w')
        K.linear_dw(dq2, S_['h2b'], M, D, D, out=G['ca_in_w'][:D])
        K.linear_dw(dkv2, S_['memb'], Mm, 2 * D, D, out=G['ca_in_w'][D:])
        dh2, _ = K.linear_dx(dq2, w_in[:D], M, D, D, want_f32=True)
        dmem = None
        if need_dmem:
            dmem, _ = K.linear_dx(dkv2, w_in[D:], Mm, 2 * D, D, want_f32=True)
            dmem = dmem.view(B, Sm, D)
        dx1, _, _, _ = K.layernorm_bwd(dh2, S_['x1'], S_['m2'], S_['r2'], W.p('n2.w'), M, D, dres=dx2, dgamma=G['n2.w'], dbeta=G['n2.b'], defer=True)
        # --- causal self attention
        dob = K.rows_mask_cast(dx1, M, D, drop=Drop(p, seed, 2), colsum=G['sa_out_b'])
        K.linear_dw(dob, S_['ctx'], M, D, D, out=G['sa_out_w'])
        _, dctx = K.linear_dx(dob, W.s('sa_out_w'), M, D, D, want_bf16=True)
        dh1 = _self_attention_bwd(W, G, S_, dctx, B, H, A, D, S_['tm'], Drop(p, seed, 1), True, None)
        dx, _, _, _ = K.layernorm_bwd(dh1, S_['x'], S_['m1'], S_['r1'], W.p('n1.w'), M, D, dres=dx1, dgamma=G['n1.w'], dbeta=G['n1.b'], defer=True)
        K.ln_reduce_flush()
        K.wgrad_join()
        return G, dx.view(B, A, D), dmem
