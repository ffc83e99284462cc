"""Typed, allocation-aware wrappers over the C ABI (one Python call = one kernel launch on torch's current stream).

PyTorch is plumbing here: it owns device memory (``torch.empty``) and the stream; every byte of arithmetic happens in
libvqa_hip.so.  All wrappers raise on a non-zero return code.
"""

import ctypes as C
from typing import Optional

import torch

from . import lib as _l

ACT_NONE, ACT_GELU, ACT_QUICK_GELU, ACT_RELU = 0, 1, 2, 3
F32 = torch.float32


def HALF():
    """torch dtype of the active library's 16-bit operand type (hip.lib.set_half): bfloat16 by default, float16 in fp16 mode."""
    return torch.float16 if _l.half() == 'fp16' else torch.bfloat16


class HipError(RuntimeError):
    pass


def _chk(rc, what):
    if rc != 0:
        raise HipError(f'{what} failed with code {rc}')


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def L():
    return _l.load()


# ---- weight prefetch (one layer ahead, on a stream of its own): MEASURED, OFF ------------------------------------------------------
# profiles/r02/gemm_cold_weights.log: every GEMM of the step reads its weights from HBM (nothing survives in the Infinity Cache between two
# uses) and pays +2 - 4 us for it; one streaming reader in front gives most of that back.  ``prefetch_weights`` is called by the block
# runners at the start of a layer with the NEXT layer's weight views and reads them once (vqa_prefetch) on the prefetch stream.  Inside a
# HIP-graph capture that stream is a ONE-level fork of the capture stream (profiles/r02/nested_fork_capture.md): ``prefetch_begin()`` (on
# the capture stream, before any tower forks) and ``prefetch_join()`` bracket every captured region.
# Result on MI355X (profiles/r02/weight_prefetch.md; cfg2, one box, captured step 7.14 - 7.17 ms without):
#   * gated (every prefetch waits for an event of the layer before it: exactly one layer ahead)   9.12 ms: the 48 cross-branch edges
#     serialise the graph's branches (the eager GEMM sum does improve: 5.56 -> 5.40 ms);
#   * self-paced (no edge; the kernel's width sets its pace: 4 / 8 / 12 / 16 / 24 / 48 workgroups)  8.73 / 7.97 / 7.82 / 7.64 / 7.54 / 7.38 ms:
#     narrow kernels outlast the region they were forked in, wide ones run layers ahead of the compute and are evicted again.
# So the cold-weight cost stays; what would remove it is a prefetch INSIDE the compute chain (e.g. the LayerNorm kernels reading the next
# GEMMs' weights), worth ~5 us per layer and direction at best.  Kept as an option for eager, single-stream use.
WEIGHT_PREFETCH = False
PREFETCH_GATED = True
PREFETCH_WORKGROUPS = 48
_prefetch_stream = None
_prefetch_live = False


def _pf_stream():
    global _prefetch_stream
    if _prefetch_stream is None:
        _prefetch_stream = torch.cuda.Stream()
    return _prefetch_stream


def prefetch_begin():
    global _prefetch_live
    if not WEIGHT_PREFETCH:
        return
    _pf_stream().wait_stream(torch.cuda.current_stream())
    _prefetch_live = True


def prefetch_join():
    global _prefetch_live
    if _prefetch_live:
        torch.cuda.current_stream().wait_stream(_pf_stream())
    _prefetch_live = False


def prefetch_weights(tensors, workgroups=64):
    """``tensors``: views into ONE arena (a layer's 16-bit weights): the span from the lowest to the highest address is read once."""
    if not WEIGHT_PREFETCH or not tensors:
        return
    capturing = torch.cuda.is_current_stream_capturing()
    if capturing and not _prefetch_live:
        return                                            # a capture that did not fork the prefetch stream: skip (never fork from a branch)
    lo = min(t.data_ptr() for t in tensors) & ~15
    hi = max(t.data_ptr() + t.numel() * t.element_size() for t in tensors)
    ps = _pf_stream()
    if PREFETCH_GATED or not capturing:
        ev = torch.cuda.Event()
        ev.record()
        ps.wait_event(ev)
    _chk(L().vqa_prefetch(lo, hi - lo, workgroups if PREFETCH_GATED else PREFETCH_WORKGROUPS, ps.cuda_stream), 'vqa_prefetch')


# ---- numerics mode of the ring GEMMs ---------------------------------------------------------------------------------------------
# train(): per-XCD k rotation ON (every weight line fetched from HBM by one XCD instead of missing in eight L2s at once: cfg2 7.16 -> 6.94 ms);
# a row's fp32 summation order then depends on the XCD that owns its tile.  eval(): OFF -- results are bit-identical whatever the batch position
# (tests/test_parity_gpu.py::test_full_size_properties_batch32) and the parity fixtures are compared in that mode.  Called by the models' forward.
TRAIN_K_ROTATE = True
FORCE_K_ROTATE = False       # tests: rotation on in eval mode too (the eval-mode parity fixtures under train-mode numerics)
K_ROTATE_PHASE = 0           # tests: 0..7, another assignment of k-loop starting points to the XCDs = another summation order of the same products
_k_rotate_state = None


def set_training_numerics(training: bool):
    global _k_rotate_state
    want = (2 if TRAIN_K_ROTATE == 2 else 1) if ((training and TRAIN_K_ROTATE) or FORCE_K_ROTATE) else 0       # 2 (lab): the grouped weight-gradient launch too
    if want:
        want |= (K_ROTATE_PHASE & 7) << 8
    if want != _k_rotate_state:
        L().vqa_set_gemm_k_rotate(want)
        _k_rotate_state = want


_gd = _l.VqaGemmDesc()
_ad = _l.VqaAttnDesc()
_fd = _l.VqaFusedAttnDesc()


class Drop:
    """Dropout site descriptor: probability + (seed, stream) key of the counter RNG."""
    __slots__ = ('p', 'seed', 'stream')

    def __init__(self, p=0.0, seed=0, stream=0):
        self.p, self.seed, self.stream = float(p), int(seed), int(stream)


NO_DROP = Drop()


def gemm(a, b, M, N, K, lda, ldb, a_kc=True, b_kc=True, out_f32=None, out_bf16=None, pre_bf16=None, bias=None,
         residual=None, act_grad_of=None, act=ACT_NONE, act_bwd=ACT_NONE, drop: Drop = NO_DROP, allow_split_k=False,
         split_k=0, tile_hint=0, ldc_f32=None, ldc_bf16=None, ld_pre=None, ld_res=None, ld_ag=None, colsum=None, c_prezeroed=False):
    d = _gd
    d.a, d.b = _p(a), _p(b)
    d.M, d.N, d.K, d.lda, d.ldb, d.a_kc, d.b_kc = M, N, K, lda, ldb, int(a_kc), int(b_kc)
    d.c_f32, d.ldc_f32 = _p(out_f32), (ldc_f32 or N)
    d.c_bf16, d.ldc_bf16 = _p(out_bf16), (ldc_bf16 or N)
    d.pre_bf16, d.ld_pre = _p(pre_bf16), (ld_pre or N)
    d.bias = _p(bias)
    d.residual, d.ld_res = _p(residual), (ld_res or N)
    d.act_grad_of, d.ld_ag = _p(act_grad_of), (ld_ag or N)
    d.act, d.act_bwd, d.alpha = act, act_bwd, 1.0
    d.drop_p, d.drop_seed, d.drop_stream = drop.p, drop.seed, drop.stream
    d.split_k, d.allow_split_k, d.tile_hint = split_k, int(allow_split_k), tile_hint
    d.colsum, d.c_prezeroed = _p(colsum), int(c_prezeroed)
    _chk(L().vqa_gemm_bf16(C.byref(d), _stream()), f'vqa_gemm_bf16(M={M},N={N},K={K})')


# ---- the three linear-layer products (torch Linear weight [N,K]) ---------------------------------------------

def linear_fwd(x_bf16, w_bf16, bias, M, N, K, *, want_f32=False, want_bf16=False, want_pre=False, act=ACT_NONE,
               residual=None, drop: Drop = NO_DROP, lda=None):
    """y = drop(act(x W^T + b)) + residual.  Returns (y_f32|None, y_bf16|None, pre_bf16|None)."""
    dev = x_bf16.device
    yf = torch.empty((M, N), dtype=F32, device=dev) if want_f32 else None
    yb = torch.empty((M, N), dtype=HALF(), device=dev) if want_bf16 else None
    pre = torch.empty((M, N), dtype=HALF(), device=dev) if want_pre else None
    gemm(x_bf16, w_bf16, M, N, K, lda or K, K, True, True, out_f32=yf, out_bf16=yb, pre_bf16=pre, bias=bias,
         residual=residual, act=act, drop=drop)
    return yf, yb, pre


def linear_dx(dy_bf16, w_bf16, M, N, K, *, want_f32=False, want_bf16=False, residual=None, act_grad_of=None,
              act_bwd=ACT_NONE, drop: Drop = NO_DROP, ldw=None, ldy=None, colsum=None, allow_split_k=False):
    """dx[M,K] = (dy[M,N] W[N,K]) * act'(act_grad_of) * dropmask + residual.  ``colsum`` (pre-zeroed [K] fp32) receives the
    column sums of dx before the residual: the bias gradient of the Linear that produced the activation input."""
    dev = dy_bf16.device
    of = torch.empty((M, K), dtype=F32, device=dev) if want_f32 else None
    ob = torch.empty((M, K), dtype=HALF(), device=dev) if want_bf16 else None
    gemm(dy_bf16, w_bf16, M, K, N, ldy or N, ldw or K, True, False, out_f32=of, out_bf16=ob, residual=residual,
         act_grad_of=act_grad_of, act_bwd=act_bwd, drop=drop, colsum=colsum, allow_split_k=allow_split_k)
    return of, ob


# ---- weight-gradient GEMMs: queued, issued grouped ------------------------------------------------------------------
# Nothing in a backward pass waits for a weight gradient (only the optimiser does).  linear_dw therefore only QUEUES its GEMM;
# wgrad_flush() issues everything queued as grouped launches (vqa_gemm_bf16_grouped: up to 32 GEMMs per grid) and the block
# runner calls it once, at the end of its backward.  One launch then pays one cold start and one tail for a whole encoder's
# weight gradients and its tens of thousands of equal tiles fill every CU, where a lone 768 x 768 output has 144 tiles.
# Operands are kept alive (referenced by the queue) until the flush.  (Issuing them on a side HIP stream instead -- per GEMM, per
# layer or per flush -- measured slower or no faster beside the parallel encoder branches and is not offered.)
WGRAD_GROUPED = True
WGRAD_GROUP_MAX = 128     # items per vqa_gemm_bf16_grouped2 call (the library cuts it into launches: <= 64 on 256 x 256 tiles, <= 32 on the ring kernel)
_wgrad = {}          # main cuda_stream handle -> [unused, unused, pending GEMM argument tuples]
_group_items = None


def _wgrad_slot():
    cur = torch.cuda.current_stream()
    slot = _wgrad.get(cur.cuda_stream)
    if slot is None:
        slot = _wgrad[cur.cuda_stream] = [None, [], []]
    return cur, slot


WGRAD_SUMSQ = None      # optional device fp32 scalar: every grouped weight-gradient launch adds the sum of squares of what it writes (FusedAdamW's
                        # global-norm reduction rides in the GEMMs: optim.FusedAdamW.fuse_wgrad_norm); the covered address ranges are noted in
WGRAD_SUMSQ_COVERED = None      # this list [(first byte, end byte)] since the optimiser's last zero_grad()


def _launch_group(pending):
    global _group_items
    if _group_items is None:
        _group_items = (_l.VqaGemmGroupItem * WGRAD_GROUP_MAX)()
    # outputs the optimiser can match against a parameter gradient by address range (contiguous [N, K] blocks) carry their sum of squares;
    # anything else (a strided slot) goes in a call of its own without it, so that nothing is ever counted that the optimiser cannot see
    fused = WGRAD_SUMSQ is not None and WGRAD_SUMSQ_COVERED is not None
    groups = [(pending, None)]
    if fused:
        # ... and rows written into a ZERO-FILLED slot (prezeroed: the v rows of a one-key attention's packed in-projection, whose q / k rows keep
        # the exact zero the reference gives them) are part of a gradient nobody else writes: counted here, the optimiser would find a range
        # that covers a third of a gradient and fall back to its full pass for the WHOLE model (round 3: every MoE config did)
        ok = [a for a in pending if a[7].is_contiguous() and not a[8]]
        groups = [(ok, WGRAD_SUMSQ), ([a for a in pending if not (a[7].is_contiguous() and not a[8])], None)]
    for todo, ssq in groups:
        for i0 in range(0, len(todo), WGRAD_GROUP_MAX):
            chunk = todo[i0:i0 + WGRAD_GROUP_MAX]
            for it, a in zip(_group_items, chunk):
                dy, x, M, N, Kd, ldy, ldx, out, _ = a
                # dW[N,K] = dy[M,N]^T x[M,K]: GEMM rows = N, columns = K, reduction over the M tokens
                it.a, it.b, it.c_f32 = _p(dy), _p(x), _p(out)
                it.M, it.N, it.K, it.lda, it.ldb, it.ldc = N, Kd, M, ldy, ldx, out.stride(0)
                if ssq is not None:
                    WGRAD_SUMSQ_COVERED.append((out.data_ptr(), out.data_ptr() + 4 * N * Kd))
            _chk(L().vqa_gemm_bf16_grouped2(_group_items, len(chunk), 0, 0, _p(ssq), _stream()), 'vqa_gemm_bf16_grouped2')


def wgrad_flush():
    """Issues the queued weight-gradient GEMMs (grouped), ordered after everything the current stream was given so far."""
    if not _wgrad:
        return
    cur = torch.cuda.current_stream()
    slot = _wgrad.get(cur.cuda_stream)
    if slot is None or not slot[2]:
        return
    _launch_group(slot[2])
    slot[2].clear()


WGRAD_DEFER_TO_STEP_END = False      # graph mode: block runners leave their queues alone; the step calls wgrad_flush_all()


def wgrad_flush_all():
    """Issues every queued weight-gradient GEMM of every stream on the CURRENT stream (after ``loss.backward()`` returned, i.e.
    after autograd joined its streams): the long grouped launches then run alone at full efficiency instead of beside --
    and in the way of -- the other encoder's short dependent dX launches."""
    pending = []
    for slot in _wgrad.values():
        pending.extend(slot[2])
        slot[2].clear()
    if pending:
        _launch_group(pending)


def wgrad_join():
    """End of a block backward: every queued weight-gradient GEMM is issued and ordered before what the current stream does
    next (the gradients are about to be handed to autograd)."""
    if not _wgrad or WGRAD_DEFER_TO_STEP_END:
        return
    wgrad_flush()


SKIP_WEIGHT_GRADS = False      # set by a block whose parameters are all frozen (requires_grad False): see skip_weight_grads


class skip_weight_grads:
    """Context of a block backward whose parameters are ALL frozen (reference training strategies freeze / unfreeze whole
    modules per epoch: training_utils.py:401-455) but whose input still needs a gradient: every weight-gradient GEMM of the
    block is skipped -- a third of its backward FLOPs -- while dX flows on unchanged."""

    def __init__(self, on: bool):
        self.on = bool(on)

    def __enter__(self):
        global SKIP_WEIGHT_GRADS
        self.prev, SKIP_WEIGHT_GRADS = SKIP_WEIGHT_GRADS, self.on or SKIP_WEIGHT_GRADS

    def __exit__(self, *exc):
        global SKIP_WEIGHT_GRADS
        SKIP_WEIGHT_GRADS = self.prev


def linear_dw(dy_bf16, x_bf16, M, N, K, out=None, ldy=None, ldx=None, prezeroed=False, count_norm=True):
    """dW[N,K] = dy[M,N]^T x[M,K]  (fp32; both operands read through the transposing LDS path).  ``prezeroed``: ``out``
    is known to be zero (a zero-filled slot of a gradient arena), so a split-K launch needs no memset.  ``count_norm=False`` (implied by
    ``prezeroed``): the output stays out of the fused clipping norm (optim.FusedAdamW.fuse_wgrad_norm) -- rows of a zero-filled slot are part of
    a gradient nobody else writes, and a tensor autograd will SUM with another gradient of a tied weight is no parameter's gradient at all.
    With WGRAD_GROUPED (default) and a caller-provided ``out`` the GEMM is only queued: the caller must end its backward with wgrad_join()."""
    if SKIP_WEIGHT_GRADS and out is not None:
        return out                                      # frozen block: nobody reads this gradient
    if out is None:
        out = torch.empty((N, K), dtype=F32, device=dy_bf16.device)
        gemm(dy_bf16, x_bf16, N, K, M, ldy or N, ldx or K, False, False, out_f32=out, allow_split_k=True, c_prezeroed=False)
        return out
    if WGRAD_GROUPED and out.stride(-1) == 1 and N % 8 == 0 and K % 8 == 0:        # any token count M: the reduction's ragged tail is zero-filled per lane
        _, slot = _wgrad_slot()
        slot[2].append((dy_bf16, x_bf16, M, N, K, ldy or N, ldx or K, out, bool(prezeroed) or not count_norm))      # [8]: keep out of the fused norm
        return out
    gemm(dy_bf16, x_bf16, N, K, M, ldy or N, ldx or K, False, False, out_f32=out, allow_split_k=True, c_prezeroed=bool(prezeroed))
    return out


def colsum_bf16(x, M, N, ld=None, out=None):
    if out is None:
        out = torch.empty((N,), dtype=F32, device=x.device)
    _chk(L().vqa_colsum_bf16(_p(x), M, N, ld or N, _p(out), _stream()), 'vqa_colsum_bf16')
    return out


def colsum_f32(x, M, N, ld=None, out=None):
    if out is None:
        out = torch.empty((N,), dtype=F32, device=x.device)
    _chk(L().vqa_colsum_f32(_p(x), M, N, ld or N, _p(out), _stream()), 'vqa_colsum_f32')
    return out


def cast_bf16(x_f32, out=None):
    x_f32 = x_f32.contiguous()
    if out is None:
        out = torch.empty(x_f32.shape, dtype=HALF(), device=x_f32.device)
    _chk(L().vqa_cast_f32_bf16(_p(x_f32), _p(out), x_f32.numel(), _stream()), 'vqa_cast_f32_bf16')
    return out


def cast_f32(x_bf16):
    out = torch.empty(x_bf16.shape, dtype=F32, device=x_bf16.device)
    _chk(L().vqa_cast_bf16_f32(_p(x_bf16), _p(out), x_bf16.numel(), _stream()), 'vqa_cast_bf16_f32')
    return out


def cast_multi(jobs_dev, njobs, max_n):
    _chk(L().vqa_cast_multi(_p(jobs_dev), njobs, max_n, _stream()), 'vqa_cast_multi')


def add_f32(a, b, want_f32=True, want_bf16=False):
    y = torch.empty_like(a) if want_f32 else None
    yb = torch.empty(a.shape, dtype=HALF(), device=a.device) if want_bf16 else None
    _chk(L().vqa_add_f32(_p(a), _p(b), _p(y), _p(yb), a.numel(), _stream()), 'vqa_add_f32')
    return y, yb


def gather_rows(src, idx_i32, n, D, ld_src=None, want_f32=False, want_bf16=True):
    dst = torch.empty((n, D), dtype=F32, device=src.device) if want_f32 else None
    dstb = torch.empty((n, D), dtype=HALF(), device=src.device) if want_bf16 else None
    _chk(L().vqa_gather_rows_f32(_p(src), _p(idx_i32), _p(dst), _p(dstb), n, D, ld_src or D, _stream()), 'vqa_gather_rows_f32')
    return dst, dstb


# ---- row kernels of the expert runners (csrc/expert_ops.hip) --------------------------------------------------------

def rows_mask_cast(dy, M, N, *, pre=None, act=ACT_NONE, drop: Drop = NO_DROP, colsum=None, ld=None):
    """bf16(dy * act'(pre) * dropmask) [M, N] and, fused, its column sums ADDED into ``colsum`` (a zero-filled arena slot)."""
    out = torch.empty((M, N), dtype=HALF(), device=dy.device)
    _chk(L().vqa_rows_mask_cast(_p(dy), ld or N, _p(pre), act, _p(out), _p(colsum), M, N, drop.p, drop.seed, drop.stream, _stream()),
         'vqa_rows_mask_cast')
    return out


def glu_fwd(h, T, H, drop: Drop = NO_DROP):
    y = torch.empty((T, H), dtype=F32, device=h.device)
    _chk(L().vqa_glu_fwd(_p(h), _p(y), T, H, drop.p, drop.seed, drop.stream, _stream()), 'vqa_glu_fwd')
    return y


def glu_bwd(dy, h, T, H, drop: Drop = NO_DROP):
    dh = torch.empty((T, 2 * H), dtype=F32, device=h.device)
    _chk(L().vqa_glu_bwd(_p(dy), _p(h), _p(dh), T, H, drop.p, drop.seed, drop.stream, _stream()), 'vqa_glu_bwd')
    return dh


def head_keep_fwd(v, T, R, H, Dh, drop: Drop = NO_DROP):
    out = torch.empty((T * R, H * Dh), dtype=HALF(), device=v.device)
    _chk(L().vqa_head_keep_fwd(_p(v), _p(out), T, R, H, Dh, drop.p, drop.seed, drop.stream, _stream()), 'vqa_head_keep_fwd')
    return out


def head_keep_bwd(dout, T, R, H, Dh, drop: Drop = NO_DROP):
    dv = torch.empty((T, H * Dh), dtype=HALF(), device=dout.device)
    _chk(L().vqa_head_keep_bwd(_p(dout), _p(dv), T, R, H, Dh, drop.p, drop.seed, drop.stream, _stream()), 'vqa_head_keep_bwd')
    return dv


def repeat_rows(src, out_rows, D, R, mode, alpha=1.0, want_f32=True, want_bf16=False):
    dst = torch.empty((out_rows, D), dtype=F32, device=src.device) if want_f32 else None
    dstb = torch.empty((out_rows, D), dtype=HALF(), device=src.device) if want_bf16 else None
    _chk(L().vqa_repeat_rows_f32(_p(src), _p(dst), _p(dstb), out_rows, D, R, mode, alpha, _stream()), 'vqa_repeat_rows_f32')
    return dst, dstb


def rows_mean(x, R, T, D, out=None, out_bf16=None, ld_out=None):
    _chk(L().vqa_rows_mean_f32(_p(x), R, _p(out), _p(out_bf16), ld_out or D, T, D, _stream()), 'vqa_rows_mean_f32')


def take_stride(src_h16, n, stride, offset):
    dst = torch.empty((n,), dtype=HALF(), device=src_h16.device)
    _chk(L().vqa_take_stride_bf16(_p(src_h16), _p(dst), n, stride, offset, _stream()), 'vqa_take_stride_bf16')
    return dst


def scatter_stride(src_f32, dst_f32, n, stride, offset):
    _chk(L().vqa_scatter_stride_f32(_p(src_f32), _p(dst_f32), n, stride, offset, _stream()), 'vqa_scatter_stride_f32')


def _ptr_array(ts):
    return (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])


def moe_dense_combine_fwd(ys, w_all, T, E, D):
    out = torch.empty((T, D), dtype=F32, device=w_all.device)
    _chk(L().vqa_moe_dense_combine_fwd(_ptr_array(ys), _p(w_all), _p(out), T, E, D, _stream()), 'vqa_moe_dense_combine_fwd')
    return out


def moe_dense_combine_bwd(dout, ys, w_all, T, E, D):
    dys = [torch.empty((T, D), dtype=F32, device=dout.device) for _ in range(E)]
    dw = torch.empty((E, T), dtype=F32, device=dout.device)
    _chk(L().vqa_moe_dense_combine_bwd(_p(dout), _ptr_array(ys), _p(w_all), _ptr_array(dys), _p(dw), T, E, D, _stream()), 'vqa_moe_dense_combine_bwd')
    return dys, dw


# ---- LayerNorm ------------------------------------------------------------------------------------------------

def layernorm_fwd(x, gamma, beta, rows, cols, *, add=None, want_f32=True, want_bf16=False, eps=1e-5, drop: Drop = NO_DROP):
    dev = x.device
    y = torch.empty((rows, cols), dtype=F32, device=dev) if want_f32 else None
    yb = torch.empty((rows, cols), dtype=HALF(), device=dev) if want_bf16 else None
    mean = torch.empty((rows,), dtype=F32, device=dev)
    rstd = torch.empty((rows,), dtype=F32, device=dev)
    _chk(L().vqa_layernorm_fwd(_p(x), _p(add), _p(gamma), _p(beta), _p(y), _p(yb), _p(mean), _p(rstd), rows, cols, eps,
                               drop.p, drop.seed, drop.stream, _stream()), 'vqa_layernorm_fwd')
    return y, yb, mean, rstd


LN_REDUCE_GROUPED = True
_ln_pending = {}
_ln_items = None


def ln_reduce_flush():
    """Sums the partials of every deferred layernorm_bwd of the current stream (grouped launches of up to 32)."""
    global _ln_items
    pend = _ln_pending.get(torch.cuda.current_stream().cuda_stream)
    if not pend:
        return
    if _ln_items is None:
        _ln_items = (_l.VqaLnReduceItem * 32)()
    for i0 in range(0, len(pend), 32):
        chunk = pend[i0:i0 + 32]
        for it, (ws, nb, cols, dg, db, cs) in zip(_ln_items, chunk):
            it.ws, it.nblocks, it.cols, it.dgamma, it.dbeta, it.dx_colsum = _p(ws), nb, cols, _p(dg), _p(db), _p(cs)
        _chk(L().vqa_layernorm_reduce_grouped(_ln_items, len(chunk), _stream()), 'vqa_layernorm_reduce_grouped')
    pend.clear()


def layernorm_bwd(dy, x, mean, rstd, gamma, rows, cols, *, dres=None, want_f32=True, want_bf16=False, want_affine=True,
                  drop: Drop = NO_DROP, drop_mode=0, dgamma=None, dbeta=None, dx_colsum=None, accumulate=False, defer=False):
    """``accumulate``: dgamma / dbeta / dx_colsum (all caller-provided, initialised -- slots of a zero-filled GradArena) are
    ADDED to with fp32 atomics: one launch, no workspace (see include/vqa_hip.h)."""
    dev = dy.device
    dx = torch.empty((rows, cols), dtype=F32, device=dev) if want_f32 else None
    dxb = torch.empty((rows, cols), dtype=HALF(), device=dev) if want_bf16 else None
    ws = None
    if LN_REDUCE_GROUPED and defer and dgamma is not None and dbeta is not None:
        # partial sums only; ln_reduce_flush() (end of the block's backward) sums the partials of all deferred calls in one launch
        ws = torch.empty((L().vqa_layernorm_bwd_ws_floats(cols),), dtype=F32, device=dev)
        _chk(L().vqa_layernorm_bwd_partials(_p(dy), _p(x), _p(mean), _p(rstd), _p(gamma), _p(dres), _p(dx), _p(dxb),
                                            1 if dx_colsum is not None else 0, _p(ws), rows, cols, drop.p, drop.seed, drop.stream,
                                            drop_mode, _stream()), 'vqa_layernorm_bwd_partials')
        _ln_pending.setdefault(torch.cuda.current_stream().cuda_stream, []).append(
            (ws, L().vqa_layernorm_bwd_blocks(rows), cols, dgamma, dbeta, dx_colsum))
        return dx, dxb, dgamma, dbeta
    if accumulate:
        assert dgamma is not None and dbeta is not None
        _chk(L().vqa_layernorm_bwd(_p(dy), _p(x), _p(mean), _p(rstd), _p(gamma), _p(dres), _p(dx), _p(dxb), _p(dgamma), _p(dbeta),
                                   _p(dx_colsum), None, rows, cols, drop.p, drop.seed, drop.stream, drop_mode, _stream()), 'vqa_layernorm_bwd')
        return dx, dxb, dgamma, dbeta
    if want_affine:
        if dgamma is None:
            dgamma = torch.empty((cols,), dtype=F32, device=dev)
        if dbeta is None:
            dbeta = torch.empty((cols,), dtype=F32, device=dev)
        ws = torch.empty((L().vqa_layernorm_bwd_ws_floats(cols),), dtype=F32, device=dev)
    if ws is None and dx_colsum is not None:
        ws = torch.empty((L().vqa_layernorm_bwd_ws_floats(cols),), dtype=F32, device=dev)
    _chk(L().vqa_layernorm_bwd(_p(dy), _p(x), _p(mean), _p(rstd), _p(gamma), _p(dres), _p(dx), _p(dxb), _p(dgamma), _p(dbeta),
                               _p(dx_colsum), _p(ws), rows, cols, drop.p, drop.seed, drop.stream, drop_mode, _stream()), 'vqa_layernorm_bwd')
    return dx, dxb, dgamma, dbeta


# ---- attention ---------------------------------------------------------------------------------------------------

def attention_fwd(q, k, v, ldq, ldk, ldv, B, H, Sq, Skv, Dh, mask_u8=None, drop: Drop = NO_DROP, out=None, causal=False):
    if out is None:
        out = torch.empty((B * Sq, H * Dh), dtype=HALF(), device=q.device)
    d = _ad
    d.q, d.k, d.v, d.o = _p(q), _p(k), _p(v), _p(out)
    d.ldq, d.ldk, d.ldv, d.ldo = ldq, ldk, ldv, H * Dh
    d.B, d.H, d.Sq, d.Skv, d.Dh = B, H, Sq, Skv, Dh
    d.key_padding_mask, d.scale = _p(mask_u8), 0.0
    d.drop_p, d.drop_seed, d.drop_stream = drop.p, drop.seed, drop.stream
    d.d_o = d.dq = d.dk = d.dv = None
    d.causal = int(causal)
    _chk(L().vqa_attention_fwd(C.byref(d), _stream()), 'vqa_attention_fwd')
    return out


def attention_bwd(q, k, v, d_o, ldq, ldk, ldv, B, H, Sq, Skv, Dh, dq, dk, dv, lddq, lddk, lddv, mask_u8=None,
                  drop: Drop = NO_DROP, dq_colsum=None, dk_colsum=None, dv_colsum=None, causal=False):
    """``d*_colsum`` (fp32 [H*Dh], zero on entry -- gradient-arena slots): bias gradients of the Q/K/V projections, fused."""
    d = _ad
    d.q, d.k, d.v, d.o = _p(q), _p(k), _p(v), None
    d.ldq, d.ldk, d.ldv, d.ldo = ldq, ldk, ldv, H * Dh
    d.B, d.H, d.Sq, d.Skv, d.Dh = B, H, Sq, Skv, Dh
    d.key_padding_mask, d.scale = _p(mask_u8), 0.0
    d.drop_p, d.drop_seed, d.drop_stream = drop.p, drop.seed, drop.stream
    d.d_o, d.ldd_o = _p(d_o), H * Dh
    d.dq, d.dk, d.dv, d.lddq, d.lddk, d.lddv = _p(dq), _p(dk), _p(dv), lddq, lddk, lddv
    d.dq_colsum, d.dk_colsum, d.dv_colsum = _p(dq_colsum), _p(dk_colsum), _p(dv_colsum)
    nws = L().vqa_attention_bwd_ws_floats(B, H, Sq, Skv, Dh)
    ws = torch.empty((nws,), dtype=F32, device=q.device) if nws else None
    d.ws = _p(ws)
    d.causal = int(causal)
    _chk(L().vqa_attention_bwd(C.byref(d), _stream()), 'vqa_attention_bwd')
    d.dq_colsum = d.dk_colsum = d.dv_colsum = d.ws = None


FUSED_INPROJ_ATTENTION = True      # master switch of the fused in-projection + attention launch (csrc/fused_attn.h)
FUSED_ATTENTION_FUSION = True      # ... in the CrossModalAttention fusion block (self- and cross-attention, both layer forms)
FUSED_ATTENTION_ENCODERS = True    # ... in the CLIP ViT / PhoBERT layer stacks


def fused_attention_covers(D, H, Sq, Skv):
    return FUSED_INPROJ_ATTENTION and D % 64 == 0 and D % H == 0 and D // H in (64, 96) and 1 <= Sq <= 64 and 1 <= Skv <= 64


def fused_inproj_attention_fwd(xq, xkv, w_in, b_in, B, H, Sq, Skv, D, mask_u8=None, drop: Drop = NO_DROP, *, q=None, k=None, v=None,
                               ldq=None, ldk=None, ldv=None, ldxq=None, ldxkv=None, out=None, causal=False):
    """attention(xq Wq^T + bq, xkv Wk^T + bk, xkv Wv^T + bv) per (sample, head) in ONE launch (csrc/fused_attn.h); ``q`` / ``k`` /
    ``v``: optional bf16 destinations of the projections (what backward reads).  Returns the bf16 context [B*Sq, D]."""
    if out is None:
        out = torch.empty((B * Sq, D), dtype=HALF(), device=xq.device)
    d = _fd
    d.xq, d.ldxq, d.xkv, d.ldxkv = _p(xq), ldxq or D, _p(xkv), ldxkv or D
    d.w_in, d.ldw, d.b_in = _p(w_in), D, _p(b_in)
    d.q, d.k, d.v = _p(q), _p(k), _p(v)
    d.ldq, d.ldk, d.ldv = ldq or D, ldk or D, ldv or D
    d.o, d.ldo = _p(out), D
    d.B, d.H, d.Sq, d.Skv, d.D = B, H, Sq, Skv, D
    d.key_padding_mask, d.scale = _p(mask_u8), 0.0
    d.drop_p, d.drop_seed, d.drop_stream = drop.p, drop.seed, drop.stream
    d.causal = int(causal)
    _chk(L().vqa_fused_inproj_attention_fwd(C.byref(d), _stream()), 'vqa_fused_inproj_attention_fwd')
    return out


# ---- CLIP / RoBERTa front ends ------------------------------------------------------------------------------------

def patchify(pixels, ps):
    B, Cc, H, W = pixels.shape
    out = torch.empty((B * (H // ps) * (W // ps), Cc * ps * ps), dtype=HALF(), device=pixels.device)
    _chk(L().vqa_patchify_bf16(_p(pixels), _p(out), B, Cc, H, W, ps, _stream()), 'vqa_patchify_bf16')
    return out


def clip_assemble(E, cls, pos, B, P, D):
    u = torch.empty((B * (P + 1), D), dtype=F32, device=E.device)
    _chk(L().vqa_clip_assemble(_p(E), _p(cls), _p(pos), _p(u), B, P, D, _stream()), 'vqa_clip_assemble')
    return u


def clip_assemble_bwd(du, B, P, D, dcls, dpos):
    dE = torch.empty((B * P, D), dtype=HALF(), device=du.device)
    _chk(L().vqa_clip_assemble_bwd(_p(du), _p(dE), _p(dcls), _p(dpos), B, P, D, _stream()), 'vqa_clip_assemble_bwd')
    return dE


_status = {}


def _norm_device(device):
    dev = torch.device(device)
    return torch.device('cuda', torch.cuda.current_device()) if (dev.type == 'cuda' and dev.index is None) else dev


def status_word(device):
    """Per-device int32 word the index-consuming kernels clear when they meet an out-of-range token id / position id / label
    (they never dereference it; torch would device-assert).  ``check_device_status`` reads it (one host sync)."""
    dev = _norm_device(device)
    w = _status.get(dev)
    if w is None:
        w = _status[dev] = torch.ones(1, dtype=torch.int32, device=dev)
    return w


def check_device_status(device=None):
    """Raises IndexError if any kernel since the last check saw an out-of-range id / label (host sync: call it where the
    reference's loop already syncs, e.g. next to ``loss.item()``); resets the word."""
    for dev, w in _status.items():
        if device is not None and _norm_device(device) != dev:
            continue
        if int(w.item()) != 1:
            w.fill_(1)
            raise IndexError('HIP path: a token id, position id or label was out of range (ids must be in [0, vocab), '
                             'labels in [0, num_answers) or -100)')


def roberta_embed_fwd(ids, word, pos, type0, B, S, D, pad_id=1):
    pos_ids = torch.empty((B, S), dtype=torch.int32, device=ids.device)
    u = torch.empty((B * S, D), dtype=F32, device=ids.device)
    _chk(L().vqa_roberta_embed_fwd(_p(ids), _p(word), _p(pos), _p(type0), _p(pos_ids), _p(u), B, S, D, pad_id, word.shape[0], pos.shape[0],
                                   _p(status_word(ids.device)), _stream()), 'vqa_roberta_embed_fwd')
    return u, pos_ids


def roberta_embed_bwd(du, ids, pos_ids, dword, dpos, dtype0, B, S, D, pad_id=1):
    _chk(L().vqa_roberta_embed_bwd(_p(du), _p(ids), _p(pos_ids), _p(dword), _p(dpos), _p(dtype0), B, S, D, pad_id, dword.shape[0], dpos.shape[0],
                                   _stream()), 'vqa_roberta_embed_bwd')


# ---- loss ------------------------------------------------------------------------------------------------------------

def ce_argmax_fwd(logits, labels, B, Cn, label_smoothing=0.0):
    dev = logits.device
    row_loss = torch.empty((B,), dtype=F32, device=dev)
    loss2 = torch.empty((2,), dtype=F32, device=dev)          # {mean over the non-ignored rows, their count}
    pred = torch.empty((B,), dtype=torch.int64, device=dev)
    lse = torch.empty((B,), dtype=F32, device=dev)
    _chk(L().vqa_softmax_ce_argmax_fwd(_p(logits), Cn, _p(labels), _p(row_loss), _p(loss2) if labels is not None else None,
                                       _p(pred), _p(lse), B, Cn, _p(status_word(dev)), float(label_smoothing), _stream()), 'vqa_softmax_ce_argmax_fwd')
    return (loss2[0] if labels is not None else None), pred, lse, (loss2[1:] if labels is not None else None)


def ce_bwd(logits, labels, lse, dloss, B, Cn, nvalid=None, want_f32=True, want_bf16=False, label_smoothing=0.0):
    dev = logits.device
    dl = torch.empty((B, Cn), dtype=F32, device=dev) if want_f32 else None
    dlb = torch.empty((B, Cn), dtype=HALF(), device=dev) if want_bf16 else None
    _chk(L().vqa_softmax_ce_bwd(_p(logits), Cn, _p(labels), _p(lse), _p(dloss), _p(nvalid), _p(dl), _p(dlb), B, Cn, float(label_smoothing), _stream()),
         'vqa_softmax_ce_bwd')
    return dl, dlb


# ---- misc ---------------------------------------------------------------------------------------------------------------

def dropout_f32(x, drop: Drop, want_f32=True, want_bf16=False):
    y = torch.empty_like(x) if want_f32 else None
    yb = torch.empty(x.shape, dtype=HALF(), device=x.device) if want_bf16 else None
    _chk(L().vqa_dropout_f32(_p(x), _p(y), _p(yb), x.numel(), drop.p, drop.seed, drop.stream, _stream()), 'vqa_dropout_f32')
    return y, yb


def randn(shape, seed, stream, device):
    out = torch.empty(shape, dtype=F32, device=device)
    _chk(L().vqa_randn_f32(_p(out), out.numel(), seed, stream, _stream()), 'vqa_randn_f32')
    return out
