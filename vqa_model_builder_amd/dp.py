"""Data-parallel gradient exchange: one process per GPU, replicas of all parameters, one sum-all-reduce of the
gradients per step over RCCL/xGMI (``torch.distributed`` backend "nccl" IS RCCL on ROCm; "gloo" on CPU in tests).

The reference has no multi-device code at all: this is new work, shaped by SURVEY sections 5/8e:
  * static, flat fp32 buckets laid out in reverse-backward order (answer head -> MoE -> fusion -> encoders) so the
    first buckets to fill are the first to go on the wire;
  * parameters that receive no gradient in a step (frozen modules, an expert no token was routed to, dead
    parameters -- F9) are ZERO-FILLED in their bucket slot, never skipped: every rank always reduces the same
    layout, so data-dependent ``None`` gradients cannot desynchronise the collective;
  * after the reduce, ``p.grad`` is re-pointed at its (averaged) bucket slice -- no copy back; parameters that had
    no gradient on ANY rank keep ``grad = None`` so the optimiser skips them exactly like the reference's would
    (decided by a tiny all-reduced presence bitmap);
  * xGMI is point-to-point (7 links/GPU): buckets are sized in tens of MB so RCCL's direct all-to-all-style
    algorithms keep all links busy, and each bucket's all-reduce is launched asynchronously as soon as it is packed;
  * overlap, eager step (``attach()``): a post-accumulate hook per parameter packs its gradient into its bucket the moment
    autograd produces it; a bucket goes on the wire when the gradients it EXPECTS have arrived.  Expected = every parameter
    that has had a gradient on any rank in an earlier step (learned from the all-reduced presence bitmap, so all ranks agree):
    parameters that never get one (RoBERTa's pooler, CLIP's post_layernorm, MultimodalExpert's dead branch -- the first
    parameters of their encoders in send order) do not hold their bucket, or any later one, back until ``finalize()``;
  * overlap, captured step (``prepare_static`` / ``reduce_segment``): the backward is cut into per-block HIP graphs
    (graph.GraphedTrainStep) and each block's gradient arenas are all-reduced IN PLACE while the next block's graph replays;
    ``grad_dtype='bf16'`` sends bfloat16 copies (half the bytes; fp32 masters stay in the arenas).
"""

from typing import Dict, List, Optional, Sequence

import torch
import torch.distributed as dist


class _Bucket:
    __slots__ = ('params', 'offsets', 'numel', 'flat', 'work', 'filled', 'pending', 'expected')

    def __init__(self):
        self.params, self.offsets, self.numel, self.flat, self.work = [], [], 0, None, None
        self.filled, self.pending, self.expected = [], 0, []


class GradReducer:
    def __init__(self, params: List[torch.nn.Parameter], bucket_mb: float = 64.0, process_group=None, average: bool = True,
                 grad_dtype: str = 'fp32', force_collectives: bool = False):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        # ``force_collectives``: issue every collective even in a ONE-rank process group (a one-GPU box is the only hardware the build
        # sees: this is how the RCCL code path -- communicator, streams, bf16 sums, all-gathers beside graph replays -- runs there)
        self.single = self.world == 1 and not (force_collectives and dist.is_available() and dist.is_initialized())
        self.average = average
        if grad_dtype not in ('fp32', 'bf16'):
            raise ValueError("grad_dtype must be 'fp32' or 'bf16'")
        self.grad_dtype = grad_dtype
        cap = int(bucket_mb * 1024 * 1024 / 4)
        self.buckets: List[_Bucket] = []
        cur = _Bucket()
        for p in reversed([p for p in params if p.requires_grad]):        # reverse registration ~ backward order
            n = (p.numel() + 3) // 4 * 4                                   # 16-byte aligned slots
            if cur.numel and cur.numel + n > cap:
                self.buckets.append(cur)
                cur = _Bucket()
            cur.params.append(p)
            cur.offsets.append(cur.numel)
            cur.numel += n
        if cur.numel:
            self.buckets.append(cur)
        for b in self.buckets:
            b.expected = [True] * len(b.params)               # until the first reduced presence bitmap says otherwise
        self.nparams = sum(len(b.params) for b in self.buckets)
        self._presence = None
        self._where = {}
        for bi, b in enumerate(self.buckets):
            for si, p in enumerate(b.params):
                self._where[id(p)] = (bi, si)
        self._hooks = []
        self._armed = False
        self._next = 0
        self._learned = False
        self.sent_before_finalize = 0                         # diagnostics / tests: buckets on the wire when finalize() was called
        self._segments = None
        self._stats = {'bytes': 0}

    # ---- overlap mode (eager step) -----------------------------------------------------------------------------------
    def attach(self):
        """Registers the per-parameter hooks (idempotent).  Use ``finalize()`` instead of ``reduce()`` afterwards."""
        if self._hooks or self.single:
            return self
        for b in self.buckets:
            for p in b.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        return self

    def _arm(self, device):
        self._ensure(device)
        for b in self.buckets:
            b.filled = [False] * len(b.params)
            b.pending = sum(b.expected)
            b.work = None
        self._next = 0
        self._armed = True

    @torch.no_grad()
    def _on_grad(self, p):
        if p.grad is None:
            return
        if not self._armed:
            self._arm(p.grad.device)
        bi, si = self._where[id(p)]
        b = self.buckets[bi]
        if b.filled[si] or bi < self._next:             # second accumulation in one step, or a gradient nobody expected that arrives
            return                                      # behind its bucket: both are settled in finalize()
        off = b.offsets[si]
        b.flat[off:off + p.numel()].copy_(p.grad.reshape(-1))
        b.filled[si] = True
        if b.expected[si]:
            b.pending -= 1
        self._launch_ready()

    def _launch_ready(self):
        # Collectives are matched across ranks BY ORDER, and which gradients exist is data dependent (an expert no token
        # was routed to on this rank): buckets therefore go on the wire strictly in index order -- a bucket waits for its
        # own EXPECTED gradients AND for every earlier bucket; whatever is still held back is sent, in the same order, by
        # finalize().  Slots of parameters without a gradient are zeroed before the send.
        while self._next < len(self.buckets) and self.buckets[self._next].pending == 0:
            b = self.buckets[self._next]
            for si, (p, off) in enumerate(zip(b.params, b.offsets)):
                if not b.filled[si]:
                    b.flat[off:off + p.numel()].zero_()
            b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._next += 1

    @torch.no_grad()
    def finalize(self):
        """Call after ``loss.backward()`` in overlap mode."""
        if self.single:
            return
        device = self.buckets[0].params[0].device
        if not self._armed:
            self._arm(device)
        self.sent_before_finalize = self._next
        flags = [0.0 if p.grad is None else 1.0 for b in self.buckets for p in b.params]
        for b in self.buckets[self._next:]:          # buckets an absent gradient (here or earlier) kept off the wire, in order
            for si, (p, off) in enumerate(zip(b.params, b.offsets)):
                view = b.flat[off:off + p.numel()]
                if p.grad is None:
                    view.zero_()
                elif not b.filled[si]:
                    view.copy_(p.grad.reshape(-1))
                    b.filled[si] = True
            b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._next = len(self.buckets)
        self._presence.copy_(torch.tensor(flags, dtype=torch.float32), non_blocking=True)
        pres_work = dist.all_reduce(self._presence, op=dist.ReduceOp.SUM, group=self.group, async_op=True)   # always LAST in the sequence
        pres_work.wait()
        present = self._presence.tolist()
        # A parameter nobody expected that got a gradient somewhere (its first ever: e.g. an expert routed to for the first time)
        # may have produced it AFTER its bucket left on some rank.  Every rank derives the same list from the REDUCED bitmap and the
        # (rank-consistent) expected set, and contributes what its own bucket send did not contain; the sum is added to the slot.
        fix, i = [], 0
        for bi, b in enumerate(self.buckets):
            for si in range(len(b.params)):
                if present[i] > 0 and not b.expected[si]:
                    fix.append((bi, si))
                i += 1
        if fix:
            for b in self.buckets:
                if b.work is not None:
                    b.work.wait()
            parts = []
            for bi, si in fix:
                b = self.buckets[bi]
                p = b.params[si]
                missed = p.grad is not None and not b.filled[si]
                parts.append(p.grad.reshape(-1).float() if missed else torch.zeros(p.numel(), dtype=torch.float32, device=device))
            pack = torch.cat(parts)
            dist.all_reduce(pack, op=dist.ReduceOp.SUM, group=self.group)
            o = 0
            for bi, si in fix:
                b = self.buckets[bi]
                n = b.params[si].numel()
                b.flat[b.offsets[si]:b.offsets[si] + n].add_(pack[o:o + n])
                o += n
        self._learn(present)
        self._assign(present)
        self._armed = False

    def _learn(self, present):
        """Expected set := every parameter that has had a gradient on some rank in some step so far (monotone, rank-consistent)."""
        i = 0
        for b in self.buckets:
            for si in range(len(b.params)):
                b.expected[si] = (present[i] > 0) if not self._learned else (b.expected[si] or present[i] > 0)
                i += 1
        self._learned = True

    def _assign(self, present):
        scale = 1.0 / self.world if self.average else 1.0
        i = 0
        for b in self.buckets:
            if b.work is not None:
                b.work.wait()
            if scale != 1.0:
                b.flat.mul_(scale)
            for p, off in zip(b.params, b.offsets):
                p.grad = b.flat[off:off + p.numel()].view(p.shape) if present[i] > 0 else None
                i += 1

    def _ensure(self, device):
        for b in self.buckets:
            if b.flat is None or b.flat.device != device:
                b.flat = torch.zeros(b.numel, dtype=torch.float32, device=device)
        if self._presence is None or self._presence.device != device:
            self._presence = torch.zeros(self.nparams, dtype=torch.float32, device=device)

    @torch.no_grad()
    def reduce(self):
        """Call after ``loss.backward()``.  Sums (and averages) every gradient across ranks."""
        if self.single:
            return
        device = self.buckets[0].params[0].device
        self._ensure(device)
        flags = []
        for b in self.buckets:
            for p in b.params:
                flags.append(0.0 if p.grad is None else 1.0)
        self._presence.copy_(torch.tensor(flags, dtype=torch.float32), non_blocking=True)
        pres_work = dist.all_reduce(self._presence, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        # pack + launch, bucket by bucket
        for b in self.buckets:
            srcs, dsts, zero_slots = [], [], []
            for p, off in zip(b.params, b.offsets):
                view = b.flat[off:off + p.numel()]
                if p.grad is None:
                    zero_slots.append(view)
                else:
                    srcs.append(p.grad.reshape(-1))
                    dsts.append(view)
            if zero_slots:
                torch._foreach_zero_(zero_slots)
            if srcs:
                torch._foreach_copy_(dsts, srcs)
            b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        pres_work.wait()
        self._assign(self._presence.tolist())

    # ---- captured step: per-segment in-place exchange ------------------------------------------------------------------
    @torch.no_grad()
    def prepare_static(self, segment_of: Optional[Dict[int, str]] = None, order: Sequence[str] = ('all',), sparse=()):
        """HIP-graph mode (graph.GraphedTrainStep), called once after the backward graphs were captured: from then on every
        replay writes the gradients to the SAME addresses.  The block runners' gradient arenas (one storage per encoder / fusion
        layer, holding every gradient of the block) are all-reduced IN PLACE, whole -- no packing pass, no second copy of 0.9 GB
        of gradients; the few parameters with a storage of their own are packed into one small buffer per segment and ``p.grad``
        re-pointed at it.  ``segment_of`` maps id(parameter) -> segment name (the backward graph that produces its gradient),
        ``order`` lists the segments in the order their graphs replay: ``reduce_segment(name)`` exchanges one of them.  The set
        of present gradients is the capture's (a captured step has no data-dependent control flow).
        ``sparse``: callables returning ``(parameter, ids [M], rows [M, D] fp32, pad_id)`` for embedding tables whose gradient is a
        scatter of M rows (the 64 001 x 768 word table: 196 MB dense, 6 MB of rows at batch 32): the table is left out of the
        all-reduce; every rank all-gathers (ids, rows) and adds the other ranks' rows to its own table gradient.  The sum is the one the
        dense all-reduce gives, in a different (fp32) summation order."""
        self._segments = {}
        sparse_of = {}
        for getter in sparse:
            got = getter()
            if got is not None and got[0].grad is not None:
                sparse_of[id(got[0])] = getter
        for name in order:
            by_storage, sparse_here = {}, []
            for b in self.buckets:
                for p in b.params:
                    if p.grad is None or (segment_of is not None and segment_of.get(id(p), order[0]) != name):
                        continue
                    if id(p) in sparse_of:
                        sparse_here.append((p, sparse_of[id(p)]))
                        continue
                    us = p.grad.untyped_storage()
                    by_storage.setdefault(us.data_ptr(), [us, []])[1].append(p)
            inplace, loose, where = [], [], []
            for us, ps in by_storage.values():
                g0 = ps[0].grad
                if len(ps) > 1 and all(q.grad.dtype == torch.float32 and q.grad.is_contiguous() for q in ps) and us.nbytes() % 4 == 0:
                    # the segment's gradients inside this arena as maximal contiguous runs (a block whose backward is split over two
                    # graphs contributes a run of its accumulated slots and a run of its weight slots to each segment; a whole
                    # block is one run): each run is all-reduced in place
                    whole = torch.empty(0, dtype=torch.float32, device=g0.device).set_(us, 0, (us.nbytes() // 4,))
                    spans = sorted((q.grad.storage_offset(), q.grad.storage_offset() + q.grad.numel()) for q in ps)
                    runs = [list(spans[0])]
                    for a, e in spans[1:]:
                        if a <= runs[-1][1] + 8:              # slots are 16-byte aligned: <= 3 floats of padding between neighbours
                            runs[-1][1] = max(runs[-1][1], e)
                        else:
                            runs.append([a, e])
                    inplace.extend(whole[a:e] for a, e in runs)
                    for q in ps:                                  # which run holds each gradient, and where: for wire_gradient_ptrs()
                        for ri, (a, e) in enumerate(runs):
                            if a <= q.grad.storage_offset() < e:
                                where.append((q, len(inplace) - len(runs) + ri, q.grad.storage_offset() - a))
                else:
                    loose.extend(ps)
            pack = None
            if loose:
                n = sum((q.numel() + 3) // 4 * 4 for q in loose)
                flat = torch.zeros(n, dtype=torch.float32, device=loose[0].grad.device)
                dsts, srcs, off = [], [], 0
                for q in loose:
                    view = flat[off:off + q.numel()]
                    srcs.append(q.grad.reshape(-1))
                    dsts.append(view)
                    q.grad = view.view(q.shape)
                    where.append((q, len(inplace), off))
                    off += (q.numel() + 3) // 4 * 4
                pack = (flat, dsts, srcs)
                inplace.append(flat)
            stage = None
            if self.grad_dtype == 'bf16':
                stage = [torch.empty(f.numel(), dtype=torch.bfloat16, device=f.device) for f in inplace]
            nbytes = sum(f.numel() for f in inplace) * (2 if stage is not None else 4)
            gather = 0
            for p, getter in sparse_here:
                _, ids, rows, _ = getter()
                gather += rows.numel() * 4 + ids.numel() * 4          # per rank; every rank receives world x this
            nbytes += self.world * gather
            self._segments[name] = {'flats': inplace, 'pack': pack, 'stage': stage, 'works': [], 'bytes': nbytes, 'sparse': sparse_here, 'gathered': [], 'gather_bytes_per_rank': gather,
                                    'where': where}
        self._inplace = [f for s in self._segments.values() for f in s['flats']]
        return self

    def wire_gradient_ptrs(self) -> Dict[int, int]:
        """bf16 buckets: id(parameter) -> device address of the parameter's gradient inside the segment's bfloat16 staging buffer (what the
        all-reduce leaves there IS the summed gradient).  An optimiser that reads it there (FusedAdamW.wire_grads) makes the copy back into
        the fp32 arenas unnecessary: call ``consume_on_wire()`` and ``wait_segment`` skips it -- 1.2 GB of copy traffic per step at 244 M
        parameters, and the optimiser's two gradient reads shrink from 4 to 2 bytes per parameter.  ``p.grad`` then keeps the LOCAL
        gradient of the step (nobody reads it in the captured step)."""
        out = {}
        for seg in (self._segments or {}).values():
            if seg['stage'] is None:
                continue
            for q, fi, off in seg['where']:
                out[id(q)] = seg['stage'][fi].data_ptr() + 2 * off
        return out

    def consume_on_wire(self, on: bool = True):
        self._consume_on_wire = bool(on)
        return self

    def segment_needs_pack(self, name: str) -> bool:
        """Whether ``pack_segment`` issues any device work for this segment (stand-alone gradients to gather, or bf16 wire copies)."""
        seg = self._segments[name]
        return seg['pack'] is not None or seg['stage'] is not None

    def pack_segment(self, name: str):
        """Device-side preparation of a segment's exchange (capturable: the tail of the segment's backward graph): gathers the
        stand-alone gradients into the segment's pack buffer and, with bf16 buckets, writes the bfloat16 copies to send."""
        seg = self._segments[name]
        if seg['pack'] is not None:
            _, dsts, srcs = seg['pack']
            torch._foreach_copy_(dsts, srcs)
        if seg['stage'] is not None:
            for f, s in zip(seg['flats'], seg['stage']):
                s.copy_(f)                                   # fp32 -> bf16 (plumbing for the wire format, not model arithmetic)

    @torch.no_grad()
    def reduce_segment(self, name: str):
        """Launches the (asynchronous) all-reduce of one segment on the communication stream: it is ordered after everything the
        current stream has been given so far (the segment's backward graph) and runs beside whatever is enqueued next."""
        if self.single:
            return
        seg = self._segments[name]
        bufs = seg['stage'] if seg['stage'] is not None else seg['flats']
        seg['works'] = [dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True) for t in bufs]
        seg['gathered'] = []
        for p, getter in seg['sparse']:
            _, ids, rows, pad = getter()
            ids32 = torch.where(ids == pad, torch.full_like(ids, -1), ids).to(torch.int32)       # padding_idx rows carry no gradient
            rows = rows.contiguous()
            ids_all = [torch.empty_like(ids32) for _ in range(self.world)]
            rows_all = [torch.empty_like(rows) for _ in range(self.world)]
            seg['works'].append(dist.all_gather(ids_all, ids32, group=self.group, async_op=True))
            seg['works'].append(dist.all_gather(rows_all, rows, group=self.group, async_op=True))
            seg['gathered'].append((p, ids_all, rows_all))

    @torch.no_grad()
    def wait_segment(self, name: str):
        seg = self._segments[name]
        for w in seg['works']:
            w.wait()
        seg['works'] = []
        if seg['stage'] is not None and not getattr(self, '_consume_on_wire', False):
            for f, s in zip(seg['flats'], seg['stage']):
                f.copy_(s)                                   # bf16 sum -> the fp32 arena the optimiser reads
        if seg['gathered']:
            from .hip import kernels as K
            rank = dist.get_rank(self.group)
            for p, ids_all, rows_all in seg['gathered']:
                V, D = p.grad.shape
                for r in range(self.world):
                    if r != rank:                            # this rank's own rows are in its table gradient already
                        K._chk(K.L().vqa_embedding_rows_bwd(rows_all[r].data_ptr(), ids_all[r].data_ptr(), p.grad.data_ptr(), ids_all[r].numel(), D, V,
                                                            K._stream()), 'vqa_embedding_rows_bwd')
                if self.average and self.world > 1:
                    p.grad.mul_(1.0 / self.world)
            seg['gathered'] = []
        if self.average and self.world > 1:
            for f in seg['flats']:
                f.mul_(1.0 / self.world)

    @torch.no_grad()
    def reduce_static(self):
        """All segments back to back (no overlap): the round-1 exchange between the backward and the optimiser graph."""
        if self.single:
            return
        for name in self._segments:
            self.pack_segment(name)
            self.reduce_segment(name)
        for name in self._segments:
            self.wait_segment(name)

    def segment_gather_bytes(self) -> Dict[str, int]:
        """Bytes each rank contributes to a segment's all-gathers (sparse embedding rows); included world-fold in segment_bytes()."""
        return {k: s.get('gather_bytes_per_rank', 0) for k, s in (self._segments or {}).items()}

    def segment_bytes(self) -> Dict[str, int]:
        return {k: v['bytes'] for k, v in (self._segments or {}).items()}

    def bytes_per_step(self) -> int:
        if self._segments:
            return sum(v['bytes'] for v in self._segments.values())
        return sum(b.numel for b in self.buckets) * 4
