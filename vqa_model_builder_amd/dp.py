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
    (decided by a tiny all-reduced presence bitmap that rides in the first bucket);
  * xGMI is point-to-point (7 links/GPU): buckets are sized in tens of MB so RCCL's direct all-to-all-style
    algorithms keep all links busy, and each bucket's all-reduce is launched asynchronously as soon as it is packed;
  * overlap (``attach()``): a post-accumulate hook per parameter packs its gradient into the bucket the moment autograd
    produces it and the bucket goes on the wire when its last expected gradient has arrived -- the answer head / MoE /
    fusion buckets and the first encoder's buckets travel while the remaining backward still computes; ``finalize()``
    after ``backward()`` sends whatever is left (zero-filling absent gradients), waits, averages and re-points ``.grad``.
"""

from typing import List, Optional

import torch
import torch.distributed as dist


class _Bucket:
    __slots__ = ('params', 'offsets', 'numel', 'flat', 'work', 'filled', 'pending')

    def __init__(self):
        self.params, self.offsets, self.numel, self.flat, self.work = [], [], 0, None, None
        self.filled, self.pending = [], 0


class GradReducer:
    def __init__(self, params: List[torch.nn.Parameter], bucket_mb: float = 64.0, process_group=None, average: bool = True):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.average = average
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
        self.nparams = sum(len(b.params) for b in self.buckets)
        self._presence = None
        self._where = {}
        for bi, b in enumerate(self.buckets):
            for si, p in enumerate(b.params):
                self._where[id(p)] = (bi, si)
        self._hooks = []
        self._armed = False
        self._next = 0

    # ---- overlap mode ---------------------------------------------------------------------------------------------
    def attach(self):
        """Registers the per-parameter hooks (idempotent).  Use ``finalize()`` instead of ``reduce()`` afterwards."""
        if self._hooks or self.world == 1:
            return self
        for b in self.buckets:
            for p in b.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        return self

    def _arm(self, device):
        self._ensure(device)
        for b in self.buckets:
            b.filled = [False] * len(b.params)
            b.pending = len(b.params)
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
        if b.filled[si] or bi < self._next:             # second accumulation in one step (grad accumulation): handled in finalize
            return
        off = b.offsets[si]
        b.flat[off:off + p.numel()].copy_(p.grad.reshape(-1))
        b.filled[si] = True
        b.pending -= 1
        self._launch_ready()

    def _launch_ready(self):
        # Collectives are matched across ranks BY ORDER, and which gradients exist is data dependent (an expert no token
        # was routed to on this rank): buckets therefore go on the wire strictly in index order -- a bucket waits for its
        # own gradients AND for every earlier bucket; whatever is still held back is sent, in the same order, by finalize().
        while self._next < len(self.buckets) and self.buckets[self._next].pending == 0:
            b = self.buckets[self._next]
            b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._next += 1

    @torch.no_grad()
    def finalize(self):
        """Call after ``loss.backward()`` in overlap mode."""
        if self.world == 1:
            return
        device = self.buckets[0].params[0].device
        if not self._armed:
            self._arm(device)
        flags = [0.0 if p.grad is None else 1.0 for b in self.buckets for p in b.params]
        for b in self.buckets[self._next:]:          # buckets an absent gradient (here or earlier) kept off the wire, in order
            for si, (p, off) in enumerate(zip(b.params, b.offsets)):
                view = b.flat[off:off + p.numel()]
                if p.grad is None:
                    view.zero_()
                elif not b.filled[si]:
                    view.copy_(p.grad.reshape(-1))
            b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._next = len(self.buckets)
        self._presence.copy_(torch.tensor(flags, dtype=torch.float32), non_blocking=True)
        pres_work = dist.all_reduce(self._presence, op=dist.ReduceOp.SUM, group=self.group, async_op=True)   # always LAST in the sequence
        pres_work.wait()
        self._assign(self._presence.tolist())
        self._armed = False

    def _assign(self, present):
        scale = 1.0 / self.world if self.average else 1.0
        i = 0
        for b in self.buckets:
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
        if self.world == 1:
            return
        device = self.buckets[0].params[0].device
        self._ensure(device)
        # presence bitmap: which parameters got a gradient on this rank
        flags, k = [], 0
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

    @torch.no_grad()
    def prepare_static(self):
        """HIP-graph mode (graph.GraphedTrainStep), called once after the forward+backward graph was captured: from then on
        every replay writes the gradients to the SAME addresses.  The block runners' gradient arenas (one storage per encoder /
        fusion layer, holding every gradient of the block) are all-reduced IN PLACE, whole -- no packing pass, no second copy
        of 0.9 GB of gradients; the few parameters with a storage of their own are packed into one small buffer and
        ``p.grad`` re-pointed at it.  The set of present gradients is the capture's (a captured step has no data-dependent
        control flow)."""
        by_storage = {}
        for b in self.buckets:
            for p in b.params:
                if p.grad is None:
                    continue
                us = p.grad.untyped_storage()
                by_storage.setdefault(us.data_ptr(), [us, []])[1].append(p)
        self._inplace, loose = [], []
        for us, ps in by_storage.values():
            g0 = ps[0].grad
            if len(ps) > 1 and all(q.grad.dtype == torch.float32 for q in ps) and us.nbytes() % 4 == 0:
                self._inplace.append(torch.empty(0, dtype=torch.float32, device=g0.device).set_(us, 0, (us.nbytes() // 4,)))
            else:
                loose.extend(ps)
        self._loose = None
        if loose:
            n = sum((q.numel() + 3) // 4 * 4 for q in loose)
            flat = torch.zeros(n, dtype=torch.float32, device=loose[0].grad.device)
            dsts, srcs, off = [], [], 0
            for q in loose:
                view = flat[off:off + q.numel()]
                srcs.append(q.grad.reshape(-1))
                dsts.append(view)
                q.grad = view.view(q.shape)
                off += (q.numel() + 3) // 4 * 4
            self._loose = (flat, dsts, srcs)
        return self

    @torch.no_grad()
    def reduce_static(self):
        """After every replay of the forward+backward graph: sum (and, with ``average``, scale) the gradients where they lie."""
        if self.world == 1:
            return
        works = [dist.all_reduce(f, op=dist.ReduceOp.SUM, group=self.group, async_op=True) for f in self._inplace]
        flats = list(self._inplace)
        if self._loose is not None:
            flat, dsts, srcs = self._loose
            torch._foreach_copy_(dsts, srcs)
            works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            flats.append(flat)
        for w in works:
            w.wait()
        if self.average and self.world > 1:
            for f in flats:
                f.mul_(1.0 / self.world)

    def bytes_per_step(self) -> int:
        return sum(b.numel for b in self.buckets) * 4
