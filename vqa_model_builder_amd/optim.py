"""Fused AdamW + global-norm clipping + weight-shadow refresh: the optimiser half of the training step as two HIP
launches per parameter group instead of torch's clip (norm + scale) + fused-AdamW + per-forward shadow casts.

Semantics = ``torch.nn.utils.clip_grad_norm_(params, max_norm)`` followed by ``torch.optim.AdamW.step()`` (the pair the
reference's loop runs, training_pipeline.py:497-502; param groups by name substring, :239-252): checked against torch in
tests/test_kernels_gpu.py::test_fused_adamw_matches_torch.  HBM-bound: 28 B/param (p, g, m, v read; p, m, v written)
+ 2 B/param for the bf16 shadow the next forward's GEMMs read, so the shadow refresh costs no extra pass.
State layout (``step`` / ``exp_avg`` / ``exp_avg_sq`` per parameter) matches torch's AdamW, so optimiser checkpoints
are interchangeable.
"""

import os
import struct

import torch

from .hip import kernels as K
from .hip import lib as _hl
from .hip import ops as _ops
from .hip import shadow as _shadow


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=2e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, max_grad_norm=None, loss_scale=None,
                 growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
        """``loss_scale`` (fp16 operand mode; None = off): a float = static scale, ``'dynamic'`` = torch.amp.GradScaler's
        policy (start 65536, x ``backoff_factor`` after a step with non-finite gradients -- which is skipped --, x
        ``growth_factor`` after ``growth_interval`` clean steps), kept in device memory and applied inside the update kernels:
        ``scale_loss(loss).backward()`` then ``step()`` replaces the reference loop's scaler.scale / unscale_ / step / update
        sequence (training_pipeline.py:466-502) without a host read of found_inf.  A caller that drives its own
        ``GradScaler`` needs none of this: the gradients are ordinary fp32 tensors."""
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.max_grad_norm = max_grad_norm
        self._amp_cfg = None
        if loss_scale is not None:
            dynamic = loss_scale == 'dynamic'
            self._amp_cfg = (65536.0 if dynamic else float(loss_scale), dynamic, float(growth_factor), float(backoff_factor), int(growth_interval))
        self._amp = None                                  # device floats {scale, growth_tracker, found_inf of the last step}
        self._shadow_sets = []
        self._tables = {}
        self._norm2 = None
        self._hyper = None                                # device-side schedule: one {lr, step} float pair per STEP CLASS (the parameters of a
                                                          # group that share a step count); None = lr / bias corrections passed by value
        self._hyper_gi, self._hyper_lr, self._hyper_of = [], [], {}      # class -> group index, the lr last written to it, id(parameter) -> class
        self._hyper_key, self._epoch = [], 0              # class -> (group, step count at creation, step() call that created it)
        self._staging = {}                                # HIP-graph mode: pinned host rows per job table
        self._wnorm2 = None                               # fuse_wgrad_norm(): device scalar the weight-gradient GEMMs add their sum of squares to
        self._touched = {}                                # id(parameter) -> (touched-granule map, the exp_avg it was built for): _touched_map()
        self.sparse_row_updates = os.environ.get('VQA_SPARSE_ROWS', '1') != '0'      # False: every granule of every parameter takes the full update path (tests, A/B)
        self.grad_prescale = 1.0                          # DP: gradients hold the all-reduced SUM; 1/world is applied here
        self.wire_grads = None                            # DP captured step with bf16 buckets: id(parameter) -> device address of its all-reduced
                                                          # bfloat16 gradient in the exchange's staging buffer (read there: no copy back to fp32)

    def fuse_wgrad_norm(self, on: bool = True, device=None):
        """The global-norm reduction of ``clip_grad_norm_`` for the WEIGHT gradients rides in the grouped GEMMs that produce them (their epilogues
        add the sum of squares of what they store to a device scalar: csrc/gemm_dw256.h, gemm_epilogue), instead of a pass that re-reads 4 B per
        parameter: 0.98 GB per cfg2 step.  ``zero_grad()`` clears the scalar and the list of covered address ranges; ``step()`` leaves the covered
        gradients out of its own norm pass and adds the scalar -- but only when every covered byte IS the gradient of a parameter it updates
        (a tied weight whose two gradients autograd summed, a gradient a hook replaced, a second backward accumulated into the first: the
        covered ranges and the gradients no longer coincide and the step falls back to the full pass, which is always correct).  One optimiser
        at a time (the scalar hangs off hip.kernels); not for gradients that are all-reduced after the GEMMs (data parallel: the norm must be
        the reduced gradients').  ``GraphedTrainStep`` switches it on for its single-GPU step."""
        if on:
            dev = torch.device(device) if device is not None else next((p.device for g in self.param_groups for p in g['params']), None)
            self._wnorm2 = torch.zeros(_hl.SUMSQ_SLOTS * _hl.SUMSQ_STRIDE, dtype=torch.float32, device=dev)      # slotted partial sums (include/vqa_hip.h)
            K.WGRAD_SUMSQ, K.WGRAD_SUMSQ_COVERED = self._wnorm2, []
        else:
            # also when the accumulators are ANOTHER optimiser's (an earlier model of this process): the GEMMs of the step being built must not
            # keep adding to a buffer nobody reads; that optimiser then finds the global gone and takes its full norm pass (always correct)
            K.WGRAD_SUMSQ, K.WGRAD_SUMSQ_COVERED = None, None
            self._wnorm2 = None
        return self

    def zero_grad(self, set_to_none: bool = True):
        super().zero_grad(set_to_none)
        if self._wnorm2 is not None and K.WGRAD_SUMSQ is self._wnorm2:
            self._wnorm2.zero_()
            K.WGRAD_SUMSQ_COVERED.clear()

    def _norm_counted(self, grads):
        """{id(parameter)} whose gradient lies inside an address range the weight-gradient GEMMs already took the sum of squares of, or None when
        the covered ranges and the gradients do not coincide byte for byte (then the scalar is ignored).  ``grads``: [(id, first byte, end byte)]."""
        if self._wnorm2 is None or K.WGRAD_SUMSQ is not self._wnorm2 or not K.WGRAD_SUMSQ_COVERED:
            return None
        import bisect
        spans = sorted(set(K.WGRAD_SUMSQ_COVERED))
        if len(spans) != len(K.WGRAD_SUMSQ_COVERED) or any(a[1] > b[0] for a, b in zip(spans, spans[1:])):
            return None                                    # a range written twice (two backward passes since zero_grad) or overlapping ranges
        merged = []                                       # ranges that touch are one range: a packed gradient (q | k | v rows of an in-projection)
        for a, e in spans:                                # may be written by two GEMMs, each covering its rows
            if merged and merged[-1][1] == a:
                merged[-1][1] = e
            else:
                merged.append([a, e])
        spans = [(a, e) for a, e in merged]
        starts = [a for a, _ in spans]
        counted, nbytes = set(), 0
        for pid, a, e in grads:
            i = bisect.bisect_right(starts, a) - 1
            if i >= 0 and e <= spans[i][1]:
                counted.add(pid)
                nbytes += e - a
        return counted if nbytes == sum(e - a for a, e in spans) else None

    def attach_shadows(self, model):
        """Lets the update kernel write the bf16 / packed-fp32 weight shadows of the block runners directly."""
        self._shadow_sets = [m._W.shadows for m in model.modules() if hasattr(m, '_W')]
        # nodes built lazily at the first forward (the model's tail: vqa_model._Tail) hang off their host module's ``_tails``
        self._tail_hosts = [m for m in model.modules() if hasattr(type(m), '_tail')]
        return self

    def make_capturable(self, device):
        """Device-side schedule: learning rate and step count live in device memory -- one {lr, step} pair per STEP CLASS, i.e. per set of
        parameters of a group that share a step count (torch's AdamW keeps ``step`` per parameter: a checkpoint may hold several values, and a
        parameter that receives its first gradient late starts its own bias corrections) -- the count is advanced by a device-side add that is
        part of the (captured) step and the update kernel derives the bias corrections from it: a replayed graph follows the schedule instead
        of repeating the captured step's constants, and a step skipped for non-finite fp16 gradients does not count.  Call before a capture,
        after any eager warm-up steps.  ``step()`` copies a changed ``group['lr']`` to the device words itself whenever it runs outside a
        stream capture; between REPLAYS of a captured step call ``refresh_lr()`` (``GraphedTrainStep`` does)."""
        self.sync_step_counts()                           # already device-side: the host copies first, they define the classes
        self._hyper, self._hyper_gi, self._hyper_lr, self._hyper_of, self._hyper_key = [], [], [], {}, []
        self._hyper_dev = torch.device(device)
        self._epoch += 1
        for gi, group in enumerate(self.param_groups):
            for p in group['params']:
                if self.state.get(p):                    # a parameter that has never been updated gets its class with its FIRST gradient (step()):
                    self._class_of(gi, p)                # its bias corrections start there, not with the parameters that train from step one
        self._staging = {slot: torch.empty(tuple(c[1].shape), dtype=torch.int64).pin_memory() for slot, c in self._tables.items()}
        return self

    def _class_of(self, gi, p):
        """Step class of parameter ``p`` (group ``gi``): created on first sight -- never inside a stream capture, where neither the allocation
        nor the initialising fill would be legal / replay-safe."""
        ci = self._hyper_of.get(id(p))
        if ci is not None:
            return ci
        st = self.state.get(p)
        # (an expert under dense MoE dispatch counts its own updates on the device -- ``_vqa_step``, which overrides the class count in the
        # kernel's bias corrections --; it stays in the class its host-side step puts it in, so the job tables keep the partition the eager
        # warm-up steps built and a capture finds every table allocated)
        step0 = int(st['step']) if st else 0
        key = (gi, step0, self._epoch)                     # classes are shared only among parameters that join in the same call: an older
        if key in self._hyper_key:                         # class with the same starting count has advanced since
            ci = self._hyper_key.index(key)
        else:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError('FusedAdamW: a parameter without a device-side step class received its first gradient inside a stream capture; '
                                   'run one eager step in the capture\'s configuration before make_capturable()')
            lr = float(self.param_groups[gi]['lr'])
            self._hyper.append(torch.tensor([lr, float(max(step0, 0))], dtype=torch.float32).to(self._hyper_dev))
            self._hyper_gi.append(gi)
            self._hyper_lr.append(lr)
            self._hyper_key.append(key)
            ci = len(self._hyper) - 1
        self._hyper_of[id(p)] = ci
        return ci

    # ---- loss scaling (fp16 operand mode) -----------------------------------------------------------------------------
    def _amp_state(self, device):
        if self._amp is None or self._amp.device != torch.device(device):
            self._amp = torch.tensor([self._amp_cfg[0], 0.0, 0.0, 0.0], dtype=torch.float32).to(device)
        return self._amp

    def scale_loss(self, loss: torch.Tensor) -> torch.Tensor:
        """loss x the current scale (a device multiply: replays of a captured step follow the scale)."""
        if self._amp_cfg is None:
            return loss
        return loss * self._amp_state(loss.device)[0]

    @property
    def loss_scale(self):
        return None if self._amp_cfg is None else (float(self._amp[0]) if self._amp is not None else self._amp_cfg[0])

    def found_inf(self) -> bool:
        """Whether the last step was skipped for non-finite gradients (host read: diagnostics / tests only)."""
        return self._amp is not None and float(self._amp[2]) != 0.0

    def refresh_lr(self):
        """Copies every group's current ``lr`` to the device words of its step classes where it changed (one small fill per changed class;
        nothing when the schedule did not move).  ``step()`` calls it outside stream captures; call it between replays of a captured step."""
        if self._hyper is not None:
            for i, h in enumerate(self._hyper):
                lr = float(self.param_groups[self._hyper_gi[i]]['lr'])
                if lr != self._hyper_lr[i]:
                    h[0:1].fill_(lr)
                    self._hyper_lr[i] = lr

    def note_replays(self, n=1):
        """Kept for callers of round 1: the step count of a capturable optimiser lives on the device (``hyper[1]``) and the
        python-side ``state[p]['step']`` is read back from it on demand (``sync_step_counts``), so nothing to note."""

    def sync_step_counts(self):
        """Device-side schedule: ``state[p]['step']`` := the device step count of the parameter's step class -- the number of updates
        actually APPLIED (the capture pass applies none; a step skipped for non-finite fp16 gradients does not count).  One
        host read per class; called by ``state_dict()``."""
        if self._hyper is None:
            return
        counts = [int(round(float(h[1]))) for h in self._hyper]
        for group in self.param_groups:
            for p in group['params']:
                ci = self._hyper_of.get(id(p))
                if ci is not None and p in self.state and self.state[p]:
                    own = getattr(p, '_vqa_step', None)      # an expert's parameters: the updates the expert received
                    self.state[p]['step'] = counts[ci] if own is None else int(round(float(own)))

    def state_dict(self):
        self.sync_step_counts()
        return super().state_dict()

    def _shadow_map(self):
        out = {}
        lazy = [t._W.shadows for h in getattr(self, '_tail_hosts', ()) for t in h.__dict__.get('_tails', {}).values()]
        sets = [ss for ss in self._shadow_sets + lazy if ss._jobs is not None]   # others: not materialised yet, their first refresh casts
        for ss in sorted(sets, key=lambda ss: ss.last_use):                      # a parameter in two sets: the most recently used one wins
            for _, p, arena, off, _n in ss._plan:
                t = ss._arena[arena]
                out[id(p)] = (t.data_ptr() + off * t.element_size(), 0 if arena == 'bf16' else 1, ss)
        return out, sets

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        lib, st = K.L(), K._stream()
        self._epoch += 1
        if self._amp_cfg is not None and self._hyper is None:
            # a step skipped for non-finite gradients must not advance the bias corrections: count on the device from the start
            first = next((p for g in self.param_groups for p in g['params'] if p.grad is not None), None)
            if first is not None:
                self.make_capturable(first.device)
        if self._hyper is not None and not torch.cuda.is_current_stream_capturing():
            self.refresh_lr()                              # a scheduler moved group['lr'] since the last step (reference loop: LambdaLR warm-up
                                                           # from 0, scheduler.step() per step -- training_pipeline.py:311-318,510)
        shadows, shadow_sets = self._shadow_map()
        updated = set()
        launches, dev = {}, None                        # (group index, step count) -> rows: torch's bias correction is per-parameter
        counted = {}                                    # MoE layers under dense dispatch: id(steps) -> (active [E], steps [E], {expert: step so far})
        standalone = []
        spans = []                                      # (id(parameter), first byte, end byte) of every fp32 gradient read where autograd left it
        for gi, group in enumerate(self.param_groups):
            for p in group['params']:
                g = p.grad
                if g is None:
                    continue
                if not g.is_contiguous() or g.dtype != torch.float32:
                    g = p.grad = g.contiguous().float()
                state = self.state[p]
                if not state:
                    state['step'] = 0
                    state['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                prev_step = int(state['step'])
                if self._hyper is None:                     # capturable mode counts on the device (bias corrections come from there)
                    state['step'] = prev_step + 1
                sp, sk, _ = shadows.get(id(p), (0, 0, None))
                gptr = g.data_ptr()
                wire = self.wire_grads.get(id(p)) if self.wire_grads else None
                if wire is not None:                        # data-parallel captured step: the all-reduced bf16 sum, where RCCL left it (dp.py)
                    gptr, wire_flag = wire, 0x100           # VQA_OPT_GRAD_BF16
                else:
                    wire_flag = 0
                    spans.append((id(p), gptr, gptr + 4 * p.numel()))
                updated.add(id(p))
                if sp == 0:                                 # stand-alone parameter (tail ops, experts): its cached bf16 copy, if any
                    sh = _ops.standalone_shadow(p)
                    if sh is not None and sh.numel() == p.numel():
                        sp, sk = sh.data_ptr(), 0
                        standalone.append(p)
                wd_kind = struct.unpack('<q', struct.pack('<fI', float(group['weight_decay']), sk | wire_flag))[0]
                act = getattr(p, '_vqa_active', None)       # device word: routed-token count of the parameter's expert (dense MoE dispatch)
                own = getattr(p, '_vqa_step', None)         # device word: the expert's own update count (bias corrections)
                if own is not None:
                    a_all, s_all, e = p._vqa_counts
                    ent = counted.get(id(s_all))
                    if ent is None:
                        ent = counted[id(s_all)] = (a_all, s_all, {})
                    ent[2].setdefault(e, prev_step)
                launches.setdefault((gi, state['step']) if self._hyper is None else (gi, -1 - self._class_of(gi, p)), []).append(
                    (p.data_ptr(), gptr, state['exp_avg'].data_ptr(), state['exp_avg_sq'].data_ptr(), sp, p.numel(), wd_kind,
                     act.data_ptr() if act is not None else 0, own.data_ptr() if own is not None else 0, self._touched_map(p, state), id(p)))
                dev = p.device
        if not launches:
            return loss
        if self._norm2 is None or self._norm2.device != dev:
            self._norm2 = torch.zeros(1, dtype=torch.float32, device=dev)
        tables = []
        live = set()
        # weight gradients whose sum of squares the GEMMs that wrote them have already taken (fuse_wgrad_norm): left out of the norm pass below
        in_gemm = None if self.wire_grads else self._norm_counted(spans)
        for (gi, step), rows9 in launches.items():
            rows = [r[:10] for r in rows9]
            flags = tuple(in_gemm is not None and r[10] in in_gemm for r in rows9)
            key = (tuple(rows), flags)
            slot = (gi, len(tables))
            live.add(slot)
            cached = self._tables.get(slot)
            if cached is None or cached[0] != key:
                ch = lib.vqa_opt_chunk_elems()
                chunks = [(ji, off) for ji, r in enumerate(rows) for off in range(0, r[5], ch)]
                nrm = [c for c in chunks if not flags[c[0]]]
                stage = self._staging.get(slot)
                if stage is not None and cached is not None and stage.shape[0] == len(rows) and cached[3] == len(chunks) and cached[0][1] == flags:
                    # HIP-graph mode: same tensors, new addresses (gradients allocated from the graph's pool).  No
                    # allocation is legal inside a stream capture: refill the pre-pinned staging rows and copy them
                    # over the existing device table (a memcpy node that replays harmlessly).
                    stage.copy_(torch.tensor(rows, dtype=torch.int64))
                    cached[1].copy_(stage, non_blocking=True)
                    cached = (key,) + tuple(cached[1:])
                else:
                    cached = (key, torch.tensor(rows, dtype=torch.int64).to(dev), torch.tensor(chunks, dtype=torch.int32).to(dev), len(chunks),
                              torch.tensor(nrm, dtype=torch.int32).to(dev) if nrm and len(nrm) < len(chunks) else None, len(nrm))
                self._tables[slot] = cached
            tables.append((self.param_groups[gi], step, cached[1], cached[2], cached[3], gi, cached[4], cached[5]))
        clip = self.max_grad_norm is not None and self.max_grad_norm > 0
        amp = None
        if self._amp_cfg is not None:
            amp = self._amp_state(dev)
        if clip or amp is not None:
            if in_gemm is not None:
                torch.sum(self._wnorm2, dim=0, keepdim=True, out=self._norm2)      # the weight gradients' share, summed by the GEMMs that stored them
            else:
                self._norm2.zero_()
            for _, _, tab, chunks, nch, _, nrm, nnrm in tables:
                if nnrm == nch:
                    K._chk(lib.vqa_sumsq_multi(tab.data_ptr(), chunks.data_ptr(), nch, self._norm2.data_ptr(), st), 'vqa_sumsq_multi')
                elif nnrm > 0:
                    K._chk(lib.vqa_sumsq_multi(tab.data_ptr(), nrm.data_ptr(), nnrm, self._norm2.data_ptr(), st), 'vqa_sumsq_multi')
        for a_all, s_all, first in counted.values():
            if not getattr(s_all, '_vqa_seeded', False):   # first sight (never inside a capture: the warm-up steps run eagerly): start
                for e, n0 in first.items():                 # each expert's device count from what its parameters have received so far
                    s_all[e:e + 1].fill_(float(n0))
                s_all._vqa_seeded = True
            K._chk(lib.vqa_opt_advance_counts(s_all.data_ptr(), a_all.data_ptr(), s_all.numel(), self._norm2.data_ptr() if amp is not None else None, st),
                   'vqa_opt_advance_counts')
        for group, step, tab, chunks, nch, gi, _nrm, _nnrm in tables:
            b1, b2 = group['betas']
            hyper = None
            if self._hyper is not None:
                hyper = self._hyper[-1 - step]                 # the table's step class (keys of device-side tables are (group, -1 - class))
                # device-side step += 1 (captured with the step) unless this step's gradients are non-finite
                K._chk(lib.vqa_opt_advance(hyper.data_ptr(), self._norm2.data_ptr() if amp is not None else None, st), 'vqa_opt_advance')
            K._chk(lib.vqa_adamw_multi(tab.data_ptr(), chunks.data_ptr(), nch, self._norm2.data_ptr() if (clip or amp is not None) else None,
                                       float(self.max_grad_norm or 0.0) if clip else 0.0, float(group['lr']), b1, b2, group['eps'],
                                       1.0 - b1 ** max(step, 1), 1.0 - b2 ** max(step, 1), hyper.data_ptr() if hyper is not None else None,
                                       float(self.grad_prescale), amp.data_ptr() if amp is not None else None, st),
                   'vqa_adamw_multi')
        if amp is not None:
            _, dynamic, growth, backoff, interval = self._amp_cfg
            if dynamic:
                K._chk(lib.vqa_amp_update(amp.data_ptr(), self._norm2.data_ptr(), growth, backoff, interval, st), 'vqa_amp_update')
            else:                                           # static scale: only record found_inf
                K._chk(lib.vqa_amp_update(amp.data_ptr(), self._norm2.data_ptr(), 1.0, 1.0, 1 << 30, st), 'vqa_amp_update')
        _ops.bump_shadow_generation()                      # stand-alone bf16 shadows this step did not write are stale now
        gen = _shadow.bump_generation()
        for ss in shadow_sets:                             # sets whose every updated parameter was written here stay current
            if all(id(p) not in updated or shadows[id(p)][2] is ss for _, p, _, _, _ in ss._plan):
                ss._gen = gen
        for p in standalone:
            _ops.mark_shadow_fresh(p)
        return loss

    def _touched_map(self, p, state):
        """Device address of the parameter's touched-granule map (VqaOptJob::touched), or 0.  Only for parameters the model marks
        ``_vqa_sparse_rows`` (embedding tables whose gradient is non-zero in a few rows per step): one byte per 256 elements, zero while the
        granule's moments are exactly zero; the update kernel skips the moments' 16 B per parameter for granules with an all-zero gradient and
        sets the byte at the first non-zero value.  Built from the moments themselves (a fresh state: all zero; a loaded checkpoint: where
        they are non-zero), never part of ``state_dict()``."""
        if not getattr(p, '_vqa_sparse_rows', False) or self.wire_grads or not self.sparse_row_updates:
            return 0
        t = self._touched.get(id(p))
        if t is None or t[1] is not state['exp_avg']:
            n, g = p.numel(), (p.numel() + 255) // 256
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError('FusedAdamW: the touched-granule map of a sparse-row parameter must exist before a stream capture (run a warm-up step)')
            nz = ((state['exp_avg'].reshape(-1) != 0) | (state['exp_avg_sq'].reshape(-1) != 0))
            pad = torch.zeros(g * 256, dtype=torch.bool, device=p.device)
            pad[:n] = nz
            t = self._touched[id(p)] = (pad.view(g, 256).any(dim=1).to(torch.uint8).contiguous(), state['exp_avg'])
        return t[0].data_ptr()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._touched.clear()                              # rebuilt from the loaded moments at the next step

    def norm_coverage(self):
        """(elements the clipping norm's own pass reads, elements of all gradients) of the last step's job tables: with the weight gradients' sum of
        squares taken in their GEMMs (fuse_wgrad_norm) the first number is what is left -- embedding tables, biases, LayerNorm affine."""
        ch = _hl.load().vqa_opt_chunk_elems()
        read = total = 0
        for key, rows, chunks, nch, nrm, nnrm in self._tables.values():
            n = sum(r[5] for r in key[0])
            total += n
            read += n if nnrm == nch else sum(min(ch, key[0][ji][5] - off) for ji, off in (nrm.tolist() if nrm is not None else []))
        return read, total

    def grad_norm(self):
        """Global gradient norm of the last clipped step (device scalar)."""
        if self._norm2 is None:
            return None
        return self._norm2.sqrt() * self.grad_prescale / (self._amp[0] if self._amp is not None else 1.0)
