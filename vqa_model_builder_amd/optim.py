"""Fused AdamW + global-norm clipping + weight-shadow refresh: the optimiser half of the training step as two HIP
launches per parameter group instead of torch's clip (norm + scale) + fused-AdamW + per-forward shadow casts.

Semantics = ``torch.nn.utils.clip_grad_norm_(params, max_norm)`` followed by ``torch.optim.AdamW.step()`` (the pair the
reference's loop runs, training_pipeline.py:497-502; param groups by name substring, :239-252): checked against torch in
tests/test_kernels_gpu.py::test_fused_adamw_matches_torch.  HBM-bound: 28 B/param (p, g, m, v read; p, m, v written)
+ 2 B/param for the bf16 shadow the next forward's GEMMs read, so the shadow refresh costs no extra pass.
State layout (``step`` / ``exp_avg`` / ``exp_avg_sq`` per parameter) matches torch's AdamW, so optimiser checkpoints
are interchangeable.
"""

import struct

import torch

from .hip import kernels as K
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
        self._hyper = None                                # HIP-graph mode: per group device floats {lr, step}
        self._staging = {}                                # HIP-graph mode: pinned host rows per job table
        self.grad_prescale = 1.0                          # DP: gradients hold the all-reduced SUM; 1/world is applied here
        self.wire_grads = None                            # DP captured step with bf16 buckets: id(parameter) -> device address of its all-reduced
                                                          # bfloat16 gradient in the exchange's staging buffer (read there: no copy back to fp32)

    def attach_shadows(self, model):
        """Lets the update kernel write the bf16 / packed-fp32 weight shadows of the block runners directly."""
        self._shadow_sets = [m._W.shadows for m in model.modules() if hasattr(m, '_W')]
        # nodes built lazily at the first forward (the model's tail: vqa_model._Tail) hang off their host module's ``_tails``
        self._tail_hosts = [m for m in model.modules() if hasattr(type(m), '_tail')]
        return self

    def make_capturable(self, device):
        """HIP-graph mode: learning rate and step count live in device memory ({lr, step} per group), are advanced by a
        device-side add that is part of the captured step, and the update kernel derives the bias corrections from them
        -- a replayed graph then follows the schedule instead of repeating the captured step's constants.  Call before
        the capture, after any eager warm-up steps; all parameters of a group must share one step count."""
        self._hyper = []
        for group in self.param_groups:
            steps = {int(self.state[p]['step']) for p in group['params'] if p in self.state and self.state[p] and not hasattr(p, '_vqa_step')}
            if len(steps) > 1:
                raise RuntimeError('FusedAdamW.make_capturable: parameters of one group have different step counts')
            self._hyper.append(torch.tensor([float(group['lr']), float(steps.pop() if steps else 0)], dtype=torch.float32).to(device))
        self._staging = {slot: torch.empty(tuple(c[1].shape), dtype=torch.int64).pin_memory() for slot, c in self._tables.items()}
        return self

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
        """Copies the groups' current ``lr`` to the device words (call between replays when a scheduler changed it)."""
        if self._hyper is not None:
            for group, h in zip(self.param_groups, self._hyper):
                h[0:1].fill_(float(group['lr']))

    def note_replays(self, n=1):
        """Kept for callers of round 1: the step count of a capturable optimiser lives on the device (``hyper[1]``) and the
        python-side ``state[p]['step']`` is read back from it on demand (``sync_step_counts``), so nothing to note."""

    def sync_step_counts(self):
        """Capturable mode: ``state[p]['step']`` := the device step count of the parameter's group -- the number of updates
        actually APPLIED (the capture pass applies none; a step skipped for non-finite fp16 gradients does not count).  One
        host read per group; called by ``state_dict()``."""
        if self._hyper is None:
            return
        for group, h in zip(self.param_groups, self._hyper):
            n = int(round(float(h[1])))
            for p in group['params']:
                if p in self.state and self.state[p]:
                    own = getattr(p, '_vqa_step', None)      # an expert's parameters: the updates the expert received
                    self.state[p]['step'] = n if own is None else int(round(float(own)))

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
        if self._amp_cfg is not None and self._hyper is None:
            # a step skipped for non-finite gradients must not advance the bias corrections: count on the device from the start
            first = next((p for g in self.param_groups for p in g['params'] if p.grad is not None), None)
            if first is not None:
                self.make_capturable(first.device)
        shadows, shadow_sets = self._shadow_map()
        updated = set()
        launches, dev = {}, None                        # (group index, step count) -> rows: torch's bias correction is per-parameter
        counted = {}                                    # MoE layers under dense dispatch: id(steps) -> (active [E], steps [E], {expert: step so far})
        standalone = []
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
                launches.setdefault((gi, state['step']), []).append(
                    (p.data_ptr(), gptr, state['exp_avg'].data_ptr(), state['exp_avg_sq'].data_ptr(), sp, p.numel(), wd_kind,
                     act.data_ptr() if act is not None else 0, own.data_ptr() if own is not None else 0))
                dev = p.device
        if not launches:
            return loss
        if self._norm2 is None or self._norm2.device != dev:
            self._norm2 = torch.zeros(1, dtype=torch.float32, device=dev)
        tables = []
        live = set()
        for (gi, step), rows in launches.items():
            key = tuple(rows)
            slot = (gi, len(tables))
            live.add(slot)
            cached = self._tables.get(slot)
            if cached is None or cached[0] != key:
                ch = lib.vqa_opt_chunk_elems()
                chunks = [(ji, off) for ji, r in enumerate(rows) for off in range(0, r[5], ch)]
                stage = self._staging.get(slot)
                if stage is not None and cached is not None and stage.shape[0] == len(rows) and cached[3] == len(chunks):
                    # HIP-graph mode: same tensors, new addresses (gradients allocated from the graph's pool).  No
                    # allocation is legal inside a stream capture: refill the pre-pinned staging rows and copy them
                    # over the existing device table (a memcpy node that replays harmlessly).
                    stage.copy_(torch.tensor(rows, dtype=torch.int64))
                    cached[1].copy_(stage, non_blocking=True)
                    cached = (key, cached[1], cached[2], cached[3])
                else:
                    cached = (key, torch.tensor(rows, dtype=torch.int64).to(dev), torch.tensor(chunks, dtype=torch.int32).to(dev), len(chunks))
                self._tables[slot] = cached
            tables.append((self.param_groups[gi], step, cached[1], cached[2], cached[3], gi))
        if self._hyper is not None and len(tables) != len({t[5] for t in tables}):
            raise RuntimeError('FusedAdamW (capturable): parameters of one group have different step counts')
        clip = self.max_grad_norm is not None and self.max_grad_norm > 0
        amp = None
        if self._amp_cfg is not None:
            amp = self._amp_state(dev)
        if clip or amp is not None:
            self._norm2.zero_()
            for _, _, tab, chunks, nch, _ in tables:
                K._chk(lib.vqa_sumsq_multi(tab.data_ptr(), chunks.data_ptr(), nch, self._norm2.data_ptr(), st), 'vqa_sumsq_multi')
        for a_all, s_all, first in counted.values():
            if not getattr(s_all, '_vqa_seeded', False):   # first sight (never inside a capture: the warm-up steps run eagerly): start
                for e, n0 in first.items():                 # each expert's device count from what its parameters have received so far
                    s_all[e:e + 1].fill_(float(n0))
                s_all._vqa_seeded = True
            K._chk(lib.vqa_opt_advance_counts(s_all.data_ptr(), a_all.data_ptr(), s_all.numel(), self._norm2.data_ptr() if amp is not None else None, st),
                   'vqa_opt_advance_counts')
        for group, step, tab, chunks, nch, gi in tables:
            b1, b2 = group['betas']
            hyper = None
            if self._hyper is not None:
                hyper = self._hyper[gi]
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

    def grad_norm(self):
        """Global gradient norm of the last clipped step (device scalar)."""
        if self._norm2 is None:
            return None
        return self._norm2.sqrt() * self.grad_prescale / (self._amp[0] if self._amp is not None else 1.0)
