"""bf16 weight shadows.

The parameters the callers see (optimiser, clip_grad_norm_, state_dict) stay ordinary fp32 ``nn.Parameter``s
(SURVEY.md section 8b "Ownership"); the MFMA GEMMs read bf16 copies.  A ShadowSet owns ONE bf16 arena and ONE fp32
arena per block, knows which parameter lands where (including packed layouts such as q|k|v -> [3D, D]) and refreshes
every stale copy with a single multi-tensor cast launch -- the equivalent of autocast's per-forward weight cast,
minus the per-tensor launches and re-casts of unchanged weights.
"""

from typing import Dict, List, Tuple

import torch

from . import kernels as K


# A fused optimiser updates parameters through raw pointers (no autograd version bump) and writes the bf16 copy of ONE set per
# parameter itself; it then bumps the generation and stamps the sets it kept current.  A set that shares a parameter with the stamped
# one (the same Linear reached through two different nodes) sees the generation move on and re-casts at its next use.
_generation = 0


def bump_generation() -> int:
    global _generation
    _generation += 1
    return _generation


class ShadowSet:
    _uses = 0

    def __init__(self):
        self._plan: List[Tuple[str, torch.nn.Parameter, str, int, int]] = []   # (key, param, arena, offset, numel)
        self._views: Dict[str, Tuple[str, int, Tuple[int, ...]]] = {}          # key -> (arena, offset, shape)
        self._size = {'bf16': 0, 'f32': 0}
        self._arena = {}
        self._jobs = None
        self._sig = None
        self._max_n = 0
        self._gen = 0                     # generation (above) at which the arenas were last known current
        self.last_use = 0                 # ordinal of the latest refresh(): the optimiser keeps the most recently used set current

    # -- planning (construction time, device-agnostic) --------------------------------------------------------
    def _reserve(self, arena: str, numel: int) -> int:
        off = self._size[arena]
        self._size[arena] = off + ((numel + 7) // 8) * 8            # keep every view 16-byte aligned
        return off

    def add(self, key: str, params, shape, arena: str = 'bf16'):
        """``params``: one parameter or a list packed back-to-back (row-wise concat) into a view of ``shape``."""
        if not isinstance(params, (list, tuple)):
            params = [params]
        total = sum(p.numel() for p in params)
        n = 1
        for s in shape:
            n *= s
        assert n == total, (key, shape, total)
        off = self._reserve(arena, total)
        self._views[key] = (arena, off, tuple(shape))
        o = off
        for p in params:
            self._plan.append((key, p, arena, o, p.numel()))
            o += p.numel()

    # -- run time ------------------------------------------------------------------------------------------------
    def _materialise(self, device):
        self._arena = {'bf16': torch.empty(max(self._size['bf16'], 8), dtype=K.HALF(), device=device),
                       'f32': torch.empty(max(self._size['f32'], 8), dtype=torch.float32, device=device)}
        rows = []
        for _, p, arena, off, n in self._plan:
            dst = self._arena[arena]
            rows.append([p.data_ptr(), dst.data_ptr() + off * dst.element_size(), n, 0 if arena == 'bf16' else 1])
        self._jobs = torch.tensor(rows, dtype=torch.int64).to(device)
        self._max_n = max(r[2] for r in rows)
        self._ptr_sig = tuple(p.data_ptr() for _, p, _, _, _ in self._plan)

    def refresh(self, device):
        """Re-casts if any source parameter changed (version counter or storage)."""
        ptr_sig = tuple(p.data_ptr() for _, p, _, _, _ in self._plan)
        if (self._jobs is None or ptr_sig != getattr(self, '_ptr_sig', None) or self._arena['bf16'].device != device
                or self._arena['bf16'].dtype != K.HALF()):        # operand type switched (hip.lib.set_half): new arena, new casts
            for _, p, _, _, _ in self._plan:
                if p.device != device:
                    raise RuntimeError(f'parameter on {p.device}, expected {device}: move the module to the GPU first')
            self._materialise(device)
            self._sig = None
        ShadowSet._uses += 1
        self.last_use = ShadowSet._uses
        sig = tuple(p._version for _, p, _, _, _ in self._plan)
        if sig != self._sig or self._gen != _generation:
            K.cast_multi(self._jobs, len(self._plan), self._max_n)
            self._sig, self._gen = sig, _generation

    def get(self, key: str) -> torch.Tensor:
        arena, off, shape = self._views[key]
        n = 1
        for s in shape:
            n *= s
        return self._arena[arena][off:off + n].view(shape)
