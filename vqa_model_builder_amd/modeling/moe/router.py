"""Expert routers on the HIP path.  Same classes / constructor signatures / return contract as the reference's
``src/modeling/moe/router.py`` (BaseRouter :14-72, TopKRouter :75-178, SoftRouter :181-248, NoisyTopKRouter
:251-366, ExpertChoiceRouter :369-449, create_router :452-494):

    routing_weights [B,S,K] fp32, expert_indices [B,S,K] int64, aux_outputs: dict = router(x)

The gate GEMV, softmax, top-k, renormalisation and their gradients run in fp32 kernels (csrc/moe.hip) so the
selected expert ids match the fp32 reference wherever its own margins allow.  The load-balance statistic is
returned as a detached scalar: in the reference it never reaches the classification loss.
"""

import inspect
from abc import ABC, abstractmethod
from typing import Any, Dict, Tuple

import torch
import torch.nn as nn

from ...hip import kernels as K
from ...hip.blocks import new_seed


def _lib():
    return K.L()


class _GateTopKFn(torch.autograd.Function):
    """x [T,D], gate [E,D], w_noise [E,D]|None, noise [T,E]|None -> weights [T,K], indices [T,K], probs_clean [T,E]"""

    @staticmethod
    def forward(ctx, x, gate, w_noise, noise, noise_std, top_k, soft, temperature, forced=None):
        T, D = x.shape
        E = gate.shape[0]
        dev, st = x.device, K._stream()
        x, gate = x.contiguous().float(), gate.contiguous()
        clean = torch.empty((T, E), dtype=torch.float32, device=dev)
        noisy = torch.empty_like(clean)
        nraw = torch.empty_like(clean) if noise is not None else None
        K._chk(_lib().vqa_router_gate_fwd(x.data_ptr(), gate.data_ptr(), K._p(w_noise), K._p(noise), noise_std, clean.data_ptr(),
                                          noisy.data_ptr(), K._p(nraw), T, E, D, st), 'vqa_router_gate_fwd')
        logits = noisy
        if soft and temperature != 1.0:
            logits = noisy / temperature          # scalar scale of a [T,E] tensor: plumbing
        kk = E if soft else top_k
        w = torch.empty((T, kk), dtype=torch.float32, device=dev)
        idx = torch.empty((T, kk), dtype=torch.int64, device=dev)
        probs = torch.empty((T, E), dtype=torch.float32, device=dev)
        K._chk(_lib().vqa_router_topk_fwd(logits.data_ptr(), w.data_ptr(), idx.data_ptr(), probs.data_ptr(), T, E, kk, st), 'vqa_router_topk_fwd')
        probs_clean = probs
        if noise is not None:
            probs_clean = torch.empty_like(probs)
            w2, i2 = torch.empty_like(w), torch.empty_like(idx)
            K._chk(_lib().vqa_router_topk_fwd(clean.data_ptr(), w2.data_ptr(), i2.data_ptr(), probs_clean.data_ptr(), T, E, kk, st), 'vqa_router_topk_fwd')
        if soft:
            # SoftRouter returns the weights in expert order with indices arange(E) (router.py:223-228)
            w_out = probs
            probs_clean = probs_clean.clone() if probs_clean is probs else probs_clean
            idx_out = torch.arange(E, device=dev).expand(T, E).contiguous()
        else:
            w_out, idx_out = w, idx
            if forced is not None:
                # parity tests only (BaseRouter._forced_indices): the discrete choice is GIVEN -- the reference's own, where a 16-bit run lands
                # inside a numerical tie of two router probabilities --; the weights are this run's probabilities of those experts,
                # renormalised (router.py:313-320), and backward differentiates exactly that (vqa_router_topk_bwd takes the indices as data)
                idx_out = forced.reshape(T, kk).to(torch.int64).contiguous()
                w_out = probs.gather(1, idx_out)
                w_out = w_out / w_out.sum(dim=1, keepdim=True)
        ctx.save_for_backward(x, gate, w_noise, noise, nraw, logits, idx_out)
        ctx.meta = (T, E, D, kk, noise_std, soft, temperature)
        ctx.mark_non_differentiable(idx_out, probs_clean)
        return w_out, idx_out, probs_clean

    @staticmethod
    def backward(ctx, dw, _di, _dp):
        x, gate, w_noise, noise, nraw, logits, idx = ctx.saved_tensors
        T, E, D, kk, noise_std, soft, temperature = ctx.meta
        dev, st = x.device, K._stream()
        dlog = torch.empty((T, E), dtype=torch.float32, device=dev)
        dw = dw.contiguous().float()
        K._chk(_lib().vqa_router_topk_bwd(logits.data_ptr(), idx.data_ptr(), dw.data_ptr(), dlog.data_ptr(), T, E, kk, st), 'vqa_router_topk_bwd')
        if soft and temperature != 1.0:
            dlog = dlog / temperature
        dgate = torch.empty_like(gate)
        dwn = torch.empty_like(w_noise) if w_noise is not None else None
        dx = torch.empty_like(x)
        K._chk(_lib().vqa_router_gate_bwd(x.data_ptr(), gate.data_ptr(), K._p(w_noise), K._p(noise), noise_std, K._p(nraw), dlog.data_ptr(),
                                          dgate.data_ptr(), K._p(dwn), dx.data_ptr(), T, E, D, st), 'vqa_router_gate_bwd')
        return dx, dgate, (dwn if noise is not None else None), None, None, None, None, None, None


def _aux_loss(probs_clean, idx, E, kk, weight):
    T = probs_clean.shape[0]
    out = torch.empty((), dtype=torch.float32, device=probs_clean.device)
    K._chk(_lib().vqa_router_aux_loss(probs_clean.data_ptr(), idx.data_ptr(), T, E, kk, weight, out.data_ptr(), K._stream()), 'vqa_router_aux_loss')
    return out


class BaseRouter(ABC, nn.Module):
    """Reference router.py:14-72 (bias-free ``gate``)."""

    def __init__(self, input_dim: int, num_experts: int, top_k: int = 2):
        super().__init__()
        self.input_dim, self.num_experts, self.top_k = input_dim, num_experts, top_k
        self.gate = nn.Linear(input_dim, num_experts, bias=False)
        self._forced_indices = None       # tests: int64 [B,S,K] standing for torch.topk's choice (see _GateTopKFn.forward)

    @abstractmethod
    def forward(self, x: torch.Tensor, **kwargs) -> Tuple[torch.Tensor, torch.Tensor, Dict[str, Any]]:
        pass

    def _route(self, x, w_noise=None, noise=None, noise_std=1.0, soft=False, temperature=1.0):
        if not x.is_cuda:
            raise RuntimeError('router: HIP path needs GPU tensors; no CPU fallback on the product path')
        B, S, D = x.shape
        w, idx, probs = _GateTopKFn.apply(x.reshape(B * S, D), self.gate.weight, w_noise, noise, noise_std, self.top_k, soft, temperature,
                                          None if soft else self._forced_indices)
        kk = w.shape[-1]
        return w.view(B, S, kk), idx.view(B, S, kk), probs.view(B, S, -1)


class TopKRouter(BaseRouter):
    """Reference router.py:75-178."""

    def __init__(self, input_dim, num_experts, top_k=2, use_aux_loss=True, load_balance_weight=0.01):
        super().__init__(input_dim, num_experts, top_k)
        self.use_aux_loss, self.load_balance_weight = use_aux_loss, load_balance_weight

    def forward(self, x, **kwargs):
        w, idx, probs = self._route(x)
        aux = {}
        if self.use_aux_loss:
            aux['load_balance_loss'] = _aux_loss(probs.reshape(-1, self.num_experts), idx.reshape(-1, self.top_k), self.num_experts,
                                                 self.top_k, self.load_balance_weight)
            aux['router_probs'] = probs
        return w, idx, aux


class SoftRouter(BaseRouter):
    """Reference router.py:181-248: every expert, softmax(logits / temperature)."""

    def __init__(self, input_dim, num_experts, temperature=1.0):
        super().__init__(input_dim, num_experts, num_experts)
        self.temperature = temperature

    def forward(self, x, **kwargs):
        w, idx, _ = self._route(x, soft=True, temperature=self.temperature)
        wd = w.detach()
        return w, idx, {'router_probs': w, 'entropy': (-(wd * torch.log(wd + 1e-10)).sum(-1)).mean()}


class NoisyTopKRouter(BaseRouter):
    """Reference router.py:251-366: train-mode logits += randn * softplus(w_noise(x)) * noise_std."""

    def __init__(self, input_dim, num_experts, top_k=2, noise_std=1.0, use_aux_loss=True, load_balance_weight=0.01):
        super().__init__(input_dim, num_experts, top_k)
        self.noise_std, self.use_aux_loss, self.load_balance_weight = noise_std, use_aux_loss, load_balance_weight
        self.w_noise = nn.Linear(input_dim, num_experts, bias=False)
        self._injected_noise = None       # tests: a [B,S,E] tensor standing for torch.randn_like

    def forward(self, x, **kwargs):
        noise = None
        if self.training:
            B, S, _ = x.shape
            if self._injected_noise is not None:
                noise = self._injected_noise.reshape(B * S, self.num_experts).contiguous().float()
            else:
                noise = K.randn((B * S, self.num_experts), new_seed(), 55, x.device)
        w, idx, probs = self._route(x, self.w_noise.weight if noise is not None else None, noise, self.noise_std)
        aux = {}
        if self.use_aux_loss:
            aux['load_balance_loss'] = _aux_loss(probs.reshape(-1, self.num_experts), idx.reshape(-1, self.top_k), self.num_experts,
                                                 self.top_k, self.load_balance_weight)
            aux['router_probs'] = probs
            aux['noise_scale'] = 0.0
        return w, idx, aux


class ExpertChoiceRouter(BaseRouter):
    """Reference router.py:369-449 (ablation-only).  Gate logits come from the HIP gate kernel; the per-expert
    token selection is a [T,E] bookkeeping loop kept in torch ops (softmax over the SEQUENCE axis, K=1)."""

    def __init__(self, input_dim, num_experts, capacity_factor=1.25):
        super().__init__(input_dim, num_experts, 1)
        self.capacity_factor = capacity_factor

    def forward(self, x, **kwargs):
        B, S, D = x.shape
        T = B * S
        capacity = int(self.capacity_factor * T / self.num_experts)
        if not x.is_cuda:
            raise RuntimeError('router: HIP path needs GPU tensors; no CPU fallback on the product path')
        clean = torch.empty((T, self.num_experts), dtype=torch.float32, device=x.device)
        noisy = torch.empty_like(clean)
        xx = x.reshape(T, D).contiguous().float()
        K._chk(_lib().vqa_router_gate_fwd(xx.data_ptr(), self.gate.weight.data_ptr(), None, None, 1.0, clean.data_ptr(), noisy.data_ptr(),
                                          None, T, self.num_experts, D, K._stream()), 'vqa_router_gate_fwd')
        scores = torch.softmax(clean.view(B, S, -1), dim=1)
        flat = scores.view(T, self.num_experts)
        idx = torch.zeros(T, dtype=torch.long, device=x.device)
        w = torch.zeros(T, device=x.device)
        for e in range(self.num_experts):
            top_s, top_i = torch.topk(flat[:, e], min(capacity, T), dim=0)
            idx[top_i] = e
            w[top_i] = top_s
        return w.view(B, S, 1), idx.view(B, S, 1), {'router_probs': scores, 'capacity': capacity}


def create_router(router_type: str, input_dim: int, num_experts: int, **kwargs) -> BaseRouter:
    """Reference router.py:452-494: unknown kwargs are dropped, unknown type raises ValueError."""
    routers = {'topk': TopKRouter, 'soft': SoftRouter, 'noisy_topk': NoisyTopKRouter, 'expert_choice': ExpertChoiceRouter}
    if router_type not in routers:
        raise ValueError(f"Unknown router type: {router_type}. Available: {list(routers.keys())}")
    cls = routers[router_type]
    valid = set(inspect.signature(cls.__init__).parameters) - {'self'}
    return cls(input_dim, num_experts, **{k: v for k, v in kwargs.items() if k in valid})
