"""Mixture-of-Experts layers on the HIP path (reference src/modeling/moe/moe_layer.py: MOELayer :29-196,
SparseMOELayer :199-358, VQAMOELayer :551-692).

Contracts kept for the ablation harness (ablation_trainer.py:112-305): ``self.router`` is looked up and called as
``self.router(x)`` on every forward (it gets monkey-patched and swapped), any K <= E and arbitrary (also zero)
routing weights are accepted, ``expert_indices == -1`` routes nowhere, ``aux_outputs`` / ``get_aux_loss()`` /
``get_expert_usage()`` / ``experts`` / ``input_dim`` / ``num_experts`` / ``top_k`` exist.

Execution: the reference runs every selected expert on ALL tokens and multiplies by a weight that is 0 for unrouted
ones (moe_layer.py:151-168).  Here tokens are dispatched: each expert runs only on the rows routed to it, which is
numerically equivalent whenever the expert does not mix tokens across the sequence axis -- always at S = 1 (the
classification model, vqa_model.py:674) and for token-local experts at any S.  Attention-bearing
experts at S > 1 take the reference's dense route so their cross-token attention sees the same rows.
"""

from typing import Any, Dict, Optional

import torch
import torch.nn as nn

from ...hip import kernels as K
from ...hip import ops
from .experts import MultimodalExpert, TextExpert, VisionExpert, create_expert
from .router import NoisyTopKRouter, create_router


class _GatherRowsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x2, lst, n):
        T, D = x2.shape
        out, _ = K.gather_rows(x2, lst, n, D, want_f32=True, want_bf16=False)
        ctx.save_for_backward(lst)
        ctx.meta = (T, D, n)
        return out

    @staticmethod
    def backward(ctx, dy):
        (lst,) = ctx.saved_tensors
        T, D, n = ctx.meta
        dx = torch.zeros((T, D), dtype=torch.float32, device=dy.device)
        ones = torch.ones((T,), dtype=torch.float32, device=dy.device)
        dy = dy.contiguous().float()
        K._chk(K.L().vqa_moe_scatter_add(dy.data_ptr(), lst.data_ptr(), ones.data_ptr(), dx.data_ptr(), n, D, K._stream()), 'vqa_moe_scatter_add')
        return dx, None, None


class _CombineFn(torch.autograd.Function):
    """out[t] = sum_e w_all[e,t] * y_e[row of t]  -- one node for the whole weighted scatter (moe_layer.py:160-168)."""

    @staticmethod
    def forward(ctx, weights, indices, w_all, lists, counts, T, D, *ys):
        dev = weights.device
        out = torch.zeros((T, D), dtype=torch.float32, device=dev)
        st = K._stream()
        ys = [y.contiguous().float() if y is not None else None for y in ys]
        for e, y in enumerate(ys):
            if y is None:
                continue
            K._chk(K.L().vqa_moe_scatter_add(y.data_ptr(), lists[e].data_ptr(), w_all[e].data_ptr(), out.data_ptr(), counts[e], D, st),
                   'vqa_moe_scatter_add')
        ctx.save_for_backward(indices, w_all, lists, *[y for y in ys if y is not None])
        ctx.meta = (counts, T, D, [y is not None for y in ys], weights.shape)
        return out

    @staticmethod
    def backward(ctx, dout):
        indices, w_all, lists, *ys_saved = ctx.saved_tensors
        counts, T, D, present, wshape = ctx.meta
        dev, st = dout.device, K._stream()
        dout = dout.contiguous().float()
        E = w_all.shape[0]
        dw_all = torch.zeros((E, T), dtype=torch.float32, device=dev)
        dys, it = [], iter(ys_saved)
        for e, has in enumerate(present):
            if not has:
                dys.append(None)
                continue
            y = next(it)
            dy = torch.empty_like(y)
            K._chk(K.L().vqa_moe_combine_bwd(dout.data_ptr(), y.data_ptr(), lists[e].data_ptr(), w_all[e].data_ptr(), dy.data_ptr(), None,
                                             dw_all[e].data_ptr(), counts[e], D, st), 'vqa_moe_combine_bwd')
            dys.append(dy)
        Kk = indices.shape[-1]
        dweights = torch.empty((T, Kk), dtype=torch.float32, device=dev)
        K._chk(K.L().vqa_moe_route_weight_grad(dw_all.data_ptr(), indices.data_ptr(), dweights.data_ptr(), T, E, Kk, st), 'vqa_moe_route_weight_grad')
        return (dweights.view(wshape), None, None, None, None, None, None) + tuple(dys)


class _DenseCombineFn(torch.autograd.Function):
    """Dense dispatch: out[t] = sum_e w_all[e,t] * y_e[t] over ALL experts in one launch (w_all is 0 where an expert was not chosen:
    the reference's own formulation, moe_layer.py:151-168); backward in one launch + the routing-weight gather."""

    @staticmethod
    def forward(ctx, weights, indices, w_all, T, D, *ys):
        shapes = [y.shape for y in ys]
        ys = [y.reshape(T, D).contiguous().float() for y in ys]
        out = K.moe_dense_combine_fwd(ys, w_all, T, len(ys), D)
        ctx.save_for_backward(indices, w_all, *ys)
        ctx.meta = (T, D, weights.shape, shapes)
        return out

    @staticmethod
    def backward(ctx, dout):
        indices, w_all, *ys = ctx.saved_tensors
        T, D, wshape, shapes = ctx.meta
        E, Kk = len(ys), indices.shape[-1]
        dys, dw_all = K.moe_dense_combine_bwd(dout.contiguous().float(), ys, w_all, T, E, D)
        dweights = torch.empty((T, Kk), dtype=torch.float32, device=dout.device)
        K._chk(K.L().vqa_moe_route_weight_grad(dw_all.data_ptr(), indices.data_ptr(), dweights.data_ptr(), T, E, Kk, K._stream()), 'vqa_moe_route_weight_grad')
        return (dweights.view(wshape), None, None, None, None) + tuple(dy.view(sh) for dy, sh in zip(dys, shapes))


class MOELayer(nn.Module):
    """Reference moe_layer.py:29-196."""

    def __init__(self, config=None, input_dim=768, hidden_dim=3072, output_dim=768, num_experts=8, top_k=2, router_type='topk',
                 expert_type='feedforward', dropout=0.1, use_aux_loss=True, load_balance_weight=0.01):
        super().__init__()
        if config is not None:
            input_dim, hidden_dim, output_dim = config.input_dim, config.hidden_dim, config.output_dim
            num_experts, top_k, dropout = config.num_experts, config.num_experts_per_token, config.expert_dropout
            if config.router_config:
                router_type = config.router_config.router_type
                use_aux_loss = config.router_config.use_aux_loss
                load_balance_weight = config.router_config.load_balance_weight
        self.input_dim, self.hidden_dim, self.output_dim = input_dim, hidden_dim, output_dim
        self.num_experts, self.top_k = num_experts, top_k
        self.router = create_router(router_type=router_type, input_dim=input_dim, num_experts=num_experts, top_k=top_k,
                                    use_aux_loss=use_aux_loss, load_balance_weight=load_balance_weight)
        self.experts = nn.ModuleList([create_expert(expert_type=expert_type, input_dim=input_dim, hidden_dim=hidden_dim,
                                                    output_dim=output_dim, expert_id=i, dropout=dropout) for i in range(num_experts)])
        self.output_norm = nn.LayerNorm(output_dim)
        self.aux_outputs: Dict[str, Any] = {}

    # ---- dense dispatch (HIP-graph mode) -----------------------------------------------------------------------------
    # The sparse dispatch reads the per-expert token counts on the host (the reference syncs per expert, moe_layer.py:156):
    # a captured graph cannot.  On this path the MoE sees ONE token per sample (the fused vector, S = 1) and every expert's
    # GEMMs have M <= batch rows -- they are bound by reading the expert's weights, not by its rows -- so running every expert
    # on every token and combining with the routing weights (zero where an expert was not chosen) costs no time and needs no
    # host decision.  Outputs and gradients are those of the sparse dispatch; an expert NO token chose gets zero gradients
    # instead of none, and its parameters carry `_vqa_active` (device word = its routed-token count) so that FusedAdamW
    # leaves them alone exactly as it skips a grad-is-None parameter, and `_vqa_step` (device word = the number of updates the
    # expert received) so that its AdamW bias corrections follow ITS step count, as torch's per-parameter `step` does.
    dense_dispatch = False
    parallel_branches = 0          # dense dispatch: experts on side streams (0 off, 1 specialised experts, 2 every expert; profiles/r02/moe.md)

    def enable_dense_dispatch(self, on: bool = True):
        self.dense_dispatch = on
        if on and getattr(self, '_active', None) is None:
            dev = self.output_norm.weight.device
            self._active = torch.ones((len(self.experts),), dtype=torch.float32, device=dev)
            self._steps = torch.zeros((len(self.experts),), dtype=torch.float32, device=dev)     # per-expert update counts (FusedAdamW)
            for e, expert in enumerate(self.experts):
                for prm in expert.parameters():
                    prm._vqa_active = self._active[e:e + 1]
                    prm._vqa_step = self._steps[e:e + 1]
                    prm._vqa_counts = (self._active, self._steps, e)
        return self

    def _forward_dense(self, x, mask, w2, i2, B, S, D, kwargs):
        T, E = B * S, len(self.experts)
        dev = x.device
        w_all = torch.empty((E, T), dtype=torch.float32, device=dev)       # w_all[e,t] = sum_k w2[t,k] * (i2[t,k] == e); -1 routes nowhere
        lists = torch.empty((E, T), dtype=torch.int32, device=dev)
        counts = torch.empty((E,), dtype=torch.int32, device=dev)
        K._chk(K.L().vqa_moe_expert_tokens(w2.detach().data_ptr(), i2.data_ptr(), T, w2.shape[-1], E, w_all.data_ptr(), lists.data_ptr(),
                                           counts.data_ptr(), K._stream()), 'vqa_moe_expert_tokens')
        self._active.copy_(counts)                                          # routed-token counts: FusedAdamW skips an expert nobody chose
        def run(expert):
            if S == 1 or expert.token_local:
                return expert(x.reshape(T, 1, D), **kwargs)
            return expert(x, mask=mask, **kwargs)
        # parallel_branches 1: the specialised experts (a 2-3 layer decoder: ~4x the dependent launches of the others) on a side HIP
        # stream, the short experts on this one; 2: every expert but the first on a stream of its own.  Chains of latency-floor
        # launches side by side; each expert's backward follows it to its stream.
        mode = int(self.parallel_branches)
        side_ids = [e for e, ex in enumerate(self.experts) if (getattr(ex, 'long_chain', False) if mode == 1 else e > 0)] if mode else []
        if side_ids and len(side_ids) < E:
            main = torch.cuda.current_stream()
            if getattr(self, '_side_streams', None) is None:
                self._side_streams = {}
            ys = [None] * E
            for j, e in enumerate(side_ids):
                key = 0 if mode == 1 else j
                side = self._side_streams.get(key)
                if side is None:
                    side = self._side_streams[key] = torch.cuda.Stream()
                if mode != 1 or j == 0:
                    side.wait_stream(main)
                with torch.cuda.stream(side):
                    ys[e] = run(self.experts[e])
            for e, expert in enumerate(self.experts):
                if ys[e] is None:
                    ys[e] = run(expert)
            for side in self._side_streams.values():
                main.wait_stream(side)
            for e in side_ids:
                ys[e].record_stream(main)
        else:
            ys = [run(expert) for expert in self.experts]
        out = _DenseCombineFn.apply(w2, i2, w_all, T, self.output_dim, *ys)
        out = ops.layer_norm(out, self.output_norm.weight, self.output_norm.bias, self.output_norm.eps)
        return out.view(B, S, self.output_dim)

    def forward(self, x: torch.Tensor, mask: Optional[torch.Tensor] = None, **kwargs) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError('MOELayer: HIP path needs GPU tensors; no CPU fallback on the product path')
        B, S, D = x.shape
        T, E = B * S, len(self.experts)
        routing_weights, expert_indices, aux_outputs = self.router(x)
        self.aux_outputs = aux_outputs
        Kk = expert_indices.shape[-1]
        w2 = routing_weights.reshape(T, Kk).contiguous().float()
        i2 = expert_indices.reshape(T, Kk).contiguous().long()
        dev = x.device
        if self.dense_dispatch:
            return self._forward_dense(x, mask, w2, i2, B, S, D, kwargs)
        w_all = torch.empty((E, T), dtype=torch.float32, device=dev)
        lists = torch.empty((E, T), dtype=torch.int32, device=dev)
        counts_dev = torch.empty((E,), dtype=torch.int32, device=dev)
        K._chk(K.L().vqa_moe_expert_tokens(w2.detach().data_ptr(), i2.data_ptr(), T, Kk, E, w_all.data_ptr(), lists.data_ptr(),
                                           counts_dev.data_ptr(), K._stream()), 'vqa_moe_expert_tokens')
        counts = counts_dev.tolist()      # the reference syncs once per expert (`.any()`, moe_layer.py:156); here once per layer
        x2 = x.reshape(T, D)
        ys = []
        for e, expert in enumerate(self.experts):
            n = counts[e]
            if n == 0:                    # unrouted expert: skipped, its parameters get no gradient (F9)
                ys.append(None)
                continue
            if S == 1 or expert.token_local:
                xe = _GatherRowsFn.apply(x2, lists[e], n)
                ys.append(expert(xe.view(n, 1, D), **kwargs).reshape(n, self.output_dim))
            else:                         # cross-token attention inside the expert: dense route like the reference
                ye = expert(x, mask=mask, **kwargs).reshape(T, self.output_dim)
                ys.append(_GatherRowsFn.apply(ye, lists[e], n))
        out = _CombineFn.apply(w2, i2, w_all, lists, counts, T, self.output_dim, *ys)
        out = ops.layer_norm(out, self.output_norm.weight, self.output_norm.bias, self.output_norm.eps)
        return out.view(B, S, self.output_dim)

    def get_aux_loss(self) -> torch.Tensor:
        if 'load_balance_loss' in self.aux_outputs:
            return self.aux_outputs['load_balance_loss']
        return torch.tensor(0.0)

    def get_expert_usage(self) -> Dict[int, float]:
        return {i: e.get_usage_ratio() for i, e in enumerate(self.experts)}


class VQAMOELayer(MOELayer):
    """Reference moe_layer.py:551-692: one NoisyTopKRouter + {Vision, Text, Multimodal, specialised} experts."""

    def __init__(self, input_dim=768, hidden_dim=3072, output_dim=768, num_vision_experts=2, num_text_experts=2,
                 num_multimodal_experts=2, num_specialized_experts=2, top_k=2, dropout=0.1, vietnamese_optimized=True):
        nn.Module.__init__(self)
        total = num_vision_experts + num_text_experts + num_multimodal_experts + num_specialized_experts
        self.input_dim, self.hidden_dim, self.output_dim = input_dim, hidden_dim, output_dim
        self.num_experts, self.top_k = total, top_k
        self.router = NoisyTopKRouter(input_dim=input_dim, num_experts=total, top_k=top_k, use_aux_loss=True)
        from .experts import ObjectDetectionExpert, OCRExpert, SceneUnderstandingExpert, SegmentationExpert
        kinds = ([VisionExpert] * num_vision_experts + [TextExpert] * num_text_experts + [MultimodalExpert] * num_multimodal_experts)
        spec = [SegmentationExpert, ObjectDetectionExpert, OCRExpert, SceneUnderstandingExpert]
        kinds += [spec[i % len(spec)] for i in range(num_specialized_experts)]
        self.experts = nn.ModuleList()
        for eid, cls in enumerate(kinds):
            kw = dict(input_dim=input_dim, hidden_dim=hidden_dim, output_dim=output_dim, expert_id=eid, dropout=dropout)
            if cls is OCRExpert:
                kw['vietnamese_optimized'] = vietnamese_optimized
            self.experts.append(cls(**kw))
        self.output_norm = nn.LayerNorm(output_dim)
        self.aux_outputs: Dict[str, Any] = {}


class SparseMOELayer(nn.Module):
    """Reference moe_layer.py:199-358: NoisyTopKRouter + token dispatch with a per-expert capacity
    ``int(capacity_factor * tokens * top_k / num_experts)``: an expert over its capacity keeps the tokens with the largest routing
    weight (``torch.topk`` on the weights, :321-327), the rest contribute nothing (and their routing weight gets no gradient).
    Each expert sees its tokens as ONE sequence ``[1, C, D]`` (:334) -- irrelevant for token-local experts (the default
    'feedforward'), reproduced for attention-bearing expert types.  (The generative model's own construction of this class passes
    ``config=`` and raises ``TypeError`` in the reference: SURVEY F11 -- same here.)"""

    def __init__(self, input_dim: int = 768, hidden_dim: int = 3072, output_dim: int = 768, num_experts: int = 8, top_k: int = 2,
                 capacity_factor: float = 1.25, dropout: float = 0.1, use_aux_loss: bool = True, expert_type: str = 'feedforward'):
        super().__init__()
        self.input_dim, self.hidden_dim, self.output_dim = input_dim, hidden_dim, output_dim
        self.num_experts, self.top_k, self.capacity_factor = num_experts, top_k, capacity_factor
        self.router = NoisyTopKRouter(input_dim=input_dim, num_experts=num_experts, top_k=top_k, use_aux_loss=use_aux_loss)
        self.experts = nn.ModuleList([create_expert(expert_type=expert_type, input_dim=input_dim, hidden_dim=hidden_dim, output_dim=output_dim,
                                                    expert_id=i, dropout=dropout) for i in range(num_experts)])
        self.output_norm = nn.LayerNorm(output_dim)
        self.aux_outputs: Dict[str, Any] = {}

    def _compute_capacity(self, num_tokens: int) -> int:
        return int(self.capacity_factor * num_tokens * self.top_k / self.num_experts)

    def forward(self, x: torch.Tensor, mask: Optional[torch.Tensor] = None, **kwargs) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError('SparseMOELayer: HIP path needs GPU tensors; no CPU fallback on the product path')
        B, S, D = x.shape
        T, E = B * S, self.num_experts
        capacity = self._compute_capacity(T)
        routing_weights, expert_indices, aux_outputs = self.router(x)
        self.aux_outputs = aux_outputs
        Kk = expert_indices.shape[-1]
        w2 = routing_weights.reshape(T, Kk).contiguous().float()
        i2 = expert_indices.reshape(T, Kk).contiguous().long()
        dev = x.device
        w_all = torch.empty((E, T), dtype=torch.float32, device=dev)
        lists = torch.empty((E, T), dtype=torch.int32, device=dev)
        counts_dev = torch.empty((E,), dtype=torch.int32, device=dev)
        K._chk(K.L().vqa_moe_expert_tokens(w2.detach().data_ptr(), i2.data_ptr(), T, Kk, E, w_all.data_ptr(), lists.data_ptr(),
                                           counts_dev.data_ptr(), K._stream()), 'vqa_moe_expert_tokens')
        counts = counts_dev.tolist()              # one host sync per layer (the reference: `.any()` per expert + `len(token_indices)`)
        x2 = x.reshape(T, D)
        for e in range(E):                        # capacity cut first: the token lists are saved by the autograd nodes below
            n = counts[e]
            if n > capacity:                      # keep the `capacity` largest routing weights, in topk's order (:321-327)
                cand = lists[e, :n].long()
                top = torch.topk(w_all[e].index_select(0, cand), capacity).indices
                lists[e, :capacity] = cand.index_select(0, top).to(torch.int32)
                counts[e] = capacity
        ys = []
        for e, expert in enumerate(self.experts):
            n = counts[e]
            if n == 0:
                ys.append(None)
                continue
            xe = _GatherRowsFn.apply(x2, lists[e], n)
            if expert.token_local:
                ys.append(expert(xe.view(n, 1, D), **kwargs).reshape(n, self.output_dim))
            else:
                ys.append(expert(xe.view(1, n, D), **kwargs).reshape(n, self.output_dim))
        out = _CombineFn.apply(w2, i2, w_all, lists, counts, T, self.output_dim, *ys)
        out = ops.layer_norm(out, self.output_norm.weight, self.output_norm.bias, self.output_norm.eps)
        return out.view(B, S, self.output_dim)

    def get_aux_loss(self) -> torch.Tensor:
        if 'load_balance_loss' in self.aux_outputs:
            return self.aux_outputs['load_balance_loss']
        return torch.tensor(0.0)


class HierarchicalMOE(nn.Module):
    """Reference moe_layer.py:361-548 (examples): a TopKRouter over expert GROUPS, one TopKRouter per group over its experts, every
    chosen expert run on ALL tokens and weighted by group weight x expert weight where both routers chose it, then Linear + LayerNorm.

    out[t] = sum_k sum_g [group_k(t) = g] gw_k(t) * sum_j sum_e [expert^g_j(t) = e] ew^g_j(t) * expert_{g,e}(x)[t]
    The reference evaluates the same sum with one expert call per (k, g, j, e) combination (:470-507); here each expert that any
    token selects is called ONCE and its weights are summed first -- identical in eval mode (in train mode the reference draws
    fresh dropout masks per repeated call).  The load-balance term adds a group's router loss once per top-k slot that reaches the
    group (:480-484), as there.  Routers, experts, the output projection and LayerNorm are the HIP modules / ops; the routing masks
    and the weighted sum are plain torch tensor arithmetic."""

    def __init__(self, input_dim: int = 768, hidden_dim: int = 3072, output_dim: int = 768, num_expert_groups: int = 4,
                 experts_per_group: int = 4, top_k_groups: int = 2, top_k_experts: int = 1, dropout: float = 0.1, expert_types=None):
        super().__init__()
        from .router import TopKRouter
        self.input_dim, self.hidden_dim, self.output_dim = input_dim, hidden_dim, output_dim
        self.num_expert_groups, self.experts_per_group = num_expert_groups, experts_per_group
        self.top_k_groups, self.top_k_experts = top_k_groups, top_k_experts
        if expert_types is None:
            expert_types = ['vision', 'text', 'multimodal', 'feedforward']
            expert_types = (expert_types * num_expert_groups)[:num_expert_groups]
        self.group_router = TopKRouter(input_dim=input_dim, num_experts=num_expert_groups, top_k=top_k_groups, use_aux_loss=True)
        self.expert_routers = nn.ModuleList([TopKRouter(input_dim=input_dim, num_experts=experts_per_group, top_k=top_k_experts, use_aux_loss=True)
                                             for _ in range(num_expert_groups)])
        self.expert_groups = nn.ModuleList([
            nn.ModuleList([create_expert(expert_type=expert_types[g], input_dim=input_dim, hidden_dim=hidden_dim, output_dim=output_dim,
                                         expert_id=g * experts_per_group + e, dropout=dropout) for e in range(experts_per_group)])
            for g in range(num_expert_groups)])
        self.output_proj = nn.Linear(output_dim, output_dim)
        self.output_norm = nn.LayerNorm(output_dim)
        self.aux_outputs: Dict[str, Any] = {}

    def forward(self, x: torch.Tensor, mask: Optional[torch.Tensor] = None, **kwargs) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError('HierarchicalMOE: HIP path needs GPU tensors; no CPU fallback on the product path')
        B, S, _ = x.shape
        group_weights, group_indices, group_aux = self.group_router(x)
        zero = torch.zeros((), dtype=torch.float32, device=x.device)
        total_aux = group_aux.get('load_balance_loss', zero)
        # [G, B, S]: summed group weight of each token for each group, and how many top-k slots reach the group at all
        gsel = torch.stack([(group_indices == g) for g in range(self.num_expert_groups)])             # [G, B, S, Kg]
        gw = (gsel.to(group_weights.dtype) * group_weights.unsqueeze(0)).sum(-1)
        slots = gsel.flatten(1, 2).any(dim=1).sum(dim=1).tolist()                                     # host sync: the reference's `.any()` per (k, g)
        out = None
        for g in range(self.num_expert_groups):
            if slots[g] == 0:
                continue
            ew, ei, e_aux = self.expert_routers[g](x)
            total_aux = total_aux + slots[g] * e_aux.get('load_balance_loss', zero)
            esel = torch.stack([(ei == e) for e in range(self.experts_per_group)])                    # [Eg, B, S, Ke]
            cw = (esel.to(ew.dtype) * ew.unsqueeze(0)).sum(-1) * gw[g].unsqueeze(0)                   # [Eg, B, S]
            used = esel.flatten(1).any(dim=1).tolist()
            for e in range(self.experts_per_group):
                if not used[e]:
                    continue
                y = self.expert_groups[g][e](x, mask=mask, **kwargs) * cw[e].unsqueeze(-1)
                out = y if out is None else out + y
        if out is None:
            out = torch.zeros((B, S, self.output_dim), dtype=torch.float32, device=x.device)
        self.aux_outputs = {'load_balance_loss': total_aux, 'group_probs': group_aux.get('router_probs', None)}
        out = ops.linear(out, self.output_proj.weight, self.output_proj.bias)
        return ops.layer_norm(out, self.output_norm.weight, self.output_norm.bias, self.output_norm.eps)

    def get_aux_loss(self) -> torch.Tensor:
        return self.aux_outputs.get('load_balance_loss', torch.tensor(0.0))
