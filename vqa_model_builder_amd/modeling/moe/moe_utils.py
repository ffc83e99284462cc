"""Helpers of the reference's ``src/modeling/moe/moe_utils.py`` (exported by ``src.modeling.moe``, moe/__init__.py:38-48,86-96;
``analyze_routing_patterns`` is imported by examples/moe_examples.py:292).

They are host-side bookkeeping over the router's SMALL outputs ([B, S, E] probabilities, [B, S, K] indices) -- auxiliary-loss
arithmetic, usage statistics, checkpoint wrappers -- not part of the forward/backward hot path, so they are plain tensor
expressions that work on whatever device the router outputs live on.  Same names, signatures, return types and values as the
reference (pinned by ``tests/golden/parts.npz``: oracle/gen_golden.py run_parts, tests/test_oracle_golden.py), formulated
without the reference's per-token Python loops.
"""

from typing import Any, Dict, List, Optional, Tuple

import torch
import torch.nn as nn


def compute_expert_capacity(num_tokens: int, num_experts: int, top_k: int, capacity_factor: float = 1.25) -> int:
    """Reference moe_utils.py:12-31: floor(capacity_factor * num_tokens * top_k / num_experts)."""
    return int(capacity_factor * num_tokens * top_k / num_experts)


def _assignment_counts(expert_indices: torch.Tensor, num_experts: int) -> torch.Tensor:
    """[num_experts] float: how many (token, slot) pairs chose each expert."""
    flat = expert_indices.reshape(-1)
    return torch.zeros(num_experts, dtype=torch.float32, device=flat.device).scatter_add_(0, flat, torch.ones_like(flat, dtype=torch.float32))


def compute_load_balance_loss(router_probs: torch.Tensor, expert_indices: torch.Tensor, num_experts: int, weight: float = 0.01) -> torch.Tensor:
    """Reference moe_utils.py:34-74: weight * E * sum_e (assignments_e / tokens) * mean_t probs[t, e] (differentiable in probs)."""
    num_tokens = router_probs.shape[0] * router_probs.shape[1]
    fraction = _assignment_counts(expert_indices, num_experts) / num_tokens
    mean_prob = router_probs.mean(dim=(0, 1))
    return weight * (num_experts * torch.sum(fraction * mean_prob))


def compute_router_z_loss(router_logits: torch.Tensor, weight: float = 0.001) -> torch.Tensor:
    """Reference moe_utils.py:77-94: weight * mean(logsumexp(logits)^2)."""
    return weight * torch.logsumexp(router_logits, dim=-1).pow(2).mean()


def get_expert_utilization(expert_indices: torch.Tensor, num_experts: int) -> Dict[int, float]:
    """Reference moe_utils.py:97-120: share of all (token, slot) assignments per expert id (one host read instead of E)."""
    flat = expert_indices.reshape(-1)
    ok = (flat >= 0) & (flat < num_experts)          # ids outside [0, E) (the ablation harness writes -1) count for nobody
    counts = torch.zeros(num_experts, dtype=torch.float32, device=flat.device).scatter_add_(0, flat.clamp(0, num_experts - 1), ok.float())
    total = expert_indices.numel()
    return {e: c / total for e, c in enumerate(counts.tolist())}


def compute_expert_entropy(router_probs: torch.Tensor) -> torch.Tensor:
    """Reference moe_utils.py:123-139: mean over tokens of -sum_e p log(p + 1e-10)."""
    return -(router_probs * torch.log(router_probs + 1e-10)).sum(dim=-1).mean()


class ExpertDropout(nn.Module):
    """Reference moe_utils.py:142-191: drops whole experts in training (one Bernoulli(1 - drop_rate) draw per expert per call),
    zeroes their routing weights and renormalises by the remaining sum (+1e-10); indices are returned unchanged."""

    def __init__(self, num_experts: int, drop_rate: float = 0.1):
        super().__init__()
        self.num_experts = num_experts
        self.drop_rate = drop_rate

    def forward(self, expert_weights: torch.Tensor, expert_indices: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        if not self.training or self.drop_rate == 0:
            return expert_weights, expert_indices
        keep = torch.bernoulli(torch.full((self.num_experts,), 1 - self.drop_rate, device=expert_indices.device))
        kept = expert_weights * keep[expert_indices]
        return kept / (kept.sum(dim=-1, keepdim=True) + 1e-10), expert_indices


class ExpertParallelWrapper(nn.Module):
    """Reference moe_utils.py:194-254: places expert i on device ``device_ids[min(i // (E // len(ids)), len(ids) - 1)]`` and
    round-trips the input through that device.  Nothing in the reference calls it; data parallelism over RCCL
    (vqa_model_builder_amd/dp.py) is how this build scales.  Kept because the name is exported."""

    def __init__(self, experts: nn.ModuleList, device_ids: Optional[List[int]] = None):
        super().__init__()
        self.experts = experts
        self.num_experts = len(experts)
        if device_ids is None:
            device_ids = list(range(torch.cuda.device_count()))
        self.device_ids = device_ids
        if len(device_ids) > 0:
            per_device = max(1, self.num_experts // len(device_ids))
            for i, expert in enumerate(experts):
                expert.to(f'cuda:{device_ids[min(i // per_device, len(device_ids) - 1)]}')

    def forward(self, x: torch.Tensor, expert_id: int, **kwargs) -> torch.Tensor:
        expert = self.experts[expert_id]
        device = next(expert.parameters()).device
        return expert(x.to(device), **kwargs).to(x.device)


def save_moe_checkpoint(moe_layer: nn.Module, path: str, additional_info: Optional[Dict] = None):
    """Reference moe_utils.py:257-280: torch.save of {'state_dict', 'num_experts', 'input_dim', 'output_dim'[, 'additional_info']}."""
    checkpoint = {'state_dict': moe_layer.state_dict(), 'num_experts': moe_layer.num_experts,
                  'input_dim': moe_layer.input_dim, 'output_dim': moe_layer.output_dim}
    if additional_info:
        checkpoint['additional_info'] = additional_info
    torch.save(checkpoint, path)


def load_moe_checkpoint(moe_layer: nn.Module, path: str, strict: bool = True) -> Dict:
    """Reference moe_utils.py:283-302.  The file holds tensors and plain containers only, so it is read with
    ``weights_only=True`` (nothing in the file is executed)."""
    checkpoint = torch.load(path, map_location='cpu', weights_only=True)
    moe_layer.load_state_dict(checkpoint['state_dict'], strict=strict)
    return checkpoint.get('additional_info', {})


def analyze_routing_patterns(router_probs: torch.Tensor, expert_indices: torch.Tensor, num_experts: int) -> Dict[str, Any]:
    """Reference moe_utils.py:305-341: utilisation, mean routing entropy, mean of the per-token max / min probability and the
    symmetric expert co-selection matrix (how often experts e1, e2 sit in two different slots of the same token; the diagonal
    counts a token that lists one expert twice, twice -- as the reference's pair loop does)."""
    analysis = {
        'expert_utilization': get_expert_utilization(expert_indices, num_experts),
        'routing_entropy': compute_expert_entropy(router_probs).item(),
        'max_prob_mean': router_probs.max(dim=-1).values.mean().item(),
        'min_prob_mean': router_probs.min(dim=-1).values.mean().item(),
    }
    flat = expert_indices.reshape(-1, expert_indices.size(-1))
    onehot = torch.zeros(flat.size(0), num_experts, device=flat.device).scatter_add_(1, flat, torch.ones_like(flat, dtype=torch.float32))
    co = onehot.t() @ onehot                                   # counts ordered slot pairs (j, k), j == k included
    co = co - torch.diag(onehot.sum(0))                        # remove j == k: what is left is every unordered pair, both ways
    analysis['expert_co_selection'] = co.cpu().numpy().tolist()
    return analysis
