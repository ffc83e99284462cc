"""MoE configuration dataclasses (field-for-field mirror of reference src/modeling/moe/moe_config.py:10-158)."""

from dataclasses import dataclass, field
from typing import List, Optional


@dataclass
class ExpertConfig:
    expert_type: str = 'feedforward'
    input_dim: int = 768
    hidden_dim: int = 3072
    output_dim: int = 768
    num_layers: int = 2
    dropout: float = 0.1
    activation: str = 'gelu'
    use_layer_norm: bool = True
    expert_capacity: Optional[int] = None


@dataclass
class RouterConfig:
    router_type: str = 'topk'
    num_experts: int = 8
    top_k: int = 2
    noise_std: float = 1.0
    load_balance_weight: float = 0.01
    capacity_factor: float = 1.25
    use_aux_loss: bool = True
    jitter_noise: bool = True


@dataclass
class MOEConfig:
    input_dim: int = 768
    hidden_dim: int = 3072
    output_dim: int = 768
    num_experts: int = 8
    num_experts_per_token: int = 2
    expert_configs: Optional[List[ExpertConfig]] = None
    router_config: Optional[RouterConfig] = None
    use_sparse_moe: bool = True
    expert_dropout: float = 0.1
    combine_method: str = 'weighted_sum'
    shared_expert: bool = False
    hierarchical: bool = False
    num_hierarchical_levels: int = 2

    def __post_init__(self):
        if self.router_config is None:
            self.router_config = RouterConfig(num_experts=self.num_experts, top_k=self.num_experts_per_token)
        if self.expert_configs is None:
            self.expert_configs = [ExpertConfig(input_dim=self.input_dim, hidden_dim=self.hidden_dim, output_dim=self.output_dim)
                                   for _ in range(self.num_experts)]


@dataclass
class VQAMOEConfig(MOEConfig):
    vision_expert_indices: List[int] = field(default_factory=lambda: [0, 1])
    text_expert_indices: List[int] = field(default_factory=lambda: [2, 3])
    multimodal_expert_indices: List[int] = field(default_factory=lambda: [4, 5, 6, 7])
    use_segmentation_expert: bool = True
    use_detection_expert: bool = True
    use_ocr_expert: bool = True
    use_scene_expert: bool = True
    use_spatial_expert: bool = True
    vietnamese_optimized: bool = True

    def get_expert_type_for_index(self, index: int) -> str:
        for kind, idxs in (('vision', self.vision_expert_indices), ('text', self.text_expert_indices),
                           ('multimodal', self.multimodal_expert_indices)):
            if index in idxs:
                return kind
        return 'feedforward'
