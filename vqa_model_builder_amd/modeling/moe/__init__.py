"""``src.modeling.moe`` surface of the reference (moe/__init__.py:51-97) on the HIP path."""

from .experts import (BaseExpert, CountingExpert, ExpertWithCapacity, FeedForwardExpert, GatedLinearExpert, MultimodalExpert,
                      ObjectDetectionExpert, OCRExpert, SceneUnderstandingExpert, SegmentationExpert, SpatialReasoningExpert,
                      TextExpert, VisionExpert, create_expert)
from .moe_config import ExpertConfig, MOEConfig, RouterConfig, VQAMOEConfig
from .moe_layer import HierarchicalMOE, MOELayer, SparseMOELayer, VQAMOELayer
from .moe_utils import (ExpertDropout, ExpertParallelWrapper, analyze_routing_patterns, compute_expert_capacity, compute_expert_entropy,
                        compute_load_balance_loss, compute_router_z_loss, get_expert_utilization, load_moe_checkpoint, save_moe_checkpoint)
from .router import BaseRouter, ExpertChoiceRouter, NoisyTopKRouter, SoftRouter, TopKRouter, create_router

__all__ = [
    'BaseExpert', 'ExpertWithCapacity', 'BaseRouter', 'TopKRouter', 'SoftRouter', 'NoisyTopKRouter', 'ExpertChoiceRouter',
    'create_router', 'VisionExpert', 'TextExpert', 'MultimodalExpert', 'FeedForwardExpert', 'GatedLinearExpert', 'create_expert',
    'SegmentationExpert', 'ObjectDetectionExpert', 'OCRExpert', 'SceneUnderstandingExpert', 'SpatialReasoningExpert',
    'CountingExpert', 'MOELayer', 'SparseMOELayer', 'HierarchicalMOE', 'VQAMOELayer', 'MOEConfig', 'ExpertConfig', 'RouterConfig',
    'VQAMOEConfig', 'compute_expert_capacity', 'compute_load_balance_loss', 'compute_router_z_loss', 'get_expert_utilization',
    'compute_expert_entropy', 'ExpertDropout', 'ExpertParallelWrapper', 'save_moe_checkpoint', 'load_moe_checkpoint',
    'analyze_routing_patterns',
]
