"""``src.modeling.meta_arch`` surface of the reference (meta_arch/__init__.py:39-71), classification model."""

from .vqa_config import (AnswerHeadConfig, BackboneType, FusionConfig, FusionType, KnowledgeConfig, MOEConfig, TextEncoderConfig,
                         TextEncoderType, VisualEncoderConfig, VQAModelConfig, get_default_vietnamese_vqa_config)
from .vqa_model import (AnswerHead, CrossModalAttention, MultimodalFusion, TextEncoder, VietnameseVQAModel, VisualEncoder, VQAOutput,
                        create_vqa_model)

__all__ = [
    'BackboneType', 'TextEncoderType', 'FusionType', 'VisualEncoderConfig', 'TextEncoderConfig', 'FusionConfig', 'MOEConfig',
    'KnowledgeConfig', 'AnswerHeadConfig', 'VQAModelConfig', 'get_default_vietnamese_vqa_config', 'VQAOutput', 'VisualEncoder',
    'TextEncoder', 'CrossModalAttention', 'MultimodalFusion', 'AnswerHead', 'VietnameseVQAModel', 'create_vqa_model',
]
