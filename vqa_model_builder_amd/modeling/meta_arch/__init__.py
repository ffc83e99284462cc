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


_GENERATIVE = ('GenerativeVQAConfig', 'GenerativeVQAOutput', 'GenerativeVQAModel', 'create_generative_vqa_model',
               'get_default_generative_vqa_config')


def __getattr__(name):
    """The generative names of the reference's ``src.modeling.meta_arch`` (meta_arch/__init__.py:39-71) resolve -- lazily, so the
    classification path never imports it -- to this package's HIP implementation (generative_vqa_model.py: started this round;
    default configuration and the fusion-MoE variants, not ``moe_type='sparse'``).  ``vqa_model_builder_amd.install_as_src()``
    keeps binding the reference's OWN generative module under ``src.modeling.meta_arch`` when the reference tree is importable
    (``install_as_src(generative='hip')`` binds this one)."""
    if name in _GENERATIVE:
        from . import generative_vqa_model as g
        return getattr(g, name)
    raise AttributeError(name)
