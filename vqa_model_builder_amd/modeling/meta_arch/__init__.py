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
    """The generative names of the reference's ``src.modeling.meta_arch`` (meta_arch/__init__.py:39-71): not re-implemented on the HIP
    path.  ``vqa_model_builder_amd.install_as_src()`` binds them to the reference's own implementation
    when the reference tree is importable; without it, asking for one fails loudly instead of resolving to nothing."""
    if name in _GENERATIVE:
        raise ImportError(f'{name}: the generative VQA path is not part of this build; put the reference tree on sys.path and call '
                          'vqa_model_builder_amd.install_as_src() to use its own implementation alongside the HIP classification model')
    raise AttributeError(name)
