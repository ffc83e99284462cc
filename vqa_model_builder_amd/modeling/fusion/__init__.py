"""``src.modeling.fusion`` surface of the reference (fusion/__init__.py:5-19)."""

from .fusion_approaches import BaseFusion, CrossAttentionBlock, CrossAttentionFusion, QFormerFusion, SingleStreamFusion, create_fusion_model

__all__ = ['BaseFusion', 'CrossAttentionFusion', 'QFormerFusion', 'SingleStreamFusion', 'create_fusion_model']
