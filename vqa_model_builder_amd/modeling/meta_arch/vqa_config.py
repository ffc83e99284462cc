"""Config surface of the VQA hot path (field-for-field mirror of the reference).

Restates the dataclasses of reference ``src/modeling/meta_arch/vqa_config.py:11-285``
(enums :11-36, per-component configs :39-168, ``VQAModelConfig`` :171-232,
``get_default_vietnamese_vqa_config`` :235-285).  Field names, order and defaults are the
drop-in contract: ``ModelPipeline._build_model_config`` (reference
``src/core/model_pipeline.py:185-301``) constructs these by keyword.
"""

from dataclasses import asdict, dataclass, field
from enum import Enum
from typing import Any, Dict, List, Optional


class BackboneType(Enum):
    RESNET = 'resnet'
    VIT = 'vit'
    SWIN = 'swin'
    CLIP = 'clip'
    DINO = 'dino'


class TextEncoderType(Enum):
    PHOBERT = 'phobert'
    BERT = 'bert'
    ROBERTA = 'roberta'
    BARTPHO = 'bartpho'
    CLIP_TEXT = 'clip_text'


class FusionType(Enum):
    CONCAT = 'concat'
    BILINEAR = 'bilinear'
    ATTENTION = 'attention'
    CROSS_ATTENTION = 'cross_attention'
    MCAN = 'mcan'      # enum string only: MultimodalFusion has no 'mcan' branch
    MUTAN = 'mutan'


@dataclass
class VisualEncoderConfig:
    backbone_type: str = 'vit'
    model_name: str = 'openai/clip-vit-base-patch32'
    pretrained: bool = True
    freeze_backbone: bool = False
    output_dim: int = 768
    use_spatial_features: bool = True
    num_spatial_tokens: int = 196


@dataclass
class TextEncoderConfig:
    encoder_type: str = 'phobert'
    model_name: str = 'vinai/phobert-base'
    pretrained: bool = True
    freeze_encoder: bool = False
    output_dim: int = 768
    max_length: int = 128
    pooling_strategy: str = 'cls'


@dataclass
class FusionConfig:
    fusion_type: str = 'cross_attention'
    hidden_dim: int = 512
    output_dim: int = 512
    num_heads: int = 8
    num_layers: int = 2
    dropout: float = 0.1
    use_layer_norm: bool = True


@dataclass
class MOEConfig:
    use_moe: bool = False
    num_experts: int = 8
    top_k: int = 2
    router_type: str = 'top_k'           # ignored by _init_moe, as in the reference (F11)
    expert_type: str = 'feedforward'     # ignored by _init_moe, as in the reference (F11)
    hidden_dim: int = 2048
    load_balance_weight: float = 0.01


@dataclass
class KnowledgeConfig:
    use_knowledge: bool = False
    num_contexts: int = 5
    retriever_type: str = 'hybrid'
    vector_store_type: str = 'faiss'
    context_fusion: str = 'attention'
    knowledge_base_path: Optional[str] = None


@dataclass
class AnswerHeadConfig:
    num_answers: int = 3000
    hidden_dims: List[int] = field(default_factory=lambda: [512, 256])
    dropout: float = 0.3
    use_sigmoid: bool = False
    classifier_type: str = 'mlp'


_SUBCONFIGS = (
    ('visual_encoder', VisualEncoderConfig),
    ('text_encoder', TextEncoderConfig),
    ('fusion', FusionConfig),
    ('moe', MOEConfig),
    ('knowledge', KnowledgeConfig),
    ('answer_head', AnswerHeadConfig),
)


@dataclass
class VQAModelConfig:
    visual_encoder: VisualEncoderConfig = field(default_factory=VisualEncoderConfig)
    text_encoder: TextEncoderConfig = field(default_factory=TextEncoderConfig)
    fusion: FusionConfig = field(default_factory=FusionConfig)
    moe: MOEConfig = field(default_factory=MOEConfig)
    knowledge: KnowledgeConfig = field(default_factory=KnowledgeConfig)
    answer_head: AnswerHeadConfig = field(default_factory=AnswerHeadConfig)
    embed_dim: int = 768
    dropout: float = 0.1

    @classmethod
    def from_dict(cls, config_dict: Dict[str, Any]) -> 'VQAModelConfig':
        parts = {name: klass(**config_dict.get(name, {})) for name, klass in _SUBCONFIGS}
        return cls(embed_dim=config_dict.get('embed_dim', 768),
                   dropout=config_dict.get('dropout', 0.1), **parts)

    def to_dict(self) -> Dict[str, Any]:
        return asdict(self)


def get_default_vietnamese_vqa_config() -> VQAModelConfig:
    """Defaults of reference ``vqa_config.py:235-285`` (MoE-8 and knowledge both on)."""
    return VQAModelConfig(
        visual_encoder=VisualEncoderConfig(backbone_type='vit', model_name='openai/clip-vit-base-patch32',
                                           pretrained=True, freeze_backbone=False, output_dim=768),
        text_encoder=TextEncoderConfig(encoder_type='phobert', model_name='vinai/phobert-base', pretrained=True,
                                       freeze_encoder=False, output_dim=768, max_length=128,
                                       pooling_strategy='cls'),
        fusion=FusionConfig(fusion_type='cross_attention', hidden_dim=768, output_dim=768, num_heads=8,
                            num_layers=2, dropout=0.1),
        moe=MOEConfig(use_moe=True, num_experts=8, top_k=2, router_type='top_k', expert_type='feedforward'),
        knowledge=KnowledgeConfig(use_knowledge=True, num_contexts=5, retriever_type='hybrid',
                                  context_fusion='attention'),
        answer_head=AnswerHeadConfig(num_answers=3000, hidden_dims=[768, 512], dropout=0.3),
    )


__all__ = [
    'BackboneType', 'TextEncoderType', 'FusionType', 'VisualEncoderConfig', 'TextEncoderConfig',
    'FusionConfig', 'MOEConfig', 'KnowledgeConfig', 'AnswerHeadConfig', 'VQAModelConfig',
    'get_default_vietnamese_vqa_config',
]
