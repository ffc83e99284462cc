"""The flat config surface "VQAPipeline" exposes for the hot path (field-for-field mirrors of the reference's
``ModelPipelineConfig`` src/core/model_pipeline.py:17-63, ``TrainingPipelineConfig`` src/core/training_pipeline.py:25-65,
``DataPipelineConfig`` src/core/data_pipeline.py:22-60, ``VQAPipelineConfig(.from_yaml)`` src/core/vqa_pipeline.py:30-74)
plus ``build_model_config``: the mapping ``ModelPipeline._build_model_config`` applies (model_pipeline.py:185-301).
Only the surface is mirrored: the orchestration around it (data loading, logging, checkpoints) stays the reference's.
"""

from dataclasses import dataclass, field
from typing import List, Optional, Tuple

from ..modeling.meta_arch.vqa_config import (AnswerHeadConfig, FusionConfig, KnowledgeConfig, MOEConfig, TextEncoderConfig,
                                             VisualEncoderConfig, VQAModelConfig)


@dataclass
class ModelPipelineConfig:
    visual_backbone: str = "vit"
    visual_model_name: str = "openai/clip-vit-base-patch32"
    visual_output_dim: int = 768
    freeze_visual: bool = False
    text_encoder_type: str = "phobert"
    text_model_name: str = "vinai/phobert-base"
    text_output_dim: int = 768
    text_max_length: int = 64
    freeze_text: bool = False
    fusion_type: str = "cross_attention"
    fusion_hidden_dim: int = 768
    fusion_num_heads: int = 8
    fusion_num_layers: int = 2
    fusion_dropout: float = 0.1
    use_moe: bool = False
    moe_num_experts: int = 8
    moe_top_k: int = 2
    moe_hidden_dim: int = 2048
    moe_load_balance_weight: float = 0.01
    use_knowledge: bool = False
    knowledge_num_contexts: int = 5
    knowledge_retriever_type: str = "dense"
    num_answers: int = 3000
    answer_hidden_dims: List[int] = field(default_factory=lambda: [768, 512])
    answer_dropout: float = 0.3
    embed_dim: int = 768
    dropout: float = 0.1
    device: str = "auto"


@dataclass
class TrainingPipelineConfig:
    num_epochs: int = 20
    gradient_accumulation_steps: int = 1
    max_grad_norm: float = 1.0
    optimizer_name: str = "adamw"
    learning_rate: float = 2e-5
    weight_decay: float = 0.01
    betas: Tuple[float, float] = (0.9, 0.999)
    scheduler_name: str = "cosine"
    warmup_ratio: float = 0.1
    warmup_steps: int = 0
    use_amp: bool = True
    amp_dtype: str = "float16"
    early_stopping: bool = True
    patience: int = 5
    min_delta: float = 0.001
    checkpoint_dir: str = "checkpoints"
    save_best: bool = True
    save_every_epoch: bool = True
    metric_for_best: str = "vqa_accuracy"
    log_interval: int = 50
    eval_interval: int = 1
    seed: int = 42


@dataclass
class DataPipelineConfig:
    images_dir: str = "data/raw/images"
    text_file: str = "data/raw/texts/evaluate_60k_data_balanced_preprocessed.csv"
    train_ratio: float = 0.8
    val_ratio: float = 0.1
    test_ratio: float = 0.1
    batch_size: int = 32
    eval_batch_size: int = 64
    num_workers: int = 4
    pin_memory: bool = True
    image_size: Tuple[int, int] = (224, 224)
    normalize_mean: List[float] = field(default_factory=lambda: [0.485, 0.456, 0.406])
    normalize_std: List[float] = field(default_factory=lambda: [0.229, 0.224, 0.225])
    augmentation_strength: str = "medium"
    tokenizer_name: str = "vinai/phobert-base"
    max_seq_length: int = 64
    min_answer_freq: int = 5
    validate_samples: int = 5
    seed: int = 42


@dataclass
class VQAPipelineConfig:
    mode: str = "train"
    data: DataPipelineConfig = field(default_factory=DataPipelineConfig)
    model: ModelPipelineConfig = field(default_factory=ModelPipelineConfig)
    training: TrainingPipelineConfig = field(default_factory=TrainingPipelineConfig)
    output_dir: str = "outputs"
    log_dir: str = "logs/pipeline"
    resume_from: Optional[str] = None

    @classmethod
    def from_yaml(cls, yaml_path: str) -> "VQAPipelineConfig":
        """Same behaviour as the reference, including its TypeError on keys that are not dataclass fields."""
        import yaml
        with open(yaml_path, 'r') as f:
            d = yaml.safe_load(f)
        return cls(mode=d.get('mode', 'train'), data=DataPipelineConfig(**d.get('data', {})),
                   model=ModelPipelineConfig(**d.get('model', {})), training=TrainingPipelineConfig(**d.get('training', {})),
                   output_dir=d.get('output_dir', 'outputs'), log_dir=d.get('log_dir', 'logs/pipeline'), resume_from=d.get('resume_from'))


def build_model_config(c: ModelPipelineConfig) -> VQAModelConfig:
    """ModelPipelineConfig -> VQAModelConfig exactly as model_pipeline.py:185-301 maps it."""
    return VQAModelConfig(
        visual_encoder=VisualEncoderConfig(backbone_type=c.visual_backbone, model_name=c.visual_model_name, pretrained=True,
                                           freeze_backbone=c.freeze_visual, output_dim=c.visual_output_dim),
        text_encoder=TextEncoderConfig(encoder_type=c.text_encoder_type, model_name=c.text_model_name, pretrained=True,
                                       freeze_encoder=c.freeze_text, output_dim=c.text_output_dim, max_length=c.text_max_length),
        fusion=FusionConfig(fusion_type=c.fusion_type, hidden_dim=c.fusion_hidden_dim, output_dim=c.fusion_hidden_dim,
                            num_heads=c.fusion_num_heads, num_layers=c.fusion_num_layers, dropout=c.fusion_dropout),
        moe=MOEConfig(use_moe=c.use_moe, num_experts=c.moe_num_experts, top_k=c.moe_top_k, hidden_dim=c.moe_hidden_dim,
                      load_balance_weight=c.moe_load_balance_weight),
        knowledge=KnowledgeConfig(use_knowledge=c.use_knowledge, num_contexts=c.knowledge_num_contexts,
                                  retriever_type=c.knowledge_retriever_type),
        answer_head=AnswerHeadConfig(num_answers=c.num_answers, hidden_dims=c.answer_hidden_dims, dropout=c.answer_dropout),
        embed_dim=c.embed_dim, dropout=c.dropout)
