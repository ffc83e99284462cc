from .pipeline_config import (DataPipelineConfig, ModelPipelineConfig, TrainingPipelineConfig, VQAPipelineConfig,  # noqa: F401
                              build_model_config)
