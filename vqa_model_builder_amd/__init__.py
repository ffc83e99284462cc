"""vqa_model_builder_amd -- MI355X-native (gfx950) implementation of the AutoViVQA forward/backward hot path.

Public surface mirrors the reference's ``src.modeling.meta_arch`` / ``src.modeling.moe`` packages
(SURVEY.md section 8b); ``install_as_src()`` aliases them under those dotted names so the reference's own
training loop imports this implementation unchanged.
"""

import importlib
import sys
import types

__version__ = '0.2.0'


def set_compute_dtype(kind: str = 'bf16'):
    """'bf16' (default) or 'fp16': the 16-bit type of every GEMM / attention operand (fp32 accumulation, fp32 LayerNorm / softmax /
    loss / residual stream and fp32 master weights and gradients either way).  'fp16' is the dtype of the reference's main loop
    (autocast fp16 + GradScaler, training_pipeline.py:346-347,457): the caller scales the loss (``GradScaler`` or
    ``optim.FusedAdamW(loss_scale=...)``) exactly as it does for the reference.  Process-wide; call before the model is built."""
    from .hip import lib
    lib.set_half(kind)
    return kind


def compute_dtype() -> str:
    from .hip import lib
    return lib.half()

_ALIASES = {
    'src.modeling.meta_arch': 'vqa_model_builder_amd.modeling.meta_arch',
    'src.modeling.meta_arch.vqa_config': 'vqa_model_builder_amd.modeling.meta_arch.vqa_config',
    'src.modeling.meta_arch.vqa_model': 'vqa_model_builder_amd.modeling.meta_arch.vqa_model',
    'src.modeling.moe': 'vqa_model_builder_amd.modeling.moe',
    'src.modeling.moe.router': 'vqa_model_builder_amd.modeling.moe.router',
    'src.modeling.moe.moe_layer': 'vqa_model_builder_amd.modeling.moe.moe_layer',
    'src.modeling.moe.moe_config': 'vqa_model_builder_amd.modeling.moe.moe_config',
    'src.modeling.moe.expert_types': 'vqa_model_builder_amd.modeling.moe.experts',
    'src.modeling.moe.specialized_experts': 'vqa_model_builder_amd.modeling.moe.experts',
    'src.modeling.moe.base_expert': 'vqa_model_builder_amd.modeling.moe.experts',
}


def install_as_src(force: bool = False):
    """Registers this package's modules under the dotted names the reference imports them by
    (``from src.modeling.meta_arch import ...`` model_pipeline.py:189-197,307; ``from src.modeling.moe import VQAMOELayer``
    vqa_model.py:529; ``from src.modeling.moe.router import create_router`` ablation_trainer.py:205).  Parent packages
    ``src`` / ``src.modeling`` are created as namespace stubs only when they are not importable already, so the rest of
    the reference tree (``src.core``, ``src.data`` ...) keeps resolving to the reference's own files."""
    for parent in ('src', 'src.modeling'):
        if parent not in sys.modules:
            try:
                importlib.import_module(parent)
            except Exception:
                m = types.ModuleType(parent)
                m.__path__ = []
                sys.modules[parent] = m
    for alias, target in _ALIASES.items():
        if force or alias not in sys.modules or not getattr(sys.modules[alias], '__file__', '').startswith(__path__[0]):
            mod = importlib.import_module(target)
            sys.modules[alias] = mod
            parent, _, leaf = alias.rpartition('.')
            if parent in sys.modules:
                setattr(sys.modules[parent], leaf, mod)
    return sorted(_ALIASES)
