"""vqa_model_builder_amd -- MI355X-native (gfx950) implementation of the AutoViVQA forward/backward hot path.

Public surface mirrors the reference's ``src.modeling.meta_arch`` / ``src.modeling.moe`` packages
(SURVEY.md section 8b); ``install_as_src()`` aliases them under those dotted names so the reference's own
training loop imports this implementation unchanged.
"""

import importlib
import sys
import types

__version__ = '0.3.0'


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


_GENERATIVE = ('GenerativeVQAConfig', 'GenerativeVQAOutput', 'GenerativeVQAModel', 'create_generative_vqa_model',
               'get_default_generative_vqa_config')


def _reference_generative():
    """The reference's OWN ``src/modeling/meta_arch/generative_vqa_model.py`` when the reference tree is importable (its ``src``
    package is on sys.path), loaded with the reference's own ``src.modeling.moe`` modules bound inside it: what
    ``install_as_src(generative='reference')`` binds on request.  None otherwise."""
    import importlib.util
    import os
    src = sys.modules.get('src')
    f = next((os.path.join(p, 'modeling', 'meta_arch', 'generative_vqa_model.py') for p in getattr(src, '__path__', [])
              if os.path.isfile(os.path.join(p, 'modeling', 'meta_arch', 'generative_vqa_model.py'))), None)
    if f is None:
        return None
    is_moe = lambda k: k == 'src.modeling.moe' or k.startswith('src.modeling.moe.')
    saved = {k: sys.modules.pop(k) for k in [k for k in sys.modules if is_moe(k)]}
    modeling = sys.modules.get('src.modeling')
    saved_attr = getattr(modeling, 'moe', None)
    try:
        spec = importlib.util.spec_from_file_location('src.modeling.meta_arch.generative_vqa_model', f)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)          # its `from src.modeling.moe...` imports load the reference's own moe files now
    except Exception:
        mod = None
    finally:
        for k in [k for k in sys.modules if is_moe(k)]:
            del sys.modules[k]                # the reference's moe modules stay alive through the generative module's globals
        sys.modules.update(saved)
        if modeling is not None and saved_attr is not None:
            modeling.moe = saved_attr
    return mod


def install_as_src(force: bool = False, generative: str = 'hip'):
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
    # the five generative names of src.modeling.meta_arch (meta_arch/__init__.py:39-71).  generative='hip' (default since round 3): this
    # package's GenerativeVQAModel -- pinned against reference-run fixtures at full size (330 M parameters, 64 000-way head) in both
    # operand types, with use_moe (moe_type 'vqa' / 'standard' / 'sparse') covered (tests/test_generative_gpu.py).
    # generative='reference' keeps the reference's own implementation when its tree is importable (falls back to 'hip' when it is not).
    gen = _reference_generative() if generative == 'reference' else None
    if gen is None:
        gen = importlib.import_module('vqa_model_builder_amd.modeling.meta_arch.generative_vqa_model')
    sys.modules['src.modeling.meta_arch.generative_vqa_model'] = gen
    ours = sys.modules['src.modeling.meta_arch']
    ours.generative_vqa_model = gen
    for name in _GENERATIVE:
        setattr(ours, name, getattr(gen, name))
    return sorted(_ALIASES)
