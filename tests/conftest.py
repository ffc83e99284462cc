import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(tag):
    """Returns (arrays: dict name->np.ndarray, meta: dict) of a committed fixture."""
    z = np.load(os.path.join(GOLDEN, tag + '.npz'), allow_pickle=False)
    arrays = {k: z[k] for k in z.files if k != 'meta'}
    return arrays, json.loads(str(z['meta']))


class CfgView:
    """Minimal VQAModelConfig-shaped object rebuilt from a fixture's meta (for the oracle)."""

    def __init__(self, meta):
        d = meta['dims']
        ns = lambda **kw: type('NS', (), kw)()
        self.fusion = ns(fusion_type=meta['fusion_type'], num_heads=d['fusion_heads'], use_layer_norm=True,
                         hidden_dim=d['D'], output_dim=d['D'], num_layers=d['fusion_layers'])
        self.moe = ns(use_moe=meta['num_experts'] > 0, num_experts=max(meta['num_experts'], 1), top_k=2,
                      hidden_dim=d['moe_hidden'])
        self.dims = d


@pytest.fixture(scope='session')
def golden_loader():
    return load_golden


@pytest.fixture(autouse=True)
def _eval_numerics_between_tests(request):
    """GPU tests start from the eval-mode numerics of the ring GEMMs (per-XCD k rotation OFF: hip/kernels.py set_training_numerics), whatever mode the
    previous test's model left the library in -- the bit-exactness tests between kernels hold for the unrotated loop."""
    if request.node.get_closest_marker('gpu') is not None:
        try:
            import torch
            if torch.cuda.is_available():
                from vqa_model_builder_amd.hip import kernels as K
                K.FORCE_K_ROTATE = False
                K._k_rotate_state = None
                K.set_training_numerics(False)
        except Exception:
            pass
    yield
