"""How much do the parity figures of one fixture move between fp32 summation orders?  The eval-mode fixture is run with the ring GEMMs' k rotation off
and on with several phases (each a different assignment of k-loop starting points to the XCDs: same products, another summation order); prints the
aggregate gradient error, the worst gradient-norm error and the router-gate norm error per realisation."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import test_parity_gpu as T
from vqa_model_builder_amd.hip import kernels as K

def one(tag, mode, phase):
    K.FORCE_K_ROTATE = phase is not None
    K.K_ROTATE_PHASE = phase or 0
    K._k_rotate_state = None
    T.NORM_TOL, T.ENV, T.ENV_GRAD_SMALL = 10.0, 100.0, 100.0          # report only
    try:
        r = T.run_case(tag, False, mode)
    finally:
        K.FORCE_K_ROTATE = False
        K.set_training_numerics(False)
    return r

for tag, mode in (('full32_cfg3_mcan_moe4', 'bf16'), ('full_cfg3_mcan_moe4', 'fp16'), ('full_cfg3_mcan_moe4', 'bf16'), ('full32_cfg2_xattn', 'bf16')):
    for phase in (None, 0, 1, 2, 3, 5):
        r = one(tag, mode, phase)
        print(f'REAL {tag} {mode} phase={phase} logits={r["logits_rel_l2"]:.3e} grad_agg={r.get("grad_global_rel_l2", float("nan")):.4f} '
              f'(ref {r.get("ref_autocast_grad_global", float("nan")):.4f}) gnorm_worst={r.get("gnorm_worst_rel", float("nan")):.4f}', flush=True)
