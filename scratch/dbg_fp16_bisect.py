"""Bisects a whole-model backward: every ops.linear call of the model records the gradient of its output and of its input; each
dX / dW / db is then recomputed in fp32 torch from the recorded dY and compared (debug helper)."""
import sys
import torch, torch.nn.functional as F
sys.path.insert(0, '.')
import vqa_model_builder_amd as vqa
from oracle import det_weights as dw
from tests.conftest import load_golden
from tests.helpers import build_model, fixture_inputs
from vqa_model_builder_amd.hip import ops
tag, mode, scale = sys.argv[1], sys.argv[2], float(sys.argv[3])
vqa.set_compute_dtype(mode)
arrays, meta = load_golden(tag)
sd = dw.make_state_dict({k: tuple(v) for k, v in meta['shapes'].items()}, meta['seed'])
px, ids, mask, labels = fixture_inputs(arrays, meta)
model = build_model(meta); model.load_state_dict(sd); model = model.cuda().eval()
calls = []
orig = ops.linear
def rec(x, weight, bias=None, act=0, drop=ops.NO_DROP):
    xin = x if x.requires_grad else x.detach().requires_grad_(True)
    y = orig(xin, weight, bias, act, drop)
    ent = {'x': xin, 'w': weight, 'b': bias, 'act': act, 'y': y}
    y.register_hook(lambda g, e=ent: e.__setitem__('dy', g.detach().clone()))
    if xin.requires_grad: xin.register_hook(lambda g, e=ent: e.__setitem__('dx', g.detach().clone()))
    calls.append(ent)
    return y
ops.linear = rec
out = model(pixel_values=px.cuda(), input_ids=ids.cuda(), attention_mask=mask.cuda(), labels=labels.cuda())
(out.loss * scale).backward()
rl = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))
for i, e in enumerate(calls):
    if 'dy' not in e: continue
    x, w, dy = e['x'].detach().float(), e['w'].detach().float(), e['dy'].float()
    x2, dy2 = x.reshape(-1, x.shape[-1]), dy.reshape(-1, dy.shape[-1])
    pre = x2 @ w.t() + (e['b'].detach() if e['b'] is not None else 0)
    g = dy2 * (pre > 0).float() if e['act'] == 3 else dy2
    msg = f'linear {i}: W{tuple(w.shape)} rows {x2.shape[0]} act {e["act"]} |dy| {float(dy2.abs().max()):.3g}'
    if 'dx' in e: msg += f'  dX err {rl(e["dx"].float().reshape(-1, x.shape[-1]), g @ w):.2e}'
    if e['w'].grad is not None: msg += f'  dW err {rl(e["w"].grad.float(), g.t() @ x2):.2e}'
    print(msg)
# direct check of the saved pre-activation of the failing call (linear 3) with the recorded tensors
from vqa_model_builder_amd.hip import kernels as K
for i in (2, 3):
    e = calls[i]
    x2 = e['x'].detach().float().reshape(-1, e['x'].shape[-1]).contiguous()
    w, b = e['w'].detach().float(), e['b'].detach().float()
    N, Kd = w.shape
    for rep in range(2):
        yf, _, pre = K.linear_fwd(K.cast_bf16(x2), K.cast_bf16(w), b, x2.shape[0], N, Kd, want_f32=True, want_pre=True, act=3)
        torch.cuda.synchronize()
        ref = x2 @ w.t() + b
        print(f'linear {i} direct rep {rep}: y err {rl(yf, ref.clamp(min=0)):.2e}  pre err {rl(pre.float(), ref):.2e}  mask mismatches {int(((pre.float() > 0) != (ref > 0)).sum())} of {ref.numel()}')
        print('   pre row0', pre.float()[0, :8].tolist()); print('   ref row0', ref[0, :8].tolist())
