"""Per-parameter gradient error of one tiny fixture in one operand mode, in model order (debug helper)."""
import sys
import numpy as np, torch
sys.path.insert(0, '.')
import vqa_model_builder_amd as vqa
from oracle import det_weights as dw, vqa_oracle as vo
from tests.conftest import CfgView, load_golden
from tests.helpers import build_model, fixture_inputs
tag, mode, scale = sys.argv[1], sys.argv[2], float(sys.argv[3])
vqa.set_compute_dtype(mode)
arrays, meta = load_golden(tag); d = meta['dims']
sd = dw.make_state_dict({k: tuple(v) for k, v in meta['shapes'].items()}, meta['seed'])
px, ids, mask, labels = fixture_inputs(arrays, meta)
model = build_model(meta); model.load_state_dict(sd); model = model.cuda().eval()
out = model(pixel_values=px.cuda(), input_ids=ids.cuda(), attention_mask=mask.cuda(), labels=labels.cuda(), return_features=True)
(out.loss * scale).backward()
_, _, _, g0 = vo.forward_backward(sd, CfgView(meta), px, ids, mask, labels, vit_heads=d['vit_heads'], text_heads=d['txt_heads'])
for n, p in model.named_parameters():
    if p.grad is None or n not in g0: continue
    a, b = p.grad.float().cpu() / scale, g0[n]
    if float(b.norm()) < 1e-6: continue
    e = float((a - b).norm() / b.norm())
    if e > 5e-3: print(f'{e:9.3e}  |ref| {float(b.norm()):9.3e}  max|scaled| {float(p.grad.abs().max()):9.3e}  {n}')
print('fused err', float((out.fused_features.float().cpu() - torch.from_numpy(arrays['fused'])).norm() / np.linalg.norm(arrays['fused'])))
