import sys, torch, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '.')
from tests.conftest import load_golden
from oracle import det_weights as dw
from tests.helpers import build_model
arrays, meta = load_golden('tiny_xattn_moe8')
model = build_model(meta)
sd = dw.make_state_dict({k: tuple(v) for k, v in meta['shapes'].items()}, meta['seed'])
model.load_state_dict(sd); model = model.cuda().eval()
d = meta['dims']
px, ids, mask, labels = dw.make_inputs(d['batch'], d['seq'], d['image'], vocab_hi=min(30000, d['vocab']), num_answers=d['num_answers'], seed=meta['seed'])
out = model(pixel_values=px.cuda(), input_ids=ids.cuda(), attention_mask=mask.cuda(), labels=labels.cuda(), return_features=True)
aux = model.moe_layer.aux_outputs
rp = aux['router_probs'].detach().cpu().numpy()
print('router probs max abs diff', np.abs(rp - arrays['router_probs']).max())
print('ref probs', np.round(arrays['router_probs'].reshape(-1, 8), 3))
print('got probs', np.round(rp.reshape(-1, 8), 3))
print('fused rel', np.linalg.norm(out.fused_features.detach().cpu().numpy() - arrays['fused']) / np.linalg.norm(arrays['fused']))
print('per-sample logits rel', [float(np.linalg.norm(out.logits[b].detach().cpu().numpy() - arrays['logits'][b]) / np.linalg.norm(arrays['logits'][b])) for b in range(d['batch'])])
