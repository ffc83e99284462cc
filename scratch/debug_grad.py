import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import det_weights as dw, vqa_oracle as vo
from tests.conftest import CfgView, load_golden
from tests.helpers import build_model
tag = sys.argv[1] if len(sys.argv) > 1 else 'full_cfg1_concat'
arrays, meta = load_golden(tag)
d = meta['dims']
sd = dw.make_state_dict({k: tuple(v) for k, v in meta['shapes'].items()}, meta['seed'])
px, ids, mask, labels = dw.make_inputs(d['batch'], d['seq'], d['image'], vocab_hi=min(30000, d['vocab']), num_answers=d['num_answers'], seed=meta['seed'])
model = build_model(meta); model.load_state_dict(sd); model = model.cuda().eval()
out = model(pixel_values=px.cuda(), input_ids=ids.cuda(), attention_mask=mask.cuda(), labels=labels.cuda())
out.loss.backward(); torch.cuda.synchronize()
torch.set_num_threads(16)
_, _, _, og = vo.forward_backward(sd, CfgView(meta), px, ids, mask, labels, vit_heads=d['vit_heads'], text_heads=d['txt_heads'])
named = dict(model.named_parameters())
def rl(a, b): return float((a - b).norm() / (b.norm() + 1e-30))
rows = []
for name, g_ref in og.items():
    g = named[name].grad
    if g is None: continue
    g = g.float().cpu()
    if g_ref.norm() < 1e-6: continue
    tot = rl(g, g_ref)
    f, fr = g.flatten(), g_ref.flatten()
    head = rl(f[:64], fr[:64])
    # error vs element magnitude: relative error restricted to the largest 10% elements
    k = max(1, fr.numel() // 10)
    idx = fr.abs().topk(k).indices
    big = rl(f[idx], fr[idx])
    rows.append((tot, head, big, name, tuple(g.shape), float(g_ref.norm())))
rows.sort(reverse=True)
print('total_relL2  head64_relL2  top10pct_relL2  name')
for r in rows[:40]: print('%.4f %.4f %.4f %s %s %.3g' % r)
print('...median total', np.median([r[0] for r in rows]))
