import sys, numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, '.')
from oracle import det_weights as dw, vqa_oracle as vo
from tests.conftest import CfgView, load_golden
class Q(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x): return x.bfloat16().float()
    @staticmethod
    def backward(ctx, g): return g.bfloat16().float()
tag, wstd, gain = sys.argv[1], float(sys.argv[2]), float(sys.argv[3])
arrays, meta = load_golden(tag)
d = meta['dims']
sd = dw.make_state_dict({k: tuple(v) for k, v in meta['shapes'].items()}, meta['seed'])
for k, v in sd.items():
    if v.dim() >= 2 and 'embedding' not in k and 'mask_tokens' not in k:
        fan_in = int(np.prod(v.shape[1:])); g = 4.0 if k.endswith('classifier.6.weight') else 1.0
        v.mul_(wstd * np.sqrt(fan_in) / g * (gain if k.endswith('classifier.6.weight') else 1.0))
    elif 'embedding' in k and v.dim() >= 1: v.mul_(wstd / 0.5)
px, ids, mask, labels = dw.make_inputs(d['batch'], d['seq'], d['image'], vocab_hi=min(30000, d['vocab']), num_answers=d['num_answers'], seed=meta['seed'])
cfg = CfgView(meta); kw = dict(vit_heads=d['vit_heads'], text_heads=d['txt_heads'])
torch.set_num_threads(8)
l0, _, p0, g0 = vo.forward_backward(sd, cfg, px, ids, mask, labels, **kw)
ol, om, oc = F.linear, torch.matmul, F.conv2d
F.linear = lambda x, w, b=None: ol(Q.apply(x), Q.apply(w), b)
torch.matmul = lambda a, b: om(Q.apply(a), Q.apply(b))
F.conv2d = lambda x, w, *a, **k: oc(Q.apply(x), Q.apply(w), *a, **k)
l1, _, p1, g1 = vo.forward_backward(sd, cfg, px, ids, mask, labels, **kw)
rl = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))
t2 = l0.topk(2, -1).values
print(tag, 'wstd', wstd, 'gain', gain, '| logits std %.3f rel-l2 %.2e maxabs %.2e | min margin %.4f | argmax equal %s' % (float(l0.std()), rl(l1, l0), float((l1 - l0).abs().max()), float((t2[:,0]-t2[:,1]).min()), bool((p0 == p1).all())))
errs = sorted(((rl(g1[k], g0[k]), k) for k in g0 if g0[k].norm() > 1e-7 * max(v.norm() for v in g0.values())), reverse=True)
print('   grad rel-l2: max %.4f (%s) median %.4f min %.4f' % (errs[0][0], errs[0][1], errs[len(errs)//2][0], errs[-1][0]))
