"""Is ~10% gradient deviation inherent to bf16 GEMM operands at this init?  Emulate on CPU: round every linear/matmul
operand to bf16 (fp32 accumulate) in the oracle, forward and backward, compare to the fp32 oracle."""
import sys, numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, '.')
from oracle import det_weights as dw, vqa_oracle as vo
from tests.conftest import CfgView, load_golden

class Q(torch.autograd.Function):           # round to bf16 in fwd AND round the incoming grad in bwd
    @staticmethod
    def forward(ctx, x): return x.bfloat16().float()
    @staticmethod
    def backward(ctx, g): return g.bfloat16().float()

tag = sys.argv[1]
arrays, meta = load_golden(tag)
d = meta['dims']
sd = dw.make_state_dict({k: tuple(v) for k, v in meta['shapes'].items()}, meta['seed'])
px, ids, mask, labels = dw.make_inputs(d['batch'], d['seq'], d['image'], vocab_hi=min(30000, d['vocab']), num_answers=d['num_answers'], seed=meta['seed'])
cfg = CfgView(meta)
kw = dict(vit_heads=d['vit_heads'], text_heads=d['txt_heads'])
l0, _, _, g0 = vo.forward_backward(sd, cfg, px, ids, mask, labels, **kw)
orig_linear, orig_matmul, orig_conv = F.linear, torch.matmul, F.conv2d
F.linear = lambda x, w, b=None: orig_linear(Q.apply(x), Q.apply(w), b)
torch.matmul = lambda a, b: orig_matmul(Q.apply(a), Q.apply(b))
F.conv2d = lambda x, w, *a, **k: orig_conv(Q.apply(x), Q.apply(w), *a, **k)
l1, _, _, g1 = vo.forward_backward(sd, cfg, px, ids, mask, labels, **kw)
rl = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))
print('logits rel l2', rl(l1, l0), 'max abs', float((l1 - l0).abs().max()))
errs = sorted(((rl(g1[k], g0[k]), k) for k in g0 if g0[k].norm() > 1e-6), reverse=True)
print('grad rel-l2: max %.4f median %.4f min %.4f' % (errs[0][0], errs[len(errs)//2][0], errs[-1][0]))
for e, k in errs[:8] + errs[-8:]: print('  %.4f %s' % (e, k))
