"""Isolates the answer head + CE backward of a tiny fixture in one operand mode against fp32 torch on the same fused features."""
import sys
import torch, torch.nn.functional as F
sys.path.insert(0, '.')
import vqa_model_builder_amd as vqa
from oracle import det_weights as dw
from tests.conftest import load_golden
from tests.helpers import build_model, fixture_inputs
from vqa_model_builder_amd.hip import ops, kernels as K
tag, mode, scale = sys.argv[1], sys.argv[2], float(sys.argv[3])
vqa.set_compute_dtype(mode)
arrays, meta = load_golden(tag)
sd = dw.make_state_dict({k: tuple(v) for k, v in meta['shapes'].items()}, meta['seed'])
px, ids, mask, labels = fixture_inputs(arrays, meta)
model = build_model(meta); model.load_state_dict(sd); model = model.cuda().eval()
fused = torch.from_numpy(arrays['fused']).cuda().requires_grad_(True)
labels = labels.cuda()
logits = model.answer_head(fused)
loss, _ = ops.cross_entropy_argmax(logits, labels)
(loss * scale).backward()
lins = [m for m in model.answer_head.classifier if isinstance(m, torch.nn.Linear)]
f2 = fused.detach().clone().requires_grad_(True)
ws = [(l.weight.detach().clone().requires_grad_(True), l.bias.detach().clone().requires_grad_(True)) for l in lins]
x = f2
for i, (w, b) in enumerate(ws):
    x = F.linear(x, w, b)
    if i < len(ws) - 1: x = F.relu(x)
F.cross_entropy(x, labels).backward()
rl = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))
print('logits', rl(logits.detach(), x.detach()), 'loss', float(loss), float(F.cross_entropy(x, labels)))
print('dfused', rl(fused.grad / scale, f2.grad), 'per-sample', [rl(fused.grad[i] / scale, f2.grad[i]) for i in range(fused.shape[0])])
for l, (w, b) in zip(lins, ws):
    print(tuple(l.weight.shape), 'dW', rl(l.weight.grad / scale, w.grad), 'db', rl(l.bias.grad / scale, b.grad))
# the dx GEMM of the last layer alone
dl = (torch.softmax(x.detach(), -1) - F.one_hot(labels, x.shape[1]).float()) / x.shape[0] * scale
M, N, Kd = dl.shape[0], 40, 40
dyp = torch.zeros((M, N), device='cuda'); dyp[:, :37] = dl
wp = torch.zeros((N, Kd), device='cuda'); wp[:37] = lins[-1].weight.detach()
dx, _ = K.linear_dx(K.cast_bf16(dyp), K.cast_bf16(wp), M, N, Kd, want_f32=True)
print('dx6 gemm', rl(dx, dyp @ wp), [rl(dx[i], (dyp @ wp)[i]) for i in range(M)], 'max|dy|', float(dyp.abs().max()), 'max|dx|', float(dx.abs().max()))
