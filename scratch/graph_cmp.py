import sys, time, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '.')
import bench
from vqa_model_builder_amd.hip import lib
lib.load()
from vqa_model_builder_amd.graph import GraphedTrainStep
dev = torch.device('cuda:0')
px, ids, mask, labels = bench.synthetic_batch(32, dev, 0)
batch = dict(pixel_values=px, input_ids=ids, attention_mask=mask, labels=labels)
mode = sys.argv[1] if len(sys.argv) > 1 else 'eval'
def fresh():
    m = bench.build_model('cfg2_xattn', dev)
    m = m.eval() if mode == 'eval' else m.train()
    return m, bench.make_optimizer(m)
N = 12
model, opt = fresh()
eager = []
for i in range(N):
    opt.zero_grad(set_to_none=True)
    out = model(**batch); out.loss.backward(); opt.step()
    eager.append(out.loss.item())
print('eager', ['%.4f' % l for l in eager], flush=True)
model, opt = fresh()
gs = GraphedTrainStep(model, opt, batch, warmup=3)
graph = [float('nan')] * 3 + [gs(batch).item() for _ in range(N - 3)]
print('graph', ['%.4f' % l for l in graph], flush=True)
