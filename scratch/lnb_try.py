import sys, time, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '.')
import bench
from vqa_model_builder_amd.hip import lib
L = lib.load()
from vqa_model_builder_amd.graph import GraphedTrainStep
cap = int(sys.argv[1])
L.vqa_set_layernorm_bwd_blocks(int(sys.argv[1]))
dev = torch.device('cuda:0')
px, ids, mask, labels = bench.synthetic_batch(32, dev, 0)
batch = dict(pixel_values=px, input_ids=ids, attention_mask=mask, labels=labels)
model = bench.build_model('cfg2_xattn', dev).train()
opt = bench.make_optimizer(model)
gs = GraphedTrainStep(model, opt, batch)
for _ in range(5): gs(batch)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): loss = gs(batch)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
print('ln bwd blocks %d: graph step %.3f ms  %.1f samples/s loss %.4f' % (cap, dt * 1e3, 32 / dt, loss.item()), flush=True)
import os; os._exit(0)
