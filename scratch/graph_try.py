import sys, time, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '.')
import bench
from vqa_model_builder_amd.hip import lib
lib.load()
from vqa_model_builder_amd.graph import GraphedTrainStep
dev = torch.device('cuda:0')
px, ids, mask, labels = bench.synthetic_batch(32, dev, 0)
batch = dict(pixel_values=px, input_ids=ids, attention_mask=mask, labels=labels)
towers, wgrad = int(sys.argv[1]), int(sys.argv[2])
model = bench.build_model('cfg2_xattn', dev).train()
model.parallel_towers = False
opt = bench.make_optimizer(model)
gs = GraphedTrainStep(model, opt, batch, parallel_towers=bool(towers), wgrad_side_stream=bool(wgrad))
losses = [gs(batch).item() for _ in range(9)]
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): loss = gs(batch)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
print('towers %d wgrad %d: graph step %.3f ms  %.1f samples/s  loss@12 %.4f loss@32 %.4f' % (towers, wgrad, dt * 1e3, 32 / dt, losses[-1], loss.item()), flush=True)
import os; os._exit(0)
