"""Does the HBM-bound AdamW overlap with the forward pass?  Three captured graphs: forward only, optimiser only, both as
parallel branches."""
import sys, time, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '.')
import bench
from vqa_model_builder_amd.hip import lib, blocks
lib.load()
dev = torch.device('cuda:0')
px, ids, mask, labels = bench.synthetic_batch(32, dev, 0)
batch = dict(pixel_values=px, input_ids=ids, attention_mask=mask, labels=labels)
model = bench.build_model('cfg2_xattn', dev).train()
model.parallel_towers = True
opt = bench.make_optimizer(model)
blocks.enable_indirect_seeds(dev)
for _ in range(3):
    opt.zero_grad(set_to_none=True)
    out = model(**batch); out.loss.backward(); opt.step()
torch.cuda.synchronize()
opt.make_capturable(dev)
def timeit(g, n=20):
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): g.replay()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
g_opt = torch.cuda.CUDAGraph()
with torch.cuda.graph(g_opt):
    opt.step()
g_fwd = torch.cuda.CUDAGraph()
with torch.cuda.graph(g_fwd):
    with torch.no_grad():
        out = model(**batch)
g_both = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
with torch.cuda.graph(g_both):
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        opt.step()
    with torch.no_grad():
        out = model(**batch)
    cur.wait_stream(side)
print('optimiser alone %.3f ms   forward alone %.3f ms   both in parallel %.3f ms' % (timeit(g_opt), timeit(g_fwd), timeit(g_both)), flush=True)
import os; os._exit(0)
