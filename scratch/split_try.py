"""Would splitting the batch into two independent half-batch chains pack the GPU better?  Times a B=32 step graph, a B=16 step
graph, and two B=16 step graphs (separate model replicas) replayed concurrently on two streams."""
import sys, time, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '.')
import bench
from vqa_model_builder_amd.hip import lib
lib.load()
from vqa_model_builder_amd.graph import GraphedTrainStep
dev = torch.device('cuda:0')
def make(B):
    px, ids, mask, labels = bench.synthetic_batch(B, dev, 0)
    batch = dict(pixel_values=px, input_ids=ids, attention_mask=mask, labels=labels)
    model = bench.build_model('cfg2_xattn', dev).train()
    opt = bench.make_optimizer(model)
    return GraphedTrainStep(model, opt, batch), batch
def timeit(fn, n=20):
    for _ in range(4): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
g32, b32 = make(32)
t32 = timeit(lambda: g32(b32))
ga, ba = make(16)
t16 = timeit(lambda: ga(ba))
gb, bb = make(16)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def both():
    with torch.cuda.stream(s1): ga.g_main.replay()
    with torch.cuda.stream(s2): gb.g_main.replay()
tb = timeit(both)
print('B=32 step %.3f ms | B=16 step %.3f ms | two B=16 steps concurrently %.3f ms' % (t32, t16, tb), flush=True)
import os; os._exit(0)
