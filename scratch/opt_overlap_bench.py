"""Upper bound of what a software-pipelined optimiser could win: the fwd+bwd graph and the optimiser graph replayed (a) one after
the other on one stream, (b) at the same time on two streams (the optimiser then runs beside the FORWARD of the step graph -- racy as
training, valid as a timing of the overlap)."""
import sys, time
import torch
sys.path.insert(0, '.')
import bench
from vqa_model_builder_amd.graph import GraphedTrainStep


class NullReducer:
    world = 1
    grad_dtype = 'fp32'
    def reduce(self): pass
    def prepare_static(self, *a, **k): pass
    def reduce_static(self): pass


dev = torch.device('cuda:0')
model = bench.build_model('cfg2_xattn', dev).train()
opt = bench.make_optimizer(model)
px, ids, mask, labels = bench.synthetic_batch(32, dev, 0)
batch = dict(pixel_values=px, input_ids=ids, attention_mask=mask, labels=labels)
g = GraphedTrainStep(model, opt, batch, warmup=3, reducer=NullReducer(), segmented=False)
assert g.g_opt is not None
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

def timeit(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

def seq():
    with torch.cuda.stream(sa):
        g.g_main.replay(); g.g_opt.replay()

def only_main():
    with torch.cuda.stream(sa):
        g.g_main.replay()

def only_opt():
    with torch.cuda.stream(sa):
        g.g_opt.replay()

def par():
    sb.wait_stream(sa)
    with torch.cuda.stream(sb):
        g.g_opt.replay()
    with torch.cuda.stream(sa):
        g.g_main.replay()
        sa.wait_stream(sb)

for rep in range(2):
    print('fwd+bwd %.3f  opt %.3f  sequential %.3f  concurrent %.3f ms' % (timeit(only_main), timeit(only_opt), timeit(seq), timeit(par)), flush=True)

# (c) the same overlap as ONE graph with a fork: optimiser (reading the previous step's gradients, kept alive here) on a side branch,
# forward + backward on the main branch
keep = [p.grad for p in model.parameters()]
g2 = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
cap = torch.cuda.Stream()
cap.wait_stream(torch.cuda.current_stream())
with torch.cuda.graph(g2, stream=cap, capture_error_mode='global'):
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        for p, gr in zip(model.parameters(), keep):
            p.grad = gr
        opt.step()
    grads_after = g._fwd_bwd()
    main.wait_stream(side)

def forked():
    with torch.cuda.stream(sa):
        g2.replay()

for rep in range(2):
    print('one graph, optimiser as a parallel branch of the forward: %.3f ms' % timeit(forked), flush=True)
