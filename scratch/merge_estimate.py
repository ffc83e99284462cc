"""What would lock-step merged launches for the two encoders buy?  fwd+bwd of the text encoder alone at B=32 (M=2048) and at
B=57 (M=3648 = 2048 + 1600 rows: the row count a merged launch would see), the vision encoder alone at B=32, and both at B=32 as
two parallel branches -- each as a captured graph."""
import sys, time, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '.')
import bench
from vqa_model_builder_amd.hip import lib, blocks, kernels as K
lib.load()
dev = torch.device('cuda:0')
model = bench.build_model('cfg2_xattn', dev).train()
blocks.enable_indirect_seeds(dev)
K.WGRAD_DEFER_TO_STEP_END = True
def graph_of(fn):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): fn()
    return g
def timeit(g, n=20):
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): g.replay()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def text_fb(B):
    ids = torch.randint(2, 30000, (B, 64), device=dev); mask = torch.ones(B, 64, dtype=torch.long, device=dev)
    def f():
        for p in model.text_encoder.parameters(): p.grad = None
        pooled, seq = model.text_encoder(ids, mask)
        (seq.float().mean() + pooled.float().mean()).backward(); K.wgrad_flush_all()
    return f
def vis_fb(B):
    px = torch.randn(B, 3, 224, 224, device=dev)
    def f():
        for p in model.visual_encoder.parameters(): p.grad = None
        pooled, sp = model.visual_encoder(px)
        (sp.float().mean() + pooled.float().mean()).backward(); K.wgrad_flush_all()
    return f
t32, t57, v32 = text_fb(32), text_fb(57), vis_fb(32)
side = torch.cuda.Stream()
def both():
    cur = torch.cuda.current_stream(); side.wait_stream(cur)
    for p in model.parameters(): p.grad = None
    ids = t32.__closure__
    with torch.cuda.stream(side):
        pooled_v, sp = model.visual_encoder(both.px)
    pooled, seq = model.text_encoder(both.ids, both.mask)
    cur.wait_stream(side)
    (seq.float().mean() + pooled.float().mean() + sp.float().mean() + pooled_v.float().mean()).backward(); K.wgrad_flush_all()
both.px = torch.randn(32, 3, 224, 224, device=dev); both.ids = torch.randint(2, 30000, (32, 64), device=dev); both.mask = torch.ones(32, 64, dtype=torch.long, device=dev)
res = {}
for name, fn in (('text B=32', t32), ('text B=57', t57), ('vision B=32', v32), ('text||vision B=32', both)):
    res[name] = timeit(graph_of(fn)); print('%-20s %.3f ms' % (name, res[name]), flush=True)
import os; os._exit(0)
