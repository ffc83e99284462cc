"""FusedAdamW step (clip + AdamW + bf16 shadows) over the cfg2 model's parameters, captured and replayed: ms per step and TB/s."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device('cuda')
model = bench.build_model(sys.argv[1] if len(sys.argv) > 1 else 'cfg2_xattn', dev)
opt = bench.make_optimizer(model)
params = [p for p in model.parameters() if p.requires_grad]
for p in params:
    p.grad = torch.randn_like(p) * 1e-3
n = sum(p.numel() for p in params)
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    opt.step(); opt.step()
    opt.make_capturable(dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(10):
            opt.step()
torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f'{os.environ.get("VQA_HIP_LIB", "default"):40s} {n/1e6:.0f} M params: {ms:.3f} ms per optimiser step = {n * 34 / ms / 1e9:.2f} TB/s (30 B/param update + 4 B/param norm pass)')
