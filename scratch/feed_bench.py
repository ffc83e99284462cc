"""PCIe-inclusive step rate: the cfg2 captured step fed (a) from HBM-resident inputs, (b) by per-step pageable .to(device) copies as the
reference's loop does, (c) through DevicePrefetcher (pinned staging, side stream, one batch ahead)."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vqa_model_builder_amd.data_feed import DevicePrefetcher, as_model_inputs
from vqa_model_builder_amd.graph import GraphedTrainStep
dev = torch.device('cuda')
model = bench.build_model('cfg2_xattn', dev).train()
opt = bench.make_optimizer(model)
B, N = 32, 60
host = [{'image': torch.randn(B, 3, 224, 224), 'input_ids': torch.randint(2, 30000, (B, 64)), 'attention_mask': torch.ones(B, 64, dtype=torch.int64),
         'label': torch.randint(0, 3000, (B,))} for _ in range(8)]
first = {k: v.to(dev) for k, v in as_model_inputs(host[0]).items()}
gs = GraphedTrainStep(model, opt, first, warmup=3)
def run(feed):
    for _ in range(5): gs(first)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for b in feed: gs(b)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e3
resident = run(first for _ in range(N))
pageable = run({k: v.to(dev) for k, v in as_model_inputs(host[i % 8]).items()} for i in range(N))
prefetch = run(as_model_inputs(b) for b in DevicePrefetcher(host[i % 8] for i in range(N)))
pinned = run(as_model_inputs(b) for b in DevicePrefetcher((host[i % 8] for i in range(N)), pin=True))
print(f'cfg2 captured step, batch 32: inputs resident in HBM {resident:.3f} ms/step ({B / resident * 1e3:.0f} samples/s); pageable .to(device) per step '
      f'{pageable:.3f} ms ({B / pageable * 1e3:.0f}); DevicePrefetcher (side stream, one batch ahead) {prefetch:.3f} ms ({B / prefetch * 1e3:.0f}); the same through its own pinned staging buffers {pinned:.3f} ms ({B / pinned * 1e3:.0f})')
