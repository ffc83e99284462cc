import sys, time, torch, shutil, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '.')
which = sys.argv[1]
if which != 'default':
    import vqa_model_builder_amd.hip.lib as L_
    L_.LIB_PATH = os.path.abspath(f'scratch/lib_chunk_{which}.so')
import bench
from vqa_model_builder_amd.hip import lib
L = lib.load()
print('chunk', L.vqa_opt_chunk_elems())
dev = torch.device('cuda:0')
model = bench.build_model('cfg2_xattn', dev).train()
opt = bench.make_optimizer(model)
px, ids, mask, labels = bench.synthetic_batch(32, dev, 0)
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        opt.zero_grad(set_to_none=True)
        out = model(pixel_values=px, input_ids=ids, attention_mask=mask, labels=labels); out.loss.backward(); opt.step()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3): opt.step()
torch.cuda.synchronize(); e0.record()
for _ in range(20): opt.step()
e1.record(); torch.cuda.synchronize()
print('%s: optimiser step %.3f ms' % (which, e0.elapsed_time(e1) / 20))
