import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '.')
import bench
from vqa_model_builder_amd.hip import lib
lib.load()
dev = torch.device('cuda:0')
px, ids, mask, labels = bench.synthetic_batch(32, dev, 0)
model = bench.build_model('cfg2_xattn', dev).train()
out = model(pixel_values=px, input_ids=ids, attention_mask=mask, labels=labels); out.loss.backward()
st = {}
for n, p in model.named_parameters():
    g = p.grad
    if g is None: continue
    us = g.untyped_storage()
    e = st.setdefault(us.data_ptr(), [us.nbytes(), 0, 0])
    e[1] += 1; e[2] += g.numel() * 4
big = [(k, v) for k, v in st.items() if v[1] > 1]
print('storages', len(st), 'shared storages', len(big))
for k, v in sorted(big, key=lambda kv: -kv[1][0])[:6]: print(' storage', hex(k), 'MB', v[0] / 1e6, 'params', v[1], 'covered MB', v[2] / 1e6)
print('single-param storages', len(st) - len(big), 'MB', sum(v[2] for v in st.values() if v[1] == 1) / 1e6)
