import sys, torch
sys.path.insert(0, '.')
import bench
from vqa_model_builder_amd.graph import GraphedTrainStep
from vqa_model_builder_amd.hip import kernels as K
dev = torch.device('cuda:0')
model = bench.build_model('cfg2_xattn', dev).train()
opt = bench.make_optimizer(model)
px, ids, mask, labels = bench.synthetic_batch(32, dev, 0)
_orig = K._launch_group
_calls = [0]
def _spy(pending):
    _calls[0] += 1
    if _calls[0] in (4, 5):
        import collections
        c = collections.Counter()
        for dy, x, M, N, Kd, ldy, ldx, out, _ in pending:
            elig = (N % 256 == 0 and Kd % 256 == 0 and M % 64 == 0 and out.is_contiguous())
            c[(M, N, Kd, 'dw256' if elig else 'ring', 'contig' if out.is_contiguous() else 'strided')] += 1
        print('group call', _calls[0], 'items', len(pending))
        for k, v in sorted(c.items(), key=lambda kv: (kv[0][3], -kv[0][1] * kv[0][2])):
            print('   ', v, 'x tokens', k[0], 'out', k[1], 'x', k[2], k[3], k[4])
    return _orig(pending)
K._launch_group = _spy
g = GraphedTrainStep(model, opt, dict(pixel_values=px, input_ids=ids, attention_mask=mask, labels=labels), warmup=3)
for _ in range(3): g()
torch.cuda.synchronize()
print('norm pass reads / all gradient elements:', opt.norm_coverage())
print('covered spans', len(K.WGRAD_SUMSQ_COVERED), 'bytes', sum(e - a for a, e in K.WGRAD_SUMSQ_COVERED))
spans = sorted(set(K.WGRAD_SUMSQ_COVERED))
print('unique', len(spans), 'overlaps', sum(a[1] > b[0] for a, b in zip(spans, spans[1:])))
gr = [(n, p.grad.data_ptr(), p.grad.numel() * 4) for n, p in model.named_parameters() if p.grad is not None]
import bisect
starts = [a for a, _ in spans]
inside = 0
for n, a, nb in gr:
    i = bisect.bisect_right(starts, a) - 1
    if i >= 0 and a + nb <= spans[i][1]: inside += nb
print('gradient bytes inside spans', inside, 'of', sum(nb for _, _, nb in gr))
