import sys, bisect, torch
sys.path.insert(0, '.')
from tests.test_graph_gpu import _setup, _moe_setup
from vqa_model_builder_amd.hip import kernels as K
for name, setup in (('xattn', lambda: _setup(False)), ('moe', _moe_setup)):
    model, opt, batch = setup()
    opt.fuse_wgrad_norm(True)
    opt.zero_grad(set_to_none=True)
    model(**batch).loss.backward()
    spans = sorted(set(K.WGRAD_SUMSQ_COVERED))
    merged = []
    for a, e in spans:
        if merged and merged[-1][1] == a: merged[-1][1] = e
        else: merged.append([a, e])
    gr = sorted((p.grad.data_ptr(), p.grad.data_ptr() + p.grad.numel() * 4, n) for n, p in model.named_parameters() if p.grad is not None)
    print(name, 'spans', len(spans), 'merged', len(merged), 'dups', len(K.WGRAD_SUMSQ_COVERED) - len(spans))
    for a, e in merged:
        inside = [(n, ga - a, ge - a) for ga, ge, n in gr if ga >= a and ge <= e]
        got = sum(y - x for _, x, y in inside)
        if got != e - a:
            over = [(n, ga - a, ge - a) for ga, ge, n in gr if ga < e and ge > a]
            print('  span bytes', e - a, 'matched', got, 'overlapping gradients:', over[:6])
    opt.step()
    print(name, 'coverage', opt.norm_coverage())
    opt.fuse_wgrad_norm(False)
