import sys, torch, itertools
sys.path.insert(0, '.')
from vqa_model_builder_amd.hip import kernels as K, lib
L = lib.load()
dev = 'cuda'
def bench(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us
shapes = [('NT', 2048, 2304, 768), ('NT', 2048, 768, 768), ('NT', 2048, 3072, 768), ('NT', 2048, 768, 3072), ('NT', 1600, 2304, 768),
          ('NN', 2048, 768, 2304), ('NN', 2048, 768, 3072), ('NN', 2048, 3072, 768),
          ('TN', 2304, 768, 2048), ('TN', 768, 768, 2048), ('TN', 3072, 768, 2048), ('TN', 768, 3072, 2048), ('TN', 768, 768, 1600)]
CFG = [(0, 2, 8), (2, 2, 8), (3, 2, 8), (2, 5, 8), (3, 5, 8), (2, 1, 8), (2, 7, 8)]
if len(sys.argv) > 1: CFG = [tuple(int(x) for x in c.split(',')) for c in sys.argv[1:]]
print('%-4s %-18s %s' % ('lay', 'M,N,K', ' | '.join('p%d/h%d/g%d' % c for c in CFG)))
for lay, M, N, Kd in shapes:
    a = torch.randn((M, Kd) if lay != 'TN' else (Kd, M), device=dev).to(torch.bfloat16)
    b = torch.randn((N, Kd) if lay == 'NT' else (Kd, N), device=dev).to(torch.bfloat16)
    out = torch.empty((M, N), device=dev)
    outb = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    res = []
    af, bf_ = a.float(), b.float()
    ref = (af if lay != 'TN' else af.t()) @ (bf_.t() if lay == 'NT' else bf_)
    for pl, h, gm in CFG:
        L.vqa_set_gemm_pipeline(pl); L.vqa_set_gemm_group_m(gm)
        if True:
            if lay == 'NT': f = lambda: K.gemm(a, b, M, N, Kd, Kd, Kd, True, True, out_bf16=outb, tile_hint=h)
            elif lay == 'NN': f = lambda: K.gemm(a, b, M, N, Kd, Kd, N, True, False, out_bf16=outb, tile_hint=h)
            else: f = lambda: K.gemm(a, b, M, N, Kd, M, N, False, False, out_f32=out, tile_hint=h, split_k=1)
            f(); torch.cuda.synchronize()
            got = outb.float() if lay != 'TN' else out
            err = ((got - ref).norm() / ref.norm()).item()
            us = bench(f)
            if err > 6e-3: res.append('ERR %.3g' % err); continue
            res.append('%5.1f %4.0f' % (us, 2.0 * M * N * Kd / us / 1e6))
    print('%-4s %-18s %s' % (lay, f'{M},{N},{Kd}', ' | '.join(res)))
L.vqa_set_gemm_pipeline(0); L.vqa_set_gemm_group_m(16)
