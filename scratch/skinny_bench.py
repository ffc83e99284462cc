import sys, torch
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
    return e0.elapsed_time(e1) / iters * 1e3
L.vqa_set_gemm_pipeline(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
print('%-4s %-16s %s' % ('lay', 'M,N,K', ' | '.join('hint%d' % h for h in (3, 8, 9, 2))))
for lay, M, N, Kd in [('NT', 32, 2048, 2048), ('NN', 32, 2048, 2048), ('NT', 32, 4096, 2048), ('NN', 32, 2048, 4096), ('NT', 32, 6144, 2048),
                      ('NN', 32, 2048, 6144), ('NT', 32, 768, 2048), ('NN', 32, 768, 2048), ('NT', 32, 2048, 768), ('NT', 32, 3000, 512)]:
    a = torch.randn((M, Kd), device=dev).to(torch.bfloat16)
    b = torch.randn((N, Kd) if lay == 'NT' else (Kd, N), device=dev).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    outb = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    ref = a.float() @ (b.float().t() if lay == 'NT' else b.float()) + bias
    res = []
    for h in (3, 8, 9, 2):
        f = (lambda: K.gemm(a, b, M, N, Kd, Kd, Kd, True, True, out_bf16=outb, bias=bias, tile_hint=h)) if lay == 'NT' else \
            (lambda: K.gemm(a, b, M, N, Kd, Kd, N, True, False, out_bf16=outb, bias=bias, tile_hint=h))
        f(); torch.cuda.synchronize()
        err = ((outb.float() - ref).norm() / ref.norm()).item()
        res.append(('%5.1f' % bench(f)) if err < 6e-3 else 'ERR%.2g' % err)
    print('%-4s %-16s %s' % (lay, f'{M},{N},{Kd}', ' | '.join(res)))
