"""In-situ penalty of the path's GEMMs: the same launch with its weight operand (a) resident (one buffer, re-used), (b) streamed from HBM (rotating over
> 600 MB of distinct weight buffers: nothing survives in the 256-MiB Infinity Cache, as in the training step, where 440 MB of 16-bit weights + 7 GB of
optimiser traffic pass between two uses), (c) as (b) but each buffer read ONCE by a single streaming kernel right before its GEMM (prefetch into the
memory-side cache).  Kernel time = the dispatch's own begin / end timestamps (vqa_gemm_profile)."""
import ctypes as C, sys, torch
sys.path.insert(0, '.')
from vqa_model_builder_amd.hip import kernels as K, lib
L = lib.load()
dev = 'cuda'
SHAPES = [('NT', 2048, 3072, 768), ('NT', 2048, 768, 3072), ('NT', 2048, 768, 768), ('NT', 2048, 2304, 768),
          ('NN', 2048, 768, 3072), ('NN', 2048, 3072, 768), ('NN', 2048, 768, 2304), ('NN', 2048, 768, 768)]
def collect():
    f, ms, n = (C.c_double * 1)(), (C.c_double * 1)(), (C.c_int * 1)()
    L.vqa_gemm_profile_collect(1, f, ms, n)
    return ms[0] / max(n[0], 1) * 1e3
print('%-4s %-16s %10s %10s %10s %10s' % ('lay', 'M,N,K', 'resident', 'W from HBM', 'prefetched', 'A+W cold'), flush=True)
for lay, M, N, Kd in SHAPES:
    wbytes = N * Kd * 2
    R = max(8, int(700e6 // wbytes))
    a = torch.randn((M, Kd), device=dev).to(torch.bfloat16)
    As = [torch.randn((M, Kd), device=dev).to(torch.bfloat16) for _ in range(max(8, int(700e6 // (M * Kd * 2))))]
    Ws = [torch.randn((N, Kd) if lay == 'NT' else (Kd, N), device=dev).to(torch.bfloat16) for _ in range(R)]
    bias = torch.randn((N,), device=dev)
    outb = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    def g(aa, b):
        if lay == 'NT': K.gemm(aa, b, M, N, Kd, Kd, Kd, True, True, out_bf16=outb, bias=bias)
        else: K.gemm(aa, b, M, N, Kd, Kd, N, True, False, out_bf16=outb)
    res = []
    for mode in range(4):
        for i in range(3): g(a, Ws[i])
        torch.cuda.synchronize()
        L.vqa_gemm_profile(1, 0)
        n = min(R, 64)
        for i in range(n):
            if mode == 0: g(a, Ws[0])
            elif mode == 1: g(a, Ws[i])
            elif mode == 2:
                Ws[i].view(torch.int32).sum()            # one streaming reader touches the weights first
                g(a, Ws[i])
            else: g(As[i % len(As)], Ws[i])
        torch.cuda.synchronize()
        res.append(collect())
        L.vqa_gemm_profile(0, 0)
    print('%-4s %-16s %s' % (lay, f'{M},{N},{Kd}', ' '.join('%7.1f us' % r for r in res)), flush=True)
    del Ws, As
    torch.cuda.empty_cache()
