"""GEMMs with 32 < M < 256 rows (Segmentation expert: 4 mask tokens x 32 samples = 128 rows): the default (register-staged 64x64)
against LDS-DMA tiles forced through tile_hint + vqa_set_gemm_pipeline.  Kernel time from the library's profiling hook."""
import ctypes as C, sys, torch
sys.path.insert(0, '.')
from vqa_model_builder_amd.hip import kernels as K, lib
L = lib.load()
dev = 'cuda'
SHAPES = [('NT', 128, 6144, 2048), ('NT', 128, 2048, 2048), ('NT', 128, 4096, 2048), ('NT', 128, 2048, 4096), ('NT', 64, 2048, 2048), ('NT', 192, 2048, 2048),
          ('NN', 128, 2048, 6144), ('NN', 128, 2048, 2048), ('NN', 128, 2048, 4096), ('NN', 128, 4096, 2048)]
# (name, tile_hint, pipeline)   tile_hint = cfg + 1:  2: 64x64, 8: 32x32, 9: 32x64, 5: 128x64
MODES = [('default', 0, 0), ('64x64 dma3', 2, 3), ('32x32 dma3', 8, 3), ('32x64 dma3', 9, 3), ('128x64 dma2', 5, 2), ('64x64 dma2', 2, 2)]
def ktime(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    L.vqa_gemm_profile(1, 0)
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    f, ms, n = (C.c_double * 1)(), (C.c_double * 1)(), (C.c_int * 1)()
    L.vqa_gemm_profile_collect(1, f, ms, n)
    L.vqa_gemm_profile(0, 0)
    return ms[0] / max(n[0], 1) * 1e3
print('%-4s %-16s %s' % ('lay', 'M,N,K', ' | '.join('%-12s' % m[0] for m in MODES)), flush=True)
for lay, M, N, Kd in SHAPES:
    a = torch.randn((M, Kd), device=dev).to(torch.bfloat16)
    b = torch.randn((N, Kd) if lay == 'NT' else (Kd, N), device=dev).to(torch.bfloat16)
    bias = torch.randn((N,), device=dev)
    outb = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    ref = a.float() @ (b.float().t() if lay == 'NT' else b.float())
    res = []
    for name, hint, pipe in MODES:
        L.vqa_set_gemm_pipeline(pipe)
        f = (lambda: K.gemm(a, b, M, N, Kd, Kd, Kd, True, True, out_bf16=outb, bias=bias, tile_hint=hint)) if lay == 'NT' else \
            (lambda: K.gemm(a, b, M, N, Kd, Kd, N, True, False, out_bf16=outb, tile_hint=hint))
        f(); torch.cuda.synchronize()
        got = outb.float() - (bias if lay == 'NT' else 0)
        err = ((got - ref).norm() / ref.norm()).item()
        us = ktime(f)
        res.append('ERR %.2g    ' % err if err > 6e-3 else '%5.1f us %4.0f' % (us, 2.0 * M * N * Kd / us / 1e6))
    L.vqa_set_gemm_pipeline(0)
    print('%-4s %-16s %s' % (lay, f'{M},{N},{Kd}', ' | '.join(res)), flush=True)
