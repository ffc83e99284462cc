"""Times every GEMM shape of the path (forward NT, dX NN) on the legacy kernels (ws off), gemm_ws auto and each forced ws tile.
Kernel time = hipExtLaunchKernel start/stop timestamps (the library's profiling hook), averaged over back-to-back launches."""
import ctypes as C, sys, torch
sys.path.insert(0, '.')
from vqa_model_builder_amd.hip import kernels as K, lib
L = lib.load()
dev = 'cuda'
SHAPES = [('NT', 2048, 2304, 768), ('NT', 2048, 768, 768), ('NT', 2048, 3072, 768), ('NT', 2048, 768, 3072),
          ('NT', 1600, 2304, 768), ('NT', 1600, 768, 768), ('NT', 1600, 3072, 768), ('NT', 1600, 768, 3072), ('NT', 1600, 1536, 768), ('NT', 1568, 768, 3072),
          ('NN', 2048, 768, 2304), ('NN', 2048, 768, 3072), ('NN', 2048, 3072, 768), ('NN', 2048, 768, 768),
          ('NN', 1600, 768, 2304), ('NN', 1600, 768, 3072), ('NN', 1600, 3072, 768), ('NN', 1600, 768, 768), ('NN', 1600, 768, 1536)]
MODES = [('legacy', 0), ('ws-auto', 1)] + [(f't{i}', 2 + i) for i in range(7)]
if len(sys.argv) > 1: MODES = [m for m in MODES if m[0] in sys.argv[1:]]
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
print('%-4s %-16s %s' % ('lay', 'M,N,K', ' | '.join('%-11s' % m[0] for m in MODES)), flush=True)
for lay, M, N, Kd in SHAPES:
    a = torch.randn((M, Kd), device=dev).to(torch.bfloat16)
    b = torch.randn((N, Kd) if lay == 'NT' else (Kd, N), device=dev).to(torch.bfloat16)
    bias = torch.randn((N,), device=dev)
    outb = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    ref = a.float() @ (b.float().t() if lay == 'NT' else b.float())
    res = []
    for name, mode in MODES:
        L.vqa_set_gemm_ws(mode)
        f = (lambda: K.gemm(a, b, M, N, Kd, Kd, Kd, True, True, out_bf16=outb, bias=bias)) if lay == 'NT' else \
            (lambda: K.gemm(a, b, M, N, Kd, Kd, N, True, False, out_bf16=outb))
        f(); torch.cuda.synchronize()
        got = outb.float() - (bias if lay == 'NT' else 0)
        err = ((got - ref).norm() / ref.norm()).item()
        us = ktime(f)
        res.append('ERR %.2g   ' % err if err > 6e-3 else '%5.1f %5.0f' % (us, 2.0 * M * N * Kd / us / 1e6))
    L.vqa_set_gemm_ws(1)
    print('%-4s %-16s %s' % (lay, f'{M},{N},{Kd}', ' | '.join(res)), flush=True)
