"""Per-XCD k rotation in the ring GEMMs: off / on, weights resident / streamed from HBM AND activations rotating too (the in-situ condition:
neither operand warm in the XCDs' L2s)."""
import ctypes as C, sys, torch
sys.path.insert(0, '.')
from vqa_model_builder_amd.hip import kernels as K, lib
L = lib.load()
dev = 'cuda'
SHAPES = [('NT', 2048, 3072, 768), ('NT', 2048, 2304, 768), ('NT', 2048, 768, 768), ('NT', 2048, 768, 3072), ('NT', 1600, 3072, 768),
          ('NN', 2048, 3072, 768), ('NN', 2048, 768, 768), ('NN', 2048, 768, 2304), ('NN', 2048, 768, 3072)]
def collect():
    f, ms, n = (C.c_double * 1)(), (C.c_double * 1)(), (C.c_int * 1)()
    L.vqa_gemm_profile_collect(1, f, ms, n)
    return ms[0] / max(n[0], 1) * 1e3
print('%-4s %-14s %-30s %-30s' % ('lay', 'M,N,K', 'rotate off: res / W cold / A+W cold', 'rotate on'), flush=True)
for lay, M, N, Kd in SHAPES:
    R = max(8, int(600e6 // (N * Kd * 2)))
    RA = max(8, int(600e6 // (M * Kd * 2)))
    As = [torch.randn((M, Kd), device=dev).to(torch.bfloat16) for _ in range(RA)]
    Ws = [torch.randn((N, Kd) if lay == 'NT' else (Kd, N), device=dev).to(torch.bfloat16) for _ in range(R)]
    bias = torch.randn((N,), device=dev)
    outb = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    def g(a, b):
        if lay == 'NT': K.gemm(a, b, M, N, Kd, Kd, Kd, True, True, out_bf16=outb, bias=bias)
        else: K.gemm(a, b, M, N, Kd, Kd, N, True, False, out_bf16=outb)
    cells, ref = [], None
    for rot in (0, 1):
        L.vqa_set_gemm_k_rotate(rot)
        g(As[0], Ws[0]); torch.cuda.synchronize()
        if ref is None: ref = outb.float().clone()
        err = ((outb.float() - ref).norm() / ref.norm()).item()
        res = []
        for mode in range(3):
            for i in range(3): g(As[0], Ws[i])
            torch.cuda.synchronize()
            L.vqa_gemm_profile(1, 0)
            for i in range(48): g(As[0] if mode < 2 else As[i % RA], Ws[0] if mode == 0 else Ws[i % R])
            torch.cuda.synchronize()
            res.append(collect())
            L.vqa_gemm_profile(0, 0)
        cells.append('%5.1f / %5.1f / %5.1f  (%.0e)' % (res[0], res[1], res[2], err))
    L.vqa_set_gemm_k_rotate(0)
    print('%-4s %-14s %-30s %-30s' % (lay, f'{M},{N},{Kd}', cells[0], cells[1]), flush=True)
    del Ws, As; torch.cuda.empty_cache()
