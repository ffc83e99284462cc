"""Row-major vs column-major tile ids under the XCD remap (an XCD owns rows / columns of the output), weights resident and streamed from HBM."""
import ctypes as C, sys, torch
sys.path.insert(0, '.')
from vqa_model_builder_amd.hip import kernels as K, lib
L = lib.load()
dev = 'cuda'
SHAPES = [('NT', 2048, 3072, 768), ('NT', 2048, 2304, 768), ('NT', 2048, 768, 768), ('NT', 2048, 768, 3072), ('NT', 1600, 3072, 768), ('NT', 1600, 768, 768),
          ('NN', 2048, 3072, 768), ('NN', 2048, 768, 768), ('NN', 2048, 768, 2304), ('NN', 2048, 768, 3072), ('NN', 1600, 3072, 768)]
def collect():
    f, ms, n = (C.c_double * 1)(), (C.c_double * 1)(), (C.c_int * 1)()
    L.vqa_gemm_profile_collect(1, f, ms, n)
    return ms[0] / max(n[0], 1) * 1e3
print('%-4s %-14s %-23s %-23s' % ('lay', 'M,N,K', 'row-major  res / cold', 'col-major  res / cold'), flush=True)
for lay, M, N, Kd in SHAPES:
    R = max(8, int(700e6 // (N * Kd * 2)))
    a = torch.randn((M, Kd), device=dev).to(torch.bfloat16)
    Ws = [torch.randn((N, Kd) if lay == 'NT' else (Kd, N), device=dev).to(torch.bfloat16) for _ in range(R)]
    bias = torch.randn((N,), device=dev)
    outb = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    ref = None
    def g(b):
        if lay == 'NT': K.gemm(a, b, M, N, Kd, Kd, Kd, True, True, out_bf16=outb, bias=bias)
        else: K.gemm(a, b, M, N, Kd, Kd, N, True, False, out_bf16=outb)
    cells = []
    for order in (1, 2):
        L.vqa_set_gemm_tile_order(order)
        g(Ws[0]); torch.cuda.synchronize()
        if ref is None: ref = outb.clone()
        ok = torch.equal(ref, outb)
        res = []
        for mode in range(2):
            for i in range(3): g(Ws[i])
            torch.cuda.synchronize()
            L.vqa_gemm_profile(1, 0)
            for i in range(min(R, 48)): g(Ws[0] if mode == 0 else Ws[i])
            torch.cuda.synchronize()
            res.append(collect())
            L.vqa_gemm_profile(0, 0)
        cells.append('%5.1f / %5.1f %s' % (res[0], res[1], 'ok ' if ok else 'DIFF'))
    L.vqa_set_gemm_tile_order(0)
    print('%-4s %-14s %-23s %-23s' % (lay, f'{M},{N},{Kd}', cells[0], cells[1]), flush=True)
    del Ws; torch.cuda.empty_cache()
