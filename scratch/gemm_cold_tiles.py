"""Tile sweep under the in-situ condition (weights streamed from HBM: rotating over > 600 MB of weight buffers) next to the resident-weights number.
cfg: 1 = 64x64, 4 = 128x64, 5 = 64x128, 0 = 128x128, 6 = 256x128 (8 waves); ws = gemm_ws auto."""
import ctypes as C, sys, torch
sys.path.insert(0, '.')
from vqa_model_builder_amd.hip import kernels as K, lib
L = lib.load()
dev = 'cuda'
SHAPES = [('NT', 2048, 768, 3072), ('NN', 2048, 768, 3072), ('NT', 2048, 3072, 768), ('NN', 2048, 3072, 768), ('NT', 2048, 768, 768), ('NN', 2048, 768, 2304), ('NT', 1600, 768, 3072)]
SETTINGS = [('default', None), ('64x64/2', (1, 2)), ('64x64/3', (1, 3)), ('64x64/4', (1, 4)), ('128x64/2', (4, 2)), ('128x64/3', (4, 3)), ('64x128/2', (5, 2)), ('128x128/2', (0, 2)), ('ws', 'ws')]
def collect():
    f, ms, n = (C.c_double * 1)(), (C.c_double * 1)(), (C.c_int * 1)()
    L.vqa_gemm_profile_collect(1, f, ms, n)
    return ms[0] / max(n[0], 1) * 1e3
print('%-4s %-14s %s' % ('lay', 'M,N,K', ' | '.join('%-11s' % s[0] for s in SETTINGS)), flush=True)
for lay, M, N, Kd in SHAPES:
    wbytes = N * Kd * 2
    R = max(8, int(700e6 // wbytes))
    a = torch.randn((M, Kd), device=dev).to(torch.bfloat16)
    Ws = [torch.randn((N, Kd) if lay == 'NT' else (Kd, N), device=dev).to(torch.bfloat16) for _ in range(R)]
    bias = torch.randn((N,), device=dev)
    outb = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    def g(b):
        if lay == 'NT': K.gemm(a, b, M, N, Kd, Kd, Kd, True, True, out_bf16=outb, bias=bias)
        else: K.gemm(a, b, M, N, Kd, Kd, N, True, False, out_bf16=outb)
    cells = []
    for name, st in SETTINGS:
        L.vqa_set_gemm_ws(0); L.vqa_set_gemm_force(-1, 2)
        if st == 'ws': L.vqa_set_gemm_ws(1)
        elif st is not None: L.vqa_set_gemm_force(st[0], st[1])
        res = []
        try:
            for mode in range(2):
                for i in range(3): g(Ws[i])
                torch.cuda.synchronize()
                L.vqa_gemm_profile(1, 0)
                for i in range(min(R, 48)): g(Ws[0] if mode == 0 else Ws[i])
                torch.cuda.synchronize()
                res.append(collect())
                L.vqa_gemm_profile(0, 0)
            cells.append('%5.1f %5.1f' % tuple(res))
        except Exception as e:
            L.vqa_gemm_profile(0, 0)
            cells.append('err        ')
    L.vqa_set_gemm_ws(0); L.vqa_set_gemm_force(-1, 2)
    print('%-4s %-14s %s' % (lay, f'{M},{N},{Kd}', ' | '.join(cells)), flush=True)
    del Ws
    torch.cuda.empty_cache()
