"""NT M x 768 x 3072 (fc2 forward): does the 6144-byte row pitch of the k-contiguous operands cost L2 channel parallelism?  Row pitch of A and of W varied
independently (padded leading dimensions), weights resident and streamed from HBM (rotating buffers)."""
import ctypes as C, sys, torch
sys.path.insert(0, '.')
from vqa_model_builder_amd.hip import kernels as K, lib
L = lib.load()
dev = 'cuda'
def collect():
    f, ms, n = (C.c_double * 1)(), (C.c_double * 1)(), (C.c_int * 1)()
    L.vqa_gemm_profile_collect(1, f, ms, n)
    return ms[0] / max(n[0], 1) * 1e3
for lay, M, N, Kd in [('NT', 2048, 768, 3072), ('NT', 2048, 3072, 768), ('NT', 2048, 768, 768), ('NT', 2048, 2304, 768)]:
    print(lay, M, N, Kd, flush=True)
    bias = torch.randn((N,), device=dev)
    outb = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    for pa in (0, 8, 64, 128, 192, 256):
        cells = []
        for pb in (0, 8, 64, 128, 192, 256):
            lda, ldb = Kd + pa, Kd + pb
            a = torch.randn((M, lda), device=dev).to(torch.bfloat16)
            R = max(8, int(500e6 // (N * ldb * 2)))
            Ws = [torch.randn((N, ldb), device=dev).to(torch.bfloat16) for _ in range(R)]
            res = []
            for mode in range(2):
                for i in range(3): K.gemm(a, Ws[i], M, N, Kd, lda, ldb, True, True, out_bf16=outb, bias=bias)
                torch.cuda.synchronize()
                L.vqa_gemm_profile(1, 0)
                for i in range(min(R, 40)): K.gemm(a, Ws[0] if mode == 0 else Ws[i], M, N, Kd, lda, ldb, True, True, out_bf16=outb, bias=bias)
                torch.cuda.synchronize()
                res.append(collect())
                L.vqa_gemm_profile(0, 0)
            cells.append('%5.1f %5.1f' % tuple(res))
            del Ws, a
            torch.cuda.empty_cache()
        print('  padA %3d | padB 0,8,64,128,192,256: ' % pa + ' | '.join(cells), flush=True)
