"""The grouped weight-gradient launch of 8 encoder layers (32 GEMMs: dW[N, K] = dY[M, N]^T X[M, K], M = 2048 tokens) on every grouped tile, distinct
buffers per GEMM (~400 MB: operands come from HBM as in the step).  Kernel time from the dispatch timestamps; one result checked against torch."""
import ctypes as C, sys, torch
sys.path.insert(0, '.')
from vqa_model_builder_amd.hip import kernels as K, lib
L = lib.load()
dev = 'cuda'
M = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
layers = 8
shapes = [(2304, 768), (768, 768), (3072, 768), (768, 3072)]         # (N = out features, K = in features)
items = []
for l in range(layers):
    for N, Kd in shapes:
        dy = (torch.randn((M, N), device=dev) * 0.1).to(torch.bfloat16)
        x = torch.randn((M, Kd), device=dev).to(torch.bfloat16)
        out = torch.empty((N, Kd), device=dev, dtype=torch.float32)
        items.append((dy, x, M, N, Kd, N, Kd, out, None))
flop = sum(2.0 * M * it[3] * it[4] for it in items)
def collect():
    f, ms, n = (C.c_double * 1)(), (C.c_double * 1)(), (C.c_int * 1)()
    L.vqa_gemm_profile_collect(1, f, ms, n)
    return ms[0] / max(n[0], 1) * 1e3
names = {3: '128x128/2 (default)', 7: '128x128/3', 8: '128x128/4', 4: '256x128/2 8w', 5: '256x128/3 8w', 6: '256x256/2 8w', 2: '128x64/2', 1: '64x64'}
for tile in (3, 7, 4, 5, 2):
    L.vqa_set_gemm_group_tile(tile)
    try:
        for it in items: it[7].zero_()
        K._launch_group(items); torch.cuda.synchronize()
        err = 0.0
        for it in (items[0], items[3], items[-1]):
            ref = it[0].float().t() @ it[1].float()
            err = max(err, ((it[7] - ref).norm() / ref.norm()).item())
        L.vqa_gemm_profile(1, 0)
        for _ in range(5): K._launch_group(items)
        torch.cuda.synchronize()
        us = collect()
        L.vqa_gemm_profile(0, 0)
        print('tile %-20s %7.1f us  %6.0f TFLOP/s  %.1f %% of peak   rel err %.1e' % (names[tile], us, flop / us / 1e6, flop / us / 1e6 / 25.0, err), flush=True)
    except Exception as e:
        L.vqa_gemm_profile(0, 0)
        print('tile', names[tile], 'failed:', e, flush=True)
L.vqa_set_gemm_group_tile(0)
