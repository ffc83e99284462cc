"""Grouped weight-gradient launch of the MoE experts' Linear layers (dW[N, K] = dY[M, N]^T X[M, K] with M = 32 tokens: rank-32 outer products, 2048 x 2048 ..
4096 x 2048 outputs): a store-bound launch (fp32 dW written once).  Grouped tiles compared; GB/s of dW stores."""
import ctypes as C, sys, torch
sys.path.insert(0, '.')
from vqa_model_builder_amd.hip import kernels as K, lib
L = lib.load()
dev = 'cuda'
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32
shapes = [(2048, 768), (6144, 2048), (2048, 2048), (2048, 2048), (2048, 2048), (768, 2048),        # vision expert
          (2048, 768), (6144, 2048), (2048, 2048), (4096, 2048), (2048, 4096), (768, 2048),        # text expert
          (2048, 768), (4096, 2048), (2048, 4096), (768, 2048)]                                      # multimodal expert
items = []
for N, Kd in shapes:
    dy = (torch.randn((M, N), device=dev) * 0.1).to(torch.bfloat16)
    x = torch.randn((M, Kd), device=dev).to(torch.bfloat16)
    out = torch.empty((N, Kd), device=dev, dtype=torch.float32)
    items.append((dy, x, M, N, Kd, N, Kd, out, None))
nbytes = sum(4.0 * it[3] * it[4] for it in items)
def collect():
    f, ms, n = (C.c_double * 1)(), (C.c_double * 1)(), (C.c_int * 1)()
    L.vqa_gemm_profile_collect(1, f, ms, n)
    return ms[0] / max(n[0], 1) * 1e3
names = {3: '128x128/2', 2: '128x64/2', 1: '64x64', 7: '128x128/3', 4: '256x128/2 8w'}
for pers in (0, 512, 1024):
    L.vqa_set_gemm_group_persistent(pers)
    for tile in (3, 2, 1, 4):
        L.vqa_set_gemm_group_tile(tile)
        K._launch_group(items); torch.cuda.synchronize()
        ref = items[1][0].float().t() @ items[1][1].float()
        err = ((items[1][7] - ref).norm() / ref.norm()).item()
        L.vqa_gemm_profile(1, 0)
        for _ in range(5): K._launch_group(items)
        torch.cuda.synchronize()
        us = collect()
        L.vqa_gemm_profile(0, 0)
        print('persistent %4d tile %-14s %7.1f us   %6.0f GB/s of dW stores   rel err %.1e' % (pers, names[tile], us, nbytes / us / 1e3, err), flush=True)
L.vqa_set_gemm_group_tile(0); L.vqa_set_gemm_group_persistent(0)
