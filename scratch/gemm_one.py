"""One GEMM shape, one tile configuration, 20 launches: the unit that scratch/pmc_gemm.sh profiles.
usage: gemm_one.py LAY PIPE HINT M N K [GROUP_M]   (LAY in NT/NN/TN)"""
import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '.')
from vqa_model_builder_amd.hip import kernels as K, lib
L = lib.load()
lay = sys.argv[1]
pl, hint, M, N, Kd = [int(x) for x in sys.argv[2:7]]
gm = int(sys.argv[7]) if len(sys.argv) > 7 else 16
L.vqa_set_gemm_pipeline(pl); L.vqa_set_gemm_group_m(gm)
a = torch.randn((M, Kd) if lay != 'TN' else (Kd, M), device='cuda').to(torch.bfloat16)
b = torch.randn((N, Kd) if lay == 'NT' else (Kd, N), device='cuda').to(torch.bfloat16)
out = torch.empty((M, N), device='cuda')
outb = torch.empty((M, N), device='cuda', dtype=torch.bfloat16)
for _ in range(20):
    if lay == 'NT': K.gemm(a, b, M, N, Kd, Kd, Kd, True, True, out_bf16=outb, tile_hint=hint)
    elif lay == 'NN': K.gemm(a, b, M, N, Kd, Kd, N, True, False, out_bf16=outb, tile_hint=hint)
    else: K.gemm(a, b, M, N, Kd, M, N, False, False, out_f32=out, tile_hint=hint, split_k=1)
torch.cuda.synchronize()
