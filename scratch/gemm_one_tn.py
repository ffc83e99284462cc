import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '.')
from vqa_model_builder_amd.hip import kernels as K, lib
L = lib.load()
pl, hint, M, N, Kd = [int(x) for x in sys.argv[1:6]]
lay = sys.argv[6]
L.vqa_set_gemm_pipeline(pl); L.vqa_set_gemm_group_m(8)
a = torch.randn((M, Kd) if lay != 'TN' else (Kd, M), device='cuda').to(torch.bfloat16)
b = torch.randn((N, Kd) if lay == 'NT' else (Kd, N), device='cuda').to(torch.bfloat16)
out = torch.empty((M, N), device='cuda')
for _ in range(20):
    if lay == 'NN': K.gemm(a, b, M, N, Kd, Kd, N, True, False, out_f32=out, tile_hint=hint)
    else: K.gemm(a, b, M, N, Kd, M, N, False, False, out_f32=out, tile_hint=hint, split_k=1)
torch.cuda.synchronize()
