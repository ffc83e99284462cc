import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '.')
from vqa_model_builder_amd.hip import kernels as K, lib
L = lib.load()
pl, hint, M, N, Kd = [int(x) for x in sys.argv[1:6]]
gm = int(sys.argv[6]) if len(sys.argv) > 6 else 16
L.vqa_set_gemm_pipeline(pl); L.vqa_set_gemm_group_m(gm)
a = torch.randn((M, Kd), device='cuda').to(torch.bfloat16)
b = torch.randn((N, Kd), device='cuda').to(torch.bfloat16)
outb = torch.empty((M, N), device='cuda', dtype=torch.bfloat16)
for _ in range(20):
    K.gemm(a, b, M, N, Kd, Kd, Kd, True, True, out_bf16=outb, tile_hint=hint)
torch.cuda.synchronize()
