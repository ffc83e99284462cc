"""k rotation on / off must agree exactly on integer data (fp32 sums of small integers are exact in any order): every layout, ragged edges, split-K."""
import sys, torch
sys.path.insert(0, '.')
from vqa_model_builder_amd.hip import kernels as K, lib
L = lib.load()
dev = 'cuda'
g = torch.Generator().manual_seed(0)
def ints(shape): return torch.randint(-3, 4, shape, generator=g).float().to(dev).to(torch.bfloat16)
bad = 0
for (M, N, Kd) in [(66, 2304, 768), (32, 2048, 4096), (2048, 768, 3072), (1600, 768, 768), (100, 96, 1000), (130, 136, 520), (2048, 3072, 768), (64, 64000 // 8, 768)]:
    for a_kc, b_kc in [(True, True), (True, False), (False, False), (False, True)]:
        a = ints((M, Kd) if a_kc else (Kd, M)); b = ints((N, Kd) if b_kc else (Kd, N))
        lda = Kd if a_kc else M; ldb = Kd if b_kc else N
        if (not a_kc and M % 8) or (not b_kc and N % 8) or Kd % 8: continue
        for split in (0, 1):
            outs = []
            for rot in (0, 1):
                L.vqa_set_gemm_k_rotate(rot)
                o = torch.zeros((M, N), device=dev)
                try:
                    K.gemm(a, b, M, N, Kd, lda, ldb, a_kc, b_kc, out_f32=o, allow_split_k=bool(split))
                except Exception as e:
                    o = None
                outs.append(o)
            torch.cuda.synchronize()
            if outs[0] is None or outs[1] is None: continue
            af = a.float() if a_kc else a.float().t(); bf = b.float().t() if b_kc else b.float()
            ref = af @ bf
            e0, e1 = (outs[0] - ref).abs().max().item(), (outs[1] - ref).abs().max().item()
            if e0 != 0 or e1 != 0:
                bad += 1
                print('MISMATCH', (M, N, Kd), 'a_kc', a_kc, 'b_kc', b_kc, 'split', split, 'err off/on', e0, e1, flush=True)
L.vqa_set_gemm_k_rotate(1)
print('bad', bad)
