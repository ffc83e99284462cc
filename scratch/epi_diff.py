"""debug: where a compile-time epilogue form differs from the generic one"""
import math, sys, torch
sys.path.insert(0, '.')
from vqa_model_builder_amd.hip import kernels as K, lib as hl
L = hl.load(); DEV = 'cuda'; BF = torch.bfloat16
g = torch.Generator().manual_seed(11)
real = lambda shape, s=1.0: (torch.randn(shape, generator=g) * s).to(DEV)
kinds = [('NT', dict(bias=1, b16=1)), ('NT', dict(f32=1)), ('NT', dict(bias=1, act=K.ACT_GELU, pre=1, drop=1, b16=1)), ('NT', dict(bias=1, act=K.ACT_GELU, drop=1, b16=1)),
         ('NT', dict(bias=1, drop=1, b16=1)), ('NT', dict(bias=1, act=K.ACT_GELU, pre=1, b16=1)),
         ('NN', dict(actb=K.ACT_GELU, drop=1, b16=1, colsum=1)), ('NT', dict(bias=1, f32=1))]
for (M, N, Kd) in [(2048, 3072, 768), (2048, 768, 768), (128, 2048, 768)]:
    a = real((M, Kd)).to(BF); w_nt = real((N, Kd), 1.0 / math.sqrt(Kd)).to(BF); w_nn = w_nt.t().contiguous()
    bias, res, z = real((N,)), real((M, N)), real((M, N)).to(BF)
    for lay, o in kinds:
        outs = []
        for mode in (1, 5, 0):
            L.vqa_set_gemm_v1_fast(mode)
            of = torch.zeros((M, N), device=DEV) if o.get('f32') else None
            ob = torch.zeros((M, N), device=DEV, dtype=BF) if o.get('b16') else None
            pre = torch.zeros((M, N), device=DEV, dtype=BF) if o.get('pre') else None
            cs = torch.zeros((N,), device=DEV) if o.get('colsum') else None
            K.gemm(a, w_nt if lay == 'NT' else w_nn, M, N, Kd, Kd, Kd if lay == 'NT' else N, True, lay == 'NT', out_f32=of, out_bf16=ob, pre_bf16=pre,
                   bias=bias if o.get('bias') else None, residual=res if o.get('res') else None, act_grad_of=z if o.get('actb') else None,
                   act=o.get('act', K.ACT_NONE), act_bwd=o.get('actb', K.ACT_NONE), drop=K.Drop(0.1, 99, 3) if o.get('drop') else K.NO_DROP, colsum=cs)
            outs.append((of, ob, pre))
        for name, i in (('f32', 0), ('b16', 1), ('pre', 2)):
            x = outs[0][i]
            if x is None: continue
            for other, oname in ((outs[1], 'fast-generic'), (outs[2], 'general')):
                y = other[i]
                d = (x.float() != y.float())
                if d.any():
                    idx = d.nonzero()[:4]
                    print(M, N, Kd, lay, o, name, 'vs', oname, 'mismatches', int(d.sum()), 'of', d.numel(), [(tuple(j.tolist()), x[tuple(j)].item(), y[tuple(j)].item()) for j in idx])
print('done')
