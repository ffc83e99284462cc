"""GEMM shapes of one encoder layer WITH the epilogues the block runners fuse (the in-situ cost, minus cold caches)."""
import sys, torch
sys.path.insert(0, '.')
from vqa_model_builder_amd.hip import kernels as K, lib
L = lib.load()
dev = 'cuda'
def bench(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
M, D, I = 2048, 768, 3072
bf = lambda *s: torch.randn(s, device=dev).to(torch.bfloat16)
x, w1, w2, g = bf(M, D), bf(I, D), bf(D, I), bf(M, I)
b1, b2 = torch.randn(I, device=dev), torch.randn(D, device=dev)
res = torch.randn(M, D, device=dev)
pre, act = torch.empty((M, I), device=dev, dtype=torch.bfloat16), torch.empty((M, I), device=dev, dtype=torch.bfloat16)
yf = torch.empty((M, D), device=dev); yb = torch.empty((M, D), device=dev, dtype=torch.bfloat16)
dyb, da = bf(M, D), torch.empty((M, I), device=dev, dtype=torch.bfloat16)
cs = torch.zeros(I, device=dev)
drop = K.Drop(0.1, 1234, 5)
cases = {
 'fc1 fwd plain (bf16 out)': lambda: K.gemm(x, w1, M, I, D, D, D, True, True, out_bf16=act),
 'fc1 fwd +bias+pre+GELU': lambda: K.gemm(x, w1, M, I, D, D, D, True, True, out_bf16=act, pre_bf16=pre, bias=b1, act=K.ACT_GELU),
 'fc1 fwd +bias+pre+QGELU': lambda: K.gemm(x, w1, M, I, D, D, D, True, True, out_bf16=act, pre_bf16=pre, bias=b1, act=K.ACT_QUICK_GELU),
 'fc2 fwd plain (f32 out)': lambda: K.gemm(g, w2, M, D, I, I, I, True, True, out_f32=yf),
 'fc2 fwd +bias+res': lambda: K.gemm(g, w2, M, D, I, I, I, True, True, out_f32=yf, bias=b2, residual=res),
 'fc2 fwd +bias+drop+res': lambda: K.gemm(g, w2, M, D, I, I, I, True, True, out_f32=yf, bias=b2, residual=res, drop=drop),
 'fc2 dX plain (bf16 out)': lambda: K.gemm(dyb, w2, M, I, D, D, I, True, False, out_bf16=da),
 'fc2 dX +GELU bwd': lambda: K.gemm(dyb, w2, M, I, D, D, I, True, False, out_bf16=da, act_grad_of=pre, act_bwd=K.ACT_GELU),
 'fc2 dX +GELU bwd+colsum': lambda: K.gemm(dyb, w2, M, I, D, D, I, True, False, out_bf16=da, act_grad_of=pre, act_bwd=K.ACT_GELU, colsum=cs),
 'fc2 dX +QGELU bwd+colsum': lambda: K.gemm(dyb, w2, M, I, D, D, I, True, False, out_bf16=da, act_grad_of=pre, act_bwd=K.ACT_QUICK_GELU, colsum=cs),
 'fc1 dX plain (f32 out)': lambda: K.gemm(da, w1, M, D, I, I, D, True, False, out_f32=yf),
 'fc1 dX +res': lambda: K.gemm(da, w1, M, D, I, I, D, True, False, out_f32=yf, residual=res),
}
for name, f in cases.items():
    print('%-28s %6.1f us' % (name, bench(f)))
