"""Microbench: fused in-projection + attention (one launch) vs packed in-projection GEMM + attention kernel, the shapes of the path."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqa_model_builder_amd.hip import kernels as K

DEV, BF = 'cuda', torch.bfloat16
PEAK = 2.5e15


def timeit(fn, n=200):
    g = torch.cuda.CUDAGraph()
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print('case                          B  H  Sq Skv   two-launch us   fused us   fused TFLOP/s (frac of bf16 peak)')
for name, B, H, Sq, Skv, D, self_attn in [('fusion self', 32, 8, 64, 64, 768, True), ('fusion cross', 32, 8, 64, 50, 768, False),
                                            ('fusion last self (1 query)', 32, 8, 1, 64, 768, False), ('fusion last cross (1 query)', 32, 8, 1, 50, 768, False),
                                            ('phobert layer', 32, 12, 64, 64, 768, True), ('clip layer', 32, 12, 50, 50, 768, True)]:
    xq = torch.randn(B * Sq, D, device=DEV).to(BF)
    xkv = xq if self_attn else torch.randn(B * Skv, D, device=DEV).to(BF)
    w = (torch.randn(3 * D, D, device=DEV) * D ** -0.5).to(BF)
    b = torch.randn(3 * D, device=DEV)
    q = torch.empty(B * Sq, D, device=DEV, dtype=BF); kv = torch.empty(B * Skv, 2 * D, device=DEV, dtype=BF)
    qkv = torch.empty(B * Sq, 3 * D, device=DEV, dtype=BF)
    Dh = D // H

    def two():
        if self_attn:
            _, t, _ = K.linear_fwd(xq, w, b, B * Sq, 3 * D, D, want_bf16=True)
            return K.attention_fwd(t[:, :D], t[:, D:2 * D], t[:, 2 * D:], 3 * D, 3 * D, 3 * D, B, H, Sq, Skv, Dh)
        _, a, _ = K.linear_fwd(xq, w[:D], b[:D], B * Sq, D, D, want_bf16=True)
        _, c, _ = K.linear_fwd(xkv, w[D:], b[D:], B * Skv, 2 * D, D, want_bf16=True)
        return K.attention_fwd(a, c[:, :D], c[:, D:], D, 2 * D, 2 * D, B, H, Sq, Skv, Dh)

    def fused():
        return K.fused_inproj_attention_fwd(xq, xkv, w, b, B, H, Sq, Skv, D, q=q, k=kv[:, :D], v=kv[:, D:], ldq=D, ldk=2 * D, ldv=2 * D)

    t2, t1 = timeit(two), timeit(fused)
    flop = 2.0 * B * (Sq * D * D + 2 * Skv * D * D) + 4.0 * B * H * Sq * Skv * Dh
    print(f'{name:28s} {B:3d} {H:2d} {Sq:3d} {Skv:3d}   {t2:10.1f}   {t1:10.1f}   {flop / t1 / 1e6:8.1f} ({flop / t1 * 1e6 / PEAK:.3f})')
