"""Weight-gradient launches of the four MoE experts alone (reduction over 32 / 128 tokens): grouped-kernel tile choices, us per flush."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqa_model_builder_amd.hip import kernels as K, lib
from vqa_model_builder_amd.modeling.moe import VQAMOELayer
L = lib.load()
dev = 'cuda'
moe = VQAMOELayer(input_dim=768, hidden_dim=2048, output_dim=768, num_vision_experts=1, num_text_experts=1, num_multimodal_experts=1,
                  num_specialized_experts=1, top_k=2).to(dev)
items = []
for e, ex in enumerate(moe.experts):
    rows = 128 if type(ex).__name__ == 'SegmentationExpert' else 32
    for n, p in ex.named_parameters():
        if p.dim() == 2 and 'cross_attention' not in n and 'modality_gate' not in n:
            N, Kd = p.shape
            T = rows if ('mask_transformer' in n and 'multihead_attn.in_proj' not in n) else 32
            if 'in_proj_weight' in n and ('spatial_attention' in n or 'self_attention' in n or 'multihead_attn' in n):
                N = N // 3                      # one-key attention: only the V rows get a GEMM
            items.append((T, N, Kd))
tot = sum(n * k for _, n, k in items)
print(len(items), 'weight gradients,', round(tot / 1e6), 'M outputs =', round(tot * 4 / 1e9, 2), 'GB of fp32 stores')
bufs = [(torch.randn(T, N, device=dev).to(torch.bfloat16), torch.randn(T, Kd, device=dev).to(torch.bfloat16), torch.empty(N, Kd, device=dev)) for T, N, Kd in items]


def flush():
    for (T, N, Kd), (dy, x, out) in zip(items, bufs):
        K.linear_dw(dy, x, T, N, Kd, out=out)
    K.wgrad_flush()


def timeit(fn, n=20):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
    torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for tile, name in ((0, 'heuristic'), (1, '64x64'), (2, '128x64'), (3, '128x128')):
    L.vqa_set_gemm_group_tile(tile)
    for pers in (0, 512):
        L.vqa_set_gemm_group_persistent(pers)
        us = timeit(flush)
        print(f'group tile {name:9s} persistent {pers:4d}: {us:7.1f} us per flush  ({tot * 4 / us / 1e6:.2f} TB/s of fp32 stores)', flush=True)
L.vqa_set_gemm_group_tile(0); L.vqa_set_gemm_group_persistent(0)


def single():
    for (T, N, Kd), (dy, x, out) in zip(items, bufs):
        K.gemm(dy, x, N, Kd, T, N, Kd, False, False, out_f32=out)
print(f'one launch per weight gradient: {timeit(single):7.1f} us', flush=True)
