import sys, ctypes as C, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '.')
from vqa_model_builder_amd.hip import lib as L_, kernels as K
L_.load()
lab = C.CDLL('scratch/lab_attn.so')
lab.lab_attn_bwd.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
B, H, S, Dh = 32, 12, 64, 64
D = H * Dh
bf = lambda *s: torch.randn(s, device='cuda').to(torch.bfloat16)
qkv, do = bf(B * S, 3 * D), bf(B * S, D)
dqkv = torch.empty((B * S, 3 * D), device='cuda', dtype=torch.bfloat16)
cs = torch.zeros(3 * D, device='cuda')
d = L_.VqaAttnDesc()
p = lambda t: C.c_void_p(t.data_ptr())
d.q, d.k, d.v = p(qkv[:, :D]), p(qkv[:, D:2 * D]), p(qkv[:, 2 * D:])
d.ldq = d.ldk = d.ldv = 3 * D; d.ldo = D
d.B, d.H, d.Sq, d.Skv, d.Dh = B, H, S, S, Dh
d.key_padding_mask = None; d.scale = 0.0; d.drop_p = 0.1; d.drop_seed = 1234; d.drop_stream = 3
d.d_o, d.ldd_o = p(do), D
d.dq, d.dk, d.dv = p(dqkv[:, :D]), p(dqkv[:, D:2 * D]), p(dqkv[:, 2 * D:])
d.lddq = d.lddk = d.lddv = 3 * D
for want in (True, False):
    d.dq_colsum, d.dk_colsum, d.dv_colsum = (p(cs[:D]), p(cs[D:2 * D]), p(cs[2 * D:])) if want else (None, None, None)
    trace = torch.zeros((B * H, 8), dtype=torch.int64, device='cuda')
    for _ in range(3): assert lab.lab_attn_bwd(C.byref(d), None, None) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): lab.lab_attn_bwd(C.byref(d), None, None)
    e1.record(); torch.cuda.synchronize()
    lab.lab_attn_bwd(C.byref(d), C.c_void_p(trace.data_ptr()), None); torch.cuda.synchronize()
    t = trace.cpu().numpy()
    names = ['stage tiles + sync', 'phase 1 (scores, dP, dS, dQ)', 'barrier wait', 'phase 2 (dK, dV)', 'colsum + drain']
    print('colsum' if want else 'no colsum', '%.1f us/launch' % (e0.elapsed_time(e1) / 20 * 1e3))
    for k, n in enumerate(names):
        dlt = t[:, k + 1] - t[:, k]
        print('   %-30s p50 %6d  p90 %6d cycles' % (n, np.median(dlt), np.percentile(dlt, 90)))
    print('   total p50', int(np.median(t[:, 5] - t[:, 0])))
