"""Runs one scratch/lab_*.so GEMM instance and prints its in-kernel timeline (s_memtime stamps per workgroup).
usage: lab_run.py LIB LAY M N K [GROUP_M]"""
import sys, ctypes as C, numpy as np, torch
import os
lib = C.CDLL(sys.argv[1]); lay = sys.argv[2]
if os.environ.get('LAB_FAST') == '0': lib.lab_set_fast(0)          # the general-form instantiation
M, N, Kd = [int(x) for x in sys.argv[3:6]]
gm = int(sys.argv[6]) if len(sys.argv) > 6 else 8
vp = C.c_void_p
lib.lab_gemm.argtypes = [vp, vp, vp, vp] + [C.c_int] * 8 + [vp, vp]
a = torch.randn((M, Kd) if lay != 'TN' else (Kd, M), device='cuda').to(torch.bfloat16)
b = torch.randn((N, Kd) if lay == 'NT' else (Kd, N), device='cuda').to(torch.bfloat16)
outb = torch.empty((M, N), device='cuda', dtype=torch.bfloat16)
outf = torch.empty((M, N), device='cuda')
trace = torch.zeros((8192, 32), dtype=torch.int64, device='cuda')
a_kc, b_kc = int(lay != 'TN'), int(lay == 'NT')
lda = Kd if a_kc else M
ldb = Kd if b_kc else N
def run(tr):
    r = lib.lab_gemm(a.data_ptr(), b.data_ptr(), outb.data_ptr() if lay != 'TN' else None, outf.data_ptr() if lay == 'TN' else None,
                     M, N, Kd, lda, ldb, a_kc, b_kc, gm, tr, None)
    assert r == 0, r
for _ in range(5): run(None)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run(None)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
af, bf = a.float(), b.float()
ref = (af if lay != 'TN' else af.t()) @ (bf.t() if lay == 'NT' else bf)
got = outb.float() if lay != 'TN' else outf
print(f'{sys.argv[1]} fast={os.environ.get("LAB_FAST", "1")} {lay} {M}x{N}x{Kd}: {us:.1f} us/launch  {2.0 * M * N * Kd / us / 1e6:.0f} TF   rel err {((got - ref).norm() / ref.norm()).item():.2e}')
run(trace.data_ptr()); torch.cuda.synchronize()
t = trace.cpu().numpy().astype(np.int64)
t = t[t[:, 0] != 0]
nwg = len(t)
t0 = t[:, 0].min()
CLK = 1.0   # ticks; report in ticks and assume 100 MHz? print both
span = t[:, 28].max() - t0
steps = [(i, t[:, 2 + i]) for i in range(24 if Kd > 1152 else 18) if (t[:, 2 + i] != 0).all()]
nst = len(steps)
print(f'workgroups {nwg}, traced k-steps {nst}, kernel span {span} ticks')
def stat(x): return f'min {x.min():7d} p50 {int(np.median(x)):7d} p90 {int(np.percentile(x, 90)):7d} max {x.max():7d}'
print('start offset      ', stat(t[:, 0] - t0))
print('init+prologue     ', stat(t[:, 1] - t[:, 0]))
print('  of it: dma_init  ', stat(t[:, 31] - t[:, 0]))
print('first tile landed ', stat(t[:, 2] - t[:, 1]))
if nst > 1:
    d = np.stack([steps[i + 1][1] - steps[i][1] for i in range(nst - 1)], 1)
    print('k-step (all)      ', stat(d.reshape(-1)))
    for i in range(nst - 1): print(f'   step {i:2d}->{i+1:2d}    ', stat(d[:, i]))
last = steps[-1][1]
print('last step->loopend', stat(t[:, 26] - last))
print('epilogue issue    ', stat(t[:, 27] - t[:, 26]))
if Kd <= 1152:
    for i, nm in enumerate(['entry->bias etc', 'acc->scratch    ', 'rows of group 0 ', 'remaining groups']):
        print('   epi', nm, stat(t[:, 20 + i] - (t[:, 26] if i == 0 else t[:, 19 + i])))
print('store drain       ', stat(t[:, 28] - t[:, 27]))
print('wg total          ', stat(t[:, 28] - t[:, 0]))
print('end offset        ', stat(t[:, 28] - t0))
xcc = t[:, 29] & 0xf
print('wgs per XCC', np.bincount(xcc.astype(int)))
