"""Runs one scratch/labws_*.so gemm_ws instance and prints its in-kernel timeline: consumer wave 0 and loader wave 4 of every workgroup.
usage: lab_run_ws.py LIB LAY M N K"""
import sys, ctypes as C, numpy as np, torch
lib = C.CDLL(sys.argv[1]); lay = sys.argv[2]
M, N, Kd = [int(x) for x in sys.argv[3:6]]
vp = C.c_void_p
lib.lab_gemm.argtypes = [vp, vp, vp, vp] + [C.c_int] * 8 + [vp, vp]
a = torch.randn((M, Kd), device='cuda').to(torch.bfloat16)
b = torch.randn((N, Kd) if lay == 'NT' else (Kd, N), device='cuda').to(torch.bfloat16)
outb = torch.empty((M, N), device='cuda', dtype=torch.bfloat16)
trace = torch.zeros((4096, 64), dtype=torch.int64, device='cuda')
b_kc = int(lay == 'NT')
def run(tr):
    r = lib.lab_gemm(a.data_ptr(), b.data_ptr(), outb.data_ptr(), None, M, N, Kd, Kd, Kd if b_kc else N, 1, b_kc, 0, tr, None)
    assert r == 0, r
for _ in range(5): run(None)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run(None)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
ref = a.float() @ (b.float().t() if lay == 'NT' else b.float())
print(f'{sys.argv[1]} {lay} {M}x{N}x{Kd}: {us:.1f} us/launch back-to-back  {2.0 * M * N * Kd / us / 1e6:.0f} TF   rel err {((outb.float() - ref).norm() / ref.norm()).item():.2e}')
run(trace.data_ptr()); torch.cuda.synchronize()
t = trace.cpu().numpy().astype(np.int64)
t = t[t[:, 0] != 0]
c, l = t[:, :32], t[:, 32:]
t0 = min(c[:, 0].min(), l[:, 0].min())
def stat(x): return f'min {x.min():7d} p50 {int(np.median(x)):7d} p90 {int(np.percentile(x, 90)):7d} max {x.max():7d}'
nst = sum(1 for i in range(24) if (c[:, 2 + i] != 0).all())
print(f'workgroups {len(t)}, traced k-steps {nst}, kernel span {c[:, 28].max() - t0} ticks (s_memtime)')
print('wg start offset       ', stat(c[:, 0] - t0))
print('LOADER init -> primed ', stat(l[:, 1] - l[:, 0]))
print('LOADER primed -> tile0', stat(l[:, 2] - l[:, 1]))
for i in range(1, nst): print(f'LOADER tile {i-1:2d}->{i:2d} landed', stat(l[:, 2 + i] - l[:, 1 + i]))
print('CONS init             ', stat(c[:, 1] - c[:, 0]))
print('CONS wait first tile  ', stat(c[:, 2] - c[:, 1]))
for i in range(1, nst): print(f'CONS k-step {i-1:2d}->{i:2d}     ', stat(c[:, 2 + i] - c[:, 1 + i]))
print('CONS last step->end   ', stat(c[:, 26] - c[:, 1 + nst]))
print('CONS epilogue issue   ', stat(c[:, 27] - c[:, 26]))
print('CONS store drain      ', stat(c[:, 28] - c[:, 27]))
print('wg total              ', stat(c[:, 28] - c[:, 0]))
print('end offset            ', stat(c[:, 28] - t0))
