"""Microbenchmark of the grouped weight-gradient launch: the cfg2 step's ~110 dW GEMMs (four groups of <= 32) on the 256 x 256 / 8-wave kernel
(csrc/gemm_dw256.h) against the 128 x 128 ring kernel, same process, interleaved rounds; prints TFLOP/s and the share of the 2.5 PFLOP/s peak."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqa_model_builder_amd.hip import lib as hl

L = hl.load()
dev = 'cuda'
BF = torch.bfloat16
torch.manual_seed(0)

def layer_items(T):
    # (tokens, out rows, in cols) of one encoder layer: qkv, out, fc1, fc2
    return [(T, 2304, 768), (T, 768, 768), (T, 3072, 768), (T, 768, 3072)]

cases = []
for _ in range(12):
    cases += layer_items(2048)
for _ in range(12):
    cases += layer_items(1600)
bufs = []
zero = os.environ.get('DW_ZERO') == '1'
mk = (lambda *s: torch.zeros(*s, device=dev).to(BF)) if zero else (lambda *s: torch.randn(*s, device=dev).to(BF))
for (T, No, Ki) in cases:
    bufs.append((mk(T, No), mk(T, Ki), torch.empty(No, Ki, device=dev)))
flop = sum(2.0 * T * No * Ki for T, No, Ki in cases)
st = torch.cuda.current_stream().cuda_stream

def run():
    step = int(os.environ.get('DW_CHUNK', '128'))
    for i0 in range(0, len(cases), step):
        chunk = list(zip(cases[i0:i0 + step], bufs[i0:i0 + step]))
        items = (hl.VqaGemmGroupItem * len(chunk))()
        for it, ((T, No, Ki), (dy, x, out)) in zip(items, chunk):
            it.a, it.b, it.c_f32 = dy.data_ptr(), x.data_ptr(), out.data_ptr()
            it.M, it.N, it.K, it.lda, it.ldb, it.ldc = No, Ki, T, No, Ki, Ki
        rc = L.vqa_gemm_bf16_grouped2(items, len(chunk), 0, 0, None, st)
        assert rc == 0, rc

res = {0: [], 1: []}
for rnd in range(6):
    for big in (1, 0):
        L.vqa_set_gemm_dw256(big)
        run(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        res[big].append((time.perf_counter() - t0) / 5)
for big in (1, 0):
    t = sorted(res[big])[len(res[big]) // 2]
    print(f'dw256={big}: {t * 1e3:.3f} ms per step-equivalent ({flop / 1e9:.0f} GFLOP) = {flop / t / 1e12:.1f} TFLOP/s = {flop / t / 2.5e15 * 100:.1f} % of peak; min {min(res[big]) * 1e3:.3f} ms')
L.vqa_set_gemm_dw256(1)
