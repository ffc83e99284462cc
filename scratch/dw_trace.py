"""In-kernel timeline of the 256 x 256 weight-gradient kernel (lab build -DDW_TRACE: scratch/ab_build.sh dwtrace -DDW_TRACE): s_memtime stamps of the first
tile of every workgroup; prints median cycle counts per segment."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vqa_model_builder_amd.hip import lib as hl
L = hl.load()
raw = C.CDLL(os.environ['VQA_HIP_LIB'])
dev, BF = 'cuda', torch.bfloat16
mode = os.environ.get('DW_MODE', 'big')
T = 2048
keep = []
if mode == 'big':                      # two 3072 x 3072 outputs: 288 tiles, row-major tile order
    shapes, share = [(3072, 3072)] * 2, False
elif mode == 'same':                   # 64 items of ONE 256 x 256 tile each, all reading the SAME two operand panels: every DMA an L2 hit
    shapes, share = [(256, 256)] * 64, True
    shapes = shapes * 1
elif mode == 'mix':                    # one encoder layer's four weight gradients x 12 (text): 1296 tiles, ~5 per workgroup
    shapes, share = [(2304, 768), (768, 768), (3072, 768), (768, 3072)] * 12, False
elif mode == 'distinct':               # 64 one-tile items with operands of their own: no reuse at all
    shapes, share = [(256, 256)] * 64, False
n = len(shapes)
items = (hl.VqaGemmGroupItem * n)()
shared = None
for it, (No, Ki) in zip(items, shapes):
    if share and shared is not None:
        dy, x = shared
    else:
        dy, x = torch.randn(T, No, device=dev).to(BF), torch.randn(T, Ki, device=dev).to(BF)
        shared = (dy, x)
    out = torch.empty(No, Ki, device=dev)
    it.a, it.b, it.c_f32 = dy.data_ptr(), x.data_ptr(), out.data_ptr()
    it.M, it.N, it.K, it.lda, it.ldb, it.ldc = No, Ki, T, No, Ki, Ki
    keep += [dy, x, out]
st = torch.cuda.current_stream().cuda_stream
print('mode', mode, 'items', n)
for _ in range(3):
    assert L.vqa_gemm_bf16_grouped2(items, n, 0, 0, None, st) == 0
torch.cuda.synchronize()
buf = (C.c_ulonglong * (256 * 16))()
assert raw.vqa_dw_trace_read(buf) == 0
tr = np.array(buf[:], dtype=np.int64).reshape(256, 16)
tr = tr[tr[:, 13] > 0]
seg = lambda a, b: int(np.median(tr[:, b] - tr[:, a]))
print('workgroups traced', len(tr), 'k-tiles', int(tr[0, 14]))
print('entry -> prologue DMA issued      ', seg(0, 1))
print('first tile landed (wait + barrier) ', seg(1, 2))
for t in range(4):
    print(f'k-tile {t}: compute {seg(2 + 2 * t if t == 0 else 4 + 2 * (t - 1), 3 + 2 * t)}  wait+barrier {seg(3 + 2 * t, 4 + 2 * t)}')
print('k-tiles 4 .. end                  ', seg(10, 11), 'per k-tile', (seg(10, 11)) // max(1, int(tr[0, 14]) - 4))
print('epilogue                          ', seg(11, 12), ' store drain', seg(12, 13))
print('whole tile                        ', seg(0, 13))
print('clock MHz (s_memtime / s_memrealtime x 100)', int(np.median((tr[:, 13] - tr[:, 0]) / np.maximum(1, tr[:, 15]) * 100)))
