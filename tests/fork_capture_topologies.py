"""Stream-fork topologies inside a HIP-graph capture, each variant in its own process.

Default: variants A and B -- the ONE-level forks the captured training step relies on (vision encoder; MoE specialised expert; siblings and
lazily created streams included): tests/test_graph_gpu.py::test_capture_fork_topologies_the_step_relies_on runs exactly these.

Variants C, D, E (a fork nested inside a forked branch: the topology of round 1's core dump, gpurun_out/g3.log 'towers 1 wgrad 1') SIGSEGV
inside the capture in this runtime whatever runs on the streams -- diagnosed once, written up in profiles/r02/nested_fork_capture.md, and NOT
re-provoked by the suite: they run only on request,  python tests/fork_capture_topologies.py C D E  (or VQA_NESTED_FORK_DIAG=1)."""
import os, subprocess, sys, textwrap

COMMON = '''
import sys, torch
sys.path.insert(0, %r)
from vqa_model_builder_amd.hip import kernels as K
dev = 'cuda'
def work(x, w):
    if PLAIN:
        return x * 2
    _, y, _ = K.linear_fwd(x, w, None, x.shape[0], w.shape[0], w.shape[1], want_bf16=True)
    return y
x = torch.randn(256, 512, device=dev).to(torch.bfloat16); w = torch.randn(512, 512, device=dev).to(torch.bfloat16)
outs = []
def branch(depth, lazy):
    # runs on the current stream; forks a child stream (created inside the capture when lazy), child forks a grandchild at depth 2
    cur = torch.cuda.current_stream()
    child = torch.cuda.Stream() if lazy else POOL.pop()
    child.wait_stream(cur)
    with torch.cuda.stream(child):
        outs.append(work(x, w))
        if depth > 1:
            branch(depth - 1, lazy)
    outs.append(work(x, w))
    cur.wait_stream(child)
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

VARIANTS = {
    'A single fork x1, streams made before the capture': 'POOL=[torch.cuda.Stream() for _ in range(8)]\nDEPTH, LAZY, TOWERS, PLAIN = 1, False, 1, False',
    'B single fork x2 (two sibling branches), streams made inside': 'POOL=[]\nDEPTH, LAZY, TOWERS, PLAIN = 1, True, 2, False',
    'C nested fork (fork inside a forked branch) x1, made before': 'POOL=[torch.cuda.Stream() for _ in range(8)]\nDEPTH, LAZY, TOWERS, PLAIN = 2, False, 1, False',
    'D nested fork x1, plain torch elementwise ops only': 'POOL=[torch.cuda.Stream() for _ in range(8)]\nDEPTH, LAZY, TOWERS, PLAIN = 2, False, 1, True',
    'E nested fork x2, streams made inside (the g3 topology)': 'POOL=[]\nDEPTH, LAZY, TOWERS, PLAIN = 2, True, 2, False',
}
TAIL = '''
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(s):
    work(x, w)
    with torch.cuda.graph(g, stream=s):
        for _ in range(TOWERS):
            branch(DEPTH, LAZY)
print('captured', flush=True)
torch.cuda.synchronize()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
print('replayed', flush=True)
ref = work(x, w)
torch.cuda.synchronize()
assert all(torch.equal(o, ref) for o in outs)
print('OK', len(outs))
'''
want = [a.upper() for a in sys.argv[1:]] or (['A', 'B', 'C', 'D', 'E'] if os.environ.get('VQA_NESTED_FORK_DIAG') == '1' else ['A', 'B'])
for name, head in VARIANTS.items():
    if name[0] not in want:
        continue
    code = COMMON + head + TAIL
    try:
        r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=120)
        last = ' / '.join(r.stdout.strip().splitlines())
        err = (r.stderr.strip().splitlines() or [''])[-1][:160]
        print(f'{name:75s} rc={r.returncode:4d} {last} {err if r.returncode else ""}', flush=True)
    except subprocess.TimeoutExpired:
        print(f'{name:75s} TIMEOUT', flush=True)
