"""Builds the HIP library in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m vqa_model_builder_amd.csrc.build [--force]

One object per .hip file and operand type (cached by mtime), linked into
    vqa_model_builder_amd/csrc/libvqa_hip.so       bfloat16 GEMM / attention operands  (torch autocast bf16 scheme)
    vqa_model_builder_amd/csrc/libvqa_hip_f16.so   IEEE fp16 operands (-DVQA_HALF_F16): what the reference's main loop runs under
Same sources, same C ABI (include/vqa_hip.h), same symbol names; ``hip.lib.set_half`` picks the handle.
No torch headers are involved.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(HERE, 'libvqa_hip.so')
LIB_F16 = os.path.join(HERE, 'libvqa_hip_f16.so')
VARIANTS = (('', [], LIB), ('_f16', ['-DVQA_HALF_F16'], LIB_F16))
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I' + os.path.join(REPO, 'include'), '-I' + HERE,
         '-Wno-unused-result']


def sources():
    return sorted(f for f in os.listdir(HERE) if f.endswith('.hip'))


def _stale(obj, deps):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hdrs = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith('.h')] + [os.path.join(REPO, 'include', 'vqa_hip.h')]
    objdir = os.path.join(HERE, 'build')
    os.makedirs(objdir, exist_ok=True)
    jobs, links = [], []
    for suffix, defs, lib in VARIANTS:
        objs = []
        for src in sources():
            obj = os.path.join(objdir, src[:-4] + suffix + '.o')
            objs.append(obj)
            if force or _stale(obj, [os.path.join(HERE, src)] + hdrs):
                jobs.append([HIPCC] + FLAGS + defs + ['-c', os.path.join(HERE, src), '-o', obj])
            elif verbose:
                print(f'[build] reused {os.path.relpath(obj, REPO)} (newer than its sources)', flush=True)
        links.append((lib, objs))

    def run(cmd):
        if verbose:
            print('[build] compiling / linking:', ' '.join(os.path.relpath(c, REPO) if os.path.isabs(c) else c for c in cmd[-3:]), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed:\n' + ' '.join(cmd) + '\n' + r.stdout + r.stderr)
    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
        list(ex.map(run, jobs))
    for lib, objs in links:
        if jobs or force or _stale(lib, objs):
            run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + objs)
        elif verbose:
            print(f'[build] reused {os.path.relpath(lib, REPO)}', flush=True)
    if verbose:
        print(f'[build] {len(jobs)} object(s) compiled, {len(VARIANTS) * len(sources()) - len(jobs)} reused', flush=True)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
