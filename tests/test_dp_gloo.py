"""Data-parallel gradient exchange on CPU: world_size 2 over gloo (the N > 1 path of bench.py uses the same
GradReducer over RCCL).  Checks: averaged gradients equal the mean of the per-rank gradients, parameters with a
gradient on only ONE rank are zero-filled on the other (never skipped: no collective mismatch), parameters with no
gradient on ANY rank keep grad None (the optimiser skips them, like the reference's single-process loop would),
frozen parameters are not part of the exchange, and buckets respect the size cap."""

import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from vqa_model_builder_amd.dp import GradReducer
        torch.manual_seed(0)
        shapes = [(7, 5), (33,), (64, 16), (3,), (10, 10), (1,)]
        params = [torch.nn.Parameter(torch.randn(s)) for s in shapes]
        params[3].requires_grad_(False)                       # frozen: not exchanged
        red = GradReducer(params, bucket_mb=0.004)            # ~1k floats per bucket -> several buckets
        assert len(red.buckets) >= 2
        assert all(b.numel * 4 <= 0.004 * 1024 * 1024 or len(b.params) == 1 for b in red.buckets)
        g = torch.Generator().manual_seed(100 + rank)
        local = {}
        for i, p in enumerate(params):
            if not p.requires_grad:
                continue
            if i == 4 and rank == 1:
                continue                                       # only rank 0 has a gradient for param 4 (skipped expert)
            if i == 5:
                continue                                       # nobody has a gradient for param 5 (dead parameter)
            p.grad = torch.randn(p.shape, generator=g)
            local[i] = p.grad.numpy().copy()
        red.reduce()
        out = {i: (None if p.grad is None else p.grad.numpy().copy()) for i, p in enumerate(params)}
        q.put((rank, local, out))
    finally:
        dist.destroy_process_group()


def test_grad_reducer_world2_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    from tests.helpers import collect_from_workers
    res = {rank: (local, out) for rank, local, out in collect_from_workers(q, procs, 2)}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (l0, o0), (l1, o1) = res[0], res[1]
    import numpy as np
    for i in (0, 1, 2):
        want = (l0[i] + l1[i]) / 2
        assert np.allclose(o0[i], want, atol=1e-6) and np.allclose(o1[i], want, atol=1e-6)
    assert np.allclose(o0[4], l0[4] / 2, atol=1e-6) and np.allclose(o1[4], l0[4] / 2, atol=1e-6)   # zero-filled on rank 1
    assert o0[5] is None and o1[5] is None                    # no gradient anywhere -> stays None
    assert o0[3] is None and o1[3] is None                    # frozen


def test_single_process_is_a_noop():
    from vqa_model_builder_amd.dp import GradReducer
    p = torch.nn.Parameter(torch.ones(4))
    p.grad = torch.full((4,), 2.0)
    GradReducer([p]).reduce()
    assert torch.equal(p.grad, torch.full((4,), 2.0))


def _overlap_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from vqa_model_builder_amd.dp import GradReducer
        torch.manual_seed(0)                                  # identical replicas
        model = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.ReLU(), torch.nn.Linear(32, 32), torch.nn.ReLU(),
                                    torch.nn.Linear(32, 4))
        dead = torch.nn.Parameter(torch.ones(5))              # never used: no gradient on any rank
        params = list(model.parameters()) + [dead]            # `dead` is LAST in registration order = FIRST in send order (like the
        red = GradReducer(params, bucket_mb=0.002).attach()   # RoBERTa pooler / CLIP post_layernorm in front of their encoders)
        assert len(red.buckets) >= 3 and red._where[id(dead)][0] == 0
        outs = []
        sent = []
        for step in range(2):                                 # two steps: buckets re-arm
            g = torch.Generator().manual_seed(10 * step + rank)
            x = torch.randn(8, 16, generator=g)
            for p in params:
                p.grad = None
            y = model[0](x) if (rank == 1 and step == 1) else model(x)     # rank 1, step 1: only the first layer gets gradients
            y.square().mean().backward()
            local = [p.grad.numpy().copy() if p.grad is not None else None for p in params]
            red.finalize()
            sent.append(red.sent_before_finalize)
            outs.append((local, [p.grad.numpy().copy() if p.grad is not None else None for p in params]))
        # step 0: nothing is known yet, the dead parameter holds bucket 0 (and, by the in-order rule, all others) until finalize;
        # step 1: the reduced presence bitmap said it never gets a gradient -> on rank 0 (all gradients arrive) every bucket is on
        # the wire BEFORE finalize(); rank 1 computes only the first layer this step, its later buckets wait for finalize
        assert sent[0] == 0, sent
        if rank == 0:
            assert sent[1] == len(red.buckets), (sent, len(red.buckets))
        q.put((rank, outs))
    finally:
        dist.destroy_process_group()


def _late_worker(rank, world, port, q):
    """A parameter that gets its FIRST gradient in step 2, on one rank only, after its bucket has been sent early on the other."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from vqa_model_builder_amd.dp import GradReducer
        torch.manual_seed(0)
        a, b, c = torch.nn.Linear(8, 8), torch.nn.Linear(8, 8), torch.nn.Linear(8, 8)      # c: the "expert" nobody routes to at first
        params = list(a.parameters()) + list(c.parameters()) + list(b.parameters())
        red = GradReducer(params, bucket_mb=0.0002).attach()                                 # one parameter per bucket
        outs = []
        for step in range(3):
            g = torch.Generator().manual_seed(7 * step + rank)
            x = torch.randn(4, 8, generator=g)
            for p in params:
                p.grad = None
            y = b(a(x))
            if step == 2 and rank == 1:
                y = y + c(a(x))                                  # first gradient ever for c, on rank 1 only
            y.square().mean().backward()
            local = [p.grad.numpy().copy() if p.grad is not None else None for p in params]
            red.finalize()
            outs.append((local, [p.grad.numpy().copy() if p.grad is not None else None for p in params]))
        q.put((rank, outs))
    finally:
        dist.destroy_process_group()


def test_first_gradient_of_an_unexpected_parameter_is_not_lost():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_late_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    from tests.helpers import collect_from_workers
    res = dict(collect_from_workers(q, procs, 2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    import numpy as np
    for step in range(3):
        (l0, o0), (l1, o1) = res[0][step], res[1][step]
        for i in range(len(l0)):
            if l0[i] is None and l1[i] is None:
                assert o0[i] is None and o1[i] is None, (step, i)
                continue
            want = ((0 if l0[i] is None else l0[i]) + (0 if l1[i] is None else l1[i])) / 2
            assert np.allclose(o0[i], want, atol=1e-6) and np.allclose(o1[i], want, atol=1e-6), (step, i)


def test_overlap_hooks_world2_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_overlap_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    from tests.helpers import collect_from_workers
    res = dict(collect_from_workers(q, procs, 2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for step in range(2):
        (l0, o0), (l1, o1) = res[0][step], res[1][step]
        for i in range(len(l0) - 1):
            if l0[i] is None and l1[i] is None:
                assert o0[i] is None and o1[i] is None
                continue
            want = ((0 if l0[i] is None else l0[i]) + (0 if l1[i] is None else l1[i])) / 2
            import numpy as np
            assert np.allclose(o0[i], want, atol=1e-6) and np.allclose(o1[i], want, atol=1e-6), (step, i)
        assert o0[-1] is None and o1[-1] is None
