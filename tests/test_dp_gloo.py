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


def _wire_worker(rank, world, port, q):
    """One of 8 ranks: the tiny model's REAL gradients of this rank's own batch (CPU oracle), laid out as the captured step lays them out (one flat
    arena per block + a few stand-alone tensors), exchanged through GradReducer.prepare_static / reduce_static with fp32 and with bf16 buckets."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from oracle import det_weights as dw
        from oracle import vqa_oracle as vo
        from tests.conftest import CfgView, load_golden
        from vqa_model_builder_amd.dp import GradReducer
        arrays, meta = load_golden('tiny_mcan_moe4')
        d = meta['dims']
        sd = dw.make_state_dict({k: tuple(v) for k, v in meta['shapes'].items()}, meta['seed'])
        px, ids, mask, labels = dw.make_inputs(d['batch'], d['seq'], d['image'], vocab_hi=d['vocab'], num_answers=d['num_answers'], seed=500 + rank)
        _, _, _, grads = vo.forward_backward(sd, CfgView(meta), px, ids, mask, labels, vit_heads=d['vit_heads'], text_heads=d['txt_heads'])
        # every rank exchanges the SAME layout (the captured step runs the MoE densely: an expert no token of this rank chose has a zero gradient, not none)
        names = list(meta['grad_names'])
        grads = {n: (grads[n] if n in grads and grads[n] is not None else torch.zeros_like(sd[n])) for n in names}
        results = {}
        for wire in ('fp32', 'bf16'):
            # arenas: every block (first three name components) holds its gradients in ONE flat buffer, as the block runners do; biases stand alone
            blocks = {}
            for n in names:
                blocks.setdefault('.'.join(n.split('.')[:3]), []).append(n)
            params = []
            for key, members in blocks.items():
                if len(members) == 1 or key.endswith('bias'):
                    for n in members:
                        p = torch.nn.Parameter(sd[n].clone())
                        p.grad = grads[n].clone()
                        params.append((n, p))
                    continue
                flat = torch.zeros(sum((grads[n].numel() + 3) // 4 * 4 for n in members))
                off = 0
                for n in members:
                    p = torch.nn.Parameter(sd[n].clone())
                    flat[off:off + grads[n].numel()].copy_(grads[n].reshape(-1))
                    p.grad = flat[off:off + grads[n].numel()].view(grads[n].shape)
                    off += (grads[n].numel() + 3) // 4 * 4
                    params.append((n, p))
            red = GradReducer([p for _, p in params], average=True, grad_dtype=wire)
            red.prepare_static()
            assert sum(len(s['flats']) for s in red._segments.values()) >= 2
            red.reduce_static()
            results[wire] = {n: p.grad.detach().clone() for n, p in params}
        num = den = 0.0
        worst = 0.0
        for n in names:
            a, b = results['bf16'][n].double(), results['fp32'][n].double()
            num += float((a - b).norm() ** 2)
            den += float(b.norm() ** 2)
            if float(b.norm()) > 0:
                worst = max(worst, float((a - b).norm() / b.norm()))
        digest = float(sum(results['bf16'][n].double().sum() for n in names))
        local_vs_mean = float(sum((grads[n].double() - results['fp32'][n].double()).norm() ** 2 for n in names) ** 0.5 / den ** 0.5)
        q.put((rank, (num / den) ** 0.5, worst, digest, local_vs_mean))
    finally:
        dist.destroy_process_group()


def test_eight_rank_bf16_wire_sums_stay_inside_the_autocast_gradient_envelope():
    """The N > 1 default of the captured step sends bfloat16 buckets and lets RCCL sum them in bf16 (dp.py prepare_static / reduce_segment): every
    rank's gradient rounded to 8 bits once and seven bf16 additions on top.  Eight gloo ranks (what an 8-GPU node runs over RCCL), the tiny
    MoE model's real gradients of eight different batches (CPU oracle), arenas + stand-alone tensors as in the captured step: against the fp32
    exchange of the same gradients the bf16 exchange must stay well inside the error the REFERENCE's own bf16 autocast puts on these gradients
    (the fixture's ``ac_bf16`` envelope), and all ranks must end up with bit-identical sums."""
    import numpy as np
    from tests.conftest import load_golden
    arrays, meta = load_golden('tiny_mcan_moe4')
    gn = np.array([float(arrays['gnorm/' + n]) for n in meta['grad_names']])
    envelope = float(np.sqrt(((arrays['ac_bf16/gs'] * gn) ** 2).sum() / (gn ** 2).sum()))
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    world = 8
    procs = [ctx.Process(target=_wire_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    from tests.helpers import collect_from_workers
    res = sorted(collect_from_workers(q, procs, world, timeout=300))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    agg, worst, digest, spread = res[0][1], max(r[2] for r in res), res[0][3], res[0][4]
    print(f'\nDP-WIRE 8 ranks: bf16-wire vs fp32 exchange aggregate rel-L2 {agg:.3e} (worst tensor {worst:.3e}); reference bf16-autocast envelope {envelope:.3e}; '
          f'a rank\'s own gradient differs from the 8-rank mean by {spread:.2f} (the batches really differ)')
    assert all(r[3] == digest for r in res), [r[3] for r in res]              # replicas stay identical
    assert spread > 0.3                                                       # eight DIFFERENT gradients were summed
    assert agg <= 0.5 * envelope, (agg, envelope)
    assert worst <= 2e-2
