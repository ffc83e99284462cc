"""The whole training step as a replayed HIP graph (vqa_model_builder_amd/graph.py) against the same step launched eagerly:
  * without dropout the replayed losses follow the eager ones step by step (same kernels, same order; fp32 atomics only);
  * with dropout every replay draws NEW masks (device-side RNG epoch) although the captured kernel arguments are frozen,
    and backward regenerates the forward's masks (the loss still goes down);
  * the optimiser's device-side step count advances per replay (bias correction of step t, not of the captured step)."""

import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(train, lr=2e-4, seed=5):
    from oracle import det_weights as dw
    from oracle.gen_golden import TINY
    from tests.helpers import build_model
    from vqa_model_builder_amd.optim import FusedAdamW
    meta = {'dims': TINY, 'fusion_type': 'cross_attention', 'num_experts': 0}
    model = build_model(meta)
    model.load_state_dict(dw.make_state_dict(dw.shapes_of(model.state_dict()), seed))
    model = model.to('cuda:0')
    model = model.train() if train else model.eval()
    d = TINY
    px, ids, mask, labels = dw.make_inputs(d['batch'], d['seq'], d['image'], vocab_hi=d['vocab'], num_answers=d['num_answers'], seed=77)
    batch = dict(pixel_values=px.cuda(), input_ids=ids.cuda(), attention_mask=mask.cuda(), labels=labels.cuda())
    opt = FusedAdamW([p for p in model.parameters() if p.requires_grad], lr=lr, weight_decay=0.01, max_grad_norm=1.0).attach_shadows(model)
    return model, opt, batch


def _eager(model, opt, batch, n):
    out = []
    for _ in range(n):
        opt.zero_grad(set_to_none=True)
        o = model(**batch)
        o.loss.backward()
        opt.step()
        out.append(o.loss.item())
    return out


def test_weight_gradient_norm_taken_in_the_gemms_equals_the_norm_pass():
    """FusedAdamW.fuse_wgrad_norm: the weight gradients' share of the clipping norm is summed by the grouped GEMMs that store them; the step must
    see the same global norm (fp32 summation order aside) and move the parameters the same way as with the optimiser's own pass over every
    gradient -- with and without MoE experts (stand-alone parameters), and a second backward accumulated into the first (the covered ranges no
    longer ARE the gradients) must fall back to the full pass instead of clipping against a stale norm."""
    for setup in (_setup, _moe_setup):
        first = []
        for fused in (False, True):
            model, opt, batch = setup(False) if setup is _setup else setup()
            if fused:
                opt.fuse_wgrad_norm(True)
            try:
                for step in range(3):
                    opt.zero_grad(set_to_none=True)
                    model(**batch).loss.backward()
                    # the norm torch would clip against, from the very gradients this step consumes (independent of how the run got here:
                    # the tiny model's trajectory amplifies a 1e-7 difference in the clip coefficient ~1000x per step)
                    want = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters() if p.grad is not None)))
                    opt.step()
                    got = float(opt.grad_norm())
                    assert abs(got - want) <= 1e-5 * want, (fused, step, got, want)
                    read, total = opt.norm_coverage()
                    # fused: the optimiser's own pass must really have shrunk to what no GEMM wrote (embedding tables, biases, LayerNorm affine);
                    # a silent fall-back to the full pass gives the right norm too (round 3: packed in-projection gradients written by two
                    # GEMMs made the byte bookkeeping miss and every step fell back)
                    assert (read < 0.6 * total) if fused else (read == total), (fused, step, read, total)
                    if step == 0:
                        first.append((got, [p.detach().clone() for p in model.parameters()]))
            finally:
                opt.fuse_wgrad_norm(False)
        (n0, p0), (n1, p1) = first
        assert abs(n0 - n1) <= 2e-6 * n0, (n0, n1)          # same gradients in both runs at the first step: same norm, same update
        for a, b in zip(p0, p1):
            assert torch.allclose(a, b, atol=1e-6, rtol=1e-5)
    # accumulation of two backward passes: the fused bookkeeping must notice and the norm must be that of the SUMMED gradients
    model, opt, batch = _setup(False)
    ref_model, ref_opt, _ = _setup(False)
    opt.fuse_wgrad_norm(True)
    try:
        for m, o in ((model, opt), (ref_model, ref_opt)):
            o.zero_grad(set_to_none=True)
            m(**batch).loss.backward()
            m(**batch).loss.backward()
            o.step()
        assert abs(float(opt.grad_norm()) - float(ref_opt.grad_norm())) <= 2e-5 * float(ref_opt.grad_norm())
    finally:
        opt.fuse_wgrad_norm(False)


@pytest.mark.parametrize('towers', [False, True])
def test_replayed_step_follows_eager_step(towers):
    from vqa_model_builder_amd.graph import GraphedTrainStep
    from vqa_model_builder_amd.hip import blocks
    try:
        n, warm = 8, 2
        ref = _eager(*_setup(train=False), n)
        model, opt, batch = _setup(train=False)
        gs = GraphedTrainStep(model, opt, batch, warmup=warm, parallel_towers=towers)
        got = [gs(batch).item() for _ in range(n - warm)]
        assert ref[0] - ref[-1] > 0.05, ref                      # the reference run itself trains
        for a, b in zip(ref[warm:], got):
            assert abs(a - b) <= 1e-2 * max(1.0, abs(a)), (ref, got)     # fp32 atomics reorder sums; AdamW amplifies it step by step
        opt.sync_step_counts()
        assert {int(s['step']) for s in opt.state.values()} == {n}         # updates actually applied: warm-up + replays (the capture pass applies none)
        assert int(round(float(opt._hyper[0][1]))) == n                       # = the device-side step count the kernels use
    finally:
        blocks.disable_indirect_seeds()


def test_replays_draw_fresh_dropout_masks():
    from vqa_model_builder_amd.graph import GraphedTrainStep
    from vqa_model_builder_amd.hip import blocks
    try:
        model, opt, batch = _setup(train=True, lr=0.0)               # frozen weights: only the masks can change the loss
        gs = GraphedTrainStep(model, opt, batch, warmup=1)
        losses = [gs(batch).item() for _ in range(6)]
        assert len({round(l, 5) for l in losses}) == len(losses), losses
        model.eval()
        model2, opt2, _ = _setup(train=False, lr=0.0)
        gs2 = GraphedTrainStep(model2, opt2, batch, warmup=1)
        l2 = [gs2(batch).item() for _ in range(3)]
        assert max(l2) - min(l2) < 1e-6, l2                           # no dropout: replays are identical
    finally:
        blocks.disable_indirect_seeds()


def test_indirect_seed_matches_between_forward_and_backward():
    """dropout(x) forward and its backward regenerate the same mask from an INDIRECT seed; bumping the epoch changes it."""
    from vqa_model_builder_amd.hip import blocks, ops
    try:
        blocks.enable_indirect_seeds('cuda:0')
        x = torch.ones(64, 256, device='cuda', requires_grad=True)
        y = ops.dropout(x, 0.5, True)
        y.sum().backward()
        assert torch.equal((y.detach() != 0), (x.grad != 0))
        keep = (y.detach() != 0).float().mean().item()
        assert 0.4 < keep < 0.6
        seed = blocks.new_seed()
        from vqa_model_builder_amd.hip import kernels as K
        a = K.dropout_f32(torch.ones(4096, device="cuda"), K.Drop(0.5, seed, 3))
        b = K.dropout_f32(torch.ones(4096, device="cuda"), K.Drop(0.5, seed, 3))
        blocks.advance_rng_epoch()
        c = K.dropout_f32(torch.ones(4096, device="cuda"), K.Drop(0.5, seed, 3))
        a, b, c = [t[0] if isinstance(t, tuple) else t for t in (a, b, c)]
        assert torch.equal(a, b) and not torch.equal(a, c)
    finally:
        blocks.disable_indirect_seeds()


def _moe_setup(seed=7):
    from oracle import det_weights as dw
    from oracle.gen_golden import TINY
    from tests.helpers import build_model
    from vqa_model_builder_amd.optim import FusedAdamW
    meta = {'dims': TINY, 'fusion_type': 'cross_attention', 'num_experts': 4}
    model = build_model(meta)
    model.load_state_dict(dw.make_state_dict(dw.shapes_of(model.state_dict()), seed))
    model = model.to('cuda:0').eval()
    d = TINY
    px, ids, mask, labels = dw.make_inputs(d['batch'], d['seq'], d['image'], vocab_hi=d['vocab'], num_answers=d['num_answers'], seed=78)
    batch = dict(pixel_values=px.cuda(), input_ids=ids.cuda(), attention_mask=mask.cuda(), labels=labels.cuda())
    opt = FusedAdamW([p for p in model.parameters() if p.requires_grad], lr=2e-4, weight_decay=0.01, max_grad_norm=1.0).attach_shadows(model)
    return model, opt, batch


@pytest.mark.parametrize('branches', [0, 1, 2])
def test_dense_moe_dispatch_equals_sparse_dispatch(branches):
    """The capturable MoE dispatch (every expert on every token, routing weights zero where not chosen) against the sparse one:
    same logits and gradients; where the sparse dispatch skipped an expert the dense one produces exact zeros; the layer's
    device-side routed-token counts equal the bincount of the router's indices.  ``branches``: experts on side HIP streams
    (1: the specialised expert, 2: every expert its own) -- same results, the streams only reorder independent launches."""
    sparse, _, batch = _moe_setup()
    dense, _, _ = _moe_setup()
    dense.moe_layer.enable_dense_dispatch(True)
    dense.moe_layer.parallel_branches = branches
    o_s, o_d = sparse(**batch), dense(**batch)
    o_s.loss.backward(); o_d.loss.backward()
    assert torch.allclose(o_s.logits, o_d.logits, atol=2e-3, rtol=2e-3)
    idx = dense.moe_layer.aux_outputs['expert_indices'].reshape(-1) if 'expert_indices' in dense.moe_layer.aux_outputs else None
    gs, gd = dict(sparse.named_parameters()), dict(dense.named_parameters())
    for n in gs:
        if gs[n].grad is None:
            assert gd[n].grad is None or float(gd[n].grad.abs().max()) == 0.0, n
        else:
            a, b = gs[n].grad.float(), gd[n].grad.float()
            assert (a - b).norm() <= 5e-2 * a.norm() + 1e-6, n
    act = dense.moe_layer._active.cpu()
    assert float(act.sum()) == batch['labels'].shape[0] * dense.moe_layer.top_k        # one token per sample, top-k slots each
    if idx is not None:
        assert torch.equal(act, torch.bincount(idx.cpu().clamp(min=0), minlength=len(act)).float())


def test_fused_adamw_skips_parameters_of_an_expert_without_tokens():
    """`_vqa_active` (device word, the expert's routed-token count): 0 -> the parameter, its moments and its shadow are left
    alone (the reference's grad-is-None skip); > 0 -> normal update.  Also under a replayed graph."""
    from vqa_model_builder_amd.optim import FusedAdamW
    torch.manual_seed(0)
    a, b = torch.nn.Parameter(torch.randn(300, 7, device='cuda')), torch.nn.Parameter(torch.randn(129, device='cuda'))
    act = torch.zeros(2, device='cuda')
    a._vqa_active, b._vqa_active = act[0:1], act[1:2]
    opt = FusedAdamW([a, b], lr=1e-2, weight_decay=0.1, max_grad_norm=1.0)
    a0, b0 = a.detach().clone(), b.detach().clone()
    a.grad, b.grad = torch.randn_like(a), torch.randn_like(b)
    act.copy_(torch.tensor([0.0, 5.0]))
    opt.step()
    assert torch.equal(a.detach(), a0) and not torch.equal(b.detach(), b0)
    assert float(opt.state[a]['exp_avg'].abs().max()) == 0.0
    act.copy_(torch.tensor([2.0, 0.0]))
    b1 = b.detach().clone()
    opt.step()
    assert not torch.equal(a.detach(), a0) and torch.equal(b.detach(), b1)


def test_fused_adamw_counts_an_experts_steps_on_the_device_like_torch_counts_them_per_parameter():
    """torch.optim.AdamW keeps `step` per parameter: a parameter without a gradient in a step (an expert no token chose) is skipped
    AND its bias corrections do not age.  Under dense dispatch the skip is a device word (`_vqa_active`); the expert's own update
    count is one too (`_vqa_step`, advanced by vqa_opt_advance_counts), eagerly and under a replayed graph."""
    from vqa_model_builder_amd.optim import FusedAdamW
    pattern = [1, 0, 1, 1, 0, 0, 1]
    for graphed in (False, True):
        torch.manual_seed(3)
        a = torch.nn.Parameter(torch.randn(70, 9, device='cuda'))
        b = torch.nn.Parameter(torch.randn(33, device='cuda'))
        ra, rb = torch.nn.Parameter(a.detach().clone()), torch.nn.Parameter(b.detach().clone())
        act, steps = torch.ones(1, device='cuda'), torch.zeros(1, device='cuda')
        a._vqa_active, a._vqa_step, a._vqa_counts = act[0:1], steps[0:1], (act, steps, 0)
        opt = FusedAdamW([a, b], lr=1e-2, weight_decay=0.1, max_grad_norm=None)
        ref = torch.optim.AdamW([ra, rb], lr=1e-2, weight_decay=0.1)
        ga, gb = torch.randn(len(pattern), 70, 9, device='cuda'), torch.randn(len(pattern), 33, device='cuda')
        a.grad, b.grad = torch.zeros_like(a), torch.zeros_like(b)
        g = None
        for i, on in enumerate(pattern):
            act.fill_(float(on))
            a.grad.copy_(ga[i] * on); b.grad.copy_(gb[i])
            ra.grad, rb.grad = (ga[i].clone() if on else None), gb[i].clone()
            ref.step()
            if not graphed or i < 2:
                opt.step()
                if graphed and i == 1:
                    opt.make_capturable('cuda')
                    torch.cuda.synchronize()
                    g = torch.cuda.CUDAGraph()
                    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(s):
                        with torch.cuda.graph(g, stream=s):
                            opt.step()
                    torch.cuda.synchronize()
                    # the capture pass applied nothing
            else:
                g.replay()
        torch.cuda.synchronize()
        assert torch.allclose(a.detach(), ra.detach(), atol=2e-6, rtol=1e-5), (graphed, float((a - ra).abs().max()))
        assert torch.allclose(b.detach(), rb.detach(), atol=2e-6, rtol=1e-5), graphed
        assert int(steps.item()) == sum(pattern)
        opt.sync_step_counts()
        if graphed:
            assert int(opt.state[a]['step']) == sum(pattern) and int(opt.state[b]['step']) == len(pattern)


def test_capture_fork_topologies_the_step_relies_on():
    """The captured step forks ONE level from the capture stream (vision encoder; MoE specialised expert) -- siblings, also with
    lazily created streams, must capture and replay (variants A and B of tests/fork_capture_topologies.py, each in its own process).
    The nested forks that fault in this runtime (variants C / D / E: profiles/r02/nested_fork_capture.md) are a manual diagnostic of that
    script and are NOT run here: a known runtime fault is diagnosed once, not re-provoked on every suite run."""
    import os, subprocess, sys
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'fork_capture_topologies.py')
    r = subprocess.run([sys.executable, script, 'A', 'B'], capture_output=True, text=True, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l[:2] in ('A ', 'B ', 'C ', 'D ', 'E ')]
    print('\n'.join(lines))
    by = {l[0]: l for l in lines}
    assert set(by) == {'A', 'B'}, (r.stdout, r.stderr)
    assert 'rc=   0' in by['A'] and 'OK' in by['A'], by['A']
    assert 'rc=   0' in by['B'] and 'OK' in by['B'], by['B']


def test_moe_model_trains_under_a_replayed_graph():
    from vqa_model_builder_amd.graph import GraphedTrainStep
    from vqa_model_builder_amd.hip import blocks
    try:
        ref_model, ref_opt, batch = _moe_setup()
        ref = _eager(ref_model, ref_opt, batch, 6)
        model, opt, _ = _moe_setup()
        gs = GraphedTrainStep(model, opt, batch, warmup=2)
        got = [gs(batch).item() for _ in range(4)]
        assert ref[0] - ref[-1] > 0.02, ref
        # Steep tiny-model trajectory (loss 4.3 -> 0.6 in six steps, 3 tokens over 4 experts: most steps leave an expert without a
        # token).  Both runs skip such an expert AND count its updates separately (eager: torch's per-parameter `step`; captured:
        # the device-side per-expert count, vqa_opt_advance_counts) -- with one shared count the trajectories drifted 1.5 % apart by
        # the third step; now (MI355X): 1.9236 / 1.2484 / 0.8609 / 0.6088 against 1.9200 / 1.2494 / 0.8572 / 0.6064.  What remains is
        # run-to-run: fp32 atomics reorder the bias-gradient sums and this trajectory forks a few steps in, for the EAGER reference
        # as well (second measured reference: 1.9273 / 1.2567 / 0.9124 / 0.6245) -- hence 12 % on the late steps, 1 % on the first.
        assert abs(ref[2] - got[0]) <= 1e-2 * ref[2], (ref, got)
        for x, y in zip(ref[2:], got):
            assert abs(x - y) <= 0.12 * max(1.0, abs(x)), (ref, got)
        assert got[0] - got[-1] > 0.5 * (ref[2] - ref[-1]), (ref, got)        # and it trains at the same pace
    finally:
        blocks.disable_indirect_seeds()


def test_replays_follow_the_batches_they_are_given():
    """Two different batches (different images, token ids, padding, labels) fed alternately: the replayed graph copies each into
    its static inputs, so it follows the same eager sequence -- nothing of the capture batch (masks, position ids, labels) is
    baked in."""
    from oracle import det_weights as dw
    from oracle.gen_golden import TINY
    from vqa_model_builder_amd.graph import GraphedTrainStep
    from vqa_model_builder_amd.hip import blocks
    try:
        d = TINY
        def mk(seed):
            px, ids, mask, labels = dw.make_inputs(d['batch'], d['seq'], d['image'], vocab_hi=d['vocab'], num_answers=d['num_answers'], seed=seed)
            return dict(pixel_values=px.cuda(), input_ids=ids.cuda(), attention_mask=mask.cuda(), labels=labels.cuda())
        b1, b2 = mk(77), mk(91)
        assert not torch.equal(b1['attention_mask'], b2['attention_mask']) or not torch.equal(b1['input_ids'], b2['input_ids'])
        seq = [b1, b2, b1, b2, b2, b1, b2]                     # seq[2] is the warm-up step GraphedTrainStep runs on its capture batch
        model, opt, _ = _setup(train=False)
        ref = []
        for b in seq:
            opt.zero_grad(set_to_none=True)
            o = model(**b); o.loss.backward(); opt.step(); ref.append(o.loss.item())
        model, opt, _ = _setup(train=False)
        warm = 2
        pre = torch.cuda.Stream()                               # (not the legacy default stream: see GraphedTrainStep's docstring)
        pre.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(pre):
            for b in seq[:warm]:                               # same first two steps, eagerly
                opt.zero_grad(set_to_none=True)
                o = model(**b); o.loss.backward(); opt.step()
        torch.cuda.current_stream().wait_stream(pre)
        torch.cuda.synchronize()
        gs = GraphedTrainStep(model, opt, b1, warmup=1)         # one eager step on b1 (= seq[2]), then the capture
        got = [gs(b).item() for b in seq[warm + 1:]]
        for x, y in zip(ref[warm + 1:], got):
            assert abs(x - y) <= 1e-2 * max(1.0, abs(x)), (ref, got)
    finally:
        blocks.disable_indirect_seeds()


@pytest.mark.parametrize('resume_as', ['eager', 'graph'])
def test_checkpoint_resume_continues_the_run(tmp_path, resume_as):
    """the layout the reference's CheckpointManager writes and reads (checkpoint_manager.py:152-199 ``_create_checkpoint_dict``:
    ``model_state_dict`` / ``optimizer_state_dict`` / counters through ``torch.save``; ``load`` :450-459: ``torch.load`` -> ``model.load_state_dict``
    -> ``optimizer.load_state_dict``) round-trips the HIP model and FusedAdamW: a run resumed from the file -- into a model built with OTHER weights,
    so the 16-bit weight shadows must follow the loaded parameters -- continues like the uninterrupted one, eagerly and as a replayed graph, and the
    optimiser's step counts (bias correction) carry on from the checkpoint."""
    from vqa_model_builder_amd.graph import GraphedTrainStep
    from vqa_model_builder_amd.hip import blocks
    n0, n1 = 3, 4
    model, opt, batch = _setup(train=False)
    first = _eager(model, opt, batch, n0)
    path = tmp_path / 'checkpoint_epoch_1.pt'
    torch.save({'epoch': 1, 'global_step': n0, 'model_state_dict': model.state_dict(), 'optimizer_state_dict': opt.state_dict(),
                'metrics': {'loss': first[-1]}, 'best_metric': first[-1]}, path)
    cont = _eager(model, opt, batch, n1)
    assert first[0] - cont[-1] > 0.03, (first, cont)
    try:
        model2, opt2, _ = _setup(train=False, seed=6)
        before = _eager(model2, opt2, batch, 1)                     # the other weights are in use (shadows materialised) before the load
        assert abs(before[0] - first[0]) > 1e-3
        ck = torch.load(path, map_location='cuda:0')                # weights_only=True (torch >= 2.6 default): tensors and numbers only
        model2.load_state_dict(ck['model_state_dict'], strict=True)
        opt2.load_state_dict(ck['optimizer_state_dict'])
        assert {int(s['step']) for s in opt2.state.values()} == {n0}
        if resume_as == 'eager':
            got = _eager(model2, opt2, batch, n1)
        else:
            gs = GraphedTrainStep(model2, opt2, batch, warmup=1)   # the warm-up step is a real training step (its loss is not returned)
            got = [float('nan')] + [gs(batch).item() for _ in range(n1 - 1)]
        for a, b in zip(cont, got):
            if b == b:
                assert abs(a - b) <= (1e-5 if resume_as == 'eager' else 1e-2) * max(1.0, abs(a)), (cont, got)
        sd = opt2.state_dict()
        assert {int(s['step']) for s in sd['state'].values()} == {n0 + n1}
        sig = lambda m: float(sum(p.detach().double().abs().sum().item() for p in m.parameters()))
        assert abs(sig(model) - sig(model2)) <= (1e-7 if resume_as == 'eager' else 1e-4) * sig(model)
    finally:
        blocks.disable_indirect_seeds()
