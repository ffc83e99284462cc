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
        assert {int(s['step']) for s in opt.state.values()} == {n + 1}     # warm-up + capture pass + replays
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
