"""Pins the CPU oracle (oracle/vqa_oracle.py) against the golden vectors produced by the reference's
own modules (oracle/gen_golden.py).  fp32 CPU vs fp32 CPU: tolerance 1e-5 relative."""

import numpy as np
import pytest
import torch

from oracle import det_weights as dw
from oracle import vqa_oracle as vo
from oracle.gen_golden import sample_grad
from tests.conftest import CfgView, load_golden
from tests.helpers import fixture_inputs

TOL = 1e-5


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


def _run_case(tag):
    arrays, meta = load_golden(tag)
    d = meta['dims']
    shapes = {k: tuple(v) for k, v in meta['shapes'].items()}
    sd = dw.make_state_dict(shapes, meta['seed'])
    assert abs(dw.checksum(sd) - meta['weights_checksum']) <= 1e-6 * abs(meta['weights_checksum']), \
        'deterministic weight generator drifted from the one that made the fixture'
    px, ids, mask, labels = fixture_inputs(arrays, meta)
    cfg = CfgView(meta)
    logits, loss, pred, grads = vo.forward_backward(sd, cfg, px, ids, mask, labels,
                                                    vit_heads=d['vit_heads'], text_heads=d['txt_heads'])
    return arrays, meta, logits, loss, pred, grads


TINY = ['tiny_concat', 'tiny_xattn', 'tiny_mcan_moe4', 'tiny_xattn_moe8', 'tiny_bilinear']
FULL = ['full_cfg1_concat', 'full_cfg2_xattn', 'full_cfg3_mcan_moe4']
FULL32 = ['full32_cfg1_concat', 'full32_cfg2_xattn', 'full32_cfg3_mcan_moe4']      # BASELINE batch: 32 selected pool samples


def _check(tag, rich):
    arrays, meta, logits, loss, pred, grads = _run_case(tag)
    assert rel_err(logits.numpy(), arrays['logits']) < TOL
    assert abs(float(loss) - float(arrays['loss'])) < TOL * max(1.0, abs(float(arrays['loss'])))
    assert np.array_equal(pred.numpy(), arrays['predictions'])          # argmax ids bit-exact
    # parameters that get no gradient in the reference get none (or exact zeros) here
    for name in meta['none_grad_names']:
        assert name not in grads or float(grads[name].abs().max()) == 0.0, name
    worst = 0.0
    for name in meta['grad_names']:
        assert name in grads, f'missing gradient for {name}'
        g = grads[name]
        gn = float(g.double().norm())
        ref_n = float(arrays['gnorm/' + name])
        assert abs(gn - ref_n) <= 2e-5 * ref_n + 1e-6, (name, gn, ref_n)   # 1e-6: fp32 noise floor of
        #   gradients that are exactly zero in exact arithmetic (e.g. k_proj.bias: softmax shift invariance)
        ref_s = arrays['g/' + name]
        got_s = sample_grad(g, rich).numpy()
        scale = max(np.abs(ref_s).max(), ref_n / np.sqrt(g.numel()), 1e-3)
        e = float(np.abs(got_s - ref_s).max() / scale)
        worst = max(worst, e)
        assert e < 1e-4, (name, e)
    return worst


@pytest.mark.parametrize('tag', TINY)
def test_oracle_matches_reference_tiny(tag):
    _check(tag, True)


@pytest.mark.parametrize('tag', FULL)
def test_oracle_matches_reference_full(tag):
    _check(tag, False)


@pytest.mark.parametrize('tag', FULL32)
def test_oracle_matches_reference_full_batch32(tag):
    _check(tag, False)


GEN_TAGS = ['generative_tiny', 'generative_full', 'generative_tiny_moe_vqa', 'generative_tiny_moe_std', 'generative_full_moe_vqa', 'generative_full_moe_std']


@pytest.mark.parametrize('tag', GEN_TAGS)
def test_generative_oracle_matches_reference(tag):
    """oracle/gen_oracle.py (concatenated-sequence pre-LN fusion, optional fusion MoE over all 114 tokens -- VQAMOELayer's expert mix or
    FeedForward experts --, causal pre-LN decoder, tied 64 000-way head, label-smoothed CE with ignore_index) against what the reference's
    own GenerativeVQAModel produced: logits, loss, memory, router probabilities / expert choice, every gradient."""
    import torch
    from oracle import det_weights as dw
    from oracle import gen_oracle as go
    arrays, meta = load_golden(tag)
    d = meta['dims']
    sd0 = dw.make_state_dict({k: tuple(v) for k, v in meta['shapes'].items()}, meta['seed'])
    assert dw.checksum(sd0) == meta['weights_checksum']
    sd = go.tie(sd0)
    leaves = {}
    for k, v in sd.items():
        if k in ('decoder.embedding.weight', 'decoder.output_projection.weight'):
            continue
        leaves[k] = v.clone().requires_grad_(v.is_floating_point() and k != 'decoder.pos_encoding.pe')
    leaves['decoder.embedding.weight'] = leaves['decoder.output_projection.weight'] = leaves['answer_embedding.weight']
    px, ids, mask, dec_in, dmask, labels = go.fixture_inputs(meta)
    aux = {}
    logits, loss, memory = go.generative_forward(leaves, px, ids, mask, dec_in, dmask, labels, vit_heads=d['vit_heads'], text_heads=d['txt_heads'],
                                                 fusion_heads=d['fusion_heads'], decoder_heads=d['gen_heads'], aux_out=aux)
    loss.backward()
    if meta.get('use_moe'):
        probs = aux['router_probs'].detach().numpy()
        assert np.abs(probs - arrays['router_probs']).max() < 1e-5
        assert abs(float(aux['load_balance_loss']) - float(arrays['load_balance_loss'])) < 1e-6
        got_idx = np.sort(np.argsort(-probs, axis=-1)[..., :2], axis=-1)
        flips = (got_idx != np.sort(arrays['expert_indices'], axis=-1)).any(-1)
        assert not (flips & (arrays['router_gap'] > 1e-5)).any()          # the oracle routes every token as the reference did
    rl = lambda a, b: float(np.linalg.norm(np.asarray(a, np.float64).ravel() - np.asarray(b, np.float64).ravel()) / (np.linalg.norm(np.asarray(b, np.float64)) + 1e-30))
    if 'logits' in arrays:
        assert rl(logits.detach().numpy(), arrays['logits']) < 1e-5
    else:
        assert rl(logits.detach().flatten()[::97].numpy(), arrays['logits_sample']) < 1e-5
    assert np.array_equal(logits.detach().argmax(-1).numpy(), arrays['argmax']) or float(arrays['margin'].min()) < 1e-4
    assert abs(float(loss) - float(arrays['loss'])) < 1e-5 * max(1.0, abs(float(arrays['loss'])))
    assert rl(memory.detach().numpy(), arrays['memory']) < 1e-5
    from oracle.gen_golden import sample_grad
    n = 0
    gmax = max(float(arrays['gnorm/' + name]) for name in meta['grad_names'])
    for name in meta['grad_names']:
        key = 'answer_embedding.weight' if name in ('decoder.embedding.weight', 'decoder.output_projection.weight') else name
        g = leaves[key].grad
        assert g is not None, name
        if float(arrays['gnorm/' + name]) < 1e-6 * gmax:          # mathematically zero (key bias under softmax): rounding noise on both sides
            assert float(g.double().norm()) < 1e-5 * gmax, name
            continue
        assert abs(float(g.double().norm()) - float(arrays['gnorm/' + name])) <= 2e-5 * float(arrays['gnorm/' + name]) + 1e-9, name
        small = float(arrays['gnorm/' + name]) < 1e-4 * gmax         # fp32 summation-order noise is relative to the LARGE terms that cancel in it
        assert rl(sample_grad(g, True).numpy(), arrays['g/' + name]) < (2e-3 if small else 2e-4), name
        n += 1
    assert n > 50


@pytest.mark.parametrize('tag', ['fusion_xattn_tiny', 'fusion_xattn_full'])
def test_standalone_cross_attention_fusion_oracle_matches_reference(tag):
    """oracle/gen_oracle.py: cross_attention_fusion (bidirectional blocks with both padding masks, mean pooling over all tokens,
    Linear -> LayerNorm -> GELU -> Linear -> LayerNorm) against the reference's own module: output, input and parameter gradients."""
    import torch
    from oracle import det_weights as dw
    from oracle import gen_oracle as go
    from oracle.gen_golden import sample_grad
    arrays, meta = load_golden(tag)
    c = meta['case']
    sd = dw.make_state_dict({k: tuple(v) for k, v in meta['shapes'].items()}, meta['seed'])
    assert dw.checksum(sd) == meta['weights_checksum']
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    v, t, vmask, tmask, gy = go.fusion_fixture_inputs(meta)
    v.requires_grad_(True); t.requires_grad_(True)
    out = go.cross_attention_fusion(leaves, v, t, vmask, tmask, num_heads=c['num_attention_heads'], fusion_method=c['fusion_method'])
    (out * gy).sum().backward()
    rl = lambda a, b: float(np.linalg.norm(np.asarray(a, np.float64).ravel() - np.asarray(b, np.float64).ravel()) / (np.linalg.norm(np.asarray(b, np.float64)) + 1e-30))
    assert rl(out.detach().numpy(), arrays['out']) < 1e-5
    assert rl(v.grad.numpy(), arrays['dv']) < 1e-4 and rl(t.grad.numpy(), arrays['dt']) < 1e-4
    gmax = max(float(arrays['gnorm/' + n]) for n in meta['grad_names'])
    for n in meta['grad_names']:
        if float(arrays['gnorm/' + n]) < 1e-6 * gmax:
            continue
        assert rl(sample_grad(leaves[n].grad, True).numpy(), arrays['g/' + n]) < (2e-3 if float(arrays['gnorm/' + n]) < 1e-4 * gmax else 2e-4), n


def test_fixtures_carry_the_reference_autocast_envelopes():
    """Every model fixture stores what the reference itself does under torch.autocast (bf16 and fp16): the GPU parity gates
    are multiples of THESE numbers.  Sanity: fp16 is the tighter one, and both are finite and in a plausible band."""
    for tag in TINY + FULL + FULL32:
        arrays, meta = load_golden(tag)
        for mode in ('ac_bf16', 'ac_fp16'):
            assert arrays[mode + '/g'].shape == (len(meta['grad_names']),) and arrays[mode + '/gs'].shape == (len(meta['grad_names']),)
            assert arrays[mode + '/predictions'].shape == arrays['predictions'].shape
        b, h = float(arrays['ac_bf16/logits_rel_l2']), float(arrays['ac_fp16/logits_rel_l2'])
        assert 1e-3 < b < 3e-2 and 1e-4 < h < 4e-3 and h < b, (tag, b, h)
        if meta.get('pool'):
            assert arrays['pool_index'].shape == (meta['dims']['batch'],)
            assert float(arrays['margin'].min()) > 4.0 * float(arrays['ac_bf16/logits_max_abs']), tag    # every answer id gateable exactly


def test_parts_routers_combine_pooling():
    arrays, meta = load_golden('parts')
    seed = meta['seed']
    dm = meta['dims']
    B, S, D, E, K = dm['B'], dm['S'], dm['D'], dm['E'], dm['K']
    x = dw.normal('parts.x', (B, S, D), seed)

    def sd_for(prefix):
        return dw.make_state_dict({k: tuple(v) for k, v in meta['cases'][prefix].items()}, seed)

    sd = sd_for('noisy.')
    w, i, aux = vo.noisy_topk_router(sd, 'noisy.', x, K)
    assert np.array_equal(i.numpy(), arrays['noisy_eval/i'])
    assert rel_err(w, arrays['noisy_eval/w']) < TOL
    assert rel_err(aux['load_balance_loss'], arrays['noisy_eval/lb']) < TOL
    assert rel_err(aux['router_probs'], arrays['noisy_eval/probs']) < TOL
    noise = dw.normal('parts.noise', (B, S, E), seed)
    w, i, aux = vo.noisy_topk_router(sd, 'noisy.', x, K, noise=noise)
    assert np.array_equal(i.numpy(), arrays['noisy_train/i'])
    assert rel_err(w, arrays['noisy_train/w']) < TOL
    assert rel_err(aux['load_balance_loss'], arrays['noisy_train/lb']) < TOL

    w, i, aux = vo.topk_router(sd_for('topk.'), 'topk.', x, 3)
    assert np.array_equal(i.numpy(), arrays['topk/i']) and rel_err(w, arrays['topk/w']) < TOL
    assert rel_err(aux['load_balance_loss'], arrays['topk/lb']) < TOL
    w, i, aux = vo.soft_router(sd_for('soft.'), 'soft.', x, 0.7)
    assert np.array_equal(i.numpy(), arrays['soft/i']) and rel_err(w, arrays['soft/w']) < TOL
    assert rel_err(aux['entropy'], arrays['soft/entropy']) < TOL

    # MOELayer with feed-forward experts; then with expert 1 disabled the way the ablation harness does it
    sd = sd_for('moeff.')
    kinds = ['feedforward'] * 4
    ro = vo.topk_router(sd, 'moeff.router.', x, 2)
    y, _ = vo.moe_layer(sd, 'moeff.', x, kinds, 2, router_out=ro)
    assert rel_err(y, arrays['moeff/out']) < TOL
    w, i, aux = ro
    dis = i == 1
    w2 = w.masked_fill(dis, 0.0)
    i2 = i.masked_fill(dis, -1)
    w2 = w2 / w2.sum(dim=-1, keepdim=True).clamp(min=1e-9)
    y, _ = vo.moe_layer(sd, 'moeff.', x, kinds, 2, router_out=(w2, i2, aux))
    assert rel_err(y, arrays['moeff/out_disabled1']) < TOL

    # CrossModalAttention with query and key/value padding masks, forward + backward
    sd = {k: v.requires_grad_(True) for k, v in sd_for('cma.').items()}
    kv = dw.normal('parts.kv', (B, 7, D), seed).requires_grad_(True)
    xq = x.clone().requires_grad_(True)
    qm = torch.zeros(B, S, dtype=torch.bool)
    qm[1, 3:] = True
    km = torch.zeros(B, 7, dtype=torch.bool)
    km[2, 5:] = True
    y = vo.cross_modal_attention(sd, 'cma.', xq, kv, 4, qm, km)
    (y * dw.normal('parts.gy', tuple(y.shape), seed)).sum().backward()
    assert rel_err(y.detach(), arrays['cma/out']) < TOL
    assert rel_err(xq.grad, arrays['cma/dquery']) < 5e-5
    assert rel_err(kv.grad, arrays['cma/dkv']) < 5e-5
    for k, v in sd.items():
        assert rel_err(v.grad, arrays['cma/g/' + k[len('cma.'):]]) < 5e-5, k

    am = torch.ones(B, S, dtype=torch.int64)
    am[1, 3:] = 0
    for strat in ('cls', 'mean', 'max'):
        assert rel_err(vo.pool_text(x, am, strat), arrays['pool/' + strat]) < TOL


def test_moe_utils_helpers_match_the_reference():
    """The ten helpers of src/modeling/moe/moe_utils.py restated in vqa_model_builder_amd.modeling.moe.moe_utils against the values
    the reference's own functions returned on the same router outputs (parts.npz, utils/*)."""
    import os
    import tempfile
    import vqa_model_builder_amd.modeling.moe as moe
    arrays, meta = load_golden('parts')
    seed, dm = meta['seed'], meta['dims']
    B, S, D, E, K = dm['B'], dm['S'], dm['D'], dm['E'], dm['K']
    x = dw.normal('parts.x', (B, S, D), seed)
    sd = dw.make_state_dict({k: tuple(v) for k, v in meta['cases']['noisy.'].items()}, seed)
    w, i, aux = vo.noisy_topk_router(sd, 'noisy.', x, K)
    probs, logits = aux['router_probs'], x @ sd['noisy.gate.weight'].t()
    assert [moe.compute_expert_capacity(96, 6, 2), moe.compute_expert_capacity(15, 4, 1, 1.0), moe.compute_expert_capacity(7, 8, 2, 2.5)] == arrays['utils/capacity'].tolist()
    assert rel_err(moe.compute_load_balance_loss(probs, i, E, 0.02), arrays['utils/load_balance']) < TOL
    assert rel_err(moe.compute_router_z_loss(logits, 0.003), arrays['utils/z_loss']) < TOL
    assert rel_err(moe.compute_expert_entropy(probs), arrays['utils/entropy']) < TOL
    util = moe.get_expert_utilization(i, E)
    assert np.allclose([util[e] for e in range(E)], arrays['utils/utilization'], atol=1e-7)
    ana = moe.analyze_routing_patterns(probs, i, E)
    assert abs(ana['routing_entropy'] - float(arrays['utils/ana_entropy'])) < 1e-5 and abs(ana['max_prob_mean'] - float(arrays['utils/ana_max'])) < 1e-6
    assert abs(ana['min_prob_mean'] - float(arrays['utils/ana_min'])) < 1e-6
    assert np.array_equal(np.array(ana['expert_co_selection']), arrays['utils/ana_cosel'])
    assert [ana['expert_utilization'][e] for e in range(E)] == [util[e] for e in range(E)]
    # a token that lists one expert twice (soft routers never do, the function must still count like the reference's pair loop)
    twice = moe.analyze_routing_patterns(probs[:1, :1], torch.tensor([[[2, 2, 1]]]), E)['expert_co_selection']
    assert twice[2][2] == 2.0 and twice[2][1] == 2.0 and twice[1][2] == 2.0 and twice[1][1] == 0.0
    # an ablation-style -1 index counts for nobody
    assert moe.get_expert_utilization(torch.tensor([[[0, -1], [1, 0]]]), 3) == {0: 0.5, 1: 0.25, 2: 0.0}
    drop = moe.ExpertDropout(E, 0.4).train()
    keep = torch.from_numpy(arrays['utils/dropout_keep'])
    orig = torch.bernoulli
    torch.bernoulli = lambda t, **kw: keep.to(t.dtype)
    try:
        wd, idx = drop(w, i)
    finally:
        torch.bernoulli = orig
    assert rel_err(wd, arrays['utils/dropout_w']) < TOL and idx is i
    assert moe.ExpertDropout(E, 0.4).eval()(w, i)[0] is w
    # checkpoint helpers: round trip of a layer-shaped module (state_dict + the three attributes)
    layer = torch.nn.Linear(4, 3)
    layer.num_experts, layer.input_dim, layer.output_dim = 2, 4, 3
    with tempfile.TemporaryDirectory() as d:
        moe.save_moe_checkpoint(layer, os.path.join(d, 'm.pt'), {'epoch': 3})
        other = torch.nn.Linear(4, 3)
        assert moe.load_moe_checkpoint(other, os.path.join(d, 'm.pt')) == {'epoch': 3}
        assert torch.equal(other.weight, layer.weight)


def test_position_ids_pad_aware():
    ids = torch.tensor([[0, 5, 1, 7, 2], [0, 9, 2, 1, 1]])
    assert vo.roberta_position_ids(ids).tolist() == [[2, 3, 1, 4, 5], [2, 3, 4, 1, 1]]
