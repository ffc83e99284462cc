"""Teacher-forced block parity on a real MI355X: every block of the path (one CLIP layer, one RoBERTa layer, the
CrossModalAttention layer, each fusion branch, the MoE layer with its experts, answer head + loss) gets the SAME exact
fp32 input and upstream gradient as the CPU oracle, so the comparison is not polluted by the chaotic amplification a
26-layer bf16 network applies to upstream rounding noise (see tests/test_parity_gpu.py).  Tolerance here is what bf16
GEMM operands give across ONE block: outputs <= 1.5e-2, gradients <= 4e-2 relative L2 (measured: 4e-3 .. 1.2e-2)."""

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import det_weights as dw  # noqa: E402
from oracle import vqa_oracle as vo  # noqa: E402
from oracle.gen_golden import FULL, TINY  # noqa: E402

OUT_TOL, GRAD_TOL = 1.5e-2, 4e-2
DEV = 'cuda'


def rl(a, b):
    a, b = a.detach().float().cpu().double(), b.detach().double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def load_det(module, seed, prefix=''):
    sd = dw.make_state_dict({prefix + k: tuple(v.shape) for k, v in module.state_dict().items()}, seed)
    module.load_state_dict({k[len(prefix):]: v for k, v in sd.items()})
    return sd


def leaves_of(sd):
    return {k: v.clone().requires_grad_(v.is_floating_point() and v.dim() > 0) for k, v in sd.items()}


def check_grads(module, leaves, prefix='', tol=GRAD_TOL, skip=()):
    gmax = max(float(v.grad.norm()) for v in leaves.values() if v.grad is not None)
    worst = (0.0, '')
    n = 0
    for name, p in module.named_parameters():
        ref = leaves[prefix + name].grad
        if any(s in name for s in skip):
            continue
        if ref is None or float(ref.norm()) < 1e-4 * gmax:
            assert p.grad is None or float(p.grad.norm()) <= 1e-2 * gmax, name
            continue
        assert p.grad is not None, f'missing gradient: {name}'
        e = rl(p.grad, ref)
        worst = max(worst, (e, name))
        n += 1
        assert e <= tol, (name, e)
    assert n > 0
    return worst


@pytest.mark.parametrize('dims,layers,B', [(TINY, 1, 3), (TINY, 2, 3), (FULL, 1, 4)])
def test_clip_layers(dims, layers, B):
    from vqa_model_builder_amd.modeling.meta_arch.backbones import ClipVisionBackbone
    d = dims
    m = ClipVisionBackbone(hidden_size=d['D'], intermediate_size=d['vit_inter'], num_hidden_layers=layers,
                           num_attention_heads=d['vit_heads'], image_size=d['image'], patch_size=d['patch'])
    sd = load_det(m, 3)
    m = m.to(DEV)
    px = dw.normal('px', (B, 3, d['image'], d['image']), 3)
    lv = leaves_of(sd)
    yo = vo.clip_vision_forward(lv, '', px, d['vit_heads'])
    gy = dw.normal('gy', tuple(yo.shape), 3)
    (yo * gy).sum().backward()
    yg = m(px.to(DEV)).last_hidden_state
    (yg * gy.to(DEV)).sum().backward()
    assert rl(yg, yo) <= OUT_TOL
    print('clip', layers, 'out', rl(yg, yo), 'worst grad', check_grads(m, lv, skip=('post_layernorm',)))


@pytest.mark.parametrize('dims,layers,B', [(TINY, 1, 3), (TINY, 2, 3), (FULL, 1, 4)])
def test_roberta_layers(dims, layers, B):
    from vqa_model_builder_amd.modeling.meta_arch.backbones import RobertaBackbone
    d = dims
    m = RobertaBackbone(vocab_size=d['vocab'], hidden_size=d['D'], num_hidden_layers=layers, num_attention_heads=d['txt_heads'],
                        intermediate_size=d['txt_inter'], max_position_embeddings=d['max_pos']).eval()
    sd = load_det(m, 4)
    m = m.to(DEV)
    _, ids, mask, _ = dw.make_inputs(B, d['seq'], 32, vocab_hi=min(30000, d['vocab']), seed=4)
    lv = leaves_of(sd)
    yo = vo.roberta_forward(lv, '', ids, mask, d['txt_heads'])
    gy = dw.normal('gy', tuple(yo.shape), 4)
    (yo * gy).sum().backward()
    yg = m(ids.to(DEV), mask.to(DEV)).last_hidden_state
    (yg * gy.to(DEV)).sum().backward()
    assert rl(yg, yo) <= OUT_TOL
    print('roberta', layers, 'out', rl(yg, yo), 'worst grad', check_grads(m, lv, skip=('pooler',)))


@pytest.mark.parametrize('D,H,B,Sq,Skv', [(64, 4, 3, 8, 10), (768, 8, 4, 64, 50)])
def test_cross_modal_attention_layer(D, H, B, Sq, Skv):
    from vqa_model_builder_amd.modeling.meta_arch import CrossModalAttention
    m = CrossModalAttention(D, H, 0.1).eval()
    sd = load_det(m, 5, 'c.')
    m = m.to(DEV)
    q, kv = dw.normal('q', (B, Sq, D), 5), dw.normal('kv', (B, Skv, D), 5)
    qm = torch.zeros(B, Sq, dtype=torch.bool)
    qm[1, Sq // 2:] = True
    km = torch.zeros(B, Skv, dtype=torch.bool)
    km[2, Skv - 3:] = True
    lv = leaves_of(sd)
    qo, kvo = q.clone().requires_grad_(True), kv.clone().requires_grad_(True)
    yo = vo.cross_modal_attention(lv, 'c.', qo, kvo, H, qm, km)
    gy = dw.normal('gy', tuple(yo.shape), 5)
    (yo * gy).sum().backward()
    qg, kvg = q.to(DEV).requires_grad_(True), kv.to(DEV).requires_grad_(True)
    yg = m(qg, kvg, qm.to(DEV), km.to(DEV))
    (yg * gy.to(DEV)).sum().backward()
    assert rl(yg, yo) <= OUT_TOL
    assert rl(qg.grad, qo.grad) <= GRAD_TOL and rl(kvg.grad, kvo.grad) <= GRAD_TOL
    print('cma', D, 'out', rl(yg, yo), 'dq', rl(qg.grad, qo.grad), 'dkv', rl(kvg.grad, kvo.grad), 'worst grad', check_grads(m, lv, 'c.'))


@pytest.mark.parametrize('D,H,B,Sq,Skv', [(64, 4, 3, 8, 10), (768, 8, 4, 64, 50)])
def test_cross_modal_attention_first_token_only(D, H, B, Sq, Skv):
    """The last fusion layer's short cut (only output row 0 is read): against the ORACLE's full block, row 0 of the output and
    every gradient when the upstream gradient lives on row 0 alone -- dead rows contribute exactly nothing in the reference."""
    from vqa_model_builder_amd.modeling.meta_arch import CrossModalAttention
    m = CrossModalAttention(D, H, 0.1).eval()
    sd = load_det(m, 6, 'c.')
    m = m.to(DEV)
    q, kv = dw.normal('q', (B, Sq, D), 6), dw.normal('kv', (B, Skv, D), 6)
    qm = torch.zeros(B, Sq, dtype=torch.bool)
    qm[1, Sq // 2:] = True
    km = torch.zeros(B, Skv, dtype=torch.bool)
    km[2, Skv - 3:] = True
    lv = leaves_of(sd)
    qo, kvo = q.clone().requires_grad_(True), kv.clone().requires_grad_(True)
    yo = vo.cross_modal_attention(lv, 'c.', qo, kvo, H, qm, km)[:, :1]
    gy = dw.normal('gy', tuple(yo.shape), 6)
    (yo * gy).sum().backward()
    qg, kvg = q.to(DEV).requires_grad_(True), kv.to(DEV).requires_grad_(True)
    yg = m(qg, kvg, qm.to(DEV), km.to(DEV), first_token_only=True)
    assert tuple(yg.shape) == (B, 1, D)
    (yg * gy.to(DEV)).sum().backward()
    assert rl(yg, yo) <= OUT_TOL
    assert rl(qg.grad, qo.grad) <= GRAD_TOL and rl(kvg.grad, kvo.grad) <= GRAD_TOL
    print('cma first-token', D, 'out', rl(yg, yo), 'dq', rl(qg.grad, qo.grad), 'dkv', rl(kvg.grad, kvo.grad), 'worst grad', check_grads(m, lv, 'c.'))


def test_frozen_block_skips_weight_gradient_gemms_and_keeps_dx():
    """Reference training strategies freeze whole modules per epoch (training_utils.py:401-455).  A frozen CrossModalAttention whose
    inputs still need gradients (trainable encoders below a frozen fusion): the block issues NO weight-gradient GEMM -- counted
    through the library's launch recorder -- its parameters get no gradient, and dquery / dkey_value are bit-identical to the
    unfrozen block's.  A frozen encoder under a frozen input needs no backward at all (autograd never calls the block)."""
    import ctypes as C
    from vqa_model_builder_amd.hip import lib
    from vqa_model_builder_amd.modeling.meta_arch import CrossModalAttention
    L = lib.load()
    D, H, B, Sq, Skv = 768, 8, 4, 64, 50
    m = CrossModalAttention(D, H, 0.1).eval()
    load_det(m, 5, 'c.')
    m = m.to(DEV)
    q, kv, gy = dw.normal('q', (B, Sq, D), 5).to(DEV), dw.normal('kv', (B, Skv, D), 5).to(DEV), dw.normal('gy', (B, Sq, D), 5).to(DEV)

    def run(frozen):
        for p in m.parameters():
            p.requires_grad_(not frozen)
            p.grad = None
        qg, kvg = q.clone().requires_grad_(True), kv.clone().requires_grad_(True)
        y = m(qg, kvg)
        torch.cuda.synchronize()
        L.vqa_gemm_profile(1, 0)
        (y * gy).sum().backward()
        torch.cuda.synchronize()
        flop, ms, n = (C.c_double * 1)(), (C.c_double * 1)(), (C.c_int * 1)()
        L.vqa_gemm_profile_collect(1, flop, ms, n)
        L.vqa_gemm_profile(0, 0)
        return qg.grad.clone(), kvg.grad.clone(), flop[0], n[0], [p.grad is not None for p in m.parameters()]

    dq1, dkv1, flop1, n1, has1 = run(False)
    dq0, dkv0, flop0, n0, has0 = run(True)
    assert all(has1) and not any(has0)
    assert torch.equal(dq0, dq1) and torch.equal(dkv0, dkv1)
    assert n0 == 7 and n1 - n0 in (1, 2), (n0, n1)             # 7 dX GEMMs; the 7 weight gradients are ONE grouped call (two launches when some
                                                               # outputs qualify for the 256 x 256 tiles and some -- the 200-token vision side -- do not)
    assert abs(flop0 / flop1 - 0.5) < 0.02, (flop0, flop1)    # dW = dX in FLOPs for every Linear of the block


@pytest.mark.parametrize('fusion_type', ['cross_attention', 'concat', 'mcan'])
def test_fusion_branches(fusion_type):
    from vqa_model_builder_amd.modeling.meta_arch import FusionConfig, MultimodalFusion
    D, H, B = 64, 4, 3
    m = MultimodalFusion(FusionConfig(fusion_type=fusion_type, hidden_dim=D, output_dim=D, num_heads=H, num_layers=2)).eval()
    sd = load_det(m, 6, 'fusion.')
    m = m.to(DEV)
    v, t = dw.normal('v', (B, 10, D), 6), dw.normal('t', (B, 8, D), 6)
    tm = torch.zeros(B, 8, dtype=torch.bool)
    tm[1, 5:] = True
    lv = leaves_of(sd)
    vo_, to_ = v.clone().requires_grad_(True), t.clone().requires_grad_(True)
    yo = vo.multimodal_fusion(lv, 'fusion.', fusion_type, H, vo_, to_, text_mask=tm)
    gy = dw.normal('gy', tuple(yo.shape), 6)
    (yo * gy).sum().backward()
    vg, tg = v.to(DEV).requires_grad_(True), t.to(DEV).requires_grad_(True)
    yg = m(vg, tg, text_mask=tm.to(DEV))
    (yg * gy.to(DEV)).sum().backward()
    assert rl(yg, yo) <= OUT_TOL
    assert rl(vg.grad, vo_.grad) <= GRAD_TOL and rl(tg.grad, to_.grad) <= GRAD_TOL
    check_grads(m, lv, 'fusion.')


@pytest.mark.parametrize('D,Hm,B,E', [(64, 128, 3, 4), (64, 128, 16, 4), (768, 2048, 8, 4)])
def test_moe_layer(D, Hm, B, E):
    """VQAMOELayer: fp32 noisy top-k router (eval), Vision/Text/Multimodal/Segmentation experts, dispatch + combine + LN."""
    from vqa_model_builder_amd.modeling.moe import VQAMOELayer
    nv, nt, nm, ns = vo.expert_split(E)
    moe = VQAMOELayer(input_dim=D, hidden_dim=Hm, output_dim=D, num_vision_experts=nv, num_text_experts=nt,
                      num_multimodal_experts=nm, num_specialized_experts=ns, top_k=2, dropout=0.1).eval()
    sd = load_det(moe, 8, 'moe_layer.')
    moe = moe.to(DEV)
    x = dw.normal('x', (B, 1, D), 8)
    lv = leaves_of(sd)
    xo = x.clone().requires_grad_(True)
    kinds = vo.vqa_moe_expert_kinds(nv, nt, nm, ns)
    mo, aux_o = vo.moe_layer(lv, 'moe_layer.', xo, kinds, 2)
    gy = dw.normal('gy', tuple(mo.shape), 8)
    (mo * gy).sum().backward()
    xg = x.to(DEV).requires_grad_(True)
    mg = moe(xg)
    (mg * gy.to(DEV)).sum().backward()
    assert rl(mg, mo) <= OUT_TOL
    assert rl(moe.aux_outputs['router_probs'].reshape(B, -1), aux_o['router_probs'].reshape(B, -1)) < 1e-4     # fp32 router
    assert abs(float(moe.aux_outputs['load_balance_loss']) - float(aux_o['load_balance_loss'])) < 1e-5
    assert rl(xg.grad, xo.grad) <= GRAD_TOL
    print('moe', D, B, 'out', rl(mg, mo), 'dx', rl(xg.grad, xo.grad), 'worst', check_grads(moe, lv, 'moe_layer.'))


def _expert_classes():
    from vqa_model_builder_amd.modeling.moe import experts as E
    return E, [E.VisionExpert, E.TextExpert, E.MultimodalExpert, E.SegmentationExpert]


@pytest.mark.parametrize('D,Hm,B', [(64, 128, 5), (64, 128, 16), (768, 2048, 32)])
def test_expert_runners_match_the_op_by_op_chains(D, Hm, B):
    """Each expert at one token per row as ONE hand-scheduled autograd node (hip/expert_blocks.py: fused epilogues, attention
    over a single key reduced to out_proj(V), centre-tap convolutions, no cast / fill launches) against the op-by-op chain of
    hip/ops.py it replaces -- itself pinned to the oracle by test_moe_layer: output, input gradient and every parameter
    gradient (q / k rows of the one-key attention in-projections: exact zeros on both paths)."""
    E, classes = _expert_classes()
    for ci, cls in enumerate(classes):
        ex = cls(input_dim=D, hidden_dim=Hm, output_dim=D, expert_id=0, dropout=0.1).eval()
        load_det(ex, 20 + ci, 'e.')
        ex = ex.to(DEV)
        x = dw.normal('x', (B, 1, D), 20 + ci).to(DEV)
        gy = dw.normal('gy', (B, 1, D), 21 + ci).to(DEV)
        res = {}
        for runners in (False, True):
            E.EXPERT_RUNNERS = runners
            try:
                for p in ex.parameters():
                    p.grad = None
                xg = x.clone().requires_grad_(True)
                y = ex(xg)
                (y * gy).sum().backward()
                res[runners] = (y.detach(), xg.grad, {n: (None if p.grad is None else p.grad.clone()) for n, p in ex.named_parameters()})
            finally:
                E.EXPERT_RUNNERS = True
        (y0, dx0, g0), (y1, dx1, g1) = res[False], res[True]
        assert rl(y1, y0.cpu()) <= 2e-3, (cls.__name__, rl(y1, y0.cpu()))
        assert rl(dx1, dx0.cpu()) <= 1e-2, (cls.__name__, rl(dx1, dx0.cpu()))
        gmax = max(float(g.norm()) for g in g0.values() if g is not None)
        worst = (0.0, '')
        for n in g0:
            if g0[n] is None:
                assert g1[n] is None or float(g1[n].abs().max()) == 0.0, (cls.__name__, n)
                continue
            assert g1[n] is not None, (cls.__name__, n)
            if float(g0[n].norm()) < 1e-4 * gmax:
                assert float(g1[n].norm()) <= 1e-2 * gmax, (cls.__name__, n)
                continue
            e = rl(g1[n], g0[n].cpu())
            worst = max(worst, (e, n))
            assert e <= 1.5e-2, (cls.__name__, n, e)
        for n, g in g1.items():                    # one-key attention: the q / k thirds of in_proj get exact zeros
            if n.endswith('in_proj_weight') and 'self_attn' not in n and g is not None and 'cross_attention' not in n:
                Hh = g.shape[1]
                assert float(g[:2 * Hh].abs().max()) == 0.0, (cls.__name__, n)
        print('expert runner', cls.__name__, D, B, 'out', rl(y1, y0.cpu()), 'dx', rl(dx1, dx0.cpu()), 'worst', worst)


@pytest.mark.parametrize('D,Hm,B', [(64, 128, 16), (768, 2048, 32)])
def test_expert_runner_training_dropout_is_one_function_forward_and_backward(D, Hm, B):
    """Training mode (dropout 0.1 on attention probabilities, FFN activations and sub-layer outputs): backward must regenerate the
    forward's masks.  With the seed pinned the expert is a fixed function, so its analytic directional derivative (input and a
    weight) must match a central difference of the scalar <out, g> -- a wrong mask anywhere shows as a 10 % error; the eval /
    train outputs differ (masks are applied) and two calls with one seed agree bit for bit."""
    from vqa_model_builder_amd.hip import blocks as hb
    hb.disable_indirect_seeds()                     # seeds come from torch's CPU generator: torch.manual_seed pins the masks
    E, classes = _expert_classes()
    for ci, cls in enumerate(classes):
        ex = cls(input_dim=D, hidden_dim=Hm, output_dim=D, expert_id=0, dropout=0.1)
        load_det(ex, 30 + ci, 'e.')
        ex = ex.to(DEV).train()
        x = dw.normal('x', (B, 1, D), 30 + ci).to(DEV)
        gy = dw.normal('gy', (B, 1, D), 31 + ci).to(DEV)

        def f(xx):
            torch.manual_seed(1234)
            return ex(xx)
        xg = x.clone().requires_grad_(True)
        y = f(xg)
        (y * gy).sum().backward()
        assert torch.equal(f(x).detach(), y.detach())
        assert not torch.equal(ex.eval()(x).detach(), y.detach())
        ex.train()
        # directions ALONG the analytic gradient (a random direction makes the derivative a random-sign sum in which the bf16
        # rounding of the perturbed operand does not average out against the signal)
        eps = 0.05
        d = xg.grad / xg.grad.norm() * x.norm()
        num = float(((f(x + eps * d) - f(x - eps * d)) * gy).sum().double()) / (2 * eps)
        ana = float((xg.grad * d).sum().double())
        assert abs(num - ana) <= 0.05 * abs(ana), (cls.__name__, num, ana)
        for pname in ('output_proj.weight', 'input_proj.weight'):
            p = dict(ex.named_parameters())[pname]
            dp = p.grad / p.grad.norm() * p.detach().norm()
            ana = float((p.grad * dp).sum().double())
            ew = 0.01                      # every element moves coherently along the gradient: a small step keeps the LayerNorm behind it linear
            with torch.no_grad():
                p.add_(ew * dp)
            yp = f(x).detach()
            with torch.no_grad():
                p.sub_(2 * ew * dp)
            ym = f(x).detach()
            with torch.no_grad():
                p.add_(ew * dp)
            num = float(((yp - ym) * gy).sum().double()) / (2 * ew)
            assert abs(num - ana) <= 0.05 * abs(ana), (cls.__name__, pname, num, ana)
        print('expert runner dropout', cls.__name__, D, B, 'ok')


@pytest.mark.parametrize('D,Hm,B', [(64, 128, 3), (768, 2048, 4)])
def test_object_detection_expert(D, Hm, B):
    """ObjectDetectionExpert (reference specialized_experts.py:176-308; reachable from 8 experts): 100 learned queries through a
    3-layer decoder, then the token attends over them.  At full size its attention backward (Sq = Skv = 100, Dh = 256) does not
    fit the LDS in one piece: the generic kernel runs it as two launches through a workspace (include/vqa_hip.h)."""
    from vqa_model_builder_amd.modeling.moe.experts import ObjectDetectionExpert
    ex = ObjectDetectionExpert(input_dim=D, hidden_dim=Hm, output_dim=D, expert_id=0, dropout=0.1).eval()
    sd = load_det(ex, 11, 'e.')
    ex = ex.to(DEV)
    x = dw.normal('x', (B, 1, D), 11)
    lv = leaves_of(sd)
    xo = x.clone().requires_grad_(True)
    yo = vo.object_detection_expert(lv, 'e.', xo)
    gy = dw.normal('gy', tuple(yo.shape), 11)
    (yo * gy).sum().backward()
    xg = x.to(DEV).requires_grad_(True)
    yg = ex(xg)
    (yg * gy.to(DEV)).sum().backward()
    assert rl(yg, yo) <= OUT_TOL
    assert rl(xg.grad, xo.grad) <= 2 * GRAD_TOL
    print('detection expert', D, B, 'out', rl(yg, yo), 'dx', rl(xg.grad, xo.grad), 'worst', check_grads(ex, lv, 'e.', tol=2 * GRAD_TOL))


@pytest.mark.parametrize('D,hidden,C,B', [(64, [48, 40], 37, 3), (768, [768, 512], 3000, 32)])
def test_answer_head_and_loss(D, hidden, C, B):
    from vqa_model_builder_amd.hip import ops
    from vqa_model_builder_amd.modeling.meta_arch import AnswerHead, AnswerHeadConfig
    head = AnswerHead(AnswerHeadConfig(num_answers=C, hidden_dims=hidden), D).eval()
    sd = load_det(head, 9, 'h.')                 # ordinary 1/sqrt(fan_in) scale here: logits O(1)
    head = head.to(DEV)
    x = dw.normal('x', (B, D), 9)
    labels = dw.randint('labels', (B,), 0, C, 9)
    lv = leaves_of(sd)
    xo = x.clone().requires_grad_(True)
    lo = vo.answer_head(lv, 'h.', xo)
    loss_o = F.cross_entropy(lo, labels)
    loss_o.backward()
    xg = x.to(DEV).requires_grad_(True)
    lg = head(xg)
    loss, pred = ops.cross_entropy_argmax(lg, labels.to(DEV))
    loss.backward()
    assert rl(lg, lo) <= OUT_TOL and abs(float(loss) - float(loss_o)) < 1e-2
    top2 = lo.detach().topk(2, -1).values
    safe = (top2[:, 0] - top2[:, 1]) > 4 * float((lg.detach().cpu() - lo.detach()).abs().max())
    assert torch.equal(pred.cpu()[safe], lo.detach().argmax(-1)[safe])
    assert rl(xg.grad, xo.grad) <= 2 * GRAD_TOL          # two ReLU layers: a unit that flips sign re-draws its whole row
    check_grads(head, lv, 'h.', tol=2 * GRAD_TOL)


@pytest.mark.parametrize('prefix', ['sparse_ff.', 'sparse_glu.', 'hier.'])
def test_sparse_and_hierarchical_moe_match_the_reference(prefix):
    """SparseMOELayer (capacity cut active: 0.6 x 30 x 2 / 4 = 9 tokens per expert against ~15 routed), the same layer over GLU experts
    and HierarchicalMOE (4 groups x 2 experts: vision / text / multimodal / feed-forward at S = 5) against outputs of the reference
    itself (tests/golden/moe_variants.npz, oracle/gen_golden.py --only moe_variants): output, input gradient, load-balance loss, every
    parameter gradient; parameters the reference leaves without a gradient (unrouted experts) get none here."""
    from tests.conftest import load_golden
    from vqa_model_builder_amd.modeling.moe import HierarchicalMOE, SparseMOELayer
    arrays, meta = load_golden('moe_variants')
    seed, case = meta['seed'], meta['cases'][prefix]
    D, Hm = 64, 128
    if prefix == 'hier.':
        layer = HierarchicalMOE(input_dim=D, hidden_dim=Hm, output_dim=D, num_expert_groups=4, experts_per_group=2, top_k_groups=2, top_k_experts=1,
                                dropout=0.1)
    else:
        layer = SparseMOELayer(input_dim=D, hidden_dim=Hm, output_dim=D, num_experts=4, top_k=2, capacity_factor=0.6 if prefix == 'sparse_ff.' else 1.25,
                               dropout=0.1, expert_type='feedforward' if prefix == 'sparse_ff.' else 'glu')
    shapes = {k: tuple(v) for k, v in case['shapes'].items()}
    assert {prefix + k for k in layer.state_dict()} == set(shapes)
    sd = dw.make_state_dict(shapes, seed)
    layer.load_state_dict({k[len(prefix):]: v for k, v in sd.items()})
    layer = layer.to(DEV).eval()
    if prefix == 'sparse_ff.':
        assert layer._compute_capacity(30) == int(arrays[prefix + 'capacity']) == 9
    x = dw.normal(prefix + 'x', tuple(case['x']), seed).to(DEV).requires_grad_(True)
    gy = dw.normal(prefix + 'gy', tuple(case['x']), seed + 1).to(DEV)
    y = layer(x)
    (y * gy).sum().backward()
    t = lambda k: torch.from_numpy(np.asarray(arrays[prefix + k]))
    assert rl(y, t('out')) <= OUT_TOL, rl(y, t('out'))
    assert rl(x.grad, t('dx')) <= 2 * GRAD_TOL, rl(x.grad, t('dx'))
    assert abs(float(layer.get_aux_loss()) - float(arrays[prefix + 'aux'])) <= 1e-5
    refs = {k[len(prefix) + 2:]: torch.from_numpy(np.asarray(v)) for k, v in arrays.items() if k.startswith(prefix + 'g/')}
    gmax = max(float(g.norm()) for g in refs.values())
    worst = (0.0, '')
    for n, p in layer.named_parameters():
        ref = refs.get(n)
        if ref is None or float(ref.norm()) < 1e-4 * gmax:
            assert p.grad is None or float(p.grad.norm()) <= 1e-2 * gmax, n
            continue
        assert p.grad is not None, n
        e = rl(p.grad, ref)
        worst = max(worst, (e, n))
        assert e <= 2 * GRAD_TOL, (n, e)
    print('moe variant', prefix, 'out', rl(y, t('out')), 'dx', rl(x.grad, t('dx')), 'worst grad', worst)


class _TailHost(torch.nn.Module):
    """The three attributes vqa_model._Tail reads off the model (fusion.output_proj / layer_norm, dropout, answer_head)."""

    def __init__(self, D, hidden, C, layer_norm=True):
        super().__init__()
        from vqa_model_builder_amd.modeling.meta_arch import AnswerHead, AnswerHeadConfig, FusionConfig, MultimodalFusion
        self.fusion = MultimodalFusion(FusionConfig(fusion_type='cross_attention', hidden_dim=D, output_dim=D, num_heads=8, num_layers=1,
                                                    use_layer_norm=layer_norm))
        self.fusion.fusion_layers = torch.nn.ModuleList()          # only the projection + norm are under test
        self.answer_head = AnswerHead(AnswerHeadConfig(num_answers=C, hidden_dims=hidden, dropout=0.3), D)
        self.dropout = torch.nn.Dropout(0.1)

    def chain(self, x, project):
        from vqa_model_builder_amd.hip import ops
        if project:
            x = ops.linear(x, self.fusion.output_proj.weight, self.fusion.output_proj.bias)
            if self.fusion.layer_norm is not None:
                x = ops.layer_norm(x, self.fusion.layer_norm.weight, self.fusion.layer_norm.bias, self.fusion.layer_norm.eps)
        return self.answer_head(ops.dropout(x, self.dropout.p, self.training))


@pytest.mark.parametrize('project,layer_norm', [(True, True), (True, False), (False, True)])
def test_tail_runner_matches_the_op_chain(project, layer_norm):
    """Eval mode: one node (hip.blocks.TailRunner) against projection -> norm -> dropout -> classifier issued op by op -- the same
    kernels with the same rounding points, so logits agree to bf16 resolution and every gradient to GEMM-order noise."""
    from vqa_model_builder_amd.modeling.meta_arch import vqa_model as vm
    D, hidden, C, B = 768, [768, 512], 3000, 32
    host = _TailHost(D, hidden, C, layer_norm)
    load_det(host, 41)
    host = host.to(DEV).eval()
    x = dw.normal('x', (B, D), 41).to(DEV)
    gy = dw.normal('gy', (B, C), 42).to(DEV)
    tail = vm._Tail(host, project, True)
    assert tail.covers() and tail.current()
    x0 = x.clone().requires_grad_(True)
    y0 = host.chain(x0, project)
    (y0 * gy).sum().backward()
    g0 = {n: p.grad.clone() for n, p in host.named_parameters() if p.grad is not None}
    host.zero_grad()
    x1 = x.clone().requires_grad_(True)
    y1 = tail(x1)
    (y1 * gy).sum().backward()
    assert rl(y1, y0.cpu()) <= 2e-3
    assert rl(x1.grad, x0.grad.cpu()) <= 1e-2
    names = set()
    for n, p in host.named_parameters():
        if n in g0:
            assert p.grad is not None and rl(p.grad, g0[n].cpu()) <= 1e-2, n
            names.add(n)
    assert ('fusion.output_proj.weight' in names) == project and 'answer_head.classifier.6.bias' in names
    # a re-assigned parameter retires the tail
    host.answer_head.classifier[0].weight = torch.nn.Parameter(host.answer_head.classifier[0].weight.detach().clone())
    assert not tail.current()


def test_tail_runner_training_dropout_is_one_function_forward_and_backward():
    """Training mode: the input dropout (fused into the LayerNorm forward, replayed by its backward on load) and the classifier's
    dropouts (GEMM epilogues) must be regenerated bit for bit by backward -- central differences along the analytic gradient."""
    from vqa_model_builder_amd.hip import blocks as hb
    from vqa_model_builder_amd.modeling.meta_arch import vqa_model as vm
    hb.disable_indirect_seeds()
    D, hidden, C, B = 768, [768, 512], 3000, 32
    for project in (True, False):
        host = _TailHost(D, hidden, C)
        load_det(host, 43)
        host = host.to(DEV).train()
        tail = vm._Tail(host, project, True)
        x = dw.normal('x', (B, D), 43).to(DEV)
        gy = dw.normal('gy', (B, C), 44).to(DEV)

        def f(xx):
            torch.manual_seed(4321)
            return tail(xx)
        xg = x.clone().requires_grad_(True)
        y = f(xg)
        (y * gy).sum().backward()
        assert torch.equal(f(x).detach(), y.detach())
        host.eval()
        assert not torch.equal(tail(x).detach(), y.detach())
        host.train()
        eps = 0.02                      # ReLU: keep the units that change sign between the two probes few
        d = xg.grad / xg.grad.norm() * x.norm()
        num = float(((f(x + eps * d) - f(x - eps * d)) * gy).sum().double()) / (2 * eps)
        ana = float((xg.grad * d).sum().double())
        assert abs(num - ana) <= 0.05 * abs(ana), (project, num, ana)
        pnames = ['answer_head.classifier.0.weight', 'answer_head.classifier.6.weight'] + (['fusion.output_proj.weight'] if project else [])
        for pname in pnames:
            p = dict(host.named_parameters())[pname]
            dp = p.grad / p.grad.norm() * p.detach().norm()
            ana = float((p.grad * dp).sum().double())
            errs = []
            for ew in (0.01, 0.0025):       # ReLU units that change sign between the probes bend the secant: the error must SHRINK with the step
                with torch.no_grad():
                    p.add_(ew * dp)
                yp = f(x).detach()
                with torch.no_grad():
                    p.sub_(2 * ew * dp)
                ym = f(x).detach()
                with torch.no_grad():
                    p.add_(ew * dp)
                num = float(((yp - ym) * gy).sum().double()) / (2 * ew)
                errs.append(abs(num - ana) / abs(ana))
            print('tail dropout', project, pname, errs)
            assert errs[1] <= 0.05 and errs[1] <= max(0.6 * errs[0], 0.02), (project, pname, errs)


def test_router_variants_and_ablation_contract():
    """The ablation harness swaps / monkey-patches ``moe.router`` (ablation_trainer.py:172-224): topk with K in {1,2,4},
    soft (K = E), noise-injected noisy_topk in train mode, and a patched router that disables an expert with index -1."""
    from vqa_model_builder_amd.modeling.moe import MOELayer, create_router
    from tests.conftest import load_golden
    arrays, meta = load_golden('parts')
    seed, dm = meta['seed'], meta['dims']
    B, S, D, E, Kk = dm['B'], dm['S'], dm['D'], dm['E'], dm['K']
    x = dw.normal('parts.x', (B, S, D), seed).to(DEV)

    def load(mod, prefix):
        sd = dw.make_state_dict({k: tuple(v) for k, v in meta['cases'][prefix].items()}, seed)
        mod.load_state_dict({k[len(prefix):]: v for k, v in sd.items()})
        return mod.to(DEV)

    r = load(create_router('noisy_topk', D, E, top_k=Kk), 'noisy.').eval()
    w, i, aux = r(x)
    assert np.array_equal(i.cpu().numpy(), arrays['noisy_eval/i']) and np.allclose(w.detach().cpu().numpy(), arrays['noisy_eval/w'], atol=1e-5)
    assert abs(float(aux['load_balance_loss']) - float(arrays['noisy_eval/lb'])) < 1e-6
    r.train()
    r._injected_noise = dw.normal('parts.noise', (B, S, E), seed).to(DEV)
    w, i, aux = r(x)
    assert np.array_equal(i.cpu().numpy(), arrays['noisy_train/i']) and np.allclose(w.detach().cpu().numpy(), arrays['noisy_train/w'], atol=1e-5)
    r = load(create_router('topk', D, E, top_k=3), 'topk.').eval()
    w, i, aux = r(x)
    assert np.array_equal(i.cpu().numpy(), arrays['topk/i']) and np.allclose(w.detach().cpu().numpy(), arrays['topk/w'], atol=1e-5)
    r = load(create_router('soft', D, E, temperature=0.7), 'soft.').eval()
    w, i, aux = r(x)
    assert np.array_equal(i.cpu().numpy(), arrays['soft/i']) and np.allclose(w.detach().cpu().numpy(), arrays['soft/w'], atol=1e-5)
    # MOELayer with feed-forward experts (token-local -> sparse dispatch at S > 1), then ablation-style patched router
    layer = load(MOELayer(input_dim=D, hidden_dim=48, output_dim=D, num_experts=4, top_k=2, router_type='topk', expert_type='feedforward'),
                 'moeff.').eval()
    y = layer(x)
    assert rl(y, torch.from_numpy(arrays['moeff/out'])) <= OUT_TOL
    inner = layer.router.forward

    def patched(inp, **kw):
        w, i, aux = inner(inp, **kw)
        dis = i == 1
        w = w.masked_fill(dis, 0.0)
        i = i.masked_fill(dis, -1)
        return w / w.sum(dim=-1, keepdim=True).clamp(min=1e-9), i, aux
    layer.router.forward = patched
    assert rl(layer(x), torch.from_numpy(arrays['moeff/out_disabled1'])) <= OUT_TOL
