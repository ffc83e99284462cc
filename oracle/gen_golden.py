"""Golden-vector generator.  Runs ONLY in the build container, where /root/reference is importable.

    mkdir -p /tmp/golden_cwd && cd /tmp/golden_cwd && \
    PYTHONPATH=/root/reference:/root/repo python3 -B /root/repo/oracle/gen_golden.py [--only tiny|full|full32|parts|moe_variants|generative|fusion]

It imports the reference's own ``src.modeling.meta_arch`` / ``src.modeling.moe`` modules (SURVEY.md
§8c, Appendix C), replaces only the two hub-NAME loaders by local random-weight construction of the
same HF classes, loads the deterministic weights of ``oracle/det_weights.py``, runs the reference in
``eval()`` with autograd enabled and writes inputs-free fixtures (expected outputs + gradients) to
``tests/golden/*.npz``.  Weights and inputs are NOT stored: they are regenerated bit-identically from
names/shapes/seeds recorded in the fixture's ``meta`` JSON.  Nothing from the reference is copied.
"""

import argparse
import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import det_weights as dw  # noqa: E402

OUT = os.path.join(REPO, 'tests', 'golden')

TINY = dict(D=64, vit_heads=4, vit_layers=2, vit_inter=128, image=48, patch=16,
            vocab=120, txt_heads=4, txt_layers=2, txt_inter=128, max_pos=20, seq=8,
            fusion_heads=4, fusion_layers=2, moe_hidden=128, num_answers=37, answer_hidden=[48, 40], batch=3)
FULL = dict(D=768, vit_heads=12, vit_layers=12, vit_inter=3072, image=224, patch=32,
            vocab=64001, txt_heads=12, txt_layers=12, txt_inter=3072, max_pos=258, seq=64,
            fusion_heads=8, fusion_layers=2, moe_hidden=2048, num_answers=3000, answer_hidden=[768, 512], batch=8)


def build_reference_model(dims, fusion_type, num_experts, pooling='cls'):
    from transformers import CLIPVisionConfig, CLIPVisionModel, RobertaConfig, RobertaModel
    import src.modeling.meta_arch.vqa_model as vm
    from src.modeling.meta_arch.vqa_config import (VQAModelConfig, VisualEncoderConfig, TextEncoderConfig,
                                                   FusionConfig, MOEConfig, KnowledgeConfig, AnswerHeadConfig)
    d = dims

    def vis_init(self):
        self.backbone = CLIPVisionModel(CLIPVisionConfig(
            hidden_size=d['D'], intermediate_size=d['vit_inter'], num_hidden_layers=d['vit_layers'],
            num_attention_heads=d['vit_heads'], image_size=d['image'], patch_size=d['patch'],
            hidden_act='quick_gelu', layer_norm_eps=1e-5))
        self.backbone_dim, self.processor = d['D'], None

    def txt_init(self):
        self.encoder = RobertaModel(RobertaConfig(
            vocab_size=d['vocab'], hidden_size=d['D'], num_hidden_layers=d['txt_layers'],
            num_attention_heads=d['txt_heads'], intermediate_size=d['txt_inter'],
            max_position_embeddings=d['max_pos'], type_vocab_size=1, pad_token_id=1, bos_token_id=0,
            eos_token_id=2, layer_norm_eps=1e-5))
        self.encoder_dim, self.tokenizer = d['D'], None

    vm.VisualEncoder._init_backbone, vm.TextEncoder._init_encoder = vis_init, txt_init
    cfg = VQAModelConfig(
        visual_encoder=VisualEncoderConfig(output_dim=d['D']),
        text_encoder=TextEncoderConfig(output_dim=d['D'], max_length=d['seq'], pooling_strategy=pooling),
        fusion=FusionConfig(fusion_type=fusion_type, hidden_dim=d['D'], output_dim=d['D'],
                            num_heads=d['fusion_heads'], num_layers=d['fusion_layers'], dropout=0.1),
        moe=MOEConfig(use_moe=num_experts > 0, num_experts=max(num_experts, 1), top_k=2, hidden_dim=d['moe_hidden']),
        knowledge=KnowledgeConfig(use_knowledge=False),
        answer_head=AnswerHeadConfig(num_answers=d['num_answers'], hidden_dims=list(d['answer_hidden']), dropout=0.3))
    return vm.VietnameseVQAModel(cfg).eval(), cfg


def sample_grad(g, rich):
    """Gradient sample stored in a fixture: the whole tensor when small, else head + strided sample
    (``oracle.det_weights``-independent, so tests re-apply it to their own gradients)."""
    f = g.detach().flatten()
    head, nstr, full_below = (128, 256, 2048) if rich else (64, 64, 0)
    if f.numel() <= max(full_below, head + nstr):
        return f
    stride = f.numel() // nstr
    return torch.cat([f[:head], f[::stride][:nstr]])


class _RoundBF16(torch.autograd.Function):
    """bf16 round-trip of a GEMM operand in forward and of its incoming gradient in backward."""

    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


def bf16_envelope(sd, cfg, px, ids, mask, labels, dims, grad_names):
    """What bf16 GEMM operands with fp32 accumulation cost, measured on the (reference-pinned) CPU oracle itself:
    every linear / matmul / conv operand is rounded to bf16, everything else stays fp32.  The HIP path uses exactly
    this numeric scheme, so these deviations from the fp32 result are the floor any bf16 implementation (torch
    autocast included) sits on; parity tests hold the HIP path to a small multiple of them."""
    import torch.nn.functional as F
    from oracle import vqa_oracle as vo
    kw = dict(vit_heads=dims['vit_heads'], text_heads=dims['txt_heads'])
    l0, _, _, g0 = vo.forward_backward(sd, cfg, px, ids, mask, labels, **kw)
    ol, om, oc = F.linear, torch.matmul, F.conv2d
    F.linear = lambda x, w, b=None: ol(_RoundBF16.apply(x), _RoundBF16.apply(w), b)
    torch.matmul = lambda a, b: om(_RoundBF16.apply(a), _RoundBF16.apply(b))
    F.conv2d = lambda x, w, *a, **k: oc(_RoundBF16.apply(x), _RoundBF16.apply(w), *a, **k)
    try:
        l1, _, _, g1 = vo.forward_backward(sd, cfg, px, ids, mask, labels, **kw)
    finally:
        F.linear, torch.matmul, F.conv2d = ol, om, oc
    rl = lambda a, b: float((a - b).double().norm() / (b.double().norm() + 1e-30))
    out = {'emul/logits_rel_l2': np.float64(rl(l1, l0)), 'emul/logits_max_abs': np.float64(float((l1 - l0).abs().max()))}
    out['emul/g'] = np.array([rl(g1[k], g0[k]) if (k in g0 and k in g1) else np.nan for k in grad_names], dtype=np.float64)
    return out


AC_MODES = (('ac_bf16', torch.bfloat16, 1.0), ('ac_fp16', torch.float16, 1024.0))


def autocast_envelope(model, px, ids, mask, labels, logits0, grads0, rich):
    """The REFERENCE ITSELF under ``torch.autocast`` -- the context its training loops run the model in
    (reference src/core/training_pipeline.py:457 fp16 + GradScaler; src/pipeline/trainer/vqa_trainer.py:760-764 fp16|bf16) --
    compared with its own fp32 result on the same weights and inputs.  (The old operand-rounding emulation inside our own oracle, ``bf16_envelope``, is kept as an
    informative second number only.)  This is what "the reference at 16-bit operand
    precision" deviates from "the reference at fp32" by: the tolerance the HIP path is held to comes from these numbers,
    not from any emulation of ours.  fp16 runs with a fixed loss scale of 1024 (GradScaler's role; gradients unscaled
    before the comparison).  Stored per mode: logits rel-L2 / max-abs, answer ids, and per parameter the rel-L2 error of the
    whole gradient (``/g``) and of the fixture's sample of it (``/gs``, what the GPU test can compare), as vectors in the order
    of the fixture's ``grad_names``."""
    rl = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
    out = {}
    for tag, dt, scale in AC_MODES:
        model.zero_grad(set_to_none=True)
        with torch.enable_grad():
            with torch.autocast('cpu', dtype=dt):
                o = model(pixel_values=px, input_ids=ids, attention_mask=mask, labels=labels)
            (o.loss.float() * scale).backward()
        l1 = o.logits.detach().float()
        out[tag + '/logits_rel_l2'] = np.float64(rl(l1, logits0))
        out[tag + '/logits_max_abs'] = np.float64(float((l1 - logits0).abs().max()))
        out[tag + '/predictions'] = l1.argmax(-1).numpy()
        named = dict(model.named_parameters())
        gfull, gsamp = [], []
        for name in grads0:                                   # order = meta['grad_names']
            g1 = named[name].grad.detach().float() / scale
            gfull.append(rl(g1, grads0[name]))
            gsamp.append(rl(sample_grad(g1, rich), sample_grad(grads0[name], rich)))
        out[tag + '/g'] = np.array(gfull, dtype=np.float64)   # per parameter, whole gradient
        out[tag + '/gs'] = np.array(gsamp, dtype=np.float64)  # per parameter, on the fixture's sample of it
    model.zero_grad(set_to_none=True)
    return out


def select_samples(model, dims, num_experts, seed, n_cand, n_keep):
    """Rows of ``det_weights.make_input_pool`` kept for a batch-32 fixture: the candidates with the LARGEST top-1/top-2
    logit margin in the reference (argmax ids can then be gated bit-exact on every sample: the margins end up an order of
    magnitude above the 16-bit logit error), excluding -- with a MoE -- samples whose k-th / (k+1)-th router probabilities
    are closer than 0.02 (a discrete expert choice that 16-bit rounding could flip).  Samples are independent in eval mode,
    so selecting rows changes nothing about any row's expected output."""
    px, ids, mask, labels = dw.make_input_pool(n_cand, dims['seq'], dims['image'], vocab_hi=min(30000, dims['vocab']),
                                               num_answers=dims['num_answers'], seed=seed)
    margins, gaps = [], []
    with torch.no_grad():
        for i0 in range(0, n_cand, 32):
            sl = slice(i0, min(n_cand, i0 + 32))
            o = model(pixel_values=px[sl], input_ids=ids[sl], attention_mask=mask[sl])
            t2 = o.logits.topk(2, dim=-1).values
            margins.append(t2[:, 0] - t2[:, 1])
            if num_experts > 0:
                pr = model.moe_layer.aux_outputs['router_probs'].reshape(t2.shape[0], -1).sort(dim=-1, descending=True).values
                gaps.append(pr[:, 1] - pr[:, 2])
    margins = torch.cat(margins)
    ok = torch.ones(n_cand, dtype=torch.bool) if not gaps else torch.cat(gaps) > 0.02
    score = torch.where(ok, margins, torch.full_like(margins, -1.0))
    idx = torch.sort(score.topk(n_keep).indices).values
    assert bool(ok[idx].all())
    assert int((mask[idx].sum(1) < dims['seq']).sum()) >= 2, 'selection lost the padded rows'
    print(f'[gen_golden] selected {n_keep}/{n_cand}: min margin {float(margins[idx].min()):.3f} (pool median {float(margins.median()):.3f}), '
          f'{int((mask[idx].sum(1) < dims["seq"]).sum())} padded rows, {int((ids[idx] == 1).any(1).sum())} rows with pad ids')
    return idx, (px[idx], ids[idx], mask[idx], labels[idx])


def min_relu_margin(model, px, ids, mask):
    """Smallest |pre-activation| over every nn.ReLU of the reference model (answer head, concat fusion) on this batch: a unit this
    close to zero is a numerical coin toss at 16-bit operands -- its 0/1 derivative flips and, on a 3-sample fixture, moves the
    whole gradient by 10-20 % (measured: tiny_concat, seed 11, fp16: ONE of 120 units of classifier.3 at |pre| < 4e-4)."""
    vals, hooks = [], []
    for m in model.modules():
        if isinstance(m, torch.nn.ReLU):
            hooks.append(m.register_forward_pre_hook(lambda _m, inp: vals.append(float(inp[0].detach().abs().min()))))
    with torch.no_grad():
        model(pixel_values=px, input_ids=ids, attention_mask=mask)
    for h in hooks:
        h.remove()
    return min(vals) if vals else float('inf')


def run_model_case(tag, dims, fusion_type, num_experts, seed, full_grads, pool=None, emulate=True, relu_margin=None):
    """``relu_margin`` (tiny fixtures): the seed is advanced in steps of 1000 until no ReLU pre-activation of the reference lies
    within that distance of zero (fixture margins >> tolerance, as for the answer ids): the seed actually used is stored."""
    model, cfg = build_reference_model(dims, fusion_type, num_experts)
    shapes = dw.shapes_of(model.state_dict())
    for attempt in range(400):
        sd = dw.make_state_dict(shapes, seed)
        model.load_state_dict(sd)
        if relu_margin is None:
            break
        px, ids, mask, labels = dw.make_inputs(dims['batch'], dims['seq'], dims['image'],
                                               vocab_hi=min(30000, dims['vocab']), num_answers=dims['num_answers'], seed=seed)
        mm = min_relu_margin(model, px, ids, mask)
        if mm >= relu_margin:
            print(f'[gen_golden] {tag}: seed {seed} (attempt {attempt}): smallest ReLU pre-activation {mm:.3e}')
            break
        seed += 1000
    else:
        raise RuntimeError('no seed with the requested ReLU margin')
    sel = None
    if pool is None:
        px, ids, mask, labels = dw.make_inputs(dims['batch'], dims['seq'], dims['image'],
                                               vocab_hi=min(30000, dims['vocab']), num_answers=dims['num_answers'], seed=seed)
    else:
        sel, (px, ids, mask, labels) = select_samples(model, dims, num_experts, seed, pool, dims['batch'])
    with torch.enable_grad():
        out = model(pixel_values=px, input_ids=ids, attention_mask=mask, labels=labels, return_features=True)
        out.loss.backward()
    arrays = {'logits': out.logits.detach().numpy(), 'loss': out.loss.detach().numpy(),
              'predictions': out.predictions.numpy(),
              'visual_pooled': out.visual_features.detach().numpy(),
              'text_pooled': out.text_features.detach().numpy(),
              'fused': out.fused_features.detach().numpy()}
    top2 = out.logits.detach().topk(2, dim=-1).values
    arrays['margin'] = (top2[:, 0] - top2[:, 1]).numpy()
    grad_names, none_names = [], []
    for name, p in model.named_parameters():
        if p.grad is None:
            none_names.append(name)
            continue
        grad_names.append(name)
        g = p.grad.detach()
        arrays['gnorm/' + name] = np.float64(g.double().norm().item())
        arrays['g/' + name] = sample_grad(g, full_grads).numpy().copy()
    if num_experts > 0:
        aux = model.moe_layer.aux_outputs
        arrays['router_probs'] = aux['router_probs'].detach().numpy()
        arrays['load_balance_loss'] = aux['load_balance_loss'].detach().numpy()
    from tests.conftest import CfgView
    grads0 = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    ac = autocast_envelope(model, px, ids, mask, labels, out.logits.detach().clone(), grads0, full_grads)
    arrays.update(ac)
    env = {}
    if emulate:
        env = bf16_envelope(sd, CfgView(dict(dims=dims, fusion_type=fusion_type, num_experts=num_experts)), px, ids, mask, labels, dims, grad_names)
        arrays.update(env)
    if sel is not None:
        arrays['pool_index'] = sel.numpy()
    for mode, _, _ in AC_MODES:
        gn = dict(zip(grad_names, ac[mode + '/g'].tolist()))
        num = sum((gn[n] * float(arrays['gnorm/' + n])) ** 2 for n in gn)
        den = sum(float(arrays['gnorm/' + n]) ** 2 for n in gn)
        worst = max(gn.items(), key=lambda kv: kv[1] if float(arrays['gnorm/' + kv[0]]) > 1e-4 * max(float(arrays['gnorm/' + n]) for n in gn) else 0.0)
        print(f'[gen_golden] {tag}: reference under autocast {mode[3:]}: logits rel-L2 {float(ac[mode + "/logits_rel_l2"]):.2e} max-abs '
              f'{float(ac[mode + "/logits_max_abs"]):.2e}, answer ids kept {int((ac[mode + "/predictions"] == arrays["predictions"]).sum())}/{len(arrays["predictions"])}, '
              f'gradients aggregate {np.sqrt(num / den):.2e} worst {worst[1]:.2e} ({worst[0]})')
    meta = dict(tag=tag, dims=dims, fusion_type=fusion_type, num_experts=num_experts, seed=seed, pool=pool,
                full_grads=full_grads, shapes={k: list(v) for k, v in shapes.items()},
                grad_names=grad_names, none_grad_names=none_names, weights_checksum=dw.checksum(sd),
                torch=torch.__version__, transformers=__import__('transformers').__version__)
    arrays['meta'] = np.array(json.dumps(meta))
    path = os.path.join(OUT, f'{tag}.npz')
    np.savez_compressed(path, **arrays)
    if env:
        print(f'[gen_golden] {tag}: bf16 envelope logits {float(env["emul/logits_rel_l2"]):.2e}, grads median '
              f'{float(np.nanmedian(env["emul/g"])):.2e}')
    print(f'[gen_golden] {tag}: loss={float(out.loss):.6f} min-margin={float(arrays["margin"].min()):.4f} '
          f'params={sum(p.numel() for p in model.parameters())} -> {path} ({os.path.getsize(path) / 1e3:.0f} kB)')


def run_parts(seed=7):
    """Component-level fixtures: routers (eval, injected-noise train, topk, soft), MOELayer combine with
    ablation-style -1 indices, stand-alone CrossModalAttention with masks, text pooling variants."""
    import src.modeling.meta_arch.vqa_model as vm
    from src.modeling.moe.router import NoisyTopKRouter, TopKRouter, SoftRouter
    from src.modeling.moe.moe_layer import MOELayer
    arrays, meta = {}, {'seed': seed, 'cases': {}}

    def load(mod, prefix):
        shapes = {prefix + k: tuple(v.shape) for k, v in mod.state_dict().items()}
        sd = dw.make_state_dict(shapes, seed)
        mod.load_state_dict({k[len(prefix):]: v for k, v in sd.items()})
        meta['cases'][prefix] = {k: list(v) for k, v in shapes.items()}
        return sd

    B, S, D, E, K = 3, 5, 32, 6, 2
    x = dw.normal('parts.x', (B, S, D), seed)
    # --- routers
    r = NoisyTopKRouter(D, E, K).eval()
    load(r, 'noisy.')
    w, i, aux = r(x)
    arrays.update({'noisy_eval/w': w.detach().numpy(), 'noisy_eval/i': i.numpy(),
                   'noisy_eval/lb': aux['load_balance_loss'].detach().numpy(),
                   'noisy_eval/probs': aux['router_probs'].detach().numpy()})
    noise = dw.normal('parts.noise', (B, S, E), seed)
    r.train()
    orig = torch.randn_like
    torch.randn_like = lambda t, **kw: noise.to(t.dtype)
    try:
        w, i, aux = r(x)
    finally:
        torch.randn_like = orig
    arrays.update({'noisy_train/w': w.detach().numpy(), 'noisy_train/i': i.numpy(),
                   'noisy_train/lb': aux['load_balance_loss'].detach().numpy()})
    r = TopKRouter(D, E, 3).eval()
    load(r, 'topk.')
    w, i, aux = r(x)
    arrays.update({'topk/w': w.detach().numpy(), 'topk/i': i.numpy(), 'topk/lb': aux['load_balance_loss'].detach().numpy()})
    r = SoftRouter(D, E, temperature=0.7).eval()
    load(r, 'soft.')
    w, i, aux = r(x)
    arrays.update({'soft/w': w.detach().numpy(), 'soft/i': i.numpy(), 'soft/entropy': aux['entropy'].detach().numpy()})
    # --- MOELayer (feed-forward experts) with an ablation-style patched router: expert 1 disabled (-1)
    layer = MOELayer(input_dim=D, hidden_dim=48, output_dim=D, num_experts=4, top_k=2, router_type='topk',
                     expert_type='feedforward').eval()
    load(layer, 'moeff.')
    y = layer(x)
    arrays['moeff/out'] = y.detach().numpy()
    inner = layer.router.forward

    def patched(inp, **kw):
        w, i, aux = inner(inp, **kw)
        dis = i == 1
        w = w.masked_fill(dis, 0.0)
        i = i.masked_fill(dis, -1)
        w = w / w.sum(dim=-1, keepdim=True).clamp(min=1e-9)
        return w, i, aux
    layer.router.forward = patched
    arrays['moeff/out_disabled1'] = layer(x).detach().numpy()
    # --- CrossModalAttention with both masks
    cma = vm.CrossModalAttention(D, 4, 0.1).eval()
    load(cma, 'cma.')
    kv = dw.normal('parts.kv', (B, 7, D), seed)
    qm = torch.zeros(B, S, dtype=torch.bool)
    qm[1, 3:] = True
    km = torch.zeros(B, 7, dtype=torch.bool)
    km[2, 5:] = True
    xq = x.clone().requires_grad_(True)
    kvq = kv.clone().requires_grad_(True)
    y = cma(xq, kvq, qm, km)
    (y * dw.normal('parts.gy', tuple(y.shape), seed)).sum().backward()
    arrays.update({'cma/out': y.detach().numpy(), 'cma/dquery': xq.grad.numpy(), 'cma/dkv': kvq.grad.numpy()})
    for n, p in cma.named_parameters():
        arrays['cma/g/' + n] = p.grad.numpy()
    # --- moe_utils helpers (src/modeling/moe/moe_utils.py) on the noisy router's eval outputs
    import src.modeling.moe.moe_utils as mu
    r = NoisyTopKRouter(D, E, K).eval()
    r.load_state_dict({k[len('noisy.'):]: v for k, v in dw.make_state_dict({k: tuple(v) for k, v in meta['cases']['noisy.'].items()}, seed).items()})
    w_u, i_u, aux_u = r(x)
    probs_u, logits_u = aux_u['router_probs'].detach(), r.gate(x).detach()
    ana = mu.analyze_routing_patterns(probs_u, i_u, E)
    dropped = mu.ExpertDropout(E, 0.4).train()
    keep = (dw.normal('parts.keep', (E,), seed) > -0.3).float()       # injected Bernoulli draw: experts with keep == 0 are dropped
    orig_b = torch.bernoulli
    torch.bernoulli = lambda t, **kw: keep.to(t.dtype)
    try:
        w_d, _ = dropped(w_u.detach(), i_u)
    finally:
        torch.bernoulli = orig_b
    arrays.update({
        'utils/capacity': np.array([mu.compute_expert_capacity(96, 6, 2), mu.compute_expert_capacity(15, 4, 1, 1.0), mu.compute_expert_capacity(7, 8, 2, 2.5)]),
        'utils/load_balance': mu.compute_load_balance_loss(probs_u, i_u, E, 0.02).numpy(),
        'utils/z_loss': mu.compute_router_z_loss(logits_u, 0.003).numpy(),
        'utils/entropy': mu.compute_expert_entropy(probs_u).numpy(),
        'utils/utilization': np.array([mu.get_expert_utilization(i_u, E)[e] for e in range(E)]),
        'utils/ana_entropy': np.float64(ana['routing_entropy']), 'utils/ana_max': np.float64(ana['max_prob_mean']),
        'utils/ana_min': np.float64(ana['min_prob_mean']), 'utils/ana_cosel': np.array(ana['expert_co_selection']),
        'utils/dropout_w': w_d.numpy(), 'utils/dropout_keep': keep.numpy()})
    # --- text pooling variants (vqa_model.py:179-204)
    te = object.__new__(vm.TextEncoder)
    am = torch.ones(B, S, dtype=torch.int64)
    am[1, 3:] = 0
    for strat in ('cls', 'mean', 'max'):
        te.config = type('C', (), {'pooling_strategy': strat})()
        arrays['pool/' + strat] = vm.TextEncoder._pool_features(te, x, am).numpy()
    meta['dims'] = dict(B=B, S=S, D=D, E=E, K=K)
    arrays['meta'] = np.array(json.dumps(meta))
    path = os.path.join(OUT, 'parts.npz')
    np.savez_compressed(path, **arrays)
    print(f'[gen_golden] parts -> {path} ({os.path.getsize(path) / 1e3:.0f} kB)')


GEN_TINY = dict(TINY, gen_vocab=96, gen_heads=4, gen_layers=2, gen_ff=96, answer_len=7)
GEN_FULL = dict(FULL, gen_vocab=64000, gen_heads=8, gen_layers=6, gen_ff=2048, answer_len=16, batch=2)


def build_reference_generative(dims, use_moe=False, moe_type='standard'):
    """The reference's ``GenerativeVQAModel`` (generative_vqa_model.py:479-598) with the two hub-name loaders replaced by local
    random-weight construction of the same HF classes, as for the classification model."""
    import transformers
    from transformers import CLIPVisionConfig, CLIPVisionModel, RobertaConfig, RobertaModel
    import src.modeling.meta_arch.generative_vqa_model as gm
    d = dims
    clip = lambda name=None: CLIPVisionModel(CLIPVisionConfig(
        hidden_size=d['D'], intermediate_size=d['vit_inter'], num_hidden_layers=d['vit_layers'], num_attention_heads=d['vit_heads'],
        image_size=d['image'], patch_size=d['patch'], hidden_act='quick_gelu', layer_norm_eps=1e-5))
    rob = lambda name=None: RobertaModel(RobertaConfig(
        vocab_size=d['vocab'], hidden_size=d['D'], num_hidden_layers=d['txt_layers'], num_attention_heads=d['txt_heads'],
        intermediate_size=d['txt_inter'], max_position_embeddings=d['max_pos'], type_vocab_size=1, pad_token_id=1, bos_token_id=0,
        eos_token_id=2, layer_norm_eps=1e-5))
    old_clip, old_auto = transformers.CLIPVisionModel.from_pretrained, gm.AutoModel
    transformers.CLIPVisionModel.from_pretrained = staticmethod(clip)
    gm.AutoModel = type('AutoModelStub', (), {'from_pretrained': staticmethod(rob)})
    try:
        cfg = gm.GenerativeVQAConfig(hidden_size=d['D'], fusion_dim=d['D'], num_decoder_layers=d['gen_layers'], num_attention_heads=d['gen_heads'],
                                     decoder_ff_dim=d['gen_ff'], max_answer_length=max(d['answer_len'], 8), fusion_num_heads=d['fusion_heads'],
                                     fusion_num_layers=d['fusion_layers'], vocab_size=d['gen_vocab'], use_moe=use_moe, moe_type=moe_type)
        model = gm.GenerativeVQAModel(cfg).eval()
    finally:
        transformers.CLIPVisionModel.from_pretrained, gm.AutoModel = old_clip, old_auto
    return model, cfg


def make_decoder_inputs(dims, seed):
    """Teacher-forcing inputs: decoder ids [B, A] starting with BOS, one right-padded row, labels = the next token (-100 on pads)."""
    B, A, V = dims['batch'], dims['answer_len'], dims['gen_vocab']
    toks = dw.randint('answer_tokens', (B, A + 1), 3, V, seed)
    toks[:, 0] = 0
    dec_in, labels = toks[:, :-1].clone(), toks[:, 1:].clone()
    dmask = torch.ones(B, A, dtype=torch.int64)
    if B > 1:
        cut = max(2, (A * 5) // 8)
        dmask[1, cut:] = 0
        dec_in[1, cut:] = 1
        labels[1, cut - 1:] = -100
    return dec_in, dmask, labels


def _router_gap(model, kw):
    """Smallest gap between the k-th and (k+1)-th router probability over all tokens of the fusion MoE (eval forward, no grad)."""
    moe = model.fusion.moe_layer
    with torch.no_grad():
        model(**{k: v for k, v in kw.items() if k != 'labels'})
    p = moe.aux_outputs['router_probs'].reshape(-1, moe.num_experts)
    srt = p.sort(dim=-1, descending=True).values
    return float((srt[:, moe.top_k - 1] - srt[:, moe.top_k]).min()), p


def run_generative_case(tag, dims, seed, full_logits, use_moe=False, moe_type='standard', tries=1):
    """``use_moe``: the reference's fusion MoE over the 114 (tiny: 13) concatenated tokens (generative_vqa_model.py:224-339): moe_type 'vqa'
    = VQAMOELayer (Vision / Text / Multimodal / Segmentation experts ATTENDING ACROSS the tokens of a sample, NoisyTopK router), 'standard'
    = MOELayer(config=...) (FeedForward experts, TopK router).  Top-k routing of B x 114 tokens is discrete and the router probabilities of
    random-weight models are close to uniform: of ``tries`` seeds the one with the LARGEST smallest gap between a token's k-th and (k+1)-th
    probability is kept, and the reference's expert choice is stored (``expert_indices``) so that a parity test can hold the discrete
    decisions fixed where a 16-bit run lands inside a numerical tie (and must agree everywhere else)."""
    model, cfg = build_reference_generative(dims, use_moe, moe_type)
    shapes = dw.shapes_of(model.state_dict())

    def setup(sd_seed):
        sd_ = dw.make_state_dict(shapes, sd_seed)
        model.load_state_dict(sd_)
        px_, ids_, mask_, _ = dw.make_inputs(dims['batch'], dims['seq'], dims['image'], vocab_hi=min(30000, dims['vocab']), num_answers=8, seed=sd_seed)
        dec_in_, dmask_, labels_ = make_decoder_inputs(dims, sd_seed)
        return sd_, dict(pixel_values=px_, input_ids=ids_, attention_mask=mask_, decoder_input_ids=dec_in_, decoder_attention_mask=dmask_, labels=labels_)
    best = (-1.0, seed)
    if use_moe and tries > 1:
        for t in range(tries):
            _, kw_t = setup(seed + 1000 * t)
            gap, _ = _router_gap(model, kw_t)
            print(f'[gen_golden] {tag}: seed {seed + 1000 * t}: smallest router gap {gap:.2e}', flush=True)
            best = max(best, (gap, seed + 1000 * t))
        seed = best[1]
    sd, kw = setup(seed)
    px, ids, mask, dec_in, dmask, labels = (kw[k] for k in ('pixel_values', 'input_ids', 'attention_mask', 'decoder_input_ids', 'decoder_attention_mask', 'labels'))
    routed = {}
    hook = model.fusion.moe_layer.router.register_forward_hook(lambda m, i, o: routed.update(w=o[0].detach(), idx=o[1].detach())) if use_moe else None
    with torch.enable_grad():
        out = model(**kw)
        out.loss.backward()
    logits0 = out.logits.detach().clone()
    arrays = {'loss': out.loss.detach().numpy(), 'memory': out.encoder_hidden_states.detach().numpy(),
              'logits' if full_logits else 'logits_sample': (logits0 if full_logits else logits0.flatten()[::97]).numpy(),
              'argmax': logits0.argmax(-1).numpy()}
    if use_moe:
        hook.remove()
        moe = model.fusion.moe_layer
        arrays['router_probs'] = moe.aux_outputs['router_probs'].detach().numpy()
        arrays['load_balance_loss'] = np.float64(float(moe.aux_outputs['load_balance_loss']))
        arrays['expert_indices'] = routed['idx'].numpy()              # what the reference's router returned in THIS forward
        arrays['routing_weights'] = routed['w'].numpy()
        srt = np.sort(arrays['router_probs'].reshape(-1, moe.num_experts), axis=-1)[:, ::-1]
        arrays['router_gap'] = (srt[:, moe.top_k - 1] - srt[:, moe.top_k]).reshape(arrays['router_probs'].shape[:2])
    top2 = logits0.topk(2, dim=-1).values
    arrays['margin'] = (top2[..., 0] - top2[..., 1]).numpy()
    grad_names, none_names, grads0 = [], [], {}
    for name, p in model.named_parameters():
        if p.grad is None:
            none_names.append(name)
            continue
        grad_names.append(name)
        g = p.grad.detach()
        grads0[name] = g.clone()
        arrays['gnorm/' + name] = np.float64(g.double().norm().item())
        arrays['g/' + name] = sample_grad(g, True).numpy().copy()
    rl = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
    for mode, dt, scale in AC_MODES:                      # the reference itself under autocast: the tolerance contract
        model.zero_grad(set_to_none=True)
        with torch.enable_grad():
            with torch.autocast('cpu', dtype=dt):
                o = model(**kw)
            (o.loss.float() * scale).backward()
        l1 = o.logits.detach().float()
        arrays[mode + '/logits_rel_l2'] = np.float64(rl(l1, logits0))
        arrays[mode + '/loss_abs'] = np.float64(abs(float(o.loss) - float(out.loss)))
        named = dict(model.named_parameters())
        arrays[mode + '/gs'] = np.array([rl(sample_grad(named[n].grad.detach().float() / scale, True), sample_grad(grads0[n], True)) for n in grad_names])
        num = sum((e * float(arrays['gnorm/' + n])) ** 2 for e, n in zip(arrays[mode + '/gs'], grad_names))
        den = sum(float(arrays['gnorm/' + n]) ** 2 for n in grad_names)
        print(f'[gen_golden] {tag}: reference under autocast {mode[3:]}: logits rel-L2 {float(arrays[mode + "/logits_rel_l2"]):.2e}, '
              f'gradients aggregate {np.sqrt(num / den):.2e}')
    model.zero_grad(set_to_none=True)
    meta = dict(tag=tag, dims=dims, seed=seed, shapes={k: list(v) for k, v in shapes.items()}, keys=list(shapes), grad_names=grad_names,
                none_grad_names=none_names, weights_checksum=dw.checksum(sd), torch=torch.__version__,
                transformers=__import__('transformers').__version__, use_moe=bool(use_moe), moe_type=moe_type, min_router_gap=float(best[0]))
    arrays['meta'] = np.array(json.dumps(meta))
    path = os.path.join(OUT, f'{tag}.npz')
    np.savez_compressed(path, **arrays)
    print(f'[gen_golden] {tag}: loss={float(out.loss):.6f} params={sum(p.numel() for p in model.parameters())} -> {path} '
          f'({os.path.getsize(path) / 1e3:.0f} kB)')


FUSION_CASES = {
    'fusion_xattn_tiny': dict(B=3, V=5, T=8, vision_dim=48, text_dim=64, output_dim=64, num_attention_heads=4, num_layers=2, intermediate_dim=96,
                              fusion_method='concat'),
    'fusion_xattn_full': dict(B=4, V=50, T=64, vision_dim=768, text_dim=768, output_dim=768, num_attention_heads=8, num_layers=4,
                              intermediate_dim=3072, fusion_method='add'),
}


def run_fusion_case(tag, case, seed):
    """The reference's stand-alone ``CrossAttentionFusion`` (fusion_approaches.py:59-281): eval forward + backward of <out, gy>."""
    from src.modeling.fusion.fusion_approaches import CrossAttentionFusion
    from oracle import gen_oracle as go
    kw = {k: case[k] for k in ('vision_dim', 'text_dim', 'output_dim', 'num_attention_heads', 'num_layers', 'intermediate_dim', 'fusion_method')}
    model = CrossAttentionFusion(dropout=0.1, **kw).eval()
    shapes = dw.shapes_of(model.state_dict())
    sd = dw.make_state_dict(shapes, seed)
    model.load_state_dict(sd)
    meta = dict(tag=tag, case=case, seed=seed, shapes={k: list(v) for k, v in shapes.items()}, keys=list(shapes), weights_checksum=dw.checksum(sd))
    v, t, vmask, tmask, gy = go.fusion_fixture_inputs(meta)
    v.requires_grad_(True); t.requires_grad_(True)
    with torch.enable_grad():
        out = model(v, t, vmask, tmask)
        (out * gy).sum().backward()
    arrays = {'out': out.detach().numpy(), 'dv': v.grad.numpy(), 'dt': t.grad.numpy()}
    out0, grads0 = out.detach().clone(), {n: p.grad.detach().clone() for n, p in model.named_parameters()}
    for n, g in grads0.items():
        arrays['gnorm/' + n] = np.float64(g.double().norm().item())
        arrays['g/' + n] = sample_grad(g, True).numpy().copy()
    rl = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
    for mode, dt, scale in AC_MODES:
        model.zero_grad(set_to_none=True)
        with torch.enable_grad():
            with torch.autocast('cpu', dtype=dt):
                o = model(v.detach(), t.detach(), vmask, tmask)
            ((o.float() * gy).sum() * scale).backward()
        arrays[mode + '/out_rel_l2'] = np.float64(rl(o.detach().float(), out0))
        named = dict(model.named_parameters())
        arrays[mode + '/gs'] = np.array([rl(sample_grad(named[n].grad.detach().float() / scale, True), sample_grad(grads0[n], True)) for n in grads0])
        print(f'[gen_golden] {tag}: reference under autocast {mode[3:]}: output rel-L2 {float(arrays[mode + "/out_rel_l2"]):.2e}')
    meta['grad_names'] = list(grads0)
    arrays['meta'] = np.array(json.dumps(meta))
    path = os.path.join(OUT, f'{tag}.npz')
    np.savez_compressed(path, **arrays)
    print(f'[gen_golden] {tag}: |out|={float(out0.norm()):.4f} params={sum(p.numel() for p in model.parameters())} -> {path} ({os.path.getsize(path) / 1e3:.0f} kB)')


def run_moe_variants(seed=17):
    """The reference's other MoE layers (moe_layer.py:199-358 SparseMOELayer with the capacity cut active, :361-548 HierarchicalMOE) and
    the GLU expert (expert_types.py:448-515) in eval mode: output, input gradient, load-balance loss, every parameter gradient."""
    from src.modeling.moe.moe_layer import HierarchicalMOE, SparseMOELayer
    arrays, meta = {}, {'seed': seed, 'cases': {}}
    D, Hm = 64, 128
    cases = {
        'sparse_ff.': (lambda: SparseMOELayer(input_dim=D, hidden_dim=Hm, output_dim=D, num_experts=4, top_k=2, capacity_factor=0.6,
                                               dropout=0.1, expert_type='feedforward'), (3, 10, D)),
        'sparse_glu.': (lambda: SparseMOELayer(input_dim=D, hidden_dim=Hm, output_dim=D, num_experts=4, top_k=2, capacity_factor=1.25,
                                                dropout=0.1, expert_type='glu'), (3, 10, D)),
        'hier.': (lambda: HierarchicalMOE(input_dim=D, hidden_dim=Hm, output_dim=D, num_expert_groups=4, experts_per_group=2,
                                           top_k_groups=2, top_k_experts=1, dropout=0.1), (3, 5, D)),
    }
    for prefix, (make, xshape) in cases.items():
        layer = make().eval()
        shapes = {prefix + k: tuple(v.shape) for k, v in layer.state_dict().items()}
        sd = dw.make_state_dict(shapes, seed)
        layer.load_state_dict({k[len(prefix):]: v for k, v in sd.items()})
        meta['cases'][prefix] = {'shapes': {k: list(v) for k, v in shapes.items()}, 'x': list(xshape)}
        x = dw.normal(prefix + 'x', xshape, seed).requires_grad_(True)
        gy = dw.normal(prefix + 'gy', xshape, seed + 1)
        with torch.enable_grad():
            y = layer(x)
            (y * gy).sum().backward()
        arrays[prefix + 'out'] = y.detach().numpy()
        arrays[prefix + 'dx'] = x.grad.numpy()
        arrays[prefix + 'aux'] = np.float64(float(layer.get_aux_loss()))
        if prefix.startswith('sparse'):
            T = xshape[0] * xshape[1]
            arrays[prefix + 'capacity'] = np.int64(layer._compute_capacity(T))
            idx = layer.aux_outputs.get('expert_indices')
        for n, p in layer.named_parameters():
            if p.grad is not None:
                arrays[prefix + 'g/' + n] = p.grad.numpy().copy()
        print(f'[gen_golden] moe_variants {prefix}: |out|={float(y.norm()):.4f} aux={float(layer.get_aux_loss()):.5f} '
              f'grads={sum(p.grad is not None for p in layer.parameters())}/{sum(1 for _ in layer.parameters())}')
    arrays['meta'] = np.array(json.dumps(meta))
    path = os.path.join(OUT, 'moe_variants.npz')
    np.savez_compressed(path, **arrays)
    print(f'[gen_golden] moe_variants -> {path} ({os.path.getsize(path) / 1e3:.0f} kB)')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default='all')
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count())
    if args.only in ('all', 'parts'):
        run_parts()
    if args.only in ('all', 'moe_variants'):
        run_moe_variants()
    if args.only in ('all', 'tiny'):
        run_model_case('tiny_concat', TINY, 'concat', 0, 11, True, relu_margin=1e-2)
        run_model_case('tiny_xattn', TINY, 'cross_attention', 0, 12, True, relu_margin=1e-2)
        run_model_case('tiny_mcan_moe4', TINY, 'mcan', 4, 13, True, relu_margin=1e-2)
        run_model_case('tiny_xattn_moe8', TINY, 'cross_attention', 8, 14, True, relu_margin=1e-2)
        run_model_case('tiny_bilinear', TINY, 'bilinear', 0, 15, True, relu_margin=1e-2)
    if args.only in ('all', 'full'):
        run_model_case('full_cfg1_concat', FULL, 'concat', 0, 21, False)
        run_model_case('full_cfg2_xattn', FULL, 'cross_attention', 0, 22, False)
        run_model_case('full_cfg3_mcan_moe4', FULL, 'mcan', 4, 23, False)
    if args.only in ('all', 'full32'):
        # BASELINE.json's batch (32 per GPU): 32 samples selected from a pool of 192 by reference margin (select_samples)
        F32 = dict(FULL, batch=32)
        run_model_case('full32_cfg1_concat', F32, 'concat', 0, 31, False, pool=192, emulate=False)
        run_model_case('full32_cfg2_xattn', F32, 'cross_attention', 0, 32, False, pool=192, emulate=False)
        run_model_case('full32_cfg3_mcan_moe4', F32, 'mcan', 4, 33, False, pool=192, emulate=False)
    if args.only in ('all', 'generative'):
        # the generative model: tiny with full logits, full-size (64 000-way head) with a logits sample
        run_generative_case('generative_tiny', GEN_TINY, 41, True)
        run_generative_case('generative_full', GEN_FULL, 42, False)
    if args.only in ('all', 'generative', 'generative_moe', 'generative_moe_tiny'):
        # BASELINE configs[4] territory: the generative model WITH its fusion MoE over the 114 concatenated tokens
        run_generative_case('generative_tiny_moe_vqa', GEN_TINY, 43, True, use_moe=True, moe_type='vqa', tries=24)
        run_generative_case('generative_tiny_moe_std', GEN_TINY, 44, True, use_moe=True, moe_type='standard', tries=24)
        if args.only != 'generative_moe_tiny':
            run_generative_case('generative_full_moe_vqa', GEN_FULL, 45, False, use_moe=True, moe_type='vqa', tries=12)
            run_generative_case('generative_full_moe_std', GEN_FULL, 46, False, use_moe=True, moe_type='standard', tries=12)
    if args.only in ('all', 'fusion'):
        for i, (tag, case) in enumerate(FUSION_CASES.items()):
            run_fusion_case(tag, case, 51 + i)


if __name__ == '__main__':
    main()
