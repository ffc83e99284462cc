import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import det_weights as dw, vqa_oracle as vo
from tests.conftest import CfgView, load_golden
from tests.helpers import build_model
tag = 'tiny_mcan_moe4'
arrays, meta = load_golden(tag)
d = meta['dims']
sd = dw.make_state_dict({k: tuple(v) for k, v in meta['shapes'].items()}, meta['seed'])
model = build_model(meta); model.load_state_dict(sd); model = model.cuda().eval()
moe = model.moe_layer
rl = lambda a, b: float((a.float().cpu() - b).norm() / (b.norm() + 1e-30))
torch.manual_seed(0)
for T in (3, 16):
    x = torch.randn(T, 1, d['D'])
    kinds = vo.vqa_moe_expert_kinds(*vo.expert_split(4))
    xo = x.clone().requires_grad_(True)
    leaves = {k: v.clone().requires_grad_(v.dim() > 0) for k, v in sd.items() if k.startswith('moe_layer.')}
    yo, aux = vo.moe_layer(leaves, 'moe_layer.', xo, kinds, 2)
    w_o, i_o, _ = vo.noisy_topk_router(leaves, 'moe_layer.router.', xo, 2)
    gy = torch.randn_like(yo)
    (yo * gy).sum().backward()
    xg = x.cuda().requires_grad_(True)
    model.zero_grad()
    yg = moe(xg)
    w_g, i_g, _ = moe.router(xg)
    (yg * gy.cuda()).sum().backward()
    print(f'T={T} indices equal {torch.equal(i_g.cpu(), i_o)} out rel {rl(yg.detach(), yo.detach()):.4f} dx rel {rl(xg.grad, xo.grad):.4f} w rel {rl(w_g.detach(), w_o.detach()):.2e}')
    # per expert forward/backward in isolation
    for e, kind in enumerate(kinds):
        xe = x.clone().requires_grad_(True)
        ye = vo.EXPERT_FNS[kind](leaves, f'moe_layer.experts.{e}.', xe)
        for v in leaves.values():
            v.grad = None
        (ye * gy).sum().backward()
        xge = x.cuda().requires_grad_(True)
        model.zero_grad()
        yge = moe.experts[e](xge)
        (yge * gy.cuda()).sum().backward()
        worst = max(((rl(p.grad, leaves[f'moe_layer.experts.{e}.' + n].grad), n) for n, p in moe.experts[e].named_parameters() if p.grad is not None and leaves[f'moe_layer.experts.{e}.' + n].grad is not None and leaves[f'moe_layer.experts.{e}.' + n].grad.norm() > 1e-6), default=(0, ''))
        print(f'   expert {e} {kind:12s} out rel {rl(yge.detach(), ye.detach()):.4f} dx rel {rl(xge.grad, xe.grad):.4f} worst param grad rel {worst[0]:.4f} {worst[1]}')
