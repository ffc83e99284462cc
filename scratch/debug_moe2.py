import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import det_weights as dw, vqa_oracle as vo
from tests.conftest import CfgView, load_golden
from tests.helpers import build_model
tag = sys.argv[1] if len(sys.argv) > 1 else 'tiny_mcan_moe4'
arrays, meta = load_golden(tag)
d = meta['dims']
sd = dw.make_state_dict({k: tuple(v) for k, v in meta['shapes'].items()}, meta['seed'])
px, ids, mask, labels = dw.make_inputs(d['batch'], d['seq'], d['image'], vocab_hi=min(30000, d['vocab']), num_answers=d['num_answers'], seed=meta['seed'])
model = build_model(meta); model.load_state_dict(sd); model = model.cuda().eval()
rl = lambda a, b: float((a.float().cpu() - b).norm() / (b.norm() + 1e-30))
cap = {}
h1 = model.fusion.register_forward_hook(lambda m, i, o: cap.__setitem__('fused', o.detach()))
h2 = model.moe_layer.register_forward_hook(lambda m, i, o: cap.__setitem__('moe', o.detach()))
orig_router = model.moe_layer.router.forward
def rt(x, **kw):
    w, i, a = orig_router(x, **kw); cap['w'], cap['i'] = w.detach(), i.detach(); return w, i, a
model.moe_layer.router.forward = rt
out = model(pixel_values=px.cuda(), input_ids=ids.cuda(), attention_mask=mask.cuda(), labels=labels.cuda())
out.loss.backward()
# oracle pieces
cfg = CfgView(meta)
vis = vo.clip_vision_forward(sd, 'visual_encoder.backbone.', px, d['vit_heads'])
txt = vo.roberta_forward(sd, 'text_encoder.encoder.', ids, mask, d['txt_heads'])
fused = vo.multimodal_fusion(sd, 'fusion.', cfg.fusion.fusion_type, cfg.fusion.num_heads, vis, txt, text_mask=~mask.bool())
kinds = vo.vqa_moe_expert_kinds(*vo.expert_split(cfg.moe.num_experts))
w, i, aux = vo.noisy_topk_router(sd, 'moe_layer.router.', fused.unsqueeze(1), 2)
moe_o, _ = vo.moe_layer(sd, 'moe_layer.', fused.unsqueeze(1), kinds, 2)
print('fused rel', rl(cap['fused'], fused), ' moe rel', rl(cap['moe'].squeeze(1), moe_o.squeeze(1)))
print('indices hip', cap['i'].flatten().tolist(), 'oracle', i.flatten().tolist())
print('weights hip', cap['w'].flatten().tolist(), 'oracle', w.flatten().tolist())
print('probs oracle', aux['router_probs'].flatten().tolist())
print('logits rel', rl(out.logits.detach(), torch.from_numpy(arrays['logits'])))
