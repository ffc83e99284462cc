import sys, numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, '.')
from oracle import det_weights as dw, vqa_oracle as vo
from tests.conftest import CfgView, load_golden
from tests.helpers import build_model
tag = 'tiny_mcan_moe4'
arrays, meta = load_golden(tag)
d = meta['dims']
sd = dw.make_state_dict({k: tuple(v) for k, v in meta['shapes'].items()}, meta['seed'])
px, ids, mask, labels = dw.make_inputs(d['batch'], d['seq'], d['image'], vocab_hi=min(30000, d['vocab']), num_answers=d['num_answers'], seed=meta['seed'])
model = build_model(meta); model.load_state_dict(sd); model = model.cuda().eval()
cfg = CfgView(meta)
vis = vo.clip_vision_forward(sd, 'visual_encoder.backbone.', px, d['vit_heads'])
txt = vo.roberta_forward(sd, 'text_encoder.encoder.', ids, mask, d['txt_heads'])
fused = vo.multimodal_fusion(sd, 'fusion.', cfg.fusion.fusion_type, cfg.fusion.num_heads, vis, txt, text_mask=~mask.bool()).detach()
kinds = vo.vqa_moe_expert_kinds(*vo.expert_split(4))
leaves = {k: v.clone().requires_grad_(v.dim() > 0) for k, v in sd.items() if k.startswith(('moe_layer.', 'answer_head.'))}
fo = fused.clone().requires_grad_(True)
mo, _ = vo.moe_layer(leaves, 'moe_layer.', fo.unsqueeze(1), kinds, 2)
lo = vo.answer_head(leaves, 'answer_head.', mo.squeeze(1))
F.cross_entropy(lo, labels).backward()
fg = fused.cuda().requires_grad_(True)
model.zero_grad()
mg = model.moe_layer(fg.unsqueeze(1)).squeeze(1)
lg = model.answer_head(mg)
from vqa_model_builder_amd.hip import ops
loss, _ = ops.cross_entropy_argmax(lg, labels.cuda())
loss.backward()
rl = lambda a, b: float((a.float().cpu() - b).norm() / (b.norm() + 1e-30))
print('logits rel', rl(lg.detach(), lo.detach()), 'dfused rel', rl(fg.grad, fo.grad))
rows = []
for n, p in list(model.moe_layer.named_parameters()) + list(model.answer_head.named_parameters()):
    key = ('moe_layer.' if not n.startswith('classifier') else 'answer_head.') + n
    g = leaves[key].grad
    if p.grad is None or g is None or g.norm() < 1e-7: continue
    rows.append((rl(p.grad, g), key))
rows.sort(reverse=True)
for r in rows[:12]: print('  %.4f %s' % r)
print('  median %.4f' % np.median([r[0] for r in rows]))
# row structure of classifier.0.weight error
p = dict(model.answer_head.named_parameters())['classifier.0.weight'].grad.float().cpu(); g = leaves['answer_head.classifier.0.weight'].grad
print('per-row rel err of classifier.0.weight:', np.round(((p - g).norm(dim=1) / (g.norm(dim=1) + 1e-12)).numpy(), 3)[:48])
