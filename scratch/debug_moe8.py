import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '.')
from oracle import det_weights as dw, vqa_oracle as vo
from oracle.gen_golden import TINY
from tests.helpers import build_model
meta = {'dims': TINY, 'fusion_type': 'cross_attention', 'num_experts': 8}
model = build_model(meta)
sd = dw.make_state_dict(dw.shapes_of(model.state_dict()), 5)
model.load_state_dict(sd)
model = model.cuda().eval()
sd32 = {k: v.float() for k, v in sd.items()}
D = TINY['D']
torch.manual_seed(0)
x = torch.randn(5, 1, D)
kinds = vo.vqa_moe_expert_kinds(2, 2, 2, 2)
print('kinds', kinds)
fn = {'vision': vo.vision_expert, 'text': vo.text_expert, 'multimodal': vo.multimodal_expert, 'segmentation': vo.segmentation_expert,
      'detection': vo.object_detection_expert}
for e, kind in enumerate(kinds):
    p = f'moe_layer.experts.{e}.'
    ref = fn[kind](sd32, p, x) if kind in fn else None
    with torch.no_grad():
        got = model.moe_layer.experts[e](x.cuda()).cpu()
    err = ((got - ref).norm() / ref.norm()).item() if ref is not None else float('nan')
    print(e, kind, type(model.moe_layer.experts[e]).__name__, 'rel err %.3e' % err)
