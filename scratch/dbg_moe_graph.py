"""Loss trajectories of the tiny MoE model: eager sparse, eager dense (+ branches), graphed dense (+ branches)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_graph_gpu import _moe_setup, _eager
from vqa_model_builder_amd.graph import GraphedTrainStep
from vqa_model_builder_amd.hip import blocks

ref_model, ref_opt, batch = _moe_setup()
print('eager sparse      ', [round(x, 4) for x in _eager(ref_model, ref_opt, batch, 6)])
for br in (0, 1, 2):
    for rep in range(2):
        m, o, _ = _moe_setup()
        m.moe_layer.enable_dense_dispatch(True)
        m.moe_layer.parallel_branches = br
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            tr = _eager(m, o, batch, 6)
        torch.cuda.synchronize()
        print(f'eager dense br={br}  ', [round(x, 4) for x in tr])
for br in (0, 1, 2):
    for rep in range(2):
        m, o, _ = _moe_setup()
        gs = GraphedTrainStep(m, o, batch, warmup=2, moe_branches=br)
        got = [gs(batch).item() for _ in range(4)]
        blocks.disable_indirect_seeds()
        print(f'graph dense br={br}  ', ['-', '-'] + [round(x, 4) for x in got])
