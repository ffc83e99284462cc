"""Host -> HBM feed (vqa_model_builder_amd/data_feed.py) for batches shaped like the reference's ``vqa_collate_fn`` output
(src/data/dataset.py:204-251): values, dtypes and non-tensor fields arrive unchanged and in order; staging buffers are reused; the
renamed dict drives the model."""

import pytest
import torch

pytestmark = pytest.mark.gpu


def _batches(n, B=4):
    g = torch.Generator().manual_seed(0)
    for i in range(n):
        yield {'image': torch.randn(B, 3, 48, 48, generator=g), 'input_ids': torch.randint(2, 100, (B, 8), generator=g),
               'attention_mask': torch.ones(B, 8, dtype=torch.int64), 'label': torch.randint(0, 37, (B,), generator=g),
               'question': [f'q{i}_{j}' for j in range(B)], 'all_answers': [['a'] * 5] * B, 'answer_counts': [{1: 5}] * B}


def test_prefetcher_delivers_the_collated_batches_unchanged_and_in_order():
    from vqa_model_builder_amd.data_feed import DevicePrefetcher, as_model_inputs
    ref = list(_batches(5))
    for pin in (False, True):
        pf = DevicePrefetcher(_batches(5), pin=pin)
        got = list(pf)
        assert len(got) == 5
        for r, g in zip(ref, got):
            for k in ('image', 'input_ids', 'attention_mask', 'label'):
                assert g[k].is_cuda and g[k].dtype == r[k].dtype and torch.equal(g[k].cpu(), r[k]), k
            assert g['question'] == r['question'] and g['answer_counts'] == r['answer_counts']
    assert all(len(p) == 4 for p in pf._pinned) and all(t.is_pinned() for p in pf._pinned for t in p.values())
    kw = as_model_inputs(got[0])
    assert set(kw) == {'pixel_values', 'input_ids', 'attention_mask', 'labels'}
    assert list(DevicePrefetcher([])) == []


def test_prefetched_batch_drives_the_model():
    from oracle import det_weights as dw
    from oracle.gen_golden import TINY
    from tests.helpers import build_model
    from vqa_model_builder_amd.data_feed import DevicePrefetcher, as_model_inputs
    model = build_model({'dims': TINY, 'fusion_type': 'concat', 'num_experts': 0})
    model.load_state_dict(dw.make_state_dict(dw.shapes_of(model.state_dict()), 3))
    model = model.to('cuda').eval()
    for b in DevicePrefetcher(_batches(2, B=3)):
        out = model(**as_model_inputs(b))
        assert out.logits.shape == (3, TINY['num_answers']) and torch.isfinite(out.loss)
