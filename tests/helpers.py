"""Builds the HIP model for a golden fixture's dimensions (no hub names: explicit architectures, random init)."""
from vqa_model_builder_amd.modeling.meta_arch import (AnswerHeadConfig, FusionConfig, KnowledgeConfig, MOEConfig, TextEncoderConfig,
                                                      VietnameseVQAModel, VisualEncoderConfig, VQAModelConfig)


def model_config_from_dims(d, fusion_type, num_experts, pooling='cls'):
    ve = VisualEncoderConfig(output_dim=d['D'], pretrained=False)
    ve.arch = dict(hidden_size=d['D'], intermediate_size=d['vit_inter'], num_hidden_layers=d['vit_layers'],
                   num_attention_heads=d['vit_heads'], image_size=d['image'], patch_size=d['patch'])
    te = TextEncoderConfig(output_dim=d['D'], max_length=d['seq'], pooling_strategy=pooling, pretrained=False)
    te.arch = dict(vocab_size=d['vocab'], hidden_size=d['D'], num_hidden_layers=d['txt_layers'], num_attention_heads=d['txt_heads'],
                   intermediate_size=d['txt_inter'], max_position_embeddings=d['max_pos'], type_vocab_size=1, pad_token_id=1)
    return VQAModelConfig(
        visual_encoder=ve, text_encoder=te,
        fusion=FusionConfig(fusion_type=fusion_type, hidden_dim=d['D'], output_dim=d['D'], num_heads=d['fusion_heads'],
                            num_layers=d['fusion_layers'], dropout=0.1),
        moe=MOEConfig(use_moe=num_experts > 0, num_experts=max(num_experts, 1), top_k=2, hidden_dim=d['moe_hidden']),
        knowledge=KnowledgeConfig(use_knowledge=False),
        answer_head=AnswerHeadConfig(num_answers=d['num_answers'], hidden_dims=list(d['answer_hidden']), dropout=0.3))


def fixture_inputs(arrays, meta):
    """The seeded inputs of a golden fixture: ``det_weights.make_inputs``, or -- batch-32 fixtures -- the rows
    ``arrays['pool_index']`` of ``det_weights.make_input_pool`` (the samples ``oracle/gen_golden.py: select_samples`` kept)."""
    import torch
    from oracle import det_weights as dw
    d = meta['dims']
    if meta.get('pool'):
        px, ids, mask, labels = dw.make_input_pool(meta['pool'], d['seq'], d['image'], vocab_hi=min(30000, d['vocab']),
                                                   num_answers=d['num_answers'], seed=meta['seed'])
        idx = torch.from_numpy(arrays['pool_index'])
        return px[idx], ids[idx], mask[idx], labels[idx]
    return dw.make_inputs(d['batch'], d['seq'], d['image'], vocab_hi=min(30000, d['vocab']), num_answers=d['num_answers'], seed=meta['seed'])


def build_model(meta):
    return VietnameseVQAModel(model_config_from_dims(meta['dims'], meta['fusion_type'], meta['num_experts']))


def collect_from_workers(q, procs, n, timeout=120):
    """Reads ``n`` results from the multiprocessing queue ``q``; fails at once (instead of sitting out the queue timeout in
    silence) when a worker has exited with an error, and after ``timeout`` seconds overall."""
    import queue as _queue
    import time
    out, t0 = [], time.time()
    while len(out) < n:
        try:
            out.append(q.get(timeout=2))
            continue
        except _queue.Empty:
            pass
        dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
        if dead:
            for p in procs:
                if p.is_alive():
                    p.terminate()
            raise AssertionError(f'worker process exited with {dead}')
        if time.time() - t0 > timeout:
            for p in procs:
                if p.is_alive():
                    p.terminate()
            raise AssertionError(f'no result from the workers within {timeout} s')
    return out


def build_generative_model(dims, **overrides):
    """The HIP GenerativeVQAModel with the architecture of a generative fixture (oracle/gen_golden.py: GEN_TINY / GEN_FULL)."""
    from vqa_model_builder_amd.modeling.meta_arch.generative_vqa_model import GenerativeVQAConfig, GenerativeVQAModel
    d = dims
    cfg = GenerativeVQAConfig(
        hidden_size=d['D'], fusion_dim=d['D'], num_decoder_layers=d['gen_layers'], num_attention_heads=d['gen_heads'], decoder_ff_dim=d['gen_ff'],
        max_answer_length=max(d['answer_len'], 8), fusion_num_heads=d['fusion_heads'], fusion_num_layers=d['fusion_layers'], vocab_size=d['gen_vocab'],
        visual_arch=dict(hidden_size=d['D'], intermediate_size=d['vit_inter'], num_hidden_layers=d['vit_layers'], num_attention_heads=d['vit_heads'],
                         image_size=d['image'], patch_size=d['patch']),
        text_arch=dict(vocab_size=d['vocab'], hidden_size=d['D'], num_hidden_layers=d['txt_layers'], num_attention_heads=d['txt_heads'],
                       intermediate_size=d['txt_inter'], max_position_embeddings=d['max_pos'], type_vocab_size=1, pad_token_id=1))
    for k, v in overrides.items():
        setattr(cfg, k, v)
    return GenerativeVQAModel(cfg)
