"""Deterministic, platform-independent weights and inputs for parity tests.  TEST INFRASTRUCTURE ONLY.

Every tensor is drawn from ``numpy.random.Generator(PCG64([seed, crc32(name)]))`` (numpy's
ziggurat normal: integer + IEEE float arithmetic, identical on every machine running this image),
so the build container (where the reference is importable and the golden vectors are produced,
``oracle/gen_golden.py``) and the GPU box (where the reference does not exist) regenerate exactly
the same fp32 state_dict from nothing but the parameter names and shapes.
"""

import zlib
from typing import Dict, Iterable, Tuple

import numpy as np
import torch


def _rng(name: str, seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))


def normal(name: str, shape, seed: int = 0, std: float = 1.0, mean: float = 0.0) -> torch.Tensor:
    n = int(np.prod(shape)) if len(shape) else 1
    a = _rng(name, seed).standard_normal(n, dtype=np.float32)
    if std != 1.0:
        a *= np.float32(std)
    if mean != 0.0:
        a += np.float32(mean)
    return torch.from_numpy(a).reshape(tuple(shape))


def randint(name: str, shape, low: int, high: int, seed: int = 0) -> torch.Tensor:
    a = _rng(name, seed).integers(low, high, size=tuple(shape), dtype=np.int64)
    return torch.from_numpy(a)


_NORM_TAGS = ('norm', 'LayerNorm', 'layrnorm')


def init_tensor(name: str, shape, seed: int = 0) -> torch.Tensor:
    """Init rule by parameter name.  Chosen so that every bias / LN affine / learned token is non-trivial,
    attention scores have O(1) spread and the answer logits have margins far above bf16 noise."""
    shape = tuple(shape)
    leaf = name.rsplit('.', 1)[-1]
    if len(shape) == 0:                                   # usage_count / total_tokens buffers
        return torch.zeros(())
    is_norm = any(t in name for t in _NORM_TAGS) and len(shape) == 1
    if is_norm and leaf == 'weight':
        return normal(name, shape, seed, 0.1, 1.0)
    if is_norm and leaf == 'bias':
        return normal(name, shape, seed, 0.05)
    if leaf in ('bias', 'in_proj_bias'):
        return normal(name, shape, seed, 0.02)
    if leaf in ('class_embedding',):
        return normal(name, shape, seed, 0.5)
    if leaf in ('mask_tokens', 'object_queries'):
        return normal(name, shape, seed, 0.5)
    if 'embedding' in name and len(shape) == 2:           # word / position / token-type tables
        return normal(name, shape, seed, 0.5)
    fan_in = int(np.prod(shape[1:]))
    gain = 1.0
    if name.startswith('answer_head.') and name.endswith(('classifier.6.weight', 'classifier.4.weight')):
        gain = 4.0                                        # widen top-1/top-2 logit gaps (SURVEY §7 hard parts)
    return normal(name, shape, seed, gain / np.sqrt(fan_in))


def make_state_dict(shapes: Dict[str, Tuple[int, ...]], seed: int = 0) -> Dict[str, torch.Tensor]:
    return {k: init_tensor(k, s, seed) for k, s in shapes.items()}


def shapes_of(state_dict: Dict[str, torch.Tensor]) -> Dict[str, Tuple[int, ...]]:
    return {k: tuple(v.shape) for k, v in state_dict.items()}


def make_inputs(batch: int, seq_len: int = 64, image_size: int = 224, vocab_hi: int = 30000,
                num_answers: int = 3000, seed: int = 0, pad_rows: bool = True):
    """Synthetic batch shaped like the reference's dummy inputs (model_pipeline.py:434-439,
    examples/complete_vqa_pipeline.py:184-204), plus the edge cases the parity fixtures force:
    a right-padded row (mask 0 + pad id 1) and a stray ``input_ids == 1`` inside an attended row."""
    px = normal('pixel_values', (batch, 3, image_size, image_size), seed)
    ids = randint('input_ids', (batch, seq_len), 0, vocab_hi, seed)
    ids[ids == 1] = 2
    mask = torch.ones(batch, seq_len, dtype=torch.int64)
    if pad_rows and batch > 1:
        cut = max(2, (seq_len * 5) // 8)
        ids[1, cut:] = 1
        mask[1, cut:] = 0
        ids[0, min(3, seq_len - 1)] = 1                   # attended pad id: shifts RoBERTa position ids
    labels = randint('labels', (batch,), 0, num_answers, seed)
    return px, ids, mask, labels


def make_input_pool(n: int, seq_len: int = 64, image_size: int = 224, vocab_hi: int = 30000, num_answers: int = 3000,
                    seed: int = 0):
    """A pool of ``n`` candidate samples for the batch-32 fixtures: ``oracle/gen_golden.py`` keeps the candidates whose
    top-1/top-2 logit margin in the REFERENCE is far above any 16-bit rounding error (and, with a MoE, whose top-k expert
    choice is not a numerical tie) and records their indices in the fixture; tests rebuild the pool from the seed and take
    the same rows.  Every sixth row is right-padded (mask 0 + pad id 1, varying lengths), another sixth carries a stray
    attended ``input_ids == 1`` (shifts the RoBERTa position ids)."""
    px = normal('pool.pixel_values', (n, 3, image_size, image_size), seed)
    ids = randint('pool.input_ids', (n, seq_len), 0, vocab_hi, seed)
    ids[ids == 1] = 2
    mask = torch.ones(n, seq_len, dtype=torch.int64)
    for r in range(n):
        if r % 6 == 1:
            cut = max(2, (seq_len * (3 + (r // 6) % 4)) // 8)
            ids[r, cut:] = 1
            mask[r, cut:] = 0
        elif r % 6 == 4:
            ids[r, min(3 + r % 5, seq_len - 1)] = 1
    labels = randint('pool.labels', (n,), 0, num_answers, seed)
    return px, ids, mask, labels


def checksum(state_dict: Dict[str, torch.Tensor], names: Iterable[str] = None) -> float:
    """Order-independent fp64 digest used by fixtures to detect generator drift."""
    tot = 0.0
    for k in (names if names is not None else sorted(state_dict)):
        v = state_dict[k].double().flatten()
        tot += float((v * torch.arange(1, v.numel() + 1, dtype=torch.float64).remainder(97.0)).sum())
    return tot
