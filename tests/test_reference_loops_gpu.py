"""The reference's two training-loop bodies, restated line for line and run on the REAL HIP model on a MI355X:

  * ``TrainingPipeline._train_epoch`` (reference src/core/training_pipeline.py:440-534) with the optimiser / schedule / AMP set-up of
    :234-347: name-substring parameter groups, ``torch.optim.AdamW``, LambdaLR warm-up (factor 0 at step 0) + cosine, ``GradScaler()``,
    ``autocast(device_type='cuda', dtype=torch.float16)``, ``scaler.unscale_`` -> ``clip_grad_norm_(model.parameters())`` -> ``scaler.step`` /
    ``update`` -> ``zero_grad`` -> ``scheduler.step``, gradient accumulation, the two ``.item()`` reads per step;
  * ``VQATrainer.train_step`` (reference src/pipeline/trainer/vqa_trainer.py:746-823) with its bf16 autocast + enabled ``GradScaler``
    (:575-590), parameter groups by its own keyword list (training_utils.py:102-122), accumulation 2, ``clip_gradients``.

What must hold for the drop-in claim of INTEGRATION.md ("autocast around the call is harmless"): no dtype error anywhere under either autocast
context, every gradient an ordinary dense fp32 tensor on the parameter's device (``unscale_`` / ``clip_grad_norm_`` / AdamW read and write
them in place), a finite loss-scale that does not collapse (steps are taken, not skipped), and a loss that goes down on a fixed batch.
Run with the library's operands in bf16 (default) AND fp16 (``set_compute_dtype('fp16')``: what the fp16 loop intends)."""

import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import det_weights as dw  # noqa: E402
from oracle.gen_golden import TINY  # noqa: E402
from tests.helpers import build_model  # noqa: E402

DEV = 'cuda'


def _model_and_batch(fusion_type='cross_attention', num_experts=4, seed=9):
    meta = {'dims': TINY, 'fusion_type': fusion_type, 'num_experts': num_experts}
    model = build_model(meta)
    model.load_state_dict(dw.make_state_dict(dw.shapes_of(model.state_dict()), seed))
    model = model.to(DEV)
    d = TINY
    px, ids, mask, labels = dw.make_inputs(6, d['seq'], d['image'], vocab_hi=d['vocab'], num_answers=d['num_answers'], seed=seed)
    batch = {'image': px.to(DEV), 'input_ids': ids.to(DEV), 'attention_mask': mask.to(DEV), 'label': labels.to(DEV)}
    return model, batch


def _check_grads(model):
    n = 0
    for name, p in model.named_parameters():
        if p.grad is None:
            continue
        assert p.grad.dtype == torch.float32 and p.grad.device == p.device and p.grad.shape == p.shape and p.grad.layout == torch.strided, name
        n += 1
    assert n > 50
    return n


def _training_pipeline_loop(model, batch, steps, accum, lr=3e-3):
    """training_pipeline.py:234-347 (set-up) and :440-534 (loop body), one fixed batch standing for the loader."""
    from torch.amp import GradScaler, autocast
    no_decay = ['bias', 'LayerNorm.weight', 'layer_norm.weight']
    groups = [{'params': [p for n, p in model.named_parameters() if not any(nd in n for nd in no_decay) and p.requires_grad], 'weight_decay': 0.01},
              {'params': [p for n, p in model.named_parameters() if any(nd in n for nd in no_decay) and p.requires_grad], 'weight_decay': 0.0}]
    optimizer = torch.optim.AdamW(groups, lr=lr, betas=(0.9, 0.999), weight_decay=0.01)
    total_steps = steps // accum
    warmup_steps = max(1, int(total_steps * 0.1))

    def lr_lambda(current_step):
        if current_step < warmup_steps:
            return float(current_step) / float(max(1, warmup_steps))
        progress = float(current_step - warmup_steps) / float(max(1, total_steps - warmup_steps))
        return max(0.0, 0.5 * (1.0 + torch.cos(torch.tensor(progress * 3.14159)).item()))
    scheduler = torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda)
    scaler = GradScaler()
    model.train()
    losses, skipped = [], 0
    for step in range(steps):
        with autocast(device_type='cuda', dtype=torch.float16, enabled=True):
            outputs = model(pixel_values=batch['image'], input_ids=batch['input_ids'], attention_mask=batch['attention_mask'], labels=batch['label'])
            loss = outputs.loss
        scaled_loss = loss / accum
        scaler.scale(scaled_loss).backward()
        losses.append(loss.item())
        predictions = outputs.logits.argmax(dim=-1)
        _ = (predictions == batch['label']).sum().item()
        if (step + 1) % accum == 0:
            scaler.unscale_(optimizer)
            _check_grads(model)
            torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
            before = scaler.get_scale()
            scaler.step(optimizer)
            scaler.update()
            skipped += scaler.get_scale() < before
            optimizer.zero_grad()
            scheduler.step()
    return losses, skipped, scaler.get_scale()


@pytest.mark.parametrize('half', ['bf16', 'fp16'])
@pytest.mark.parametrize('accum', [1, 2])
def test_training_pipeline_loop_body_trains_the_hip_model(half, accum):
    import vqa_model_builder_amd as vqa
    vqa.set_compute_dtype(half)
    try:
        torch.manual_seed(1)
        model, batch = _model_and_batch()
        steps = 24 * accum
        losses, skipped, scale = _training_pipeline_loop(model, batch, steps, accum)
        assert all(math.isfinite(l) for l in losses), losses
        head, tail = sum(losses[:3 * accum]) / (3 * accum), sum(losses[-3 * accum:]) / (3 * accum)
        print(f'REFLOOP training_pipeline half={half} accum={accum} loss {head:.4f} -> {tail:.4f} skipped_steps={skipped} loss_scale={scale}')
        assert tail < head - 0.15, (head, tail)
        assert skipped <= 2 and math.isfinite(scale) and scale >= 1024.0      # GradScaler found finite gradients: steps were taken
    finally:
        vqa.set_compute_dtype('bf16')


def _vqa_trainer_steps(model, batch, steps, accum=2, lr=3e-3, dtype=torch.bfloat16):
    """vqa_trainer.py:575-590 (scaler), training_utils.py:102-122 (groups), vqa_trainer.py:746-823 (train_step)."""
    kws = ["bias", "LayerNorm", "layernorm", "bn", "BatchNorm"]
    decay = [p for n, p in model.named_parameters() if p.requires_grad and not any(k in n for k in kws)]
    no_decay = [p for n, p in model.named_parameters() if p.requires_grad and any(k in n for k in kws)]
    optimizer = torch.optim.AdamW([{"params": decay, "lr": lr, "weight_decay": 0.01}, {"params": no_decay, "lr": lr, "weight_decay": 0.0}], lr=lr)
    scheduler = torch.optim.lr_scheduler.LambdaLR(optimizer, lambda s: min(1.0, (s + 1) / 3.0))
    scaler = torch.amp.GradScaler('cuda', enabled=torch.cuda.is_bf16_supported())
    model.train()
    losses, count = [], 0
    for _ in range(steps):
        with torch.amp.autocast(device_type='cuda', enabled=scaler is not None, dtype=dtype):
            outputs = model(pixel_values=batch.get("pixel_values", batch.get("image", batch.get("images"))), input_ids=batch.get("input_ids"),
                            attention_mask=batch.get("attention_mask"), labels=batch.get("labels", batch.get("answer_ids", batch.get("label"))))
            loss = torch.nn.functional.cross_entropy(outputs.logits, batch['label'])     # a loss_fn on the logits: they must carry the autograd graph
            if hasattr(outputs, "aux_loss") and outputs.aux_loss is not None:
                loss = loss + 0.01 * outputs.aux_loss
            loss = loss / accum
        scaler.scale(loss).backward()
        with torch.no_grad():
            _ = (outputs.logits.argmax(dim=-1) == batch['label']).float().sum()
        losses.append(loss.item() * accum)
        count += 1
        if count % accum == 0:
            scaler.unscale_(optimizer)
            _check_grads(model)
            params = [p for p in model.parameters() if p.grad is not None]
            total_norm = torch.nn.utils.clip_grad_norm_(params, 1.0, 2.0).item()
            assert math.isfinite(total_norm)
            scaler.step(optimizer)
            scaler.update()
            optimizer.zero_grad()
            scheduler.step()
    return losses


def test_vqa_trainer_train_step_body_trains_the_hip_model_under_bf16_autocast():
    torch.manual_seed(2)
    model, batch = _model_and_batch(num_experts=0)
    losses = _vqa_trainer_steps(model, batch, 48)
    head, tail = sum(losses[:6]) / 6, sum(losses[-6:]) / 6
    print(f'REFLOOP vqa_trainer bf16 accum=2 loss {head:.4f} -> {tail:.4f}')
    assert all(math.isfinite(l) for l in losses) and tail < head - 0.15, (head, tail)


def test_outputs_under_autocast_equal_outputs_without_it():
    """autocast changes nothing inside the HIP path: eval-mode logits / loss / gradients are bit-identical with and without the context."""
    model, batch = _model_and_batch()
    model.eval()
    res = []
    for ctx in (torch.autocast('cuda', dtype=torch.float16), torch.autocast('cuda', dtype=torch.bfloat16), torch.autocast('cuda', enabled=False)):
        model.zero_grad(set_to_none=True)
        with ctx:
            out = model(pixel_values=batch['image'], input_ids=batch['input_ids'], attention_mask=batch['attention_mask'], labels=batch['label'])
        out.loss.backward()
        assert out.logits.dtype == torch.float32 and out.loss.dtype == torch.float32
        res.append((out.logits.detach().clone(), out.loss.detach().clone(), model.answer_head.classifier[0].weight.grad.clone()))
    for a in res[:2]:
        assert all(torch.equal(x, y) for x, y in zip(a, res[2]))
