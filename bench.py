#!/usr/bin/env python3
"""Throughput benchmark of the AutoViVQA forward/backward hot path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one training step of the path over one synthetic batch: forward + backward in
``train()`` mode (dropout on), gradient all-reduce over RCCL when N > 1, ``clip_grad_norm_(1.0)`` and AdamW
(lr 2e-5, wd 0.01).  Inputs are resident in HBM before the timed region.  Workload = BASELINE.json configs[1]:
ViT-B/32 + PhoBERT + CrossAttention fusion, bf16 GEMM operands, batch 32 per GPU (weak scaling).
Rank 0 prints ONE JSON line (metric/value/.../roofline/cpu_baseline).
"""

import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

WORKLOADS = {
    # name: (fusion_type, num_experts, description)
    'cfg2_xattn': ('cross_attention', 0, 'ViT-B/32 + PhoBERT + CrossAttention fusion (BASELINE configs[1])'),
    'cfg1_concat': ('concat', 0, 'ViT-B/32 + PhoBERT + concat fusion (BASELINE configs[0])'),
    'cfg3_mcan_moe4': ('mcan', 4, "ViT-B/32 + PhoBERT + 'mcan'(=add) fusion + MoE 4 experts top-2 (BASELINE configs[2])"),
}
# fwd+bwd algorithmic GFLOP per sample: training = 3 x forward, FLOP = 2 x MAC
GFLOP_PER_SAMPLE = {'cfg2_xattn': 66.7, 'cfg1_concat': 59.5, 'cfg3_mcan_moe4': 59.5 + 1.4}
PEAK_BF16_TFLOPS = 2500.0          # dense MFMA peak, MI355X_MICROARCH.md


def build_model(workload, device):
    from vqa_model_builder_amd.modeling.meta_arch import (AnswerHeadConfig, FusionConfig, KnowledgeConfig, MOEConfig, TextEncoderConfig,
                                                          VietnameseVQAModel, VisualEncoderConfig, VQAModelConfig)
    fusion_type, n_exp, _ = WORKLOADS[workload]
    cfg = VQAModelConfig(
        visual_encoder=VisualEncoderConfig(model_name='openai/clip-vit-base-patch32', pretrained=False, output_dim=768),
        text_encoder=TextEncoderConfig(model_name='vinai/phobert-base', pretrained=False, output_dim=768, max_length=64),
        fusion=FusionConfig(fusion_type=fusion_type, hidden_dim=768, output_dim=768, num_heads=8, num_layers=2, dropout=0.1),
        moe=MOEConfig(use_moe=n_exp > 0, num_experts=max(n_exp, 1), top_k=2, hidden_dim=2048),
        knowledge=KnowledgeConfig(use_knowledge=False),
        answer_head=AnswerHeadConfig(num_answers=3000, hidden_dims=[768, 512], dropout=0.3))
    torch.manual_seed(0)                              # identical replicas on every rank
    model = VietnameseVQAModel(cfg).to(device)
    with torch.no_grad():                             # HF-style random init (no checkpoints offline)
        for n, p in model.named_parameters():
            if p.dim() >= 2:
                p.normal_(0.0, 0.02)
    return model


def make_optimizer(model, use_torch=False, fp16=False):
    """Param groups of the reference loop (training_pipeline.py:239-252): no decay for bias / LayerNorm weights."""
    nd = ('bias', 'LayerNorm.weight', 'layer_norm.weight')
    decay = [p for n, p in model.named_parameters() if p.requires_grad and not any(t in n for t in nd)]
    no_decay = [p for n, p in model.named_parameters() if p.requires_grad and any(t in n for t in nd)]
    from vqa_model_builder_amd.optim import FusedAdamW
    groups = [{'params': decay, 'weight_decay': 0.01}, {'params': no_decay, 'weight_decay': 0.0}]
    if use_torch:
        return torch.optim.AdamW(groups, lr=2e-5, betas=(0.9, 0.999), eps=1e-8, fused=True)
    # fp16 operands: dynamic loss scale on the device (GradScaler's policy: reference training_pipeline.py:346-347,466-502)
    return FusedAdamW(groups, lr=2e-5, betas=(0.9, 0.999), eps=1e-8, max_grad_norm=1.0,
                      loss_scale='dynamic' if fp16 else None).attach_shadows(model)


def synthetic_batch(B, device, rank):
    g = torch.Generator(device=device).manual_seed(1234 + rank)
    px = torch.randn((B, 3, 224, 224), generator=g, device=device)
    ids = torch.randint(0, 30000, (B, 64), generator=g, device=device)
    mask = torch.ones((B, 64), dtype=torch.int64, device=device)
    labels = torch.randint(0, 3000, (B,), generator=g, device=device)
    return px, ids, mask, labels


def usable_cores():
    """Cores this process may actually use: min(cpu_count, affinity mask, cgroup CPU quota)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            txt = open(path).read().split()
            if path.endswith('cpu.max'):
                if txt[0] != 'max':
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, q // int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())))
        except Exception:
            pass
    return max(1, n)


def cpu_baseline(workload, steps=6, B=32, budget_s=30.0):
    """The CPU oracle (plain fp32 torch on the host cores) running the same step definition on a bounded sample."""
    from oracle import det_weights as dw
    from oracle import vqa_oracle as vo
    from tests.conftest import CfgView, load_golden
    tag = {'cfg2_xattn': 'full_cfg2_xattn', 'cfg1_concat': 'full_cfg1_concat', 'cfg3_mcan_moe4': 'full_cfg3_mcan_moe4'}[workload]
    _, meta = load_golden(tag)
    cores = usable_cores()
    torch.set_num_threads(cores)
    print(f'[bench] cpu_baseline: oracle on {cores} host threads, batch {B} ...', file=sys.stderr, flush=True)
    sd = dw.make_state_dict({k: tuple(v) for k, v in meta['shapes'].items()}, 1)
    leaves = {k: v.requires_grad_(v.is_floating_point() and v.dim() > 0) for k, v in sd.items()}
    params = [v for v in leaves.values() if v.requires_grad]
    opt = torch.optim.AdamW(params, lr=2e-5, weight_decay=0.01)
    cfg = CfgView(meta)
    px, ids, mask, labels = dw.make_inputs(B, 64, 224, seed=3, pad_rows=False)
    times = []
    t_start = time.perf_counter()
    for i in range(steps + 1):
        if i >= 2 and time.perf_counter() - t_start > budget_s:
            break
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        _, loss, _ = vo.vqa_forward(leaves, cfg, px, ids, mask, labels)
        loss.backward()
        torch.nn.utils.clip_grad_norm_([p for p in params if p.grad is not None], 1.0)
        opt.step()
        times.append(time.perf_counter() - t0)
        print(f'[bench] cpu_baseline step {i}: {times[-1]:.2f} s', file=sys.stderr, flush=True)
    steps = len(times) - 1
    t = sorted(times[1:])[len(times[1:]) // 2]
    model_name = ''
    try:
        model_name = [l.split(':', 1)[1].strip() for l in open('/proc/cpuinfo') if l.startswith('model name')][0]
    except Exception:
        pass
    return {'value': round(B / t, 3), 'unit': 'samples/s', 'cores': cores, 'kind': 'port',
            'sample': f'{steps} timed steps (median) of batch {B}, fp32, eval-mode dropout-free oracle, fwd+bwd+clip+AdamW; {model_name}'}


def run_workload(workload, args, device, world, rank, dist, want_roofline):
    """Builds the model of ``workload``, captures / times the training step and (optionally) measures the GEMM roofline.
    Returns a dict; frees everything it allocated."""
    import gc
    from vqa_model_builder_amd.hip import blocks as _blocks, kernels as K, lib
    from vqa_model_builder_amd.dp import GradReducer
    fp16 = args.dtype == 'fp16'
    model = build_model(workload, device).train()
    opt = make_optimizer(model, args.torch_optimizer, fp16)
    params = [p for p in model.parameters() if p.requires_grad]
    n_params = sum(p.numel() for p in params)
    # the fused optimiser applies 1/world itself (grad_prescale): the all-reduced SUM is never rescaled in memory
    dist_on = world > 1 or args.force_dist           # a process group exists (N > 1, or the one-rank RCCL run of --force-dist)
    reducer = (GradReducer(params, average=args.torch_optimizer, grad_dtype=args.grad_dtype, force_collectives=args.force_dist)
               if (dist_on or getattr(args, 'force_segmented', False)) else None)
    if world > 1 and not args.torch_optimizer:
        opt.grad_prescale = 1.0 / world
    px, ids, mask, labels = synthetic_batch(args.batch, device, rank)

    def eager_step():
        opt.zero_grad(set_to_none=True)
        out = model(pixel_values=px, input_ids=ids, attention_mask=mask, labels=labels)
        (opt.scale_loss(out.loss) if hasattr(opt, 'scale_loss') else out.loss).backward()
        if reducer is not None:
            reducer.finalize()
        if args.torch_optimizer:
            torch.nn.utils.clip_grad_norm_([p for p in params if p.grad is not None], 1.0)
        opt.step()                                   # FusedAdamW: global-norm clip (1.0) + AdamW + 16-bit shadow refresh, fused
        return out.loss

    # Single GPU: the whole step -- forward, backward, clip, AdamW -- is ONE captured HIP graph with the two encoders as parallel
    # branches; every replay copies a batch into the static input buffers, draws fresh dropout masks (device-side RNG epoch)
    # and advances the optimiser's device-side step count (and, fp16, its loss scale).
    # N > 1: see vqa_model_builder_amd/graph.py (backward split into per-block graphs, each block's gradient arena all-reduced over
    # RCCL on a side stream while the remaining backward replays; optimiser graph last).  Should the capture fail beside an
    # initialised process group (the decision is all-or-nothing across ranks) every rank falls back to the eager step with the
    # gradient exchange overlapped into backward by hooks.
    use_graph = not args.eager and not args.torch_optimizer
    step, launch, graphed = eager_step, 'eager', None
    if use_graph:
        from vqa_model_builder_amd.graph import GraphedTrainStep
        batch = dict(pixel_values=px, input_ids=ids, attention_mask=mask, labels=labels)
        ok = 1
        try:
            graphed = GraphedTrainStep(model, opt, batch, warmup=3, reducer=reducer, segmented=True if getattr(args, 'force_segmented', False) else None,
                                       moe_branches=1 if args.moe_branches is None else args.moe_branches, dp_split=args.dp_split, dp_segments=args.dp_segments,
                                       exchange_on_side_stream=not args.exchange_inline, wire_optimizer=not args.no_wire_optimizer,
                                       capture_error_mode='thread_local' if dist_on else 'global')
        except Exception as e:                       # noqa: BLE001 -- any capture failure means: run eagerly
            ok = 0
            print(f'[bench] rank {rank}: HIP-graph capture failed ({type(e).__name__}: {e}); eager step', file=sys.stderr, flush=True)
        if dist_on:
            flag = torch.tensor([ok], device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = int(flag.item())
        if ok:
            launch = graphed.describe() if hasattr(graphed, 'describe') else 'hip-graph'

            def step():
                return graphed(batch)
    if launch == 'eager' and reducer is not None:
        reducer.attach()                             # overlap: buckets go on the wire during backward

    def fence():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([dt], device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms = dt / args.steps * 1e3
    res = {'name': workload, 'workload': WORKLOADS[workload][2], 'ms_per_step': round(ms, 3), 'value': round(args.batch * world * args.steps / dt, 2),
           'launch': launch, 'final_loss': round(float(loss), 4), 'params': n_params}
    if fp16 and hasattr(opt, 'loss_scale'):
        res['loss_scale'] = opt.loss_scale
    if reducer is not None:
        comm = graphed.comm_stats() if (graphed is not None and hasattr(graphed, 'comm_stats')) else {}
        ranks = torch.ones(1, device=device)
        if dist_on:
            dist.all_reduce(ranks)
        res.update({'ranks_seen': int(ranks.item()), 'allreduce_bytes': reducer.bytes_per_step(), 'grad_dtype': args.grad_dtype, **comm})

    if want_roofline:
        # Instrumented re-run of the same step, one launch chain (each GEMM's own duration, not its share of an overlap).  Every
        # launch of the MFMA GEMM family is issued through hipExtLaunchKernel with a start / stop event pair: the events carry the
        # kernel's own begin / end timestamps from its dispatch packet -- the quantity rocprofv3 --kernel-trace reports -- with no
        # host or event-record overhead inside the bracket.  Launches inside the fusion block (forward and backward) are tagged.
        model.parallel_towers = False
        L = lib.load()

        def _tag(v):
            def hook(*_a):
                L.vqa_gemm_profile(1, v)
            return hook
        n_prof = 3
        eager_step()                                  # settle allocator / shadows outside the graph (not recorded)
        hooks = [model.fusion.register_forward_pre_hook(_tag(1)), model.fusion.register_forward_hook(_tag(0)),
                 model.fusion.register_full_backward_pre_hook(_tag(1)), model.fusion.register_full_backward_hook(_tag(0))]
        L.vqa_gemm_profile(1, 0)
        for _ in range(n_prof):
            eager_step()
        torch.cuda.synchronize()
        import ctypes as C
        flop, msec, cnt, byt = (C.c_double * 2)(), (C.c_double * 2)(), (C.c_int * 2)(), (C.c_double * 2)()
        L.vqa_gemm_profile_collect2(2, flop, msec, cnt, byt)
        L.vqa_gemm_profile(0, 0)
        for h in hooks:
            h.remove()
        tot_flop, tot_ms, launches = (flop[0] + flop[1]) / n_prof, (msec[0] + msec[1]) / n_prof, (cnt[0] + cnt[1]) // n_prof
        fus_flop, fus_ms = flop[1] / n_prof, msec[1] / n_prof
        ach = tot_flop / (tot_ms * 1e-3) / 1e12 if tot_ms > 0 else 0.0
        fus = fus_flop / (fus_ms * 1e-3) / 1e12 if fus_ms > 0 else None
        res['gemm'] = {'flop_per_step': tot_flop, 'ms_per_step': tot_ms, 'launches_per_step': launches, 'tflops': ach,
                       'alg_bytes_per_step': (byt[0] + byt[1]) / n_prof,
                       'fusion_flop_per_step': fus_flop, 'fusion_ms_per_step': fus_ms, 'fusion_tflops': fus}
    # free the workload (the next one builds its own model / graph pools)
    del step, eager_step
    graphed = opt = model = reducer = None
    _blocks.disable_indirect_seeds()
    gc.collect()
    torch.cuda.empty_cache()
    return res


def run_generative(args, device):
    """SURVEY section 8f rank 3 (the reference's generative model: ViT-B/32 + PhoBERT -> 2 pre-LN fusion layers over the 114 visual + question
    tokens -> 6-layer decoder, tied 64 000-way head, label-smoothed CE), batch per GPU as the main workload, answers of 32 tokens,
    teacher-forced training step captured as one HIP graph.  One GPU; reported as the ``generative_config`` object of the line."""
    import ctypes as C
    import gc
    from vqa_model_builder_amd.graph import GraphedTrainStep
    from vqa_model_builder_amd.hip import blocks as _blocks, lib
    from vqa_model_builder_amd.modeling.meta_arch.generative_vqa_model import GenerativeVQAConfig, GenerativeVQAModel
    from vqa_model_builder_amd.optim import FusedAdamW
    B, A = args.batch, 32
    torch.manual_seed(0)
    cfg = GenerativeVQAConfig(visual_arch=dict(hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12, image_size=224, patch_size=32),
                              text_arch=dict(vocab_size=64001, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                                             max_position_embeddings=258, type_vocab_size=1, pad_token_id=1))
    model = GenerativeVQAModel(cfg).to(device).train()
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() >= 2:
                p.normal_(0.0, 0.02)
    nd = ('bias', 'LayerNorm.weight', 'layer_norm.weight', 'norm')
    named = list(model.named_parameters())
    groups = [{'params': [p for n, p in named if not any(t in n for t in nd)], 'weight_decay': 0.01},
              {'params': [p for n, p in named if any(t in n for t in nd)], 'weight_decay': 0.0}]
    opt = FusedAdamW(groups, lr=2e-5, betas=(0.9, 0.999), eps=1e-8, max_grad_norm=1.0, loss_scale='dynamic' if args.dtype == 'fp16' else None).attach_shadows(model)
    g = torch.Generator(device=device).manual_seed(4321)
    batch = dict(pixel_values=torch.randn((B, 3, 224, 224), generator=g, device=device), input_ids=torch.randint(3, 30000, (B, 64), generator=g, device=device),
                 attention_mask=torch.ones((B, 64), dtype=torch.int64, device=device),
                 decoder_input_ids=torch.randint(3, 64000, (B, A), generator=g, device=device),
                 decoder_attention_mask=torch.ones((B, A), dtype=torch.int64, device=device), labels=torch.randint(3, 64000, (B, A), generator=g, device=device))
    n_params = sum(p.numel() for p in model.parameters())

    def eager_step():
        opt.zero_grad(set_to_none=True)
        out = model(**batch)
        (opt.scale_loss(out.loss) if hasattr(opt, 'scale_loss') else out.loss).backward()
        opt.step()
        return out.loss
    graphed = GraphedTrainStep(model, opt, batch, warmup=3)
    for _ in range(max(5, args.warmup // 2)):
        graphed(batch)
    torch.cuda.synchronize()
    steps = max(10, args.steps // 2)
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = graphed(batch)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    res = {'name': 'generative_vit_phobert', 'workload': 'GenerativeVQAModel: ViT-B/32 + PhoBERT + 2 fusion layers (114 tokens) + 6-layer decoder + tied 64000-way head',
           'batch_per_gpu': B, 'answer_tokens': A, 'ms_per_step': round(ms, 3), 'value': round(B / ms * 1e3, 2), 'unit': 'samples/s', 'steps': steps,
           'launch': 'hip-graph (one graph)', 'final_loss': round(float(loss), 4), 'params': n_params}
    if not args.no_roofline:
        model.parallel_towers = False
        L = lib.load()
        eager_step()
        L.vqa_gemm_profile(1, 0)
        for _ in range(2):
            eager_step()
        torch.cuda.synchronize()
        f, m, n = (C.c_double * 1)(), (C.c_double * 1)(), (C.c_int * 1)()
        L.vqa_gemm_profile_collect(1, f, m, n)
        L.vqa_gemm_profile(0, 0)
        tf = f[0] / m[0] / 1e9 if m[0] > 0 else 0.0
        res['roofline'] = {'bound': 'mfma', 'kernel': 'gemm_v1_kernel / gemm_v1_grouped_kernel (all MFMA GEMM launches of a step)', 'achieved': round(tf, 2),
                           'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(tf / PEAK_BF16_TFLOPS, 4), 'gemm_gflop_per_step': round(f[0] / 2 / 1e9, 1),
                           'gemm_ms_per_step': round(m[0] / 2, 3), 'launches_per_step': n[0] // 2, 'traffic': None}
    graphed = opt = model = None
    _blocks.disable_indirect_seeds()
    gc.collect()
    torch.cuda.empty_cache()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--workload', default='cfg2_xattn', choices=list(WORKLOADS))
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp16'],
                    help="16-bit type of the GEMM / attention operands: bf16 (default) or fp16 (+ dynamic loss scale), the reference loop's autocast dtype")
    ap.add_argument('--no-second-workload', action='store_true',
                    help='skip the MoE config (BASELINE configs[2]) that is otherwise timed too and reported as the "moe_config" object of the line')
    ap.add_argument('--grad-dtype', default=None, choices=['fp32', 'bf16'],
                    help='wire format of the data-parallel gradient exchange (default: bf16 buckets when N > 1 -- summed in bf16 on the wire, applied to fp32 '
                         'master weights by the fp32 AdamW; fp32: what torch DDP sends)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--gemm-ws', type=int, default=None, help='diagnostics: vqa_set_gemm_ws mode (0 = legacy tiles only, 1 = auto)')
    ap.add_argument('--fused-attn', type=int, default=None, help='diagnostics: bit 0 = fused in-projection + attention in the fusion block, bit 1 = in the encoders (default: both)')
    ap.add_argument('--tail-runner', type=int, default=None, help='diagnostics: 0 = projection / norm / dropout / answer head as op-by-op chain instead of one node')
    ap.add_argument('--expert-runners', type=int, default=None, help='diagnostics: 0 = MoE experts as op-by-op chains (hip/ops.py) instead of the runners')
    ap.add_argument('--moe-branches', type=int, default=None, help='diagnostics: MoE experts on side streams in the captured step (0 off, 1 specialised experts, 2 every expert)')
    ap.add_argument('--torch-optimizer', action='store_true', help='clip_grad_norm_ + torch fused AdamW instead of the HIP FusedAdamW')
    ap.add_argument('--eager', action='store_true', help='launch every kernel from the host instead of replaying the captured HIP graph of the step')
    ap.add_argument('--dp-segments', default='tapered', choices=['tapered', 'even'], help="depth split of the encoders' backward: 'tapered' = a one-layer last segment (its exchange is the exposed one), 'even' = equal segments (round 2)")
    ap.add_argument('--dp-split', default='depth', choices=['depth', 'towers'],
                    help="how the data-parallel captured step cuts the encoders' backward: 'depth' = up to four depth segments, text and vision layers of a segment as "
                         "parallel branches of one graph (default); 'towers' = text backward then vision backward, each in two graphs")
    ap.add_argument('--no-wire-optimizer', action='store_true', help='diagnostics: copy the all-reduced bf16 sums back into the fp32 gradient arenas before the optimiser (first form of round 2)')
    ap.add_argument('--ln-bwd-blocks', type=int, default=None, help='diagnostics: vqa_set_layernorm_bwd_blocks (workgroups of the LayerNorm backward kernel)')
    ap.add_argument('--gemm-k-rotate', type=int, default=None, help='diagnostics: 0 = no per-XCD k rotation of the ring GEMMs in train() mode (kernels.TRAIN_K_ROTATE; default: on in train mode, off in eval)')
    ap.add_argument('--gemm-tile-order', type=int, default=None, help='diagnostics: vqa_set_gemm_tile_order (2 = column-major tile ids: measured slower in the step)')
    ap.add_argument('--prefetch-wgs', type=int, default=None, help='diagnostics: workgroups of a self-paced weight-prefetch kernel (kernels.PREFETCH_WORKGROUPS); -1 = event-gated prefetch')
    ap.add_argument('--no-weight-prefetch', action='store_true', help='(default) no prefetch of the next layer\'s weights on a side stream (kernels.WEIGHT_PREFETCH)')
    ap.add_argument('--exchange-inline', action='store_true', help='diagnostics: wire-format copies of the gradient exchange on the compute stream (first form of round 2)')
    ap.add_argument('--force-dist', action='store_true',
                    help='diagnostics on a one-GPU box: initialise a ONE-rank RCCL process group and run the N > 1 code path (segmented step, every '
                         'all-reduce / all-gather really issued) -- exercises RCCL, not its performance')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    import torch.distributed as dist
    # rehearsal on a one-GPU box: VQA_BENCH_REHEARSE=1 runs all ranks on cuda:0 over gloo (RCCL refuses two ranks per device);
    # it exercises the multi-process code path of this file, not its performance
    rehearse = os.environ.get('VQA_BENCH_REHEARSE') == '1'
    if rehearse:
        local_rank = 0
    if world > 1 or args.force_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29531')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        torch.cuda.set_device(local_rank)
        if rehearse:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    device = torch.device('cuda', local_rank)
    torch.cuda.set_device(device)

    import vqa_model_builder_amd as vqa
    from vqa_model_builder_amd.hip import lib
    try:
        vqa.set_compute_dtype(args.dtype)
    except lib.HipLibraryMissing:
        if rank == 0:
            import __graft_entry__
            __graft_entry__.build()
        if world > 1:
            dist.barrier()
        vqa.set_compute_dtype(args.dtype)

    if args.no_weight_prefetch or args.prefetch_wgs is not None:
        from vqa_model_builder_amd.hip import kernels as _K0
        _K0.WEIGHT_PREFETCH = args.prefetch_wgs is not None
        if args.prefetch_wgs is not None:
            _K0.PREFETCH_GATED = args.prefetch_wgs < 0
            _K0.PREFETCH_WORKGROUPS = max(1, args.prefetch_wgs)
    if args.ln_bwd_blocks is not None:
        lib.load().vqa_set_layernorm_bwd_blocks(args.ln_bwd_blocks)
    if args.gemm_tile_order is not None:
        lib.load().vqa_set_gemm_tile_order(args.gemm_tile_order)
    if args.gemm_k_rotate is not None:
        from vqa_model_builder_amd.hip import kernels as _K1
        _K1.TRAIN_K_ROTATE = args.gemm_k_rotate if args.gemm_k_rotate == 2 else bool(args.gemm_k_rotate)
    if args.gemm_ws is not None:
        lib.load().vqa_set_gemm_ws(args.gemm_ws)
    if args.fused_attn is not None:
        from vqa_model_builder_amd.hip import kernels as _K
        _K.FUSED_ATTENTION_FUSION, _K.FUSED_ATTENTION_ENCODERS = bool(args.fused_attn & 1), bool(args.fused_attn & 2)
    if args.tail_runner is not None:
        from vqa_model_builder_amd.modeling.meta_arch import vqa_model as _VM
        _VM.TAIL_RUNNER = bool(args.tail_runner)
    if args.expert_runners is not None:
        from vqa_model_builder_amd.modeling.moe import experts as _E
        _E.EXPERT_RUNNERS = bool(args.expert_runners)
    if args.grad_dtype is None:
        args.grad_dtype = 'bf16' if (world > 1 or args.force_dist) else 'fp32'
    main_res = run_workload(args.workload, args, device, world, rank, dist, want_roofline=not args.no_roofline)
    moe_res = None
    if not args.no_second_workload and args.workload != 'cfg3_mcan_moe4':
        try:
            moe_res = run_workload('cfg3_mcan_moe4', args, device, world, rank, dist, want_roofline=False)
        except Exception as e:                       # noqa: BLE001 -- the second object is a report, never a reason to lose the main number
            moe_res = {'name': 'cfg3_mcan_moe4', 'error': f'{type(e).__name__}: {e}'}

    roofline = None
    if 'gemm' in main_res:
        g = main_res.pop('gemm')
        # HBM-side bytes per GEMM launch: PMC counters cannot be read from inside this process; they are collected with
        # rocprofv3 on this same command (separate FETCH_SIZE / WRITE_SIZE passes, gfx950 correction applied) by
        # profiles/pmc_traffic.py and committed with their raw counter CSVs
        traffic, traffic_src = None, None
        for rnd in ('r03', 'r02', 'r01'):
            try:
                tj = json.load(open(os.path.join(REPO, 'profiles', rnd, 'gemm_traffic.json')))
                if args.workload == 'cfg2_xattn' and args.batch == 32:
                    traffic = tj['hbm_bytes_per_gemm_launch']
                    traffic_src = f'profiles/{rnd}/gemm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this command)'
                break
            except Exception:
                continue
        ms = main_res['ms_per_step']
        # the same quantity from the committed rocprofv3 --kernel-trace --stats summary of this command (profiles/collect.sh): the live bracket
        # (hipExtLaunchKernel start / stop events) runs ~12 % longer than rocprof's per-dispatch durations -- the start stamp is taken at
        # dispatch, ~2.8 us before the kernel's first wave on a dependent chain -- so the live `frac` is the conservative one.
        # Numerator and denominator cover the SAME launches: every row of the MFMA GEMM family the live bracket counts FLOPs for
        # (gemm_v1* / gemm_kernel / gemm_ws ring and register-staged GEMMs, the grouped launches, the fused in-projection + attention
        # and cross-attention block kernels) -- recomputable by hand from the committed CSV.
        rocprof = None
        FAMILY = ('gemm_v1', 'gemm_kernel', 'gemm_ws', 'gemm_grp', 'gemm_dw256', 'fused_inproj_attn_kernel', 'xattn_block')
        for rnd in ('r03', 'r02'):
            try:
                import csv
                path = os.path.join(REPO, 'profiles', rnd, 'eager_kernel_stats.csv')
                rows = list(csv.DictReader(open(path)))
                nsteps = sum(int(r['Calls']) for r in rows if 'adamw_multi_kernel' in r['Name']) / 2.0
                fam = [r for r in rows if any(f in r['Name'] for f in FAMILY)]
                gms = sum(int(r['TotalDurationNs']) for r in fam) / 1e6 / nsteps
                if args.workload == 'cfg2_xattn' and args.batch == 32 and gms > 0:
                    rocprof = {'gemm_ms_per_step': round(gms, 3), 'tflops': round(g['flop_per_step'] / (gms * 1e-3) / 1e12, 1),
                               'frac': round(g['flop_per_step'] / (gms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                               'launches_per_step': round(sum(int(r['Calls']) for r in fam) / nsteps, 1), 'profiled_steps': nsteps,
                               'rows': 'Name contains one of ' + ' | '.join(FAMILY),
                               'source': f'profiles/{rnd}/eager_kernel_stats.csv (rocprofv3 --kernel-trace --stats of bench.py --eager, committed; not measured in this run): '
                                         'frac = flop_per_step of this run / (sum of TotalDurationNs of those rows / profiled_steps) / peak'}
                break
            except Exception:
                rocprof = None
        roofline = {'bound': 'mfma', 'kernel': 'gemm_v1_kernel / gemm_v1_grouped_kernel (all MFMA GEMM launches of a step)',
                    'achieved': round(g['tflops'], 2), 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(g['tflops'] / PEAK_BF16_TFLOPS, 4),
                    'timing': 'kernel begin/end timestamps of every GEMM dispatch (hipExtLaunchKernel start/stop events) = what rocprofv3 --kernel-trace reports',
                    'traffic': traffic, 'traffic_unit': 'B/launch (HBM side)', 'traffic_source': traffic_src,
                    # what the launches must move at least: both operands once + every output / fused epilogue stream once, per launch like `traffic`
                    'algorithmic_bytes': round(g['alg_bytes_per_step'] / max(1, g['launches_per_step'])), 'algorithmic_bytes_per_step': round(g['alg_bytes_per_step']),
                    'launches_per_step': g['launches_per_step'], 'gemm_ms_per_step': round(g['ms_per_step'], 3),
                    'gemm_gflop_per_step': round(g['flop_per_step'] / 1e9, 1),
                    'fusion_gemm_tflops': None if g['fusion_tflops'] is None else round(g['fusion_tflops'], 2),
                    'fusion_mfma_util': None if g['fusion_tflops'] is None else round(g['fusion_tflops'] / PEAK_BF16_TFLOPS, 4),
                    'fusion_gemm_ms_per_step': round(g['fusion_ms_per_step'], 3),
                    # launched FLOPs (the dead rows of the last fusion layer are not executed and not credited) over the whole step
                    'whole_step_tflops': round(g['flop_per_step'] / (ms * 1e-3) / 1e12, 2), 'rocprof': rocprof}
    if moe_res is not None and 'ms_per_step' in moe_res:
        # the MoE config is bound by HBM bytes that do not depend on the batch: 16-bit weights read forward and backward (2 x 2 B),
        # fp32 gradients written (4 B), AdamW + shadow refresh (30 B) per parameter -- SURVEY section 8(d)
        P = moe_res['params']
        byts = (2 * 2 + 4 + 30) * P
        gbs = byts / (moe_res['ms_per_step'] * 1e-3) / 1e9
        moe_res['roofline'] = {'bound': 'hbm', 'achieved': round(gbs, 1), 'peak': 8000.0, 'unit': 'GB/s', 'frac': round(gbs / 8000.0, 4),
                               'bytes_per_step': byts, 'traffic': None,
                               'note': 'whole step against the batch-independent parameter traffic (4 + 4 + 30 B/param); activations excluded'}

    # Data-parallel readiness, measured on ONE GPU: the segmented step (what N > 1 runs: one graph per segment) without an exchange gives the
    # compute side of an N-GPU step; each segment's exchange is modelled from its byte count on the exchange stream.  Run for the main
    # workload AND for the MoE config (BASELINE configs[3] = cfg3 x 8 GPUs is what north_star's >= 6x is defined on: 477 M parameters, the
    # MoE's 252 M of them produced first, in segment H), at three assumed bus bandwidths (no 8-GPU node has been measured yet).
    dp_model = None
    if world == 1 and not args.no_second_workload and not args.eager and not args.torch_optimizer:
        def model_for(workload, one_gpu_ms):
            args.force_segmented = True
            try:
                seg = run_workload(workload, args, device, world, rank, dist, want_roofline=False)
            finally:
                args.force_segmented = False
            sb, sm, sg = seg.get('segment_bytes', {}), seg.get('segment_ms', {}), seg.get('segment_gather_bytes_per_rank', {})
            total = seg['ms_per_step']
            t_opt = max(0.0, total - sum(sm.values()))
            blocks = [k for k in sm if k != 'F']                 # in replay order: H, B1 .. B4

            def predict(wire, bus):
                # one RCCL stream: a segment's exchange starts when its graph has finished AND the previous exchange is through
                ready, end = sm.get('F', 0.0), 0.0
                for k in blocks:
                    ready += sm.get(k, 0.0)
                    dense = sb.get(k, 0) - sg.get(k, 0)              # measured at world 1: the gather part counted once
                    pack = (dense / 4) * 6 / 5.0e12 * 1e3 if wire < 1.0 else 0.0          # bf16 buckets: fp32 -> bf16 staging copy (4 B read + 2 B written per element, ~5 TB/s), on the exchange stream
                    end = max(ready, end) + pack + (2 * (7 / 8) * dense * wire + 7 * sg.get(k, 0)) / bus * 1e3      # ring all-reduce + all-gather of 8 ranks' rows
                exposed = max(0.0, end - ready)
                step = ready + exposed + t_opt
                return {'exposed_allreduce_ms': round(exposed, 3), 'ms_per_step': round(step, 3), 'scaling_vs_1gpu': round(8 * one_gpu_ms / step, 2)}
            return {'one_graph_step_ms_1gpu': one_gpu_ms, 'segmented_step_ms_1gpu': total, 'segment_ms_1gpu': sm, 'optimizer_ms': round(t_opt, 3), 'segment_bytes_fp32': sb,
                    'predicted_8gpu': {f'{int(bus / 1e9)}GBps': {'fp32_buckets': predict(1.0, bus), 'bf16_buckets': predict(0.5, bus)} for bus in (150e9, 300e9, 450e9)}}
        try:
            dp_model = {'note': 'compute = the graphs of the segmented step measured on one GPU (no exchange); exchange = per segment, on the exchange stream: (bf16 buckets) the '
                                'fp32 -> bf16 staging copy, then the ring all-reduce, started when the segment\'s graph is done and the previous segment\'s exchange is through; the '
                                'optimiser reads the bf16 sums in place.  The xGMI all-reduce bus bandwidth is an ASSUMPTION (three values; 7 links x ~153 GB/s per GPU is the '
                                'hardware ceiling), not a measurement -- the driver\'s N = 8 run is the measurement.  RCCL itself has run with one rank (bench.py --force-dist, '
                                'profiles/r03/rccl_one_rank_*.log; tests/test_dp_gpu.py).',
                        args.workload: model_for(args.workload, main_res['ms_per_step'])}
            if moe_res is not None and 'ms_per_step' in moe_res and args.workload != 'cfg3_mcan_moe4':
                dp_model['cfg3_mcan_moe4'] = model_for('cfg3_mcan_moe4', moe_res['ms_per_step'])
        except Exception as e:                       # noqa: BLE001
            args.force_segmented = False
            dp_model = {'error': f'{type(e).__name__}: {e}'}

    gen_res = None
    if world == 1 and not args.no_second_workload and not args.eager and not args.torch_optimizer:
        try:
            gen_res = run_generative(args, device)
        except Exception as e:                       # noqa: BLE001 -- a report, never a reason to lose the main number
            gen_res = {'name': 'generative_vit_phobert', 'error': f'{type(e).__name__}: {e}'}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline(args.workload)
        except Exception as e:                       # the baseline is a report, never a reason to lose the GPU number
            cpu = {'value': None, 'unit': 'samples/s', 'cores': os.cpu_count(), 'kind': 'port', 'sample': f'failed: {type(e).__name__}: {e}'}

    if rank == 0:
        line = {
            'metric': 'train samples/sec (img+question)', 'value': main_res['value'], 'unit': 'samples/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': main_res['ms_per_step'], 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': main_res['workload'], 'name': args.workload, 'batch_per_gpu': args.batch,
                       'global_batch': args.batch * world, 'image': '3x224x224', 'seq_len': 64,
                       'step': 'fwd+bwd(train mode, dropout on)+allreduce+clip_grad_norm(1.0)+AdamW', 'parallelism': f'dp{world}',
                       'launch': main_res['launch'], 'final_loss': main_res['final_loss'],
                       **{k: main_res[k] for k in ('loss_scale', 'ranks_seen', 'allreduce_bytes', 'grad_dtype', 'exposed_comm_ms', 'segment_bytes', 'segment_ms', 'segment_gather_bytes_per_rank') if k in main_res}},
            'roofline': roofline, 'cpu_baseline': cpu, 'moe_config': moe_res, 'dp_model': dp_model, 'generative_config': gen_res,
        }
        print(json.dumps(line), flush=True)
    if world > 1 or args.force_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
