"""Which (shape, layout, fused-epilogue option set) the ring GEMM is launched with in one eager cfg2 / cfg3 training step, and how often:
the input to choosing which epilogue forms deserve a compile-time specialisation.   usage: gemm_census.py [workload]"""
import sys, collections, torch
sys.path.insert(0, '.')
import bench
from vqa_model_builder_amd.hip import kernels as K

wl = sys.argv[1] if len(sys.argv) > 1 else 'cfg2_xattn'
dev = torch.device('cuda:0')
if wl == 'generative':
    from vqa_model_builder_amd.modeling.meta_arch.generative_vqa_model import GenerativeVQAConfig, GenerativeVQAModel
    from vqa_model_builder_amd.optim import FusedAdamW
    cfg = GenerativeVQAConfig(visual_arch=dict(hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12, image_size=224, patch_size=32),
                              text_arch=dict(vocab_size=64001, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                                             max_position_embeddings=258, type_vocab_size=1, pad_token_id=1))
    model = GenerativeVQAModel(cfg).to(dev).train()
    opt = FusedAdamW([{'params': list(model.parameters())}], lr=2e-5, max_grad_norm=1.0).attach_shadows(model)
    g = torch.Generator(device=dev).manual_seed(4321)
    kwargs = dict(pixel_values=torch.randn((32, 3, 224, 224), generator=g, device=dev), input_ids=torch.randint(3, 30000, (32, 64), generator=g, device=dev),
                  attention_mask=torch.ones((32, 64), dtype=torch.int64, device=dev), decoder_input_ids=torch.randint(3, 64000, (32, 32), generator=g, device=dev),
                  decoder_attention_mask=torch.ones((32, 32), dtype=torch.int64, device=dev), labels=torch.randint(3, 64000, (32, 32), generator=g, device=dev))
else:
    model = bench.build_model(wl, dev).train()
    opt = bench.make_optimizer(model)
    px, ids, mask, labels = bench.synthetic_batch(32, dev, 0)
    kwargs = dict(pixel_values=px, input_ids=ids, attention_mask=mask, labels=labels)
census = collections.Counter()
orig = K.gemm

def spy(a, b, M, N, Kd, lda, ldb, a_kc=True, b_kc=True, out_f32=None, out_bf16=None, pre_bf16=None, bias=None, residual=None, act_grad_of=None,
        act=K.ACT_NONE, act_bwd=K.ACT_NONE, drop=K.NO_DROP, allow_split_k=False, split_k=0, tile_hint=0, colsum=None, **kw):
    flags = ('NT' if a_kc and b_kc else 'NN' if a_kc else 'TN' if not b_kc else 'TK',
             'bias' if bias is not None else '-', f'act{act}' if act else '-', 'pre' if pre_bf16 is not None else '-', f'actbwd{act_bwd}' if act_grad_of is not None else '-',
             'drop' if drop.p > 0 else '-', 'res' if residual is not None else '-', 'f32' if out_f32 is not None else '-', 'b16' if out_bf16 is not None else '-',
             'colsum' if colsum is not None else '-', 'splitk' if (allow_split_k or split_k) else '-')
    tile = '128x64' if (a_kc and M >= 512 and N >= 1536) else '64x64' if (M >= 256 and N >= 64) else '32x32' if (a_kc and N >= 64) else 'other'
    bn = {'128x64': 64, '64x64': 64, '32x32': 32}.get(tile, 1)
    whole = tile != 'other' and Kd % 64 == 0 and N % bn == 0 and not (allow_split_k or split_k)
    flags = (tile, 'whole' if whole else 'ragged') + flags
    census[(M, N, Kd) + flags] += 1
    return orig(a, b, M, N, Kd, lda, ldb, a_kc, b_kc, out_f32=out_f32, out_bf16=out_bf16, pre_bf16=pre_bf16, bias=bias, residual=residual, act_grad_of=act_grad_of,
                act=act, act_bwd=act_bwd, drop=drop, allow_split_k=allow_split_k, split_k=split_k, tile_hint=tile_hint, colsum=colsum, **kw)

for _ in range(2):
    opt.zero_grad(set_to_none=True)
    out = model(**kwargs)
    out.loss.backward()
    opt.step()
K.gemm = spy
import vqa_model_builder_amd.hip.blocks as B
for mod in list(sys.modules.values()):
    if mod is not None and getattr(mod, '__name__', '').startswith('vqa_model_builder_amd') and hasattr(mod, 'gemm') and getattr(mod, 'gemm') is orig:
        mod.gemm = spy
opt.zero_grad(set_to_none=True)
out = model(**kwargs)
out.loss.backward()
torch.cuda.synchronize()
tot = sum(census.values())
print(f'{wl}: {tot} K.gemm calls in one forward + backward (weight gradients go through the grouped entry and are not counted)')
by_flags = collections.Counter()
for k, v in census.items():
    by_flags[k[3:]] += v
for k, v in sorted(by_flags.items(), key=lambda kv: -kv[1]):
    print(f'{v:5d}  ' + ' '.join(k))
print()
for k, v in sorted(census.items(), key=lambda kv: -kv[1])[:40]:
    print(f'{v:5d}  {k[0]:5d} x {k[1]:5d} x {k[2]:5d}  ' + ' '.join(k[3:]))
