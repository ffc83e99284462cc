"""First numbers for the generative model (ViT-B/32 + PhoBERT + 2-layer fusion + 6-layer decoder + tied 64 000-way head), batch 32,
question 64 tokens, answer A tokens: eager step and captured-graph step (fwd + bwd + clip + AdamW), bf16 operands, train mode."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqa_model_builder_amd.modeling.meta_arch.generative_vqa_model import GenerativeVQAConfig, GenerativeVQAModel
from vqa_model_builder_amd.optim import FusedAdamW
from vqa_model_builder_amd.graph import GraphedTrainStep
from vqa_model_builder_amd.hip import lib

B, A = 32, int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = 'cuda'
torch.manual_seed(0)
cfg = GenerativeVQAConfig(visual_arch=dict(hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12, image_size=224, patch_size=32),
                          text_arch=dict(vocab_size=64001, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                                         max_position_embeddings=258, type_vocab_size=1, pad_token_id=1))
model = GenerativeVQAModel(cfg).to(dev).train()
with torch.no_grad():
    for n, p in model.named_parameters():
        if p.dim() >= 2:
            p.normal_(0.0, 0.02)
nd = ('bias', 'LayerNorm.weight', 'layer_norm.weight', 'norm')
groups = [{'params': [p for n, p in model.named_parameters() if not any(t in n for t in nd)], 'weight_decay': 0.01},
          {'params': [p for n, p in model.named_parameters() if any(t in n for t in nd)], 'weight_decay': 0.0}]
opt = FusedAdamW(groups, lr=2e-5, max_grad_norm=1.0).attach_shadows(model)
batch = dict(pixel_values=torch.randn(B, 3, 224, 224, device=dev), input_ids=torch.randint(3, 30000, (B, 64), device=dev),
             attention_mask=torch.ones(B, 64, dtype=torch.long, device=dev), decoder_input_ids=torch.randint(3, 64000, (B, A), device=dev),
             decoder_attention_mask=torch.ones(B, A, dtype=torch.long, device=dev), labels=torch.randint(3, 64000, (B, A), device=dev))
nparams = sum(p.numel() for p in set(model.parameters()))


def eager():
    opt.zero_grad(set_to_none=True)
    out = model(**batch)
    out.loss.backward()
    opt.step()
    return out.loss


def timeit(fn, n=20, w=5):
    for _ in range(w):
        l = fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        l = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, float(l)


s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    ms_e, le = timeit(eager)
torch.cuda.synchronize()
L = lib.load()
import ctypes as C
L.vqa_gemm_profile(1, 0)
with torch.cuda.stream(s):
    for _ in range(2):
        eager()
torch.cuda.synchronize()
f, m, n = (C.c_double * 1)(), (C.c_double * 1)(), (C.c_int * 1)()
L.vqa_gemm_profile_collect(1, f, m, n); L.vqa_gemm_profile(0, 0)
print(f'generative B={B} A={A} params={nparams/1e6:.0f}M  eager: {ms_e:.2f} ms/step ({B/ms_e*1e3:.0f} samples/s) loss {le:.3f};  GEMMs: {f[0]/2/1e9:.0f} GFLOP/step in {m[0]/2:.2f} ms '
      f'= {f[0]/m[0]/1e9:.0f} TFLOP/s ({f[0]/m[0]/1e9/2500:.3f} of the bf16 MFMA peak), {n[0]//2} launches/step', flush=True)
try:
    gs = GraphedTrainStep(model, opt, batch, warmup=2)
    ms_g, lg = timeit(lambda: gs(batch))
    print(f'generative B={B} A={A} captured graph: {ms_g:.2f} ms/step ({B/ms_g*1e3:.0f} samples/s) loss {lg:.3f}', flush=True)
except Exception as e:
    print('graph capture failed:', type(e).__name__, str(e)[:300])
