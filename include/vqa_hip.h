/* vqa_hip.h -- C ABI of libvqa_hip.so: the MI355X (gfx950) kernels behind the AutoViVQA hot path.
 *
 * Drop-in boundary.  The reference has no FFI for this path: it sits behind a Python
 * nn.Module API (reference src/modeling/meta_arch/vqa_model.py:480-756).  These entry points are what the
 * reference-side binding for that path binds (INTEGRATION.md shows the ctypes stub); each one cites the
 * reference call site whose arithmetic it replaces.
 *
 * Conventions
 *  - every function returns 0 on success, a hipError_t value or VQA_ERR_ARG (1001) on bad arguments; nothing
 *    throws, nothing synchronises, nothing allocates: the caller owns every buffer (outputs, saved-for-
 *    backward tensors, workspaces) and passes the stream to launch on (vqa_stream_t = hipStream_t);
 *  - pointers are device pointers; "bf16" buffers hold IEEE bfloat16 (2 bytes); all matrices row-major with an
 *    explicit leading dimension in ELEMENTS; 16-byte aligned bases; feature dims multiples of 8;
 *  - masks are uint8 (1 = ignore this key), token ids / labels are int64 like the torch tensors they alias.
 */
#ifndef VQA_HIP_H
#define VQA_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* vqa_stream_t;

#define VQA_ACT_NONE 0
#define VQA_ACT_GELU_ERF 1    /* nn.GELU() / HF "gelu": vqa_model.py:267, modeling_roberta intermediate */
#define VQA_ACT_QUICK_GELU 2  /* CLIP hidden_act: x*sigmoid(1.702x) */
#define VQA_ACT_RELU 3        /* vqa_model.py:343,458 */

int vqa_abi_version(void);
/* 0: this library was built with bfloat16 GEMM / attention operands (libvqa_hip.so), 1: IEEE fp16 (libvqa_hip_f16.so, the dtype
 * of the reference's main loop: autocast fp16 + GradScaler, training_pipeline.py:346-347,457).  Same sources, same entry
 * points: every "bf16" in a name or field below means "the library's 16-bit operand type". */
int vqa_half_kind(void);

/* ---- GEMM: C[M,N] = epilogue(alpha * sum_k A(m,k) B(n,k)) ------------------------------------------------
 * Replaces every nn.Linear / packed in_proj on the path (torch F.linear; e.g. vqa_model.py:266-269,338,457,
 * HF q/k/v/out_proj/fc1/fc2, expert_types.py:134-156) and their autograd (dX = dY W, dW = dY^T X).
 * a_kc / b_kc: 1 = operand is k-contiguous (X(r,k) at x[r*ld+k]), 0 = r-contiguous (X(r,k) at x[k*ld+r]).
 */
typedef struct VqaGemmDesc {
    const void* a; const void* b;          /* bf16 */
    int M, N, K, lda, ldb, a_kc, b_kc;
    float* c_f32; int ldc_f32;             /* optional fp32 output */
    void* c_bf16; int ldc_bf16;            /* optional bf16 output */
    void* pre_bf16; int ld_pre;            /* optional: pre-activation (after bias) saved as bf16 */
    const float* bias;                     /* optional [N] */
    const float* residual; int ld_res;     /* optional fp32 [M,N] added after activation/dropout */
    const void* act_grad_of; int ld_ag;    /* optional bf16 [M,N]: multiply by act_bwd'(value) (backward) */
    int act;                               /* VQA_ACT_* applied in the epilogue */
    int act_bwd;                           /* VQA_ACT_* whose derivative act_grad_of selects */
    float alpha;                           /* 0 means 1 */
    float drop_p; uint64_t drop_seed; uint32_t drop_stream;   /* inverted dropout on the output, index m*N+n */
    int split_k;                           /* 0 = auto (only if allow_split_k), 1 = off, >1 forced */
    int allow_split_k;                     /* fp32 atomics into c_f32 (zeroed here); plain fp32 output only */
    int tile_hint;                         /* 0 auto; else 1 + tile id of csrc/gemm.hip: 1:128x128 2:64x64 3:32x128 4:128x32 5:128x64 6:64x128
                                              7:256x128 (8 waves) 8:32x32 9:32x64 */
    float* colsum;                         /* optional fp32 [N], PRE-ZEROED by the caller: += column sums of the output values
                                              (after bias/act'/dropout, before the residual): the bias gradient of the producer */
    int c_prezeroed;                       /* split-K only: c_f32 is already zero, skip the memset */
} VqaGemmDesc;
int vqa_gemm_bf16(const VqaGemmDesc* d, vqa_stream_t stream);

/* Up to 32 independent GEMMs C_i[M_i,N_i] (fp32, plain store) = A_i B_i^T of ONE operand layout in one launch: the
 * weight-gradient GEMMs of a backward pass (dW = dY^T X: a_kc = b_kc = 0), which nothing but the optimiser waits for and
 * which the block runners therefore queue and issue together -- one cold start and one tail for all of them, thousands of
 * equal tiles to balance over the CUs.  Same tile kernel and numerics as vqa_gemm_bf16 (64x64 LDS-DMA ring, no split-K). */
#define VQA_SUMSQ_SLOTS 64      /* power of two */
#define VQA_SUMSQ_STRIDE 32     /* floats: one 128-B line per slot */
typedef struct VqaGemmGroupItem {
    const void* a; const void* b; float* c_f32;
    int M, N, K, lda, ldb, ldc;
} VqaGemmGroupItem;
int vqa_gemm_bf16_grouped(const VqaGemmGroupItem* items, int n, int a_kc, int b_kc, vqa_stream_t s);   /* a_kc == b_kc */
/* The same with the optimiser's global-norm reduction riding along: sumsq (optional; VQA_SUMSQ_SLOTS partial accumulators VQA_SUMSQ_STRIDE floats apart,
 * caller-initialised: their SUM is the quantity -- thousands of workgroups adding to ONE address finish together and serialise in one L2
 * channel: 173 us for a 33-us launch, round 3) += the sum of squares of every value written to the outputs (torch.nn.utils.clip_grad_norm_, training_pipeline.py:497, needs exactly this over all
 * gradients: the weight gradients' share is taken where they are produced instead of re-reading 4 B per parameter).  Weight-gradient items
 * (a_kc == b_kc == 0) whose rows / columns are multiples of 256 and whose token count is a multiple of 64 run on 256 x 256 tiles, one
 * 8-wave workgroup per CU (csrc/gemm_dw256.h): up to 64 of them per launch, longest reduction first, tiles drawn from a ticket counter
 * (a 4-byte memset node in front of the launch zeroes it); vqa_set_gemm_dw256(0) sends everything through the 128 x 128 / 64 x 64 ring
 * kernel.  n <= 128 here (32 through vqa_gemm_bf16_grouped). */
int vqa_gemm_bf16_grouped2(const VqaGemmGroupItem* items, int n, int a_kc, int b_kc, float* sumsq, vqa_stream_t s);
void vqa_set_gemm_dw256(int on);
/* Measurement (bench.py): while on, every GEMM dispatch carries a start / stop event pair that receives the kernel's own begin /
 * end timestamps (hipExtLaunchKernel) -- the durations rocprofv3 --kernel-trace reports; `tag` labels the launches that follow.
 * vqa_gemm_profile_collect waits for the recorded launches and returns, per tag < ntags, their summed 2*M*N*K FLOP, kernel
 * milliseconds and count, then forgets them.  Host-side state only; not for use inside a stream capture. */
void vqa_gemm_profile(int on, int tag);
int vqa_gemm_profile_collect(int ntags, double* flop, double* ms, int* launches);
/* the same, plus per tag the launches' ALGORITHMIC bytes: both operands once + every output / fused epilogue stream once (bytes may be NULL) */
int vqa_gemm_profile_collect2(int ntags, double* flop, double* ms, int* launches, double* bytes);
void vqa_set_gemm_ws(int mode);           /* one-tile-per-CU loader/consumer GEMM (csrc/gemm.hip: gemm_ws): 0 off (default: slower inside the step, profiles/r02/gemm_ws.md), 1 auto, 2 + i: force its tile i, 0x100 * mask + ...: auto over the masked tiles */
void vqa_set_gemm_force(int cfg, int stages);   /* diagnostics: tile id (0: 128x128, 1: 64x64, 4: 128x64, 5: 64x128; -1: heuristics) for every LDS-DMA launch */
void vqa_set_gemm_tile_order(int order);         /* 0 / 1 (default): row-major tile ids; 2: column-major (an XCD owns output columns: every weight line fetched by ONE XCD) -- lab */
void vqa_set_gemm_k_rotate(int on);             /* low byte 1: workgroups of XCD x start their k loop x/8 of the way through K (one HBM fetch per weight line instead of eight concurrent misses); 2: the grouped launches too;
                                                 * bits 8..10: a phase added to x (another assignment of starting points = another fp32 summation order: tests) */
void vqa_set_gemm_group_tile(int t);      /* diagnostics: 0 heuristic, 1: 64x64, 2: 128x64, 3: 128x128 */
void vqa_set_gemm_use_tr(int on);          /* diagnostics: 0 = scalar LDS gather instead of ds_read_b64_tr_b16 */
void vqa_set_gemm_v1_fast(int mode);       /* 1 (default): ring GEMMs on whole tiles (K % 64 == 0, no split-K) run the compact-prologue instantiation, with a compile-time epilogue where the launch's option set has one; 5: compact prologue, generic epilogue; 0: the general form only (tests, A/B) */
void vqa_set_gemm_pipeline(int v1);        /* 0 = register-staged double buffer; 1 = LDS-DMA pipeline; 2/3/4 = LDS-DMA with that many stages */

/* ---- elementwise / layout --------------------------------------------------------------------------------- */
/* fp32 -> bf16 (weights shadow, activations); n elements */
int vqa_cast_f32_bf16(const float* src, void* dst, size_t n, vqa_stream_t s);
/* many tensors in one launch: jobs[i] = {src,dst,n,kind}; kind 0: f32->bf16, 1: f32->f32 copy.  jobs on device. */
typedef struct VqaCastJob { const float* src; void* dst; uint64_t n; uint64_t kind; } VqaCastJob;
int vqa_cast_multi(const VqaCastJob* jobs_dev, int njobs, uint64_t max_n, vqa_stream_t s);
int vqa_cast_bf16_f32(const void* src, float* dst, size_t n, vqa_stream_t s);
/* out[n] = sum_m x[m*ld+n]  (bias gradients); x bf16 or fp32 */
int vqa_colsum_bf16(const void* x, int M, int N, int ld, float* out, vqa_stream_t s);
int vqa_colsum_f32(const float* x, int M, int N, int ld, float* out, vqa_stream_t s);
/* y = a + b (fp32), optional bf16 copy */
int vqa_add_f32(const float* a, const float* b, float* y, void* y_bf16, size_t n, vqa_stream_t s);
/* Reads nbytes (16-byte aligned start) once with `workgroups` (0 => 64) workgroups and discards them: pulls a span of weights through the
 * memory-side cache ahead of the GEMMs that will read it (run it on a side stream, one layer ahead).  Replaces nothing in the reference. */
int vqa_prefetch(const void* p, size_t nbytes, int workgroups, vqa_stream_t s);
/* out = dy * act'(pre) * dropout_mask(index)  -- backward of y = dropout(act(pre)) outside a GEMM epilogue
 * (vqa_model.py:458-459 ReLU+Dropout, expert FFNs).  pre may be NULL (no activation). */
int vqa_act_drop_bwd(const float* dy, const void* pre_bf16, int act, float* out, void* out_bf16, size_t n, float p, uint64_t seed,
                     uint32_t stream, vqa_stream_t s);
/* rows gather/scatter on fp32 [*,D]: dst[i,:] = src[idx[i],:] ; idx int32 */
int vqa_gather_rows_f32(const float* src, const int32_t* idx, float* dst, void* dst_bf16, int n, int D, int ld_src, vqa_stream_t s);

/* CLIP patch embedding im2col (HF CLIPVisionEmbeddings conv, stride == kernel, no bias): pixels fp32
 * [B,3,H,W] -> bf16 [B*P, 3*ps*ps] with k = (c, kh, kw) as the conv weight is laid out. */
int vqa_patchify_bf16(const float* pixels, void* out, int B, int C, int H, int W, int ps, vqa_stream_t s);
/* u[b,0,:] = cls + pos[0]; u[b,1+p,:] = E[b*P+p,:] + pos[1+p]  (fp32 [B*T,D]) */
int vqa_clip_assemble(const float* E, const float* cls, const float* pos, float* u, int B, int P, int D, vqa_stream_t s);
/* backward of vqa_clip_assemble: dE bf16 [B*P,D] (for the patch GEMM dW), dcls[D], dpos[T,D] */
int vqa_clip_assemble_bwd(const float* du, void* dE_bf16, float* dcls, float* dpos, int B, int P, int D, vqa_stream_t s);

/* ---- LayerNorm (torch F.layer_norm, eps 1e-5; vqa_model.py:273-275,357 and every HF/expert LN) ------------ */
/* y = LN(x (+ add)) * gamma + beta.  x fp32 [rows, cols]; optional add (fp32); outputs optional. */
int vqa_layernorm_fwd(const float* x, const float* add, const float* gamma, const float* beta,
                      float* y_f32, void* y_bf16, float* mean, float* rstd, int rows, int cols, float eps,
                      float drop_p, uint64_t drop_seed, uint32_t drop_stream, vqa_stream_t s);
/* dx = dres (optional, fp32) + LN'(dy).  x = the LN input saved in forward.  dgamma/dbeta via partials:
 * ws must hold vqa_layernorm_bwd_ws_floats(cols) floats.  dx_bf16 = optional bf16 copy of dx.
 * drop_mode 1: dx_bf16 is additionally masked by the dropout (drop_*) that forward applied to the tensor
 *              whose gradient it is (x = r + dropout(t): dx_bf16 is dt); dx_f32 stays unmasked (dr).
 * drop_mode 2: dy is masked on load (forward was y = dropout(LN(x)) with the same drop_* at index row*cols+c).
 * dx_colsum (optional, [cols]): column sums of the (masked) gradient that dx_bf16 holds = the bias gradient of the
 *              Linear whose output was added into x; fused here so no separate pass over dx is needed.  cols <= 3328.
 * ws == NULL with dgamma / dbeta / dx_colsum given = ACCUMULATE mode: the per-workgroup partial sums are added to the
 *              outputs with fp32 atomics (outputs must be initialised).  One launch instead of two, but 256 workgroups
 *              adding to the same 3*cols addresses measured 3x SLOWER than the two-pass form on MI355X (34.9 vs 11.7 + 8.2
 *              us at 2048 x 768): only worth it for short inputs; the block runners use the two-pass form (ws != NULL). */
size_t vqa_layernorm_bwd_ws_floats(int cols);
int vqa_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                      const float* dres, float* dx_f32, void* dx_bf16, float* dgamma, float* dbeta, float* dx_colsum,
                      float* ws, int rows, int cols, float drop_p, uint64_t drop_seed, uint32_t drop_stream,
                      int drop_mode, vqa_stream_t s);
/* The same backward with the final reduction of the per-workgroup partials DEFERRED: ws (vqa_layernorm_bwd_ws_floats)
 * keeps vqa_layernorm_bwd_blocks(rows) partial rows each of dgamma | dbeta | dx_colsum (the last when want_colsum), and
 * vqa_layernorm_reduce_grouped sums the partials of up to 32 such calls in ONE launch (a block runner issues it once at the
 * end of its backward instead of one short launch behind every LayerNorm). */
int vqa_layernorm_bwd_blocks(int rows);
void vqa_set_layernorm_bwd_blocks(int n);    /* tuning: workgroups of the backward kernel, <= 1024 (default 512) */
int vqa_layernorm_bwd_partials(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                               const float* dres, float* dx_f32, void* dx_bf16, int want_colsum, float* ws, int rows, int cols,
                               float drop_p, uint64_t drop_seed, uint32_t drop_stream, int drop_mode, vqa_stream_t s);
typedef struct VqaLnReduceItem {
    const float* ws; int nblocks; int cols;
    float* dgamma; float* dbeta; float* dx_colsum;      /* each optional, [cols], overwritten */
} VqaLnReduceItem;
int vqa_layernorm_reduce_grouped(const VqaLnReduceItem* items, int n, vqa_stream_t s);

/* ---- attention: softmax(Q K^T / sqrt(Dh) + key_padding) V per (batch, head) ------------------------------
 * Replaces HF CLIP/RoBERTa self-attention and nn.MultiheadAttention's core (vqa_model.py:300,304).
 * q/k/v/o are bf16 [B*S, ld] with head h at columns [h*Dh, (h+1)*Dh).  mask: uint8 [B, Skv], 1 = ignore. */
typedef struct VqaAttnDesc {
    const void* q; const void* k; const void* v; void* o;
    int ldq, ldk, ldv, ldo;
    int B, H, Sq, Skv, Dh;
    const uint8_t* key_padding_mask;
    float scale;                            /* 0 => Dh^-0.5 */
    float drop_p; uint64_t drop_seed; uint32_t drop_stream;     /* dropout on the probabilities */
    /* backward only */
    const void* d_o; int ldd_o;
    void* dq; void* dk; void* dv; int lddq, lddk, lddv;
    /* optional, backward: fp32 [H*Dh] each, ACCUMULATED into (must be initialised, e.g. zero-filled arena slots): column
     * sums over all B*S rows of the bf16 dq / dk / dv written above = the bias gradients of the Q/K/V projections, fused so
     * that no separate pass re-reads the three tensors. */
    float* dq_colsum; float* dk_colsum; float* dv_colsum;
    /* optional, backward: workspace of vqa_attention_bwd_ws_floats(...) floats.  Needed (non-zero size) only when the tiles of one
     * (batch, head) exceed the LDS together (Sq = Skv = 100 at Dh = 256: the ObjectDetection expert's queries): the backward then
     * runs as two launches that hand the probabilities / dS over through it. */
    float* ws;
    int causal;                             /* 1: query i attends keys j <= i only (nn.TransformerDecoder's tgt_mask, generative_vqa_model.py:447-451) */
} VqaAttnDesc;
size_t vqa_attention_bwd_ws_floats(int B, int H, int Sq, int Skv, int Dh);
int vqa_attention_fwd(const VqaAttnDesc* d, vqa_stream_t s);
void vqa_set_attention_mfma(int on);       /* 1 (default): MFMA kernel for Sq,Skv <= 64, Dh in {32,64,96,128}; 0: generic kernel only */
int vqa_attention_bwd(const VqaAttnDesc* d, vqa_stream_t s);

/* ---- fused in-projection + attention, forward (the x1 row of SURVEY section 8) ---------------------------------
 * nn.MultiheadAttention up to (not including) its out-projection in ONE launch (reference vqa_model.py:300 self-attention and :304
 * cross-attention of CrossModalAttention; the encoders' self-attention has the same shape): workgroup (sample b, head h) projects
 * that head's Q from xq's rows and K | V from xkv's rows with the packed in_proj weight (MFMA, LDS-DMA ring), keeps the three
 * [64 x Dh] tiles in LDS and runs softmax(Q K^T / sqrt(Dh) + key_padding) V on them.  The projections reach HBM only as the
 * optional q / k / v copies backward needs (bf16, head h at columns [h*Dh, (h+1)*Dh)); results equal vqa_gemm_bf16 (bias, bf16
 * out) followed by vqa_attention_fwd bit for bit.  Covered: Dh = D / H in {64, 96}, Sq, Skv <= 64, D % 64 == 0; anything
 * else returns VQA_ERR_ARG (callers keep the two-launch form for it). */
typedef struct VqaFusedAttnDesc {
    const void* xq; int ldxq;            /* bf16 [B*Sq, ldxq]: rows the queries are projected from */
    const void* xkv; int ldxkv;          /* bf16 [B*Skv, ldxkv]: rows the keys / values are projected from (== xq: self-attention) */
    const void* w_in; int ldw;           /* bf16 packed in_proj weight [3D, D] (q | k | v row blocks, torch layout), k-contiguous */
    const float* b_in;                   /* fp32 [3D] or NULL */
    void* q; void* k; void* v;           /* optional bf16 outputs [B*Sq, ldq] / [B*Skv, ldk] / [B*Skv, ldv] */
    int ldq, ldk, ldv;
    void* o; int ldo;                    /* bf16 [B*Sq, ldo]: attention output (input of the out-projection) */
    int B, H, Sq, Skv, D;
    const uint8_t* key_padding_mask;     /* uint8 [B, Skv], 1 = ignore; or NULL */
    float scale;                         /* 0 => Dh^-0.5 */
    float drop_p; uint64_t drop_seed; uint32_t drop_stream;     /* dropout on the probabilities, keyed as in vqa_attention_fwd */
    int causal;                          /* as VqaAttnDesc.causal */
} VqaFusedAttnDesc;
int vqa_fused_inproj_attention_fwd(const VqaFusedAttnDesc* d, vqa_stream_t s);


/* ---- RoBERTa embeddings (HF RobertaEmbeddings: word + type0 + pad-aware positions, LN) --------------------- */
/* pos_ids out: int32 [B,S] = cumsum(ids != pad) * (ids != pad) + pad.  u = sum of the three rows (fp32).
 * V / Pmax: rows of the word / position tables.  An id outside [0,V) or a position id >= Pmax (nn.Embedding: error) is never
 * dereferenced -- the padding row is read instead -- and *ok (optional device int32, initialised to 1 by the caller) is cleared. */
int vqa_roberta_embed_fwd(const int64_t* ids, const float* word, const float* pos, const float* type0,
                          int32_t* pos_ids, float* u, int B, int S, int D, int pad_id, int V, int Pmax, int32_t* ok, vqa_stream_t s);
/* scatter-add du into dword/dpos (rows pad_id skipped: nn.Embedding padding_idx) and dtype0 = sum of all rows.
 * dword [V,D] and dpos [Pmax,D] must be zero-filled by the caller; dtype0 [D]. */
int vqa_roberta_embed_bwd(const float* du, const int64_t* ids, const int32_t* pos_ids, float* dword, float* dpos,
                          float* dtype0, int B, int S, int D, int pad_id, int V, int Pmax, vqa_stream_t s);

/* nn.Embedding backward without padding_idx (generative_vqa_model.py:497 answer_embedding, tied with the output projection):
 * dweight[ids[i], :] += dy[i, :] with atomic adds (ids repeat); dweight fp32 [V, D], zero on entry; ids outside [0, V) are skipped */
int vqa_embedding_rows_bwd(const float* dy, const int32_t* ids, float* dweight, int n, int D, int V, vqa_stream_t s);

/* ---- loss (vqa_model.py:711-716: F.cross_entropy mean + argmax) ------------------------------------------- */
/* per-row loss (fp32 [B]) and argmax (int64 [B]); loss_mean (fp32 [2]) = {mean over the rows whose label is not
 * ignore_index (-100, as F.cross_entropy), number of such rows}.  A label outside [0,C) other than -100 (torch: device assert)
 * is never dereferenced: that row's loss is NaN and *ok (optional device int32, caller-initialised to 1) is cleared.
 * label_smoothing e in [0, 1): row loss = (1 - e) * nll + e * mean_c(-log p_c)  (nn.CrossEntropyLoss(label_smoothing=e), the generative
 * model's loss, generative_vqa_model.py:507-510); 0 for the classification path. */
int vqa_softmax_ce_argmax_fwd(const float* logits, int ld, const int64_t* labels, float* row_loss, float* loss_mean,
                              int64_t* argmax, float* lse, int B, int C, int32_t* ok, float label_smoothing, vqa_stream_t s);
/* dlogits = (softmax - (1 - e) onehot - e / C) * (*dloss) / nvalid (rows with an ignored / invalid label: 0); nvalid = &loss_mean[1] of the forward
 * (NULL: B); outputs fp32 and optional 16-bit copy */
int vqa_softmax_ce_bwd(const float* logits, int ld, const int64_t* labels, const float* lse, const float* dloss, const float* nvalid,
                       float* dlogits, void* dlogits_bf16, int B, int C, float label_smoothing, vqa_stream_t s);

/* ---- MoE router + dispatch (router.py:287-366, moe_layer.py:146-168); fp32 throughout ----------------------- */
/* clean[t,e] = <x[t],gate[e]>;  noisy = clean (+ noise[t,e] * softplus(<x[t],w_noise[e]>) * noise_std when noise != NULL;
 * noise_raw keeps <x,w_noise> for backward).  x fp32 [T,D], gate / w_noise fp32 [E,D] (nn.Linear, no bias). */
int vqa_router_gate_fwd(const float* x, const float* gate, const float* w_noise, const float* noise, float noise_std,
                        float* clean, float* noisy, float* noise_raw, int T, int E, int D, vqa_stream_t s);
int vqa_router_gate_bwd(const float* x, const float* gate, const float* w_noise, const float* noise, float noise_std,
                        const float* noise_raw, const float* dlogits, float* dgate, float* dw_noise, float* dx,
                        int T, int E, int D, vqa_stream_t s);
/* logits fp32 [T,E] -> softmax -> top-k (lowest index wins ties) -> renormalise by the k-sum.  probs_all optional. */
int vqa_router_topk_fwd(const float* logits, float* weights, int64_t* indices, float* probs_all, int T, int E, int K,
                        vqa_stream_t s);
int vqa_router_topk_bwd(const float* logits, const int64_t* indices, const float* dweights, float* dlogits,
                        int T, int E, int K, vqa_stream_t s);
/* load_balance_loss = weight * E * sum_e (tokens_e/T) * mean_t probs[t,e]  (probs = softmax of the CLEAN logits) */
int vqa_router_aux_loss(const float* probs, const int64_t* indices, int T, int E, int K, float weight, float* out, vqa_stream_t s);
/* per expert e: combine weight w_all[e,t] = sum_k weights[t,k]*(indices[t,k]==e) (moe_layer.py:160-161), the
 * order-preserving list of its tokens lists[e, 0:counts[e]] and counts[e].  Indices outside [0,E) (the ablation
 * harness writes -1) route nowhere.  weights/indices may have any K <= 16. */
int vqa_moe_expert_tokens(const float* weights, const int64_t* indices, int T, int K, int E, float* w_all, int32_t* lists,
                          int32_t* counts, vqa_stream_t s);
/* out[list[i],:] += w_tok[list[i]] * y[i,:]   (fp32; moe_layer.py:167-168 on the routed rows only) */
int vqa_moe_scatter_add(const float* y, const int32_t* list, const float* w_tok, float* out, int n, int D, vqa_stream_t s);
/* backward of the line above: dy[i,:] = w*dout[list[i],:] (fp32 and/or bf16), dw_tok[list[i]] = <dout[list[i]], y[i]> */
int vqa_moe_combine_bwd(const float* dout, const float* y, const int32_t* list, const float* w_tok, float* dy, void* dy_bf16,
                        float* dw_tok, int n, int D, vqa_stream_t s);
/* dweights[t,k] = dw_all[indices[t,k], t] */
int vqa_moe_route_weight_grad(const float* dw_all, const int64_t* indices, float* dweights, int T, int E, int K, vqa_stream_t s);

/* GatedLinearExpert's gate (reference src/modeling/moe/expert_types.py:501-504: h, gate = fc1(x).chunk(2); h * sigmoid(gate); dropout):
 * y[t,j] = drop(h[t,j] * sigmoid(h[t,H+j])) over h [T,2H] fp32 (dropout keyed on t*H+j), and its backward dh [T,2H]. */
int vqa_glu_fwd(const float* h, float* y, int T, int H, float drop_p, uint64_t drop_seed, uint32_t drop_stream, vqa_stream_t s);
int vqa_glu_bwd(const float* dy, const float* h, float* dh, int T, int H, float drop_p, uint64_t drop_seed, uint32_t drop_stream, vqa_stream_t s);

/* Dense dispatch (captured-graph mode, moe_layer.py:151-168 exactly as the reference runs it: every expert on every token, combined
 * with weights that are 0 where an expert was not chosen): out[t,:] = sum_e w_all[e,t] * ys[e][t,:] in one launch (w_all [E,T] as
 * vqa_moe_expert_tokens writes it), and its backward dys[e][t,:] = w_all[e,t] * dout[t,:], dw_all[e,t] = <dout[t,:], ys[e][t,:]>.
 * ys / dys: HOST arrays of E <= 16 device pointers. */
int vqa_moe_dense_combine_fwd(const float* const* ys, const float* w_all, float* out, int T, int E, int D, vqa_stream_t s);
int vqa_moe_dense_combine_bwd(const float* dout, const float* const* ys, const float* w_all, float* const* dys, float* dw_all, int T, int E, int D,
                              vqa_stream_t s);

/* ---- row kernels of the expert runners (expert_types.py:159-199,270-312,395-445, specialized_experts.py:119-173 at one token per
 * sample): each replaces a chain of cast / fill / elementwise / column-sum launches on tensors of <= a few hundred rows ------- */
/* out_bf16[m,n] = dy[m*ld+n] * act'(pre[m,n]) * dropmask(m*N+n)  and  colsum[n] += sum_m out (optional; fp32, pre-zeroed): the
 * backward of y = drop(act(pre)) + the bias gradient of the Linear that produced pre, in one pass (M small). */
int vqa_rows_mask_cast(const float* dy, int ld, const void* pre_bf16, int act, void* out_bf16, float* colsum, int M, int N, float p,
                       uint64_t seed, uint32_t stream, vqa_stream_t s);
/* y = dropout(act(x)) where the activation does not ride in a GEMM epilogue (fusion_approaches.py:124-131: Linear -> LayerNorm -> GELU ->
 * Dropout); optional 16-bit copies of y and of x (what vqa_act_drop_bwd reads as the pre-activation); dropout keyed by element index */
int vqa_act_drop_fwd(const float* x, float* y, void* y_bf16, void* pre_bf16, size_t n, int act, float p, uint64_t seed, uint32_t stream, vqa_stream_t s);
/* nn.MultiheadAttention over ONE key per sample: softmax == 1, the context is V times the dropout keep-scale of the (sample, head,
 * query) probability.  fwd: out[(t*R + r), :] = v[t, :] * keep(t, head, r)  (R queries per sample; p == 0: a plain broadcast);
 * bwd: dv[t, :] = sum_r dout[(t*R + r), :] * keep(t, head, r).  bf16 in / out, keyed like vqa_attention_fwd's element (b, h, q, 0). */
int vqa_head_keep_fwd(const void* v, void* out, int T, int R, int H, int Dh, float p, uint64_t seed, uint32_t stream, vqa_stream_t s);
int vqa_head_keep_bwd(const void* dout, void* dv, int T, int R, int H, int Dh, float p, uint64_t seed, uint32_t stream, vqa_stream_t s);
/* dst row i = alpha * src row (mode 0: i / R, every source row R times; mode 1: i % R, the R source rows tiled); fp32 and/or bf16 */
int vqa_repeat_rows_f32(const float* src, float* dst, void* dst_bf16, int out_rows, int D, int R, int mode, float alpha, vqa_stream_t s);
/* out[t, :] = mean of rows t*R .. t*R+R-1 of x (fp32 [T*R, D]); fp32 and/or bf16 with row pitch ld_out */
int vqa_rows_mean_f32(const float* x, int R, float* out, void* out_bf16, int ld_out, int T, int D, vqa_stream_t s);
/* dst[i] = src[i*stride + offset] (bf16) / dst[i*stride + offset] = src[i] (fp32): the centre tap of a Conv1d(k=3) weight, which is
 * all a length-1 sequence ever multiplies (specialized_experts.py:66-71 at S = 1), and its gradient's way back */
int vqa_take_stride_bf16(const void* src, void* dst, size_t n, int stride, int offset, vqa_stream_t s);
int vqa_scatter_stride_f32(const float* src, float* dst, size_t n, int stride, int offset, vqa_stream_t s);
/* counter-RNG helpers: N(0,1) noise for the noisy router; stand-alone inverted dropout (vqa_model.py:705) */
int vqa_randn_f32(float* out, uint64_t n, uint64_t seed, uint32_t stream, vqa_stream_t s);
int vqa_dropout_f32(const float* x, float* y, void* y_bf16, uint64_t n, float p, uint64_t seed, uint32_t stream, vqa_stream_t s);

/* ---- fused AdamW (torch.optim.AdamW semantics; reference training_pipeline.py:234-287) --------------------- */
typedef struct VqaAdamWDesc {
    float* param; const float* grad; float* exp_avg; float* exp_avg_sq; void* param_bf16 /* optional shadow */;
    uint64_t n;
    float lr, beta1, beta2, eps, weight_decay, bias_correction1, bias_correction2;
    const float* grad_scale;                /* optional device scalar multiplied into grad (clip coefficient) */
} VqaAdamWDesc;
int vqa_adamw_step(const VqaAdamWDesc* d, vqa_stream_t s);
/* multi-tensor form: one launch over a DEVICE table of per-tensor jobs (all tensors 16-byte aligned).
 * vqa_sumsq_multi: norm2[0] += sum over all jobs of |grad|^2 (caller zeroes norm2).
 * vqa_adamw_multi: grad *= min(1, max_norm / (sqrt(norm2[0]) + 1e-6)) when norm2 != NULL and max_norm > 0
 *                  (torch.nn.utils.clip_grad_norm_, training_pipeline.py:497), then the AdamW update with the job's own
 *                  weight_decay (param groups of training_pipeline.py:239-252); shadow != NULL: also writes the
 *                  parameter's bf16 (shadow_kind 0) or packed-fp32 (shadow_kind 1) copy the GEMMs read. */
#define VQA_OPT_GRAD_BF16 0x100u  /* VqaOptJob::shadow_kind flag: ``grad`` points at bfloat16 values (the data-parallel exchange's wire format: the
                                   * all-reduced sums are consumed where RCCL left them, no copy back into the fp32 gradient); low byte = kind */
typedef struct VqaOptJob {
    float* param; const float* grad; float* exp_avg; float* exp_avg_sq; void* shadow;
    uint64_t n; float weight_decay; uint32_t shadow_kind;
    const float* active;     /* optional device word: the job is SKIPPED (parameter, moments, shadow untouched) while *active == 0.
                              * The captured MoE step runs every expert on every token (no host read of the routing counts);
                              * an expert no token chose must still be left alone by the optimiser, as the reference's
                              * grad-is-None skip does -- its routed-token count is this word. */
    const float* own_step;   /* optional device word: this job's OWN step count (number of updates it received, this one included)
                              * for the two bias corrections -- torch keeps `step` per parameter, so an expert skipped in some
                              * steps must not have its corrections aged by them.  Advanced by vqa_opt_advance_counts. */
    uint8_t* touched;        /* optional device map, one byte per 256 elements (ceil(n / 256) bytes, zero = this granule's moments are still
                              * exactly zero): a granule whose gradient is all zero in this step and whose byte is zero gets the update AdamW
                              * gives it anyway -- p *= 1 - lr * weight_decay, moments stay zero -- WITHOUT reading or writing the moments (12
                              * instead of 28 B per parameter); any non-zero gradient value sets the byte for good.  For embedding tables: of
                              * PhoBERT's 64 001 rows a step touches at most batch x seq, most of the vocabulary never.  Decided from the
                              * gradient VALUES, so any training flow stays exact; the caller owns the map (zero-filled when the moments are). */
} VqaOptJob;
/* chunks_dev: uint32 [nchunks][2] = {job index, first element}; every chunk covers vqa_opt_chunk_elems() elements of its
 * tensor (the last one of a tensor fewer): one workgroup per chunk keeps the chip streaming whatever the tensor sizes. */
int vqa_opt_chunk_elems(void);
/* steps[i] += 1 where active[i] > 0 (and, when norm2 is given, only if it is finite: a step skipped by the loss scaler counts for nobody) */
int vqa_opt_advance_counts(float* steps, const float* active, int n, const float* norm2, vqa_stream_t s);
int vqa_sumsq_multi(const VqaOptJob* jobs_dev, const uint32_t* chunks_dev, int nchunks, float* norm2, vqa_stream_t s);
/* hyper_dev (optional): device floats {lr, step}; when given they override lr and the two bias corrections (computed from
 * step on the device), so a captured HIP graph of the optimiser step follows the schedule and the step count. */
int vqa_adamw_multi(const VqaOptJob* jobs_dev, const uint32_t* chunks_dev, int nchunks, const float* norm2, float max_norm, float lr,
                    float beta1, float beta2, float eps, float bias_correction1, float bias_correction2, const float* hyper_dev,
                    float grad_prescale, const float* amp_dev, vqa_stream_t s);
/* grad_prescale (0 => 1): the gradients in memory are to be read as grad * grad_prescale (norm2 is of the UNscaled values):
 * data-parallel ranks hand over the all-reduced SUM and 1/world here, so the mean is never written out. */
/* amp_dev (optional; needs norm2): device floats {loss_scale, growth_tracker, found_inf}: the gradients in memory are
 * loss_scale x the true ones (fp16 mode; torch.amp.GradScaler's role, reference training_pipeline.py:346-347,466-502): they are
 * un-scaled on the fly, and a non-finite norm2 (any inf / nan gradient) SKIPS the whole update like GradScaler.step() does.
 * vqa_amp_update then applies GradScaler.update(): scale *= backoff_factor after a skipped step, *= growth_factor after
 * growth_interval clean ones.  vqa_opt_advance: hyper_dev[1] (the device step count) += 1 unless norm2 is non-finite. */
int vqa_amp_update(float* amp_dev, const float* norm2, float growth_factor, float backoff_factor, int growth_interval, vqa_stream_t s);
int vqa_opt_advance(float* hyper_dev, const float* norm2, vqa_stream_t s);
/* nn.Bilinear (fusion_type='bilinear', vqa_model.py:348-351): y = z W^T with z[b, i*D2+j] = x1[b,i]*x2[b,j] (bf16, one GEMM over
 * K = D1*D2) -- vqa_outer_bf16 builds z; vqa_outer_bwd contracts dz = dy W back: dx1[b,i] = <dz[b,i,:], x2[b]>, dx2[b,j] =
 * sum_i dz[b,i,j] x1[b,i].  D2 % 4 == 0. */
int vqa_outer_bf16(const float* x1, const float* x2, void* z_bf16, int B, int D1, int D2, vqa_stream_t s);
int vqa_outer_bwd(const float* dz, const float* x1, const float* x2, float* dx1, float* dx2, int B, int D1, int D2, vqa_stream_t s);
/* sum of squares of a fp32 buffer accumulated into out[0] (atomic; caller zeroes) */
int vqa_sumsq_f32(const float* x, uint64_t n, float* out, vqa_stream_t s);

#ifdef __cplusplus
}
#endif
#endif
