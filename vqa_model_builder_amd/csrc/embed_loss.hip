// RoBERTa embedding gather / scatter-add and the cross-entropy + argmax head.  HBM/latency-bound row kernels:
// one wavefront per row, 16-byte accesses, shuffles for the row reductions.
#include "common.h"
#include "vqa_hip.h"

namespace {

// gridDim = (B, chunks): every block of a sample redoes the (cheap) pad-aware position-id scan over its S <= 4096 tokens -- block
// (b, 0) also writes the ids out -- and gathers its share of the rows: u[b,s,:] = word[ids] + pos[pos_id] + type0.  One block
// per sample left 32 workgroups to pull 6 MB of random table rows (49 us); 8 chunks per sample: 256 workgroups.
// Ids outside [0, V) and position ids >= Pmax (nn.Embedding raises / device-asserts on them) are never dereferenced: the row reads
// the padding row instead and *ok (optional device word, 1 = fine) is cleared for the host (hip/ops.py: check_device_status).
__global__ void roberta_embed_fwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ word, const float* __restrict__ pos,
                                         const float* __restrict__ type0, int32_t* __restrict__ pos_ids, float* __restrict__ u,
                                         int S, int D, int pad_id, int V, int Pmax, int32_t* __restrict__ ok) {
    extern __shared__ int scan[];                 // [S]
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int s = tid; s < S; s += blockDim.x) scan[s] = ids[(size_t)b * S + s] != pad_id ? 1 : 0;
    __syncthreads();
    if (tid == 0) { int run = 0; for (int s = 0; s < S; ++s) { const int m = scan[s]; run += m; scan[s] = m ? run + pad_id : pad_id; } }
    __syncthreads();
    if (blockIdx.y == 0)
        for (int s = tid; s < S; s += blockDim.x) pos_ids[(size_t)b * S + s] = scan[s];
    const int d4 = D / 4;
    const int per = (S + gridDim.y - 1) / gridDim.y, s0 = blockIdx.y * per, s1 = min(S, s0 + per);
    for (int t = tid; t < (s1 - s0) * d4; t += blockDim.x) {
        const int s = s0 + t / d4, c = t % d4;
        int64_t id = ids[(size_t)b * S + s];
        int pid = scan[s];
        if (id < 0 || id >= V || pid >= Pmax) { if (ok && c == 0) *ok = 0; id = id < 0 || id >= V ? pad_id : id; pid = pid >= Pmax ? pad_id : pid; }
        f32x4 v = reinterpret_cast<const f32x4*>(word + (size_t)id * D)[c];
        v += reinterpret_cast<const f32x4*>(pos + (size_t)pid * D)[c];
        v += reinterpret_cast<const f32x4*>(type0)[c];
        reinterpret_cast<f32x4*>(u + ((size_t)b * S + s) * D)[c] = v;
    }
}

// scatter-add rows of du into the (pre-zeroed) dense table gradients; pad rows are skipped (padding_idx)
__global__ void roberta_embed_bwd_kernel(const float* __restrict__ du, const int64_t* __restrict__ ids, const int32_t* __restrict__ pos_ids,
                                         float* __restrict__ dword, float* __restrict__ dpos, int rows, int D, int pad_id, int V, int Pmax) {
    const size_t total = (size_t)rows * D, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const int r = (int)(t / D), d = (int)(t % D);
        const float g = du[t];
        const int64_t id = ids[r];
        const int pid = pos_ids[r];
        if (id != pad_id && id >= 0 && id < V) atomicAdd(dword + (size_t)id * D + d, g);
        if (pid != pad_id && pid < Pmax) atomicAdd(dpos + (size_t)pid * D + d, g);
    }
}

// nn.Embedding backward without padding_idx (the generative decoder's tied token table): dweight[ids[i], :] += dy[i, :]; ids repeat
// (BOS in every row, pads), so the adds are atomic.  dweight zero on entry.
__global__ void embedding_rows_bwd_kernel(const float* __restrict__ dy, const int32_t* __restrict__ ids, float* __restrict__ dweight, int n, int D, int V) {
    const size_t total = (size_t)n * D, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const int i = (int)(t / D), d = (int)(t % D);
        const int id = ids[i];
        if (id >= 0 && id < V) atomicAdd(dweight + (size_t)id * D + d, dy[t]);
    }
}

// ---- cross entropy: one 256-thread block per row (a wave per row walked the 3000 classes in 47 dependent steps, twice: 23 us) ----
constexpr int64_t VQA_IGNORE_INDEX = -100;
__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ logits, int ld, const int64_t* __restrict__ labels, float* __restrict__ row_loss,
                              int64_t* __restrict__ argmax, float* __restrict__ lse_out, int B, int C, int32_t* __restrict__ ok, float smooth) {
    __shared__ float sm[4]; __shared__ int si[4]; __shared__ float ss[4]; __shared__ float sx[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, row = blockIdx.x;
    const float* x = logits + (size_t)row * ld;
    // ONE pass over the row (online softmax: running maximum + rescaled sum of exponentials per thread), 16-byte loads where the row
    // allows: at 64 000 classes the two-pass scalar form read 2 x 262 MB at 1.1 TB/s (231 us for 1024 rows)
    float m = -INFINITY, sum = 0.f, xs = 0.f; int mi = 0x7fffffff;
    auto take = [&](float v, int c) {
        if (v > m) { sum = sum * __expf(m - v) + 1.f; m = v; mi = c; }     // first maximum of this thread (c only grows)
        else sum += __expf(v - m);
        xs += v;
    };
    if ((C & 3) == 0 && (ld & 3) == 0 && (((uintptr_t)logits) & 15) == 0) {
        for (int c4 = threadIdx.x; c4 < C / 4; c4 += 256) {
            const f32x4 v = reinterpret_cast<const f32x4*>(x)[c4];
#pragma unroll
            for (int j = 0; j < 4; ++j) take(v[j], 4 * c4 + j);
        }
    } else {
        for (int c = threadIdx.x; c < C; c += 256) take(x[c], c);
    }
    const float m_own = m;
    // arg-max with lowest-index tie-break (torch.argmax returns the first maximal index): wave, then the four waves
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(m, o, 64); const int oi = __shfl_xor(mi, o, 64);
        if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
    }
    if (lane == 0) { sm[wave] = m; si[wave] = mi; }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const float om = sm[w]; const int oi = si[w];
        if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
    }
    sum = m_own == -INFINITY ? 0.f : sum * __expf(m_own - m);           // rescale this thread's partial to the row maximum
    sum = wave_sum(sum);
    if (smooth > 0.f) xs = wave_sum(xs);
    if (lane == 0) { ss[wave] = sum; sx[wave] = xs; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float lse = m + __logf(ss[0] + ss[1] + ss[2] + ss[3]);
        if (lse_out) lse_out[row] = lse;
        if (argmax) argmax[row] = mi;
        if (row_loss) {
            // F.cross_entropy semantics: label == ignore_index (-100) contributes nothing and is not counted by the mean; any
            // other label outside [0, C) (torch: device assert) is never dereferenced -- the row's loss becomes NaN so the
            // fault is visible in the loss, and *ok is cleared
            float l = 0.f;
            if (labels) {
                const int64_t y = labels[row];
                if (y >= 0 && y < C) {
                    l = lse - x[y];
                    // label smoothing (nn.CrossEntropyLoss(label_smoothing=e), generative_vqa_model.py:507-510):
                    // (1 - e) * nll + e * mean_c(-log p_c),  mean_c(-log p_c) = lse - mean_c(x_c)
                    if (smooth > 0.f) l = (1.f - smooth) * l + smooth * (lse - (sx[0] + sx[1] + sx[2] + sx[3]) / (float)C);
                } else if (y != VQA_IGNORE_INDEX) { l = __builtin_nanf(""); if (ok) *ok = 0; }
            }
            row_loss[row] = l;
        }
    }
}

// mean over the rows whose label is not ignore_index (0 valid rows: 0/0 = NaN, as torch)
__global__ __launch_bounds__(256) void ce_mean_kernel(const float* __restrict__ row_loss, const int64_t* __restrict__ labels, float* __restrict__ loss_mean, int B) {
    __shared__ float sa[4], sc[4];
    float acc = 0.f, cnt = 0.f;
    for (int i = threadIdx.x; i < B; i += 256) { acc += row_loss[i]; cnt += labels[i] != VQA_IGNORE_INDEX ? 1.f : 0.f; }
    acc = wave_sum(acc); cnt = wave_sum(cnt);
    if ((threadIdx.x & 63) == 0) { sa[threadIdx.x >> 6] = acc; sc[threadIdx.x >> 6] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) { const float n = sc[0] + sc[1] + sc[2] + sc[3]; loss_mean[0] = (sa[0] + sa[1] + sa[2] + sa[3]) / n; loss_mean[1] = n; }
}

__global__ void ce_bwd_kernel(const float* __restrict__ logits, int ld, const int64_t* __restrict__ labels, const float* __restrict__ lse,
                              const float* __restrict__ dloss, const float* __restrict__ nvalid, float* __restrict__ dlogits, h16_t* __restrict__ dlb,
                              int B, int C, float smooth) {
    const float on = 1.f - smooth, off = smooth / (float)C;
    const float scale = dloss[0] / (nvalid ? nvalid[0] : (float)B);
    const size_t total = (size_t)B * C, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const int r = (int)(t / C), c = (int)(t % C);
        const int64_t y = labels[r];
        float g = 0.f;
        if (y >= 0 && y < C) {                               // ignored / invalid rows get no gradient
            g = __expf(logits[(size_t)r * ld + c] - lse[r]) - off;
            if (y == c) g -= on;
            g *= scale;
        }
        if (dlogits) dlogits[t] = g;
        if (dlb) dlb[t] = (h16_t)g;
    }
}

}  // namespace

extern "C" {

int vqa_roberta_embed_fwd(const int64_t* ids, const float* word, const float* pos, const float* type0, int32_t* pos_ids, float* u,
                          int B, int S, int D, int pad_id, int V, int Pmax, int32_t* ok, vqa_stream_t s) {
    if (!ids || !word || !pos || !type0 || !pos_ids || !u || B <= 0 || S <= 0 || S > 4096 || D % 4) return VQA_ERR_ARG;
    if (V <= pad_id || Pmax <= pad_id || pad_id < 0) return VQA_ERR_ARG;
    const int chunks = S >= 64 ? 8 : S >= 8 ? 2 : 1;
    hipLaunchKernelGGL(roberta_embed_fwd_kernel, dim3(B, chunks), dim3(256), (size_t)S * 4, (hipStream_t)s, ids, word, pos, type0, pos_ids, u, S, D, pad_id,
                       V, Pmax, ok);
    return (int)hipGetLastError();
}

int vqa_roberta_embed_bwd(const float* du, const int64_t* ids, const int32_t* pos_ids, float* dword, float* dpos, float* dtype0,
                          int B, int S, int D, int pad_id, int V, int Pmax, vqa_stream_t s) {
    if (!du || !ids || !pos_ids || !dword || !dpos || !dtype0) return VQA_ERR_ARG;
    const size_t total = (size_t)B * S * D;
    size_t g = (total + 255) / 256; if (g > 2048) g = 2048;
    hipLaunchKernelGGL(roberta_embed_bwd_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)s, du, ids, pos_ids, dword, dpos, B * S, D, pad_id, V, Pmax);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    return vqa_colsum_f32(du, B * S, D, D, dtype0, s);
}

int vqa_embedding_rows_bwd(const float* dy, const int32_t* ids, float* dweight, int n, int D, int V, vqa_stream_t s) {
    if (!dy || !ids || !dweight || n <= 0 || D <= 0 || V <= 0) return VQA_ERR_ARG;
    size_t g = ((size_t)n * D + 255) / 256; if (g > 2048) g = 2048;
    hipLaunchKernelGGL(embedding_rows_bwd_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)s, dy, ids, dweight, n, D, V);
    return (int)hipGetLastError();
}

int vqa_softmax_ce_argmax_fwd(const float* logits, int ld, const int64_t* labels, float* row_loss, float* loss_mean, int64_t* argmax,
                              float* lse, int B, int C, int32_t* ok, float label_smoothing, vqa_stream_t s) {
    if (!logits || B <= 0 || C <= 0 || (loss_mean && !row_loss) || label_smoothing < 0.f || label_smoothing >= 1.f) return VQA_ERR_ARG;
    hipLaunchKernelGGL(ce_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)s, logits, ld, labels, row_loss, argmax, lse, B, C, ok, label_smoothing);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    if (loss_mean && labels) {
        hipLaunchKernelGGL(ce_mean_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, row_loss, labels, loss_mean, B);
        e = hipGetLastError();
    }
    return (int)e;
}

int vqa_softmax_ce_bwd(const float* logits, int ld, const int64_t* labels, const float* lse, const float* dloss, const float* nvalid,
                       float* dlogits, void* dlogits_bf16, int B, int C, float label_smoothing, vqa_stream_t s) {
    if (!logits || !labels || !lse || !dloss || (!dlogits && !dlogits_bf16) || label_smoothing < 0.f || label_smoothing >= 1.f) return VQA_ERR_ARG;
    size_t g = ((size_t)B * C + 255) / 256; if (g > 4096) g = 4096;
    hipLaunchKernelGGL(ce_bwd_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)s, logits, ld, labels, lse, dloss, nvalid, dlogits, (h16_t*)dlogits_bf16, B, C,
                       label_smoothing);
    return (int)hipGetLastError();
}

}  // extern "C"
