// VQA MoE router + dispatch for gfx950 (reference router.py:287-366, moe_layer.py:122-173).
// Everything here is tiny (T = batch tokens, E <= 64 experts, D = 768): latency-bound, fp32 throughout so that the
// top-k indices match the fp32 reference bit-for-bit wherever its own margins allow.  The expensive part of the
// MoE -- the experts' weight-streaming GEMMs -- runs through gemm.hip on the gathered token rows.
#include "common.h"
#include "vqa_hip.h"

namespace {

constexpr int MAXE = 64;

// logits[t,e] = <x[t,:], gate[e,:]>  (+ noise[t,e] * softplus(<x[t,:], w_noise[e,:]>) * noise_std).  one wave per (t,e).
__global__ void gate_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gate, const float* __restrict__ w_noise,
                                const float* __restrict__ noise, float noise_std, float* __restrict__ clean, float* __restrict__ noisy,
                                float* __restrict__ nraw, int T, int E, int D) {
    const int lane = threadIdx.x & 63;
    const int pair = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (pair >= T * E) return;
    const int t = pair / E, e = pair % E;
    float a = 0.f, b = 0.f;
    for (int d = lane; d < D; d += 64) {
        const float xv = x[(size_t)t * D + d];
        a += xv * gate[(size_t)e * D + d];
        if (noise) b += xv * w_noise[(size_t)e * D + d];
    }
    a = wave_sum(a);
    if (noise) b = wave_sum(b);
    if (lane == 0) {
        clean[pair] = a;
        float l = a;
        if (noise) {
            nraw[pair] = b;
            const float sp = b > 20.f ? b : log1pf(__expf(b));           // F.softplus (threshold 20)
            l = a + noise[pair] * sp * noise_std;
        }
        noisy[pair] = l;
    }
}

// dgate[e,d] = sum_t dl[t,e] x[t,d];  dwn[e,d] = sum_t dn[t,e] x[t,d];  dx[t,d] = sum_e dl gate + dn w_noise
// where dn[t,e] = dl[t,e] * noise * noise_std * sigmoid(nraw).   grid: (ceil(D/256), E + T)
__global__ void gate_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gate, const float* __restrict__ w_noise,
                                const float* __restrict__ noise, float noise_std, const float* __restrict__ nraw,
                                const float* __restrict__ dl, float* __restrict__ dgate, float* __restrict__ dwn, float* __restrict__ dx,
                                int T, int E, int D) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    const int y = blockIdx.y;
    if (y < E) {
        const int e = y;
        float a = 0.f, b = 0.f;
        for (int t = 0; t < T; ++t) {
            const float g = dl[t * E + e], xv = x[(size_t)t * D + d];
            a += g * xv;
            if (noise) { const float r = nraw[t * E + e]; b += g * noise[t * E + e] * noise_std / (1.f + __expf(-r)) * xv; }
        }
        dgate[(size_t)e * D + d] = a;
        if (dwn) dwn[(size_t)e * D + d] = noise ? b : 0.f;
    } else {
        const int t = y - E;
        float a = 0.f;
        for (int e = 0; e < E; ++e) {
            const float g = dl[t * E + e];
            a += g * gate[(size_t)e * D + d];
            if (noise) { const float r = nraw[t * E + e]; a += g * noise[t * E + e] * noise_std / (1.f + __expf(-r)) * w_noise[(size_t)e * D + d]; }
        }
        dx[(size_t)t * D + d] = a;
    }
}

__global__ void topk_fwd_kernel(const float* __restrict__ logits, float* __restrict__ weights, int64_t* __restrict__ indices,
                                float* __restrict__ probs_all, int T, int E, int K) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    float p[MAXE];
    float m = -INFINITY;
    for (int e = 0; e < E; ++e) { p[e] = logits[t * E + e]; m = fmaxf(m, p[e]); }
    float s = 0.f;
    for (int e = 0; e < E; ++e) { p[e] = expf(p[e] - m); s += p[e]; }
    for (int e = 0; e < E; ++e) { p[e] /= s; if (probs_all) probs_all[t * E + e] = p[e]; }
    unsigned long long used = 0ull;
    float wsum = 0.f;
    float wk[16];
    for (int k = 0; k < K; ++k) {
        int best = -1; float bv = -1.f;
        for (int e = 0; e < E; ++e) if (!((used >> e) & 1ull) && p[e] > bv) { bv = p[e]; best = e; }   // lowest index wins ties
        if (best < 0) best = 0;
        used |= 1ull << best;
        indices[t * K + k] = best;
        wk[k] = bv; wsum += bv;
    }
    for (int k = 0; k < K; ++k) weights[t * K + k] = wk[k] / wsum;
}

__global__ void topk_bwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ indices, const float* __restrict__ dweights,
                                float* __restrict__ dlogits, int T, int E, int K) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    float p[MAXE], dp[MAXE];
    float m = -INFINITY;
    for (int e = 0; e < E; ++e) { p[e] = logits[t * E + e]; m = fmaxf(m, p[e]); dp[e] = 0.f; }
    float s = 0.f;
    for (int e = 0; e < E; ++e) { p[e] = expf(p[e] - m); s += p[e]; }
    for (int e = 0; e < E; ++e) p[e] /= s;
    float S = 0.f, dot = 0.f;
    for (int k = 0; k < K; ++k) S += p[indices[t * K + k]];
    for (int k = 0; k < K; ++k) dot += dweights[t * K + k] * p[indices[t * K + k]] / S;
    for (int k = 0; k < K; ++k) dp[indices[t * K + k]] = (dweights[t * K + k] - dot) / S;
    float pd = 0.f;
    for (int e = 0; e < E; ++e) pd += p[e] * dp[e];
    for (int e = 0; e < E; ++e) dlogits[t * E + e] = p[e] * (dp[e] - pd);
}

// one block per expert: combine weights + order-preserving token list (T <= 65536)
__global__ void expert_tokens_kernel(const float* __restrict__ weights, const int64_t* __restrict__ indices, int T, int K,
                                     float* __restrict__ w_all, int32_t* __restrict__ lists, int32_t* __restrict__ counts) {
    __shared__ int wave_cnt[4];
    __shared__ int base;
    const int e = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int t0 = 0; t0 < T; t0 += 256) {
        const int t = t0 + tid;
        float w = 0.f; bool hit = false;
        if (t < T) for (int k = 0; k < K; ++k) if (indices[(size_t)t * K + k] == e) { w += weights[(size_t)t * K + k]; hit = true; }
        if (t < T) w_all[(size_t)e * T + t] = w;
        const unsigned long long bal = __ballot(hit);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wave] = __popcll(bal);
        __syncthreads();
        int off = base;
        for (int w2 = 0; w2 < wave; ++w2) off += wave_cnt[w2];
        if (hit) lists[(size_t)e * T + off + before] = t;
        __syncthreads();
        if (tid == 0) base += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        __syncthreads();
    }
    if (tid == 0) counts[e] = base;
}

// out[list[i],:] += w_tok[list[i]] * y[i,:]
__global__ void scatter_add_kernel(const float* __restrict__ y, const int32_t* __restrict__ list, const float* __restrict__ w_tok,
                                   float* __restrict__ out, int n, int D) {
    const int d4 = D / 4;
    const size_t total = (size_t)n * d4, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const int i = (int)(t / d4), c = (int)(t % d4);
        const int tok = list[i];
        const f32x4 v = reinterpret_cast<const f32x4*>(y + (size_t)i * D)[c] * w_tok[tok];
        f32x4* o = reinterpret_cast<f32x4*>(out + (size_t)tok * D) + c;
        *o = *o + v;
    }
}

// dy[i,:] = w_tok[list[i]] * dout[list[i],:] ;  dw_tok[list[i]] = <dout[list[i],:], y[i,:]>     one wave per i
__global__ void combine_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ y, const int32_t* __restrict__ list,
                                   const float* __restrict__ w_tok, float* __restrict__ dy, h16_t* __restrict__ dyb,
                                   float* __restrict__ dw_tok, int n, int D) {
    const int lane = threadIdx.x & 63, i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= n) return;
    const int tok = list[i];
    const float w = w_tok[tok];
    float dot = 0.f;
    for (int d = lane; d < D; d += 64) {
        const float g = dout[(size_t)tok * D + d];
        dot += g * y[(size_t)i * D + d];
        if (dy) dy[(size_t)i * D + d] = w * g;
        if (dyb) dyb[(size_t)i * D + d] = (h16_t)(w * g);
    }
    dot = wave_sum(dot);
    if (lane == 0) dw_tok[tok] = dot;
}

// dweights[t,k] = dw_all[indices[t,k]][t]  (0 for indices outside [0,E): ablation's -1)
__global__ void route_weight_grad_kernel(const float* __restrict__ dw_all, const int64_t* __restrict__ indices, float* __restrict__ dweights,
                                         int T, int E, int K) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T * K) return;
    const int t = i / K;
    const int64_t e = indices[i];
    dweights[i] = (e >= 0 && e < E) ? dw_all[(size_t)e * T + t] : 0.f;
}

// load_balance_loss = w * E * sum_e (tokens_e / T) * mean_t probs[t,e]      (router.py:333-366), single block
__global__ void aux_loss_kernel(const float* __restrict__ probs, const int64_t* __restrict__ indices, int T, int E, int K, float weight,
                                float* __restrict__ out) {
    __shared__ float frac[MAXE], mp[MAXE];
    const int e = threadIdx.x;
    if (e < E) {
        float cnt = 0.f, pm = 0.f;
        for (int t = 0; t < T; ++t) {
            pm += probs[t * E + e];
            for (int k = 0; k < K; ++k) cnt += indices[t * K + k] == e ? 1.f : 0.f;
        }
        frac[e] = cnt / T; mp[e] = pm / T;
    }
    __syncthreads();
    if (threadIdx.x == 0) { float s = 0.f; for (int i = 0; i < E; ++i) s += frac[i] * mp[i]; out[0] = weight * E * s; }
}

__global__ void randn_kernel(float* __restrict__ out, uint64_t n, uint64_t seed, uint32_t stream) {
    seed = resolve_seed(seed);
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float u1 = (rng_u32(seed, stream, 2 * i) + 1.0f) * (1.0f / 4294967296.0f);     // (0,1]
        const float u2 = rng_u32(seed, stream, 2 * i + 1) * (1.0f / 4294967296.0f);
        out[i] = sqrtf(-2.f * __logf(u1)) * __cosf(6.283185307179586f * u2);
    }
}

__global__ void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, h16_t* __restrict__ yb, uint64_t n, float p, float inv_keep,
                               uint64_t seed, uint32_t stream) {
    if (p > 0.f) seed = resolve_seed(seed);
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float v = x[i] * (p > 0.f ? dropout_scale(seed, stream, i, p, inv_keep) : 1.f);
        if (y) y[i] = v;
        if (yb) yb[i] = (h16_t)v;
    }
}

}  // namespace

extern "C" {

int vqa_router_gate_fwd(const float* x, const float* gate, const float* w_noise, const float* noise, float noise_std, float* clean,
                        float* noisy, float* noise_raw, int T, int E, int D, vqa_stream_t s) {
    if (!x || !gate || !clean || !noisy || T <= 0 || E <= 0 || E > MAXE) return VQA_ERR_ARG;
    if (noise && (!w_noise || !noise_raw)) return VQA_ERR_ARG;
    hipLaunchKernelGGL(gate_fwd_kernel, dim3(ceil_div(T * E, 4)), dim3(256), 0, (hipStream_t)s, x, gate, w_noise, noise, noise_std, clean, noisy, noise_raw, T, E, D);
    return (int)hipGetLastError();
}

int vqa_router_gate_bwd(const float* x, const float* gate, const float* w_noise, const float* noise, float noise_std, const float* noise_raw,
                        const float* dlogits, float* dgate, float* dw_noise, float* dx, int T, int E, int D, vqa_stream_t s) {
    if (!x || !gate || !dlogits || !dgate || !dx) return VQA_ERR_ARG;
    hipLaunchKernelGGL(gate_bwd_kernel, dim3(ceil_div(D, 256), E + T), dim3(256), 0, (hipStream_t)s, x, gate, w_noise, noise, noise_std, noise_raw,
                       dlogits, dgate, dw_noise, dx, T, E, D);
    return (int)hipGetLastError();
}

int vqa_router_topk_fwd(const float* logits, float* weights, int64_t* indices, float* probs_all, int T, int E, int K, vqa_stream_t s) {
    if (!logits || !weights || !indices || T <= 0 || E <= 0 || E > MAXE || K <= 0 || K > E || K > 16) return VQA_ERR_ARG;
    hipLaunchKernelGGL(topk_fwd_kernel, dim3(ceil_div(T, 64)), dim3(64), 0, (hipStream_t)s, logits, weights, indices, probs_all, T, E, K);
    return (int)hipGetLastError();
}

int vqa_router_topk_bwd(const float* logits, const int64_t* indices, const float* dweights, float* dlogits, int T, int E, int K, vqa_stream_t s) {
    if (!logits || !indices || !dweights || !dlogits || E > MAXE || K > 16) return VQA_ERR_ARG;
    hipLaunchKernelGGL(topk_bwd_kernel, dim3(ceil_div(T, 64)), dim3(64), 0, (hipStream_t)s, logits, indices, dweights, dlogits, T, E, K);
    return (int)hipGetLastError();
}

int vqa_router_aux_loss(const float* probs, const int64_t* indices, int T, int E, int K, float weight, float* out, vqa_stream_t s) {
    if (!probs || !indices || !out || E > MAXE) return VQA_ERR_ARG;
    hipLaunchKernelGGL(aux_loss_kernel, dim3(1), dim3(64), 0, (hipStream_t)s, probs, indices, T, E, K, weight, out);
    return (int)hipGetLastError();
}

int vqa_moe_expert_tokens(const float* weights, const int64_t* indices, int T, int K, int E, float* w_all, int32_t* lists, int32_t* counts,
                          vqa_stream_t s) {
    if (!weights || !indices || !w_all || !lists || !counts || T <= 0 || E <= 0) return VQA_ERR_ARG;
    hipLaunchKernelGGL(expert_tokens_kernel, dim3(E), dim3(256), 0, (hipStream_t)s, weights, indices, T, K, w_all, lists, counts);
    return (int)hipGetLastError();
}

int vqa_moe_scatter_add(const float* y, const int32_t* list, const float* w_tok, float* out, int n, int D, vqa_stream_t s) {
    if (!y || !list || !w_tok || !out || D % 4) return VQA_ERR_ARG;
    if (n <= 0) return VQA_OK;
    size_t g = ((size_t)n * D / 4 + 255) / 256; if (g > 2048) g = 2048;
    hipLaunchKernelGGL(scatter_add_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)s, y, list, w_tok, out, n, D);
    return (int)hipGetLastError();
}

int vqa_moe_combine_bwd(const float* dout, const float* y, const int32_t* list, const float* w_tok, float* dy, void* dy_bf16,
                        float* dw_tok, int n, int D, vqa_stream_t s) {
    if (!dout || !y || !list || !w_tok || !dw_tok || (!dy && !dy_bf16)) return VQA_ERR_ARG;
    if (n <= 0) return VQA_OK;
    hipLaunchKernelGGL(combine_bwd_kernel, dim3(ceil_div(n, 4)), dim3(256), 0, (hipStream_t)s, dout, y, list, w_tok, dy, (h16_t*)dy_bf16, dw_tok, n, D);
    return (int)hipGetLastError();
}

int vqa_moe_route_weight_grad(const float* dw_all, const int64_t* indices, float* dweights, int T, int E, int K, vqa_stream_t s) {
    if (!dw_all || !indices || !dweights) return VQA_ERR_ARG;
    hipLaunchKernelGGL(route_weight_grad_kernel, dim3(ceil_div(T * K, 256)), dim3(256), 0, (hipStream_t)s, dw_all, indices, dweights, T, E, K);
    return (int)hipGetLastError();
}

int vqa_randn_f32(float* out, uint64_t n, uint64_t seed, uint32_t stream, vqa_stream_t s) {
    if (!out) return VQA_ERR_ARG;
    if (n == 0) return VQA_OK;
    size_t g = (n + 255) / 256; if (g > 1024) g = 1024;
    hipLaunchKernelGGL(randn_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)s, out, n, seed, stream);
    return (int)hipGetLastError();
}

int vqa_dropout_f32(const float* x, float* y, void* y_bf16, uint64_t n, float p, uint64_t seed, uint32_t stream, vqa_stream_t s) {
    if (!x || (!y && !y_bf16) || p < 0.f || p >= 1.f) return VQA_ERR_ARG;
    if (n == 0) return VQA_OK;
    size_t g = (n + 255) / 256; if (g > 2048) g = 2048;
    hipLaunchKernelGGL(dropout_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)s, x, y, (h16_t*)y_bf16, n, p, p > 0.f ? 1.f / (1.f - p) : 1.f, seed, stream);
    return (int)hipGetLastError();
}

}  // extern "C"
