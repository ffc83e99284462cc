// Small row kernels of the MoE expert runners (hip/expert_blocks.py) and the dense-dispatch combine.  All of them work on
// tensors of <= a few hundred rows (one fused token per sample, 4 mask tokens per sample): latency-bound launches whose job is
// to NOT be several launches -- each one replaces a chain of casts / fills / elementwise ops / column sums.
#include "common.h"
#include "vqa_hip.h"

namespace {

// out[m,n] = bf16(dy[m,n] * act'(pre[m,n]) * keep(m*N+n));  colsum[n] += sum_m out  (the bias gradient; pre-zeroed slot).
// Workgroup = 4 waves: lane -> 4 consecutive columns of a 256-column panel, wave w -> rows m0 + w, m0 + w + 4, ... of an RPB-row chunk, RPW rows
// per wave with ALL their loads issued before the first use (the first form walked 8 rows per thread one dependent load at a time and added
// every thread's column sums with its own atomics: 3648 x 768 took 23 us -- 350 000 atomics on 768 addresses -- and was 5 % of the generative
// model's step); the four waves' column sums meet in LDS and leave as ONE atomic per column and workgroup.
template <int RPW>
__global__ __launch_bounds__(256) void rows_mask_cast_kernel(const float* __restrict__ dy, int ld, const h16_t* __restrict__ pre, int act,
                                                             h16_t* __restrict__ outb, float* __restrict__ colsum, int M, int N, float p, float inv_keep,
                                                             uint64_t seed, uint32_t stream) {
    __shared__ f32x4 part[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c4 = blockIdx.x * 64 + lane;
    const bool cok = c4 * 4 < N;
    if (p > 0.f) seed = resolve_seed(seed);
    const int m0 = blockIdx.y * (4 * RPW) + wave;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (cok) {
        f32x4 v[RPW];
        h16x4 pv[RPW];
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int m = min(m0 + 4 * i, M - 1);                        // clamped rows are loaded and dropped
            v[i] = *reinterpret_cast<const f32x4*>(dy + (size_t)m * ld + 4 * c4);
            if (pre) pv[i] = *reinterpret_cast<const h16x4*>(pre + (size_t)m * N + 4 * c4);
        }
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int m = m0 + 4 * i;
            if (m < M) {
                f32x4 x = v[i];
                if (pre) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) x[r] *= act_bwd((float)pv[i][r], act);
                }
                if (p > 0.f) x *= dropout_scale4(seed, stream, (uint64_t)m * N + 4 * c4, p, inv_keep);
                h16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) { o[r] = (h16_t)x[r]; acc[r] += (float)o[r]; }
                *reinterpret_cast<h16x4*>(outb + (size_t)m * N + 4 * c4) = o;
            }
        }
    }
    if (colsum) {
        part[wave][lane] = acc;
        __syncthreads();
        if (wave == 0 && cok) {
            const f32x4 t = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
#pragma unroll
            for (int r = 0; r < 4; ++r) atomicAdd(colsum + 4 * c4 + r, t[r]);
        }
    }
}

// Attention over ONE key per sample: softmax of a single score is 1, so the context is V itself -- times the dropout keep-scale
// torch applies to the attention probabilities, one Bernoulli per (sample t, head h, query r), keyed like the attention
// kernels key element (b, h, q, kv = 0) and rounded to the operand type like their P matrix.
__device__ __forceinline__ float head_keep(uint64_t seed, uint32_t stream, int t, int h, int r, int H, int R, float p, float inv_keep) {
    if (p <= 0.f) return 1.f;
    return (float)(h16_t)dropout_scale(seed, stream, ((uint64_t)t * H + h) * R + r, p, inv_keep);
}
// out[(t*R + r), c] = v[t, c] * keep(t, c / Dh, r)
__global__ void head_keep_fwd_kernel(const h16_t* __restrict__ v, h16_t* __restrict__ out, int T, int R, int H, int Dh, float p, float inv_keep,
                                     uint64_t seed, uint32_t stream) {
    if (p > 0.f) seed = resolve_seed(seed);
    const int D = H * Dh, d4 = D / 4;
    const size_t total = (size_t)T * R * d4, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int c = (int)(i % d4) * 4, row = (int)(i / d4), t = row / R, r = row % R;
        const float ks = head_keep(seed, stream, t, c / Dh, r, H, R, p, inv_keep);
        const h16x4 x = *reinterpret_cast<const h16x4*>(v + (size_t)t * D + c);
        h16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (h16_t)((float)x[j] * ks);
        *reinterpret_cast<h16x4*>(out + (size_t)row * D + c) = o;
    }
}
// dv[t, c] = sum_r dout[(t*R + r), c] * keep(t, c / Dh, r)
__global__ void head_keep_bwd_kernel(const h16_t* __restrict__ dout, h16_t* __restrict__ dv, int T, int R, int H, int Dh, float p, float inv_keep,
                                     uint64_t seed, uint32_t stream) {
    if (p > 0.f) seed = resolve_seed(seed);
    const int D = H * Dh, d4 = D / 4;
    const size_t total = (size_t)T * d4, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int c = (int)(i % d4) * 4, t = (int)(i / d4);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < R; ++r) {
            const float ks = head_keep(seed, stream, t, c / Dh, r, H, R, p, inv_keep);
            const h16x4 x = *reinterpret_cast<const h16x4*>(dout + ((size_t)t * R + r) * D + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] += (float)x[j] * ks;
        }
        h16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (h16_t)acc[j];
        *reinterpret_cast<h16x4*>(dv + (size_t)t * D + c) = o;
    }
}

// dst row i = alpha * src row (mode 0: i / R -- every source row repeated R times; mode 1: i % R -- the R source rows tiled)
__global__ void repeat_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, h16_t* __restrict__ dstb, int rows, int D, int R, int mode, float alpha) {
    const int d4 = D / 4;
    const size_t total = (size_t)rows * d4, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int c = (int)(i % d4), row = (int)(i / d4), sr = mode ? row % R : row / R;
        const f32x4 v = reinterpret_cast<const f32x4*>(src + (size_t)sr * D)[c] * alpha;
        if (dst) reinterpret_cast<f32x4*>(dst + (size_t)row * D)[c] = v;
        if (dstb) {
            h16x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (h16_t)v[j];
            reinterpret_cast<h16x4*>(dstb + (size_t)row * D)[c] = o;
        }
    }
}

// out[t, :] = mean over the R rows t*R .. t*R+R-1 of x
__global__ void rows_mean_kernel(const float* __restrict__ x, int R, float* __restrict__ out, h16_t* __restrict__ outb, int ld_out, int T, int D) {
    const int d4 = D / 4;
    const size_t total = (size_t)T * d4, stride = (size_t)gridDim.x * blockDim.x;
    const float inv = 1.f / (float)R;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int c = (int)(i % d4), t = (int)(i / d4);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < R; ++r) acc += reinterpret_cast<const f32x4*>(x + ((size_t)t * R + r) * D)[c];
        acc *= inv;
        if (out) reinterpret_cast<f32x4*>(out + (size_t)t * ld_out)[c] = acc;
        if (outb) {
            h16x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (h16_t)acc[j];
            reinterpret_cast<h16x4*>(outb + (size_t)t * ld_out)[c] = o;
        }
    }
}

__global__ void take_stride_kernel(const h16_t* __restrict__ src, h16_t* __restrict__ dst, size_t n, int stride_el, int offset) {
    const size_t st = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += st) dst[i] = src[i * stride_el + offset];
}
__global__ void scatter_stride_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t n, int stride_el, int offset) {
    const size_t st = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += st) dst[i * stride_el + offset] = src[i];
}

// y = dropout(act(x)) for an activation that does not ride in a GEMM epilogue (Linear -> LayerNorm -> GELU -> Dropout of the
// stand-alone fusion modules); also writes the 16-bit copy of x the backward (vqa_act_drop_bwd) reads
__global__ void act_drop_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, h16_t* __restrict__ yb, h16_t* __restrict__ pre, size_t n, int act,
                                    float p, float inv_keep, uint64_t seed, uint32_t stream) {
    if (p > 0.f) seed = resolve_seed(seed);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float v0 = x[i];
        if (pre) pre[i] = (h16_t)v0;
        float v = act_fwd(v0, act);
        if (p > 0.f) v *= dropout_scale(seed, stream, i, p, inv_keep);
        if (y) y[i] = v;
        if (yb) yb[i] = (h16_t)v;
    }
}

// GatedLinearExpert (reference expert_types.py:501-504): y[t, j] = drop(h[t, j] * sigmoid(h[t, H + j])) over h [T, 2H] fp32
__global__ void glu_fwd_kernel(const float* __restrict__ h, float* __restrict__ y, int T, int H, float p, float inv_keep, uint64_t seed, uint32_t stream) {
    if (p > 0.f) seed = resolve_seed(seed);
    const size_t n = (size_t)T * H, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const size_t t = i / H, j = i % H;
        const float a = h[t * 2 * H + j], g = h[t * 2 * H + H + j];
        float v = a / (1.f + __expf(-g));
        if (p > 0.f) v *= dropout_scale(seed, stream, i, p, inv_keep);
        y[i] = v;
    }
}

// dh[t, j] = dy * keep * sigmoid(g);  dh[t, H + j] = dy * keep * a * sigmoid(g) * (1 - sigmoid(g))
__global__ void glu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ h, float* __restrict__ dh, int T, int H, float p, float inv_keep,
                               uint64_t seed, uint32_t stream) {
    if (p > 0.f) seed = resolve_seed(seed);
    const size_t n = (size_t)T * H, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const size_t t = i / H, j = i % H;
        const float a = h[t * 2 * H + j], g = h[t * 2 * H + H + j];
        const float sg = 1.f / (1.f + __expf(-g));
        float d = dy[i];
        if (p > 0.f) d *= dropout_scale(seed, stream, i, p, inv_keep);
        dh[t * 2 * H + j] = d * sg;
        dh[t * 2 * H + H + j] = d * a * sg * (1.f - sg);
    }
}

constexpr int MAX_E = 16;
struct PtrsC { const float* p[MAX_E]; };
struct PtrsM { float* p[MAX_E]; };

// out[t, :] = sum_e w[e, t] * y_e[t, :]      (w: [E, T], the layout vqa_moe_expert_tokens writes)
__global__ void dense_combine_fwd_kernel(PtrsC ys, const float* __restrict__ w, float* __restrict__ out, int T, int E, int D) {
    const int d4 = D / 4;
    const size_t total = (size_t)T * d4, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int c = (int)(i % d4), t = (int)(i / d4);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int e = 0; e < E; ++e) acc += reinterpret_cast<const f32x4*>(ys.p[e] + (size_t)t * D)[c] * w[(size_t)e * T + t];
        reinterpret_cast<f32x4*>(out + (size_t)t * D)[c] = acc;
    }
}
// dy_e[t, :] = w[e, t] * dout[t, :];   dw[e, t] = <dout[t, :], y_e[t, :]>.   One 256-thread workgroup per token.
__global__ __launch_bounds__(256) void dense_combine_bwd_kernel(const float* __restrict__ dout, PtrsC ys, const float* __restrict__ w, PtrsM dys,
                                                                float* __restrict__ dw, int T, int E, int D) {
    __shared__ float red[4][MAX_E];
    const int t = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, d4 = D / 4;
    float acc[MAX_E];
#pragma unroll
    for (int e = 0; e < MAX_E; ++e) acc[e] = 0.f;
    for (int c = threadIdx.x; c < d4; c += 256) {
        const f32x4 g = reinterpret_cast<const f32x4*>(dout + (size_t)t * D)[c];
#pragma unroll
        for (int e = 0; e < MAX_E; ++e) {
            if (e < E) {
                const f32x4 y = reinterpret_cast<const f32x4*>(ys.p[e] + (size_t)t * D)[c];
                acc[e] += g[0] * y[0] + g[1] * y[1] + g[2] * y[2] + g[3] * y[3];
                reinterpret_cast<f32x4*>(dys.p[e] + (size_t)t * D)[c] = g * w[(size_t)e * T + t];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < MAX_E; ++e) {
        if (e < E) {
            const float s = wave_sum(acc[e]);
            if (lane == 0) red[wave][e] = s;
        }
    }
    __syncthreads();
    if (threadIdx.x < E) dw[(size_t)threadIdx.x * T + t] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

inline int grid_of(size_t work, int tpb, int cap) {
    size_t g = (work + tpb - 1) / tpb;
    return (int)(g < 1 ? 1 : g > (size_t)cap ? cap : g);
}

}  // namespace

extern "C" {

int vqa_rows_mask_cast(const float* dy, int ld, const void* pre_bf16, int act, void* out_bf16, float* colsum, int M, int N, float p,
                       uint64_t seed, uint32_t stream, vqa_stream_t s) {
    if (!dy || !out_bf16 || M <= 0 || N <= 0 || N % 4 || ld % 4 || p < 0.f || p >= 1.f) return VQA_ERR_ARG;
    if (((uintptr_t)dy & 15) || ((uintptr_t)out_bf16 & 7) || (pre_bf16 && ((uintptr_t)pre_bf16 & 7))) return VQA_ERR_ARG;
    // rows per workgroup: 32 (8 per wave) once that still gives every CU a workgroup, else 16 / 8 (the experts' few rows: 4 x 2)
    const int cols = ceil_div(N / 4, 64);
    const float ik = p > 0.f ? 1.f / (1.f - p) : 1.f;
#define VQA_RMC(RPW) hipLaunchKernelGGL((rows_mask_cast_kernel<RPW>), dim3(cols, ceil_div(M, 4 * RPW)), dim3(256), 0, (hipStream_t)s, dy, ld, (const h16_t*)pre_bf16, act, \
                                        (h16_t*)out_bf16, colsum, M, N, p, ik, seed, stream)
    if ((long)cols * ceil_div(M, 32) >= 256) VQA_RMC(8);
    else if ((long)cols * ceil_div(M, 16) >= 128) VQA_RMC(4);
    else VQA_RMC(2);
#undef VQA_RMC
    return (int)hipGetLastError();
}

int vqa_head_keep_fwd(const void* v, void* out, int T, int R, int H, int Dh, float p, uint64_t seed, uint32_t stream, vqa_stream_t s) {
    if (!v || !out || T <= 0 || R <= 0 || H <= 0 || Dh % 4 || p < 0.f || p >= 1.f) return VQA_ERR_ARG;
    hipLaunchKernelGGL(head_keep_fwd_kernel, dim3(grid_of((size_t)T * R * H * Dh / 4, 256, 1024)), dim3(256), 0, (hipStream_t)s, (const h16_t*)v, (h16_t*)out,
                       T, R, H, Dh, p, p > 0.f ? 1.f / (1.f - p) : 1.f, seed, stream);
    return (int)hipGetLastError();
}

int vqa_head_keep_bwd(const void* dout, void* dv, int T, int R, int H, int Dh, float p, uint64_t seed, uint32_t stream, vqa_stream_t s) {
    if (!dout || !dv || T <= 0 || R <= 0 || H <= 0 || Dh % 4 || p < 0.f || p >= 1.f) return VQA_ERR_ARG;
    hipLaunchKernelGGL(head_keep_bwd_kernel, dim3(grid_of((size_t)T * H * Dh / 4, 256, 1024)), dim3(256), 0, (hipStream_t)s, (const h16_t*)dout, (h16_t*)dv,
                       T, R, H, Dh, p, p > 0.f ? 1.f / (1.f - p) : 1.f, seed, stream);
    return (int)hipGetLastError();
}

int vqa_repeat_rows_f32(const float* src, float* dst, void* dst_bf16, int out_rows, int D, int R, int mode, float alpha, vqa_stream_t s) {
    if (!src || (!dst && !dst_bf16) || out_rows <= 0 || D <= 0 || D % 4 || R <= 0 || (mode != 0 && mode != 1)) return VQA_ERR_ARG;
    hipLaunchKernelGGL(repeat_rows_kernel, dim3(grid_of((size_t)out_rows * D / 4, 256, 1024)), dim3(256), 0, (hipStream_t)s, src, dst, (h16_t*)dst_bf16,
                       out_rows, D, R, mode, alpha);
    return (int)hipGetLastError();
}

int vqa_rows_mean_f32(const float* x, int R, float* out, void* out_bf16, int ld_out, int T, int D, vqa_stream_t s) {
    if (!x || (!out && !out_bf16) || R <= 0 || T <= 0 || D <= 0 || D % 4 || ld_out % 4) return VQA_ERR_ARG;
    hipLaunchKernelGGL(rows_mean_kernel, dim3(grid_of((size_t)T * D / 4, 256, 1024)), dim3(256), 0, (hipStream_t)s, x, R, out, (h16_t*)out_bf16, ld_out, T, D);
    return (int)hipGetLastError();
}

int vqa_take_stride_bf16(const void* src, void* dst, size_t n, int stride, int offset, vqa_stream_t s) {
    if (!src || !dst || stride <= 0 || offset < 0 || offset >= stride) return VQA_ERR_ARG;
    if (n == 0) return VQA_OK;
    hipLaunchKernelGGL(take_stride_kernel, dim3(grid_of(n, 256, 4096)), dim3(256), 0, (hipStream_t)s, (const h16_t*)src, (h16_t*)dst, n, stride, offset);
    return (int)hipGetLastError();
}

int vqa_scatter_stride_f32(const float* src, float* dst, size_t n, int stride, int offset, vqa_stream_t s) {
    if (!src || !dst || stride <= 0 || offset < 0 || offset >= stride) return VQA_ERR_ARG;
    if (n == 0) return VQA_OK;
    hipLaunchKernelGGL(scatter_stride_kernel, dim3(grid_of(n, 256, 4096)), dim3(256), 0, (hipStream_t)s, src, dst, n, stride, offset);
    return (int)hipGetLastError();
}

int vqa_act_drop_fwd(const float* x, float* y, void* y_bf16, void* pre_bf16, size_t n, int act, float p, uint64_t seed, uint32_t stream, vqa_stream_t s) {
    if (!x || (!y && !y_bf16) || p < 0.f || p >= 1.f) return VQA_ERR_ARG;
    if (n == 0) return VQA_OK;
    hipLaunchKernelGGL(act_drop_fwd_kernel, dim3(grid_of(n, 256, 2048)), dim3(256), 0, (hipStream_t)s, x, y, (h16_t*)y_bf16, (h16_t*)pre_bf16, n, act, p,
                       p > 0.f ? 1.f / (1.f - p) : 1.f, seed, stream);
    return (int)hipGetLastError();
}

int vqa_glu_fwd(const float* h, float* y, int T, int H, float p, uint64_t seed, uint32_t stream, vqa_stream_t s) {
    if (!h || !y || T <= 0 || H <= 0 || p < 0.f || p >= 1.f) return VQA_ERR_ARG;
    hipLaunchKernelGGL(glu_fwd_kernel, dim3(grid_of((size_t)T * H, 256, 2048)), dim3(256), 0, (hipStream_t)s, h, y, T, H, p, p > 0.f ? 1.f / (1.f - p) : 1.f, seed, stream);
    return (int)hipGetLastError();
}

int vqa_glu_bwd(const float* dy, const float* h, float* dh, int T, int H, float p, uint64_t seed, uint32_t stream, vqa_stream_t s) {
    if (!dy || !h || !dh || T <= 0 || H <= 0 || p < 0.f || p >= 1.f) return VQA_ERR_ARG;
    hipLaunchKernelGGL(glu_bwd_kernel, dim3(grid_of((size_t)T * H, 256, 2048)), dim3(256), 0, (hipStream_t)s, dy, h, dh, T, H, p, p > 0.f ? 1.f / (1.f - p) : 1.f, seed, stream);
    return (int)hipGetLastError();
}

int vqa_moe_dense_combine_fwd(const float* const* ys, const float* w_all, float* out, int T, int E, int D, vqa_stream_t s) {
    if (!ys || !w_all || !out || T <= 0 || E <= 0 || E > MAX_E || D <= 0 || D % 4) return VQA_ERR_ARG;
    PtrsC p{};
    for (int e = 0; e < E; ++e) { if (!ys[e] || ((uintptr_t)ys[e] & 15)) return VQA_ERR_ARG; p.p[e] = ys[e]; }
    hipLaunchKernelGGL(dense_combine_fwd_kernel, dim3(grid_of((size_t)T * D / 4, 256, 1024)), dim3(256), 0, (hipStream_t)s, p, w_all, out, T, E, D);
    return (int)hipGetLastError();
}

int vqa_moe_dense_combine_bwd(const float* dout, const float* const* ys, const float* w_all, float* const* dys, float* dw_all, int T, int E, int D,
                              vqa_stream_t s) {
    if (!dout || !ys || !w_all || !dys || !dw_all || T <= 0 || E <= 0 || E > MAX_E || D <= 0 || D % 4) return VQA_ERR_ARG;
    PtrsC p{}; PtrsM q{};
    for (int e = 0; e < E; ++e) {
        if (!ys[e] || !dys[e] || (((uintptr_t)ys[e] | (uintptr_t)dys[e]) & 15)) return VQA_ERR_ARG;
        p.p[e] = ys[e]; q.p[e] = dys[e];
    }
    hipLaunchKernelGGL(dense_combine_bwd_kernel, dim3(T), dim3(256), 0, (hipStream_t)s, dout, p, w_all, q, dw_all, T, E, D);
    return (int)hipGetLastError();
}

}  // extern "C"
