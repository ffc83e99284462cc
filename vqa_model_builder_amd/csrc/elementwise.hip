// HBM-bound byte movers of the path: casts (weight shadows), column sums (bias gradients), CLIP patchify /
// token assembly.  All 16-byte vector accesses, grid-stride, <= 2048 blocks (cdna guide, guideline 11/13).
#include "common.h"
#include "vqa_hip.h"

namespace {

constexpr int TPB = 256;
inline int grid_for(size_t work_items) {
    size_t g = (work_items + TPB - 1) / TPB;
    return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

__global__ void cast_f32_bf16_kernel(const float* __restrict__ src, h16_t* __restrict__ dst, size_t n) {
    const size_t n8 = n / 8;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
        const f32x4 a = reinterpret_cast<const f32x4*>(src)[2 * i];
        const f32x4 b = reinterpret_cast<const f32x4*>(src)[2 * i + 1];
        h16x8 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) { o[j] = (h16_t)a[j]; o[4 + j] = (h16_t)b[j]; }
        reinterpret_cast<h16x8*>(dst)[i] = o;
    }
    for (size_t i = n8 * 8 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = (h16_t)src[i];
}

__global__ void cast_bf16_f32_kernel(const h16_t* __restrict__ src, float* __restrict__ dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = (float)src[i];
}

// one block-row (blockIdx.y) per job; blockIdx.x strides inside the job
__global__ void cast_multi_kernel(const VqaCastJob* __restrict__ jobs) {
    const VqaCastJob job = jobs[blockIdx.y];
    const size_t n = job.n, stride = (size_t)gridDim.x * blockDim.x;
    const size_t t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const float* src = job.src;
    const bool vec_ok = (((uintptr_t)src | (uintptr_t)job.dst) & 15) == 0;
    if (job.kind == 0) {
        h16_t* dst = (h16_t*)job.dst;
        size_t n8 = vec_ok ? n / 8 : 0;
        for (size_t i = t0; i < n8; i += stride) {
            const f32x4 a = reinterpret_cast<const f32x4*>(src)[2 * i];
            const f32x4 b = reinterpret_cast<const f32x4*>(src)[2 * i + 1];
            h16x8 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) { o[j] = (h16_t)a[j]; o[4 + j] = (h16_t)b[j]; }
            reinterpret_cast<h16x8*>(dst)[i] = o;
        }
        for (size_t i = n8 * 8 + t0; i < n; i += stride) dst[i] = (h16_t)src[i];
    } else {
        float* dst = (float*)job.dst;
        size_t n4 = vec_ok ? n / 4 : 0;
        for (size_t i = t0; i < n4; i += stride) reinterpret_cast<f32x4*>(dst)[i] = reinterpret_cast<const f32x4*>(src)[i];
        for (size_t i = n4 * 4 + t0; i < n; i += stride) dst[i] = src[i];
    }
}

// column sums: block = 64 columns x 4 row-groups... each thread owns one column, loops rows with stride 4*gridDim.y
template <typename T>
__global__ void colsum_kernel(const T* __restrict__ x, int M, int N, int ld, float* __restrict__ out) {
    __shared__ float red[4][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rg = threadIdx.x >> 6;
    float acc = 0.f;
    if (col < N) {
        for (int m = blockIdx.y * 4 + rg; m < M; m += 4 * gridDim.y) acc += (float)x[(size_t)m * ld + col];
    }
    red[rg][threadIdx.x & 63] = acc;
    __syncthreads();
    if (rg == 0 && col < N) {
        const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        if (gridDim.y == 1) out[col] = v; else atomicAdd(out + col, v);
    }
}

template <typename T>
int colsum_launch(const T* x, int M, int N, int ld, float* out, hipStream_t s) {
    if (!x || !out || M <= 0 || N <= 0) return VQA_ERR_ARG;
    const int gx = ceil_div(N, 64);
    int gy = 1;
    while (gx * gy < 512 && M / (gy * 2) >= 64) gy *= 2;
    if (gy > 1) { hipError_t e = hipMemsetAsync(out, 0, (size_t)N * 4, s); if (e != hipSuccess) return (int)e; }
    hipLaunchKernelGGL((colsum_kernel<T>), dim3(gx, gy), dim3(256), 0, s, x, M, N, ld, out);
    return (int)hipGetLastError();
}

__global__ void add_f32_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y,
                               h16_t* __restrict__ yb, size_t n) {
    const size_t n4 = n / 4, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const f32x4 v = reinterpret_cast<const f32x4*>(a)[i] + reinterpret_cast<const f32x4*>(b)[i];
        if (y) reinterpret_cast<f32x4*>(y)[i] = v;
        if (yb) { h16x4 o; for (int j = 0; j < 4; ++j) o[j] = (h16_t)v[j]; reinterpret_cast<h16x4*>(yb)[i] = o; }
    }
}

__global__ void gather_rows_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx, float* __restrict__ dst,
                                   h16_t* __restrict__ dstb, int n, int D, int ld_src) {
    const int d4 = D / 4;
    const size_t total = (size_t)n * d4, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const int i = (int)(t / d4), c = (int)(t % d4);
        const f32x4 v = reinterpret_cast<const f32x4*>(src + (size_t)idx[i] * ld_src)[c];
        if (dst) reinterpret_cast<f32x4*>(dst + (size_t)i * D)[c] = v;
        if (dstb) { h16x4 o; for (int j = 0; j < 4; ++j) o[j] = (h16_t)v[j]; reinterpret_cast<h16x4*>(dstb + (size_t)i * D)[c] = o; }
    }
}

// pixels [B,C,H,W] fp32 -> out [B*P, C*ps*ps] bf16, row (b, py, px), column (c, kh, kw).  One thread = 4 kw.
__global__ void patchify_kernel(const float* __restrict__ px, h16_t* __restrict__ out, int B, int C, int H, int W, int ps) {
    const int gw = W / ps, gh = H / ps, P = gw * gh, Kc = C * ps * ps, q4 = ps / 4;
    const size_t total = (size_t)B * P * C * ps * q4, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        size_t r = t;
        const int kw4 = (int)(r % q4); r /= q4;
        const int kh = (int)(r % ps); r /= ps;
        const int c = (int)(r % C); r /= C;
        const int p = (int)(r % P); const int b = (int)(r / P);
        const int py = p / gw, pxx = p % gw;
        const f32x4 v = *reinterpret_cast<const f32x4*>(px + (((size_t)b * C + c) * H + (py * ps + kh)) * W + pxx * ps + kw4 * 4);
        h16x4 o; for (int j = 0; j < 4; ++j) o[j] = (h16_t)v[j];
        *reinterpret_cast<h16x4*>(out + ((size_t)b * P + p) * Kc + (c * ps + kh) * ps + kw4 * 4) = o;
    }
}

__global__ void clip_assemble_kernel(const float* __restrict__ E, const float* __restrict__ cls, const float* __restrict__ pos,
                                     float* __restrict__ u, int B, int P, int D) {
    const int T = P + 1, d4 = D / 4;
    const size_t total = (size_t)B * T * d4, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const int c = (int)(t % d4); const size_t row = t / d4;
        const int tok = (int)(row % T), b = (int)(row / T);
        f32x4 v = tok == 0 ? reinterpret_cast<const f32x4*>(cls)[c]
                           : reinterpret_cast<const f32x4*>(E + ((size_t)b * P + tok - 1) * D)[c];
        v += reinterpret_cast<const f32x4*>(pos + (size_t)tok * D)[c];
        reinterpret_cast<f32x4*>(u + row * D)[c] = v;
    }
}

// dpos[tok,:] = sum_b du[b,tok,:]; dcls = dpos-like sum of tok 0; dE = bf16(du[b,1+p,:]).  grid = (T, ceil(D/256)), 64 threads x 4 columns each.
// The batch loop is unrolled by 8 with the loads of a group issued together (one dependent 4-byte load per sample and thread took 24 us for 5 MB);
// the sum over the batch stays sequential in b: deterministic.
__global__ __launch_bounds__(64) void clip_assemble_bwd_kernel(const float* __restrict__ du, h16_t* __restrict__ dE, float* __restrict__ dcls,
                                                              float* __restrict__ dpos, int B, int P, int D) {
    const int T = P + 1, tok = blockIdx.x, d = (blockIdx.y * 64 + threadIdx.x) * 4;
    if (d >= D) return;                                   // D % 4 == 0 (host)
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int b0 = 0; b0 < B; b0 += 8) {
        f32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const f32x4*>(du + ((size_t)min(b0 + j, B - 1) * T + tok) * D + d);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (b0 + j < B) {
                acc += v[j];
                if (tok > 0) {
                    h16x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (h16_t)v[j][r];
                    *reinterpret_cast<h16x4*>(dE + ((size_t)(b0 + j) * P + tok - 1) * D + d) = o;
                }
            }
        }
    }
    *reinterpret_cast<f32x4*>(dpos + (size_t)tok * D + d) = acc;
    if (tok == 0) *reinterpret_cast<f32x4*>(dcls + d) = acc;
}

// dpre = dy * act'(pre) * dropmask(idx)   (backward of y = drop(act(pre)) when it is not fused into a GEMM epilogue)
__global__ void act_drop_bwd_kernel(const float* __restrict__ dy, const h16_t* __restrict__ pre, int act, float* __restrict__ out,
                                    h16_t* __restrict__ outb, size_t n, float p, float inv_keep, uint64_t seed, uint32_t stream) {
    if (p > 0.f) seed = resolve_seed(seed);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float v = dy[i];
        if (pre) v *= act_bwd((float)pre[i], act);
        if (p > 0.f) v *= dropout_scale(seed, stream, i, p, inv_keep);
        if (out) out[i] = v;
        if (outb) outb[i] = (h16_t)v;
    }
}

__global__ void sumsq_kernel(const float* __restrict__ x, uint64_t n, float* __restrict__ out) {
    __shared__ float red[4];
    float acc = 0.f;
    const uint64_t n4 = n / 4, stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
        acc += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    for (uint64_t i = n4 * 4 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) acc += x[i] * x[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

__global__ void adamw_kernel(const VqaAdamWDesc d) {
    const float gs = d.grad_scale ? *d.grad_scale : 1.f;
    const float step_size = d.lr / d.bias_correction1;
    const float inv_sqrt_bc2 = rsqrtf(d.bias_correction2);
    const uint64_t n4 = d.n / 4, stride = (uint64_t)gridDim.x * blockDim.x;
    h16_t* shadow = (h16_t*)d.param_bf16;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        f32x4 p = reinterpret_cast<f32x4*>(d.param)[i];
        const f32x4 g = reinterpret_cast<const f32x4*>(d.grad)[i] * gs;
        f32x4 m = reinterpret_cast<f32x4*>(d.exp_avg)[i];
        f32x4 v = reinterpret_cast<f32x4*>(d.exp_avg_sq)[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            p[j] *= 1.f - d.lr * d.weight_decay;
            m[j] = d.beta1 * m[j] + (1.f - d.beta1) * g[j];
            v[j] = d.beta2 * v[j] + (1.f - d.beta2) * g[j] * g[j];
            const float denom = sqrtf(v[j]) * inv_sqrt_bc2 + d.eps;
            p[j] -= step_size * (m[j] / denom);
        }
        reinterpret_cast<f32x4*>(d.param)[i] = p;
        reinterpret_cast<f32x4*>(d.exp_avg)[i] = m;
        reinterpret_cast<f32x4*>(d.exp_avg_sq)[i] = v;
        if (shadow) { h16x4 o; for (int j = 0; j < 4; ++j) o[j] = (h16_t)p[j]; reinterpret_cast<h16x4*>(shadow)[i] = o; }
    }
    for (uint64_t i = n4 * 4 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < d.n; i += stride) {
        float p = d.param[i] * (1.f - d.lr * d.weight_decay);
        const float g = d.grad[i] * gs;
        const float m = d.beta1 * d.exp_avg[i] + (1.f - d.beta1) * g;
        const float v = d.beta2 * d.exp_avg_sq[i] + (1.f - d.beta2) * g * g;
        p -= step_size * (m / (sqrtf(v) * inv_sqrt_bc2 + d.eps));
        d.param[i] = p; d.exp_avg[i] = m; d.exp_avg_sq[i] = v;
        if (shadow) shadow[i] = (h16_t)p;
    }
}

// ---- multi-tensor optimiser: one launch over a device table of per-tensor jobs, split into 64K-element chunks -------
// chunks[c] = {job index, first element}: one workgroup per chunk, so a 49 M-element embedding table and a 768-element
// bias both keep the whole chip streaming (a per-tensor grid would leave the big tensors to a handful of workgroups).
constexpr uint32_t OPT_CHUNK = 65536;

// Gradients in the data-parallel WIRE format (VQA_OPT_GRAD_BF16 in VqaOptJob::shadow_kind): ``grad`` then points at bfloat16 values -- the
// all-reduced sums as RCCL left them in the exchange's staging buffer -- whatever the library's own 16-bit operand type is.
__device__ __forceinline__ float wire_bf16(uint16_t b) { return __uint_as_float((uint32_t)b << 16); }
__device__ __forceinline__ f32x4 wire_bf16x4(const void* base, uint64_t i) {
    const u16x4 b = *reinterpret_cast<const u16x4*>(reinterpret_cast<const uint16_t*>(base) + i);
    return (f32x4){wire_bf16(b[0]), wire_bf16(b[1]), wire_bf16(b[2]), wire_bf16(b[3])};
}
__global__ void sumsq_multi_kernel(const VqaOptJob* __restrict__ jobs, const uint32_t* __restrict__ chunks, float* __restrict__ norm2) {
    __shared__ float red[4];
    const VqaOptJob j = jobs[chunks[2 * blockIdx.x]];
    const uint64_t beg = chunks[2 * blockIdx.x + 1];
    const uint64_t end = beg + OPT_CHUNK < j.n ? beg + OPT_CHUNK : j.n;
    const float* g = j.grad;
    const bool g16 = (j.shadow_kind & VQA_OPT_GRAD_BF16) != 0;
    float acc = 0.f;
    const uint64_t e4 = beg + (end - beg) / 4 * 4;
    for (uint64_t i = beg + 4 * threadIdx.x; i < e4; i += 4 * blockDim.x) {
        const f32x4 v = g16 ? wire_bf16x4(g, i) : *reinterpret_cast<const f32x4*>(g + i);
        acc += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    for (uint64_t i = e4 + threadIdx.x; i < end; i += blockDim.x) { const float t = g16 ? wire_bf16(reinterpret_cast<const uint16_t*>(g)[i]) : g[i]; acc += t * t; }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { const float t = red[0] + red[1] + red[2] + red[3]; if (t != 0.f) atomicAdd(norm2, t); }
}

// AdamW over every chunk; clip coefficient min(1, max_norm / (sqrt(norm2) + 1e-6)) computed on device
// (torch.nn.utils.clip_grad_norm_ semantics); optionally refreshes the bf16 (or packed fp32) shadow of the parameter.
// hyper (optional, device): {lr, step}: read at run time so a captured graph follows the schedule and the step count.
__global__ void adamw_multi_kernel(const VqaOptJob* __restrict__ jobs, const uint32_t* __restrict__ chunks, const float* __restrict__ norm2,
                                   float max_norm, float lr, float beta1, float beta2, float eps, float bc1, float bc2,
                                   const float* __restrict__ hyper, float prescale, const float* __restrict__ amp) {
    if (amp) {
        // fp16 mode: the gradients in memory are loss_scale x the true ones (amp[0], GradScaler's role); norm2 is the sum of
        // squares of those scaled values: a non-finite one anywhere makes it non-finite and the whole step is skipped, exactly
        // as GradScaler.step() skips optimizer.step() (reference training_pipeline.py:495-502)
        if (!isfinite(norm2[0])) return;
        prescale /= amp[0];
    }
    if (hyper) {
        lr = hyper[0];
        const float t = hyper[1];
        bc1 = 1.f - powf(beta1, t);
        bc2 = 1.f - powf(beta2, t);
    }
    const unsigned cb = blockIdx.x;      // (walking the chunks back to front, to start where the norm pass ended in the Infinity Cache: no change)
    const VqaOptJob j = jobs[chunks[2 * cb]];
    if (j.active && j.active[0] == 0.f) return;          // expert no token was routed to: skipped like a grad-is-None parameter
    if (j.own_step) {                                    // ... and its bias corrections follow the number of updates IT received (torch keeps
        const float t = fmaxf(j.own_step[0], 1.f);       // `step` per parameter: a skipped step does not age the moments' correction)
        bc1 = 1.f - powf(beta1, t);
        bc2 = 1.f - powf(beta2, t);
    }
    const uint64_t beg = chunks[2 * cb + 1];
    const uint64_t end = beg + OPT_CHUNK < j.n ? beg + OPT_CHUNK : j.n;
    // prescale: the gradients in memory are SUMS over `1/prescale` data-parallel ranks; their mean is never materialised
    float gs = prescale;
    if (norm2 && max_norm > 0.f) { const float c = max_norm / (prescale * sqrtf(norm2[0]) + 1e-6f); gs = c < 1.f ? c * prescale : prescale; }
    const float step_size = lr / bc1, inv_sqrt_bc2 = rsqrtf(bc2), decay = 1.f - lr * j.weight_decay;
    const uint64_t e4 = beg + (end - beg) / 4 * 4;
    const bool g16 = (j.shadow_kind & VQA_OPT_GRAD_BF16) != 0;
    const uint32_t skind = j.shadow_kind & 0xffu;
#ifndef VQA_ADAMW_NT
#define VQA_ADAMW_NT 1
#endif
#if VQA_ADAMW_NT                      // streaming hints: every byte of this pass is touched exactly once (measured: profiles/r02/adamw_variants.log)
#define VQA_LD4(ptr) __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(ptr))
#define VQA_ST4(ptr, val) __builtin_nontemporal_store((val), reinterpret_cast<f32x4*>(ptr))
#else
#define VQA_LD4(ptr) (*reinterpret_cast<const f32x4*>(ptr))
#define VQA_ST4(ptr, val) (*reinterpret_cast<f32x4*>(ptr) = (val))
#endif
#ifndef VQA_ADAMW_UNROLL
#define VQA_ADAMW_UNROLL 1
#endif
#pragma unroll VQA_ADAMW_UNROLL
    for (uint64_t i = beg + 4 * threadIdx.x; i < e4; i += 4 * blockDim.x) {
        f32x4 p = VQA_LD4(j.param + i);
        const f32x4 g = (g16 ? wire_bf16x4(j.grad, i) : VQA_LD4(j.grad + i)) * gs;
        if (j.touched) {
            // One wave's 64 x 4 elements of an iteration are one 256-element granule (chunks start at multiples of 65536): a granule whose
            // moments are still exactly zero (byte 0) and whose gradient is all zero now gets what the full update would give it -- p * decay
            // - step_size * (0 / (0 + eps)) = p * decay, moments 0 -> 0 -- without touching the moments.  Wave-uniform, decided from values.
            const uint64_t gran = i >> 8;
            const bool nz = (g[0] != 0.f) | (g[1] != 0.f) | (g[2] != 0.f) | (g[3] != 0.f);
            const bool any_nz = __builtin_amdgcn_ballot_w64(nz) != 0;
            const bool was = j.touched[gran] != 0;
            if (!was && !any_nz) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { p[k] *= decay; p[k] -= step_size * (0.f / (0.f * inv_sqrt_bc2 + eps)); }
                VQA_ST4(j.param + i, p);
                if (j.shadow) {
                    if (skind == 0) { h16x4 o; for (int k = 0; k < 4; ++k) o[k] = (h16_t)p[k]; *reinterpret_cast<h16x4*>((h16_t*)j.shadow + i) = o; }
                    else *reinterpret_cast<f32x4*>((float*)j.shadow + i) = p;
                }
                continue;
            }
            if (!was && (threadIdx.x & 63) == __builtin_ctzll(__builtin_amdgcn_ballot_w64(true))) j.touched[gran] = 1;
        }
        f32x4 m = VQA_LD4(j.exp_avg + i);
        f32x4 v = VQA_LD4(j.exp_avg_sq + i);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            p[k] *= decay;
            m[k] = beta1 * m[k] + (1.f - beta1) * g[k];
            v[k] = beta2 * v[k] + (1.f - beta2) * g[k] * g[k];
            p[k] -= step_size * (m[k] / (sqrtf(v[k]) * inv_sqrt_bc2 + eps));
        }
        VQA_ST4(j.param + i, p);
        VQA_ST4(j.exp_avg + i, m);
        VQA_ST4(j.exp_avg_sq + i, v);
        if (j.shadow) {
            if (skind == 0) { h16x4 o; for (int k = 0; k < 4; ++k) o[k] = (h16_t)p[k]; *reinterpret_cast<h16x4*>((h16_t*)j.shadow + i) = o; }
            else *reinterpret_cast<f32x4*>((float*)j.shadow + i) = p;
        }
    }
    for (uint64_t i = e4 + threadIdx.x; i < end; i += blockDim.x) {
        float p = j.param[i] * decay;
        const float g = (g16 ? wire_bf16(reinterpret_cast<const uint16_t*>(j.grad)[i]) : j.grad[i]) * gs;
        const float m = beta1 * j.exp_avg[i] + (1.f - beta1) * g;
        const float v = beta2 * j.exp_avg_sq[i] + (1.f - beta2) * g * g;
        p -= step_size * (m / (sqrtf(v) * inv_sqrt_bc2 + eps));
        j.param[i] = p; j.exp_avg[i] = m; j.exp_avg_sq[i] = v;
        if (j.shadow) { if (skind == 0) ((h16_t*)j.shadow)[i] = (h16_t)p; else ((float*)j.shadow)[i] = p; }
    }
}

// GradScaler._amp_update_scale_ semantics on the device (one thread): amp = {scale, growth_tracker, found_inf of this step}.
__global__ void amp_update_kernel(float* __restrict__ amp, const float* __restrict__ norm2, float growth, float backoff, int interval) {
    const bool found = !isfinite(norm2[0]);
    float scale = amp[0], tracker = amp[1];
    if (found) { scale *= backoff; tracker = 0.f; }
    else { tracker += 1.f; if (tracker >= (float)interval) { scale *= growth; tracker = 0.f; } }
    amp[0] = scale; amp[1] = tracker; amp[2] = found ? 1.f : 0.f;
}
// device-side step count of a parameter group ({lr, step}): advances unless this step's gradients were non-finite
__global__ void opt_advance_kernel(float* __restrict__ hyper, const float* __restrict__ norm2) {
    if (!norm2 || isfinite(norm2[0])) hyper[1] += 1.f;
}
// per-expert step counts: an expert's count advances in the steps in which a token was routed to it
__global__ void opt_advance_counts_kernel(float* __restrict__ steps, const float* __restrict__ active, int n, const float* __restrict__ norm2) {
    const int i = threadIdx.x;
    if (i < n && active[i] > 0.f && (!norm2 || isfinite(norm2[0]))) steps[i] += 1.f;
}

// ---- nn.Bilinear as a GEMM: z[b, i*D2 + j] = x1[b,i] * x2[b,j] (bf16 operand of y = z W^T), and the contraction of dz back ----
__global__ void outer_bf16_kernel(const float* __restrict__ x1, const float* __restrict__ x2, h16_t* __restrict__ z, int B, int D1, int D2) {
    const size_t n = (size_t)B * D1 * D2 / 4, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += stride) {
        const size_t e = q * 4;
        const int j = (int)(e % D2), i = (int)((e / D2) % D1), b = (int)(e / ((size_t)D1 * D2));
        const float a = x1[(size_t)b * D1 + i];
        const f32x4 v = *reinterpret_cast<const f32x4*>(x2 + (size_t)b * D2 + j);
        h16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (h16_t)(a * v[r]);
        *reinterpret_cast<h16x4*>(z + e) = o;
    }
}
// dx1[b,i] = sum_j dz[b,i,j] x2[b,j]: one wave per (b,i) row of dz
__global__ void outer_bwd_x1_kernel(const float* __restrict__ dz, const float* __restrict__ x2, float* __restrict__ dx1, int rows, int D1, int D2) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int b = row / D1;
    float acc = 0.f;
    for (int j = 4 * lane; j < D2; j += 256) {
        const f32x4 d = *reinterpret_cast<const f32x4*>(dz + (size_t)row * D2 + j);
        const f32x4 v = *reinterpret_cast<const f32x4*>(x2 + (size_t)b * D2 + j);
        acc += d[0] * v[0] + d[1] * v[1] + d[2] * v[2] + d[3] * v[3];
    }
    acc = wave_sum(acc);
    if (lane == 0) dx1[row] = acc;
}
// dx2[b,j] = sum_i dz[b,i,j] x1[b,i]: thread per column j, coalesced walk down the D1 rows
__global__ void outer_bwd_x2_kernel(const float* __restrict__ dz, const float* __restrict__ x1, float* __restrict__ dx2, int D1, int D2) {
    const int b = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= D2) return;
    float acc = 0.f;
    for (int i = 0; i < D1; ++i) acc += dz[((size_t)b * D1 + i) * D2 + j] * x1[(size_t)b * D1 + i];
    dx2[(size_t)b * D2 + j] = acc;
}

}  // namespace

// Weight prefetch: ONE streaming reader pulls a span of 16-bit weights through the memory-side cache ahead of the GEMMs that will read it.
// Between two uses of a weight the step moves 440 MB of weights and 7 GB of optimiser traffic, so every GEMM finds its weights in HBM, and its
// eight XCDs all miss on the same lines at the same time: +2 - 4 us per launch (profiles/r02/gemm_cold_weights.log).  Read once by this kernel,
// on a stream of its own while the previous layer computes, the lines wait in the 256-MiB Infinity Cache.
__device__ unsigned g_prefetch_sink;
__global__ void prefetch_kernel(const u32x4* __restrict__ p, size_t n16) {
    u32x4 acc = {0u, 0u, 0u, 0u};
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 7 * stride < n16; i += 8 * stride) {
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= p[i + u * stride];
    }
    for (; i < n16; i += stride) acc ^= p[i];
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x9e3779b9u) g_prefetch_sink = acc[0];      // keeps the loads; practically never taken
}

extern "C" {

int vqa_abi_version(void) { return 6; }   // 6 (round 3): VqaOptJob::touched, slotted sumsq accumulators of vqa_gemm_bf16_grouped2, vqa_set_gemm_v1_fast, the persistent-loop knobs gone; 5: vqa_gemm_profile_collect2 (algorithmic bytes), vqa_gemm_bf16_grouped2, vqa_set_gemm_dw256; 4: VQA_OPT_GRAD_BF16 job flag, vqa_prefetch, vqa_set_gemm_k_rotate / _tile_order
int vqa_half_kind(void) { return VQA_HALF_KIND; }
int vqa_outer_bf16(const float* x1, const float* x2, void* z_bf16, int B, int D1, int D2, vqa_stream_t s) {
    if (!x1 || !x2 || !z_bf16 || B <= 0 || D1 <= 0 || D2 <= 0 || D2 % 4) return VQA_ERR_ARG;
    const size_t n4 = (size_t)B * D1 * D2 / 4;
    hipLaunchKernelGGL(outer_bf16_kernel, dim3(grid_for(n4)), dim3(TPB), 0, (hipStream_t)s, x1, x2, (h16_t*)z_bf16, B, D1, D2);
    return (int)hipGetLastError();
}
int vqa_outer_bwd(const float* dz, const float* x1, const float* x2, float* dx1, float* dx2, int B, int D1, int D2, vqa_stream_t s) {
    if (!dz || !x1 || !x2 || !dx1 || !dx2 || B <= 0 || D1 <= 0 || D2 <= 0 || D2 % 4) return VQA_ERR_ARG;
    hipLaunchKernelGGL(outer_bwd_x1_kernel, dim3(ceil_div(B * D1, 4)), dim3(256), 0, (hipStream_t)s, dz, x2, dx1, B * D1, D1, D2);
    hipLaunchKernelGGL(outer_bwd_x2_kernel, dim3(ceil_div(D2, 256), B), dim3(256), 0, (hipStream_t)s, dz, x1, dx2, D1, D2);
    return (int)hipGetLastError();
}


int vqa_opt_chunk_elems(void) { return (int)OPT_CHUNK; }

int vqa_sumsq_multi(const VqaOptJob* jobs_dev, const uint32_t* chunks_dev, int nchunks, float* norm2, vqa_stream_t s) {
    if (!jobs_dev || !chunks_dev || nchunks <= 0 || !norm2) return VQA_ERR_ARG;
    hipLaunchKernelGGL(sumsq_multi_kernel, dim3(nchunks), dim3(TPB), 0, (hipStream_t)s, jobs_dev, chunks_dev, norm2);
    return (int)hipGetLastError();
}

int vqa_adamw_multi(const VqaOptJob* jobs_dev, const uint32_t* chunks_dev, int nchunks, const float* norm2, float max_norm, float lr,
                    float beta1, float beta2, float eps, float bias_correction1, float bias_correction2, const float* hyper_dev,
                    float grad_prescale, const float* amp_dev, vqa_stream_t s) {
    if (!jobs_dev || !chunks_dev || nchunks <= 0) return VQA_ERR_ARG;
    if (amp_dev && !norm2) return VQA_ERR_ARG;            // the inf check rides on the norm
    hipLaunchKernelGGL(adamw_multi_kernel, dim3(nchunks), dim3(TPB), 0, (hipStream_t)s, jobs_dev, chunks_dev, norm2, max_norm, lr, beta1, beta2,
                       eps, bias_correction1, bias_correction2, hyper_dev, grad_prescale == 0.f ? 1.f : grad_prescale, amp_dev);
    return (int)hipGetLastError();
}
int vqa_amp_update(float* amp_dev, const float* norm2, float growth_factor, float backoff_factor, int growth_interval, vqa_stream_t s) {
    if (!amp_dev || !norm2 || growth_interval < 1) return VQA_ERR_ARG;
    hipLaunchKernelGGL(amp_update_kernel, dim3(1), dim3(1), 0, (hipStream_t)s, amp_dev, norm2, growth_factor, backoff_factor, growth_interval);
    return (int)hipGetLastError();
}
int vqa_opt_advance_counts(float* steps, const float* active, int n, const float* norm2, vqa_stream_t s) {
    if (!steps || !active || n <= 0 || n > 1024) return VQA_ERR_ARG;
    hipLaunchKernelGGL(opt_advance_counts_kernel, dim3(1), dim3((n + 63) / 64 * 64), 0, (hipStream_t)s, steps, active, n, norm2);
    return (int)hipGetLastError();
}
int vqa_opt_advance(float* hyper_dev, const float* norm2, vqa_stream_t s) {
    if (!hyper_dev) return VQA_ERR_ARG;
    hipLaunchKernelGGL(opt_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)s, hyper_dev, norm2);
    return (int)hipGetLastError();
}

int vqa_cast_f32_bf16(const float* src, void* dst, size_t n, vqa_stream_t s) {
    if (!src || !dst) return VQA_ERR_ARG;
    if (n == 0) return VQA_OK;
    if (((uintptr_t)src | (uintptr_t)dst) & 15) return VQA_ERR_ARG;
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for(n / 8 + 1)), dim3(TPB), 0, (hipStream_t)s, src, (h16_t*)dst, n);
    return (int)hipGetLastError();
}

int vqa_cast_bf16_f32(const void* src, float* dst, size_t n, vqa_stream_t s) {
    if (!src || !dst) return VQA_ERR_ARG;
    if (n == 0) return VQA_OK;
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(grid_for(n)), dim3(TPB), 0, (hipStream_t)s, (const h16_t*)src, dst, n);
    return (int)hipGetLastError();
}

int vqa_cast_multi(const VqaCastJob* jobs_dev, int njobs, uint64_t max_n, vqa_stream_t s) {
    if (!jobs_dev || njobs <= 0) return VQA_ERR_ARG;
    int gx = (int)((max_n / 8 + TPB - 1) / TPB);
    gx = gx < 1 ? 1 : (gx > 64 ? 64 : gx);
    hipLaunchKernelGGL(cast_multi_kernel, dim3(gx, njobs), dim3(TPB), 0, (hipStream_t)s, jobs_dev);
    return (int)hipGetLastError();
}

int vqa_colsum_bf16(const void* x, int M, int N, int ld, float* out, vqa_stream_t s) {
    return colsum_launch<h16_t>((const h16_t*)x, M, N, ld, out, (hipStream_t)s);
}
int vqa_colsum_f32(const float* x, int M, int N, int ld, float* out, vqa_stream_t s) {
    return colsum_launch<float>(x, M, N, ld, out, (hipStream_t)s);
}

int vqa_prefetch(const void* p, size_t nbytes, int workgroups, vqa_stream_t s) {
    if (!p || ((uintptr_t)p & 15)) return VQA_ERR_ARG;
    if (nbytes < 16) return VQA_OK;
    hipLaunchKernelGGL(prefetch_kernel, dim3(workgroups > 0 ? workgroups : 64), dim3(TPB), 0, (hipStream_t)s, (const u32x4*)p, nbytes / 16);
    return (int)hipGetLastError();
}

int vqa_add_f32(const float* a, const float* b, float* y, void* y_bf16, size_t n, vqa_stream_t s) {
    if (!a || !b || (!y && !y_bf16) || n % 4) return VQA_ERR_ARG;
    if (n == 0) return VQA_OK;
    hipLaunchKernelGGL(add_f32_kernel, dim3(grid_for(n / 4)), dim3(TPB), 0, (hipStream_t)s, a, b, y, (h16_t*)y_bf16, n);
    return (int)hipGetLastError();
}

int vqa_gather_rows_f32(const float* src, const int32_t* idx, float* dst, void* dst_bf16, int n, int D, int ld_src, vqa_stream_t s) {
    if (!src || !idx || (!dst && !dst_bf16) || D % 4 || ld_src % 4 || n < 0) return VQA_ERR_ARG;
    if (n == 0) return VQA_OK;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for((size_t)n * D / 4)), dim3(TPB), 0, (hipStream_t)s, src, idx, dst,
                       (h16_t*)dst_bf16, n, D, ld_src);
    return (int)hipGetLastError();
}

int vqa_patchify_bf16(const float* pixels, void* out, int B, int C, int H, int W, int ps, vqa_stream_t s) {
    if (!pixels || !out || B <= 0 || ps % 4 || H % ps || W % ps || W % 4) return VQA_ERR_ARG;
    const size_t total = (size_t)B * C * H * W / 4;
    hipLaunchKernelGGL(patchify_kernel, dim3(grid_for(total)), dim3(TPB), 0, (hipStream_t)s, pixels, (h16_t*)out, B, C, H, W, ps);
    return (int)hipGetLastError();
}

int vqa_clip_assemble(const float* E, const float* cls, const float* pos, float* u, int B, int P, int D, vqa_stream_t s) {
    if (!E || !cls || !pos || !u || D % 4) return VQA_ERR_ARG;
    hipLaunchKernelGGL(clip_assemble_kernel, dim3(grid_for((size_t)B * (P + 1) * D / 4)), dim3(TPB), 0, (hipStream_t)s, E, cls, pos, u, B, P, D);
    return (int)hipGetLastError();
}

int vqa_clip_assemble_bwd(const float* du, void* dE_bf16, float* dcls, float* dpos, int B, int P, int D, vqa_stream_t s) {
    if (!du || !dE_bf16 || !dcls || !dpos || D % 4 || (((uintptr_t)du | (uintptr_t)dcls | (uintptr_t)dpos) & 15) || ((uintptr_t)dE_bf16 & 7)) return VQA_ERR_ARG;
    hipLaunchKernelGGL(clip_assemble_bwd_kernel, dim3(P + 1, ceil_div(D, 256)), dim3(64), 0, (hipStream_t)s, du, (h16_t*)dE_bf16, dcls, dpos, B, P, D);
    return (int)hipGetLastError();
}

int vqa_act_drop_bwd(const float* dy, const void* pre_bf16, int act, float* out, void* out_bf16, size_t n, float p, uint64_t seed,
                     uint32_t stream, vqa_stream_t s) {
    if (!dy || (!out && !out_bf16) || p < 0.f || p >= 1.f) return VQA_ERR_ARG;
    if (n == 0) return VQA_OK;
    hipLaunchKernelGGL(act_drop_bwd_kernel, dim3(grid_for(n)), dim3(TPB), 0, (hipStream_t)s, dy, (const h16_t*)pre_bf16, act, out,
                       (h16_t*)out_bf16, n, p, p > 0.f ? 1.f / (1.f - p) : 1.f, seed, stream);
    return (int)hipGetLastError();
}

int vqa_sumsq_f32(const float* x, uint64_t n, float* out, vqa_stream_t s) {
    if (!x || !out) return VQA_ERR_ARG;
    if (n == 0) return VQA_OK;
    if ((uintptr_t)x & 15) return VQA_ERR_ARG;
    hipLaunchKernelGGL(sumsq_kernel, dim3(grid_for(n / 4 + 1)), dim3(TPB), 0, (hipStream_t)s, x, n, out);
    return (int)hipGetLastError();
}

int vqa_adamw_step(const VqaAdamWDesc* d, vqa_stream_t s) {
    if (!d || !d->param || !d->grad || !d->exp_avg || !d->exp_avg_sq) return VQA_ERR_ARG;
    if (d->n == 0) return VQA_OK;
    if (((uintptr_t)d->param | (uintptr_t)d->grad | (uintptr_t)d->exp_avg | (uintptr_t)d->exp_avg_sq) & 15) return VQA_ERR_ARG;
    if (d->param_bf16 && ((uintptr_t)d->param_bf16 & 7)) return VQA_ERR_ARG;
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(d->n / 4 + 1)), dim3(TPB), 0, (hipStream_t)s, *d);
    return (int)hipGetLastError();
}

}  // extern "C"
