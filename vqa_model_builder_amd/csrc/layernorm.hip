// LayerNorm forward / backward for gfx950.  HBM-bound: one wavefront per row, 16-byte loads, the whole row lives
// in registers between the statistics pass and the normalise pass (one read of x, one write of y), row reductions
// by 64-lane shuffles (no LDS, no barriers in forward).  fp32 statistics and arithmetic, like torch under autocast.
//
// algorithmic bytes / row:  fwd  4*cols (x) [+4*cols add] + 4*cols (y fp32) and/or 2*cols (y bf16)
//                           bwd  8*cols (dy, x) [+4*cols dres] + 4*cols and/or 2*cols (dx)
#include "common.h"
#include "vqa_hip.h"

namespace {

constexpr int WAVES = 4;                 // rows per block pass
constexpr int MAXV = 16;                 // float4 per lane -> cols <= 64*4*16 = 4096

template <int NV>
__global__ __launch_bounds__(WAVES * 64) void ln_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ add, const float* __restrict__ gamma, const float* __restrict__ beta,
    float* __restrict__ y, h16_t* __restrict__ yb, float* __restrict__ mean_out, float* __restrict__ rstd_out,
    int rows, int cols, float eps, float drop_p, float inv_keep, uint64_t seed, uint32_t stream) {
    if (drop_p > 0.f) seed = resolve_seed(seed);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c4 = cols / 4;
    for (int row = blockIdx.x * WAVES + wave; row < rows; row += gridDim.x * WAVES) {
        // Every global load of the row is issued up front, UNCONDITIONALLY (lanes past the row end read a clamped chunk and discard it): behind
        // per-chunk `if (c < c4)` branches the compiler waited for each chunk before loading the next -- NV dependent round trips per row, and
        // a second chain for gamma / beta between the stores (ISA of round 2: G G [vmcnt(0)] x NV ... S S G G [vmcnt(0)] x NV)
        f32x4 v[NV], av[NV], gv[NV], bv[NV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = reinterpret_cast<const f32x4*>(x + (size_t)row * cols)[min(lane + 64 * i, c4 - 1)];
        if (add) {
#pragma unroll
            for (int i = 0; i < NV; ++i) av[i] = reinterpret_cast<const f32x4*>(add + (size_t)row * cols)[min(lane + 64 * i, c4 - 1)];
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            gv[i] = reinterpret_cast<const f32x4*>(gamma)[min(lane + 64 * i, c4 - 1)];
            bv[i] = reinterpret_cast<const f32x4*>(beta)[min(lane + 64 * i, c4 - 1)];
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (add) v[i] += av[i];
            if (c < c4) s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
            else v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        const float mean = wave_sum(s) / cols;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < c4) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float d = v[i][j] - mean; q += d * d; }
            }
        }
        const float rstd = rsqrtf(wave_sum(q) / cols + eps);
        if (lane == 0) { if (mean_out) mean_out[row] = mean; if (rstd_out) rstd_out[row] = rstd; }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < c4) {
                const f32x4 g = gv[i], b = bv[i];
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mean) * rstd * g[j] + b[j];
                if (drop_p > 0.f) o *= dropout_scale4(seed, stream, (uint64_t)row * cols + 4 * c, drop_p, inv_keep);
                if (y) reinterpret_cast<f32x4*>(y + (size_t)row * cols)[c] = o;
                if (yb) { h16x4 ob; for (int j = 0; j < 4; ++j) ob[j] = (h16_t)o[j]; reinterpret_cast<h16x4*>(yb + (size_t)row * cols)[c] = ob; }
            }
        }
    }
}

// backward: each wave walks rows; per-lane partial dgamma/dbeta stay in registers across the whole row loop, are
// combined across the block's waves through LDS once, and written as one partial row per block:
// ws[blockIdx.x][0:cols] = dgamma partial, ws[gridDim.x + blockIdx.x][..] = dbeta partial.
template <int NV>
__global__ __launch_bounds__(WAVES * 64) void ln_bwd_kernel(
    const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
    const float* __restrict__ gamma, const float* __restrict__ dres, float* __restrict__ dx, h16_t* __restrict__ dxb,
    float* __restrict__ ws, int rows, int cols, float drop_p, float inv_keep, uint64_t seed, uint32_t stream, int drop_mode, int want_colsum,
    float* __restrict__ acc_dgamma, float* __restrict__ acc_dbeta, float* __restrict__ acc_colsum) {
    if (drop_p > 0.f) seed = resolve_seed(seed);
    extern __shared__ __attribute__((aligned(16))) float lds[];     // [WAVES][3][cols]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c4 = cols / 4;
    f32x4 dg[NV], db[NV], g[NV], cs[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        dg[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; db[i] = dg[i]; cs[i] = dg[i];
        g[i] = reinterpret_cast<const f32x4*>(gamma)[min(lane + 64 * i, c4 - 1)];        // unconditional (clamped): one batch of loads, see below
    }
    for (int row = blockIdx.x * WAVES + wave; row < rows; row += gridDim.x * WAVES) {
        const float mean = mean_in[row], rstd = rstd_in[row];
        f32x4 xh[NV], gy[NV];
        float s1 = 0.f, s2 = 0.f;
        // all global loads of the row up front and unconditional (clamped chunk for lanes past the row end; never used): per-chunk branches made
        // every chunk its own dependent round trip, and the residual-gradient loads between the stores waited for those stores too
        f32x4 dv[NV], xv_[NV], rv_[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) dv[i] = reinterpret_cast<const f32x4*>(dy + (size_t)row * cols)[min(lane + 64 * i, c4 - 1)];
#pragma unroll
        for (int i = 0; i < NV; ++i) xv_[i] = reinterpret_cast<const f32x4*>(x + (size_t)row * cols)[min(lane + 64 * i, c4 - 1)];
        if (dres) {
#pragma unroll
            for (int i = 0; i < NV; ++i) rv_[i] = reinterpret_cast<const f32x4*>(dres + (size_t)row * cols)[min(lane + 64 * i, c4 - 1)];
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            xh[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; gy[i] = xh[i];
            if (c < c4) {
                f32x4 d = dv[i];
                if (drop_mode == 2) d *= dropout_scale4(seed, stream, (uint64_t)row * cols + 4 * c, drop_p, inv_keep);
                const f32x4 xv = xv_[i];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    xh[i][j] = (xv[j] - mean) * rstd;
                    gy[i][j] = d[j] * g[i][j];
                    s1 += gy[i][j];
                    s2 += gy[i][j] * xh[i][j];
                    dg[i][j] += d[j] * xh[i][j];
                    db[i][j] += d[j];
                }
            }
        }
        s1 = wave_sum(s1) / cols;
        s2 = wave_sum(s2) / cols;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < c4) {
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = rstd * (gy[i][j] - s1 - xh[i][j] * s2);
                if (dres) o += rv_[i];
                if (dx) reinterpret_cast<f32x4*>(dx + (size_t)row * cols)[c] = o;
                if (dxb || want_colsum) {
                    h16x4 ob;
                    f32x4 ks = {1.f, 1.f, 1.f, 1.f};
                    if (drop_mode == 1) ks = dropout_scale4(seed, stream, (uint64_t)row * cols + 4 * c, drop_p, inv_keep);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float t = o[j] * ks[j];
                        ob[j] = (h16_t)t;
                        cs[i][j] += t;                 // column sum of the (masked) gradient = bias gradient of the producer Linear
                    }
                    if (dxb) reinterpret_cast<h16x4*>(dxb + (size_t)row * cols)[c] = ob;
                }
            }
        }
    }
    if (!ws && !acc_dgamma && !acc_dbeta && !acc_colsum) return;
    f32x4* l4 = reinterpret_cast<f32x4*>(lds);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < c4) { l4[(wave * 3 + 0) * c4 + c] = dg[i]; l4[(wave * 3 + 1) * c4 + c] = db[i]; l4[(wave * 3 + 2) * c4 + c] = cs[i]; }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < c4; c += WAVES * 64) {
        f32x4 a = l4[c], b = l4[c4 + c], d = l4[2 * c4 + c];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) { a += l4[(w * 3 + 0) * c4 + c]; b += l4[(w * 3 + 1) * c4 + c]; d += l4[(w * 3 + 2) * c4 + c]; }
        if (ws) {
            reinterpret_cast<f32x4*>(ws + (size_t)blockIdx.x * cols)[c] = a;
            reinterpret_cast<f32x4*>(ws + (size_t)(gridDim.x + blockIdx.x) * cols)[c] = b;
            if (want_colsum) reinterpret_cast<f32x4*>(ws + (size_t)(2 * gridDim.x + blockIdx.x) * cols)[c] = d;
        } else {
            // accumulate mode: the block's partial sums go straight into the (pre-zeroed) outputs -- no workspace round trip
            // and no second launch (the reduce kernel cost as much as the backward itself on the path's 2048 x 768 rows)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (acc_dgamma) atomicAdd(acc_dgamma + 4 * c + j, a[j]);
                if (acc_dbeta) atomicAdd(acc_dbeta + 4 * c + j, b[j]);
                if (acc_colsum) atomicAdd(acc_colsum + 4 * c + j, d[j]);
            }
        }
    }
}

// final reduce of the per-block partials: out[c] = sum_b ws[b][c].  One workgroup per 64 columns; 16 row-groups of
// partial rows are summed in parallel (coalesced 256-B reads per wave) and combined through LDS.
__global__ __launch_bounds__(1024) void ln_bwd_reduce_kernel(const float* __restrict__ ws, int nblocks, int cols, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, float* __restrict__ colsum) {
    __shared__ float red[3][16][64];
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float a = 0.f, b = 0.f, d = 0.f;
    if (c < cols) {
        for (int k = rg; k < nblocks; k += 16) {
            a += ws[(size_t)k * cols + c]; b += ws[(size_t)(nblocks + k) * cols + c];
            if (colsum) d += ws[(size_t)(2 * nblocks + k) * cols + c];
        }
    }
    red[0][rg][lane] = a; red[1][rg][lane] = b; red[2][rg][lane] = d;
    __syncthreads();
    if (rg < 3 && c < cols) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) v += red[rg][k][lane];
        float* out = rg == 0 ? dgamma : rg == 1 ? dbeta : colsum;
        if (out) out[c] = v;
    }
}

// GROUPED form of the reduce: the partials of up to 32 LayerNorm backwards (one encoder's worth) summed by ONE launch at the
// end of the block's backward instead of one 8-us launch behind each of them.  blockIdx -> (item, 64-column chunk).
constexpr int MAX_LN_GROUP = 32;
struct LnReduceItem { const float* ws; float* dgamma; float* dbeta; float* colsum; int nblocks, cols; };
struct LnReduceGroup { int n; int blk_end[MAX_LN_GROUP]; LnReduceItem it[MAX_LN_GROUP]; };
__global__ __launch_bounds__(1024) void ln_bwd_reduce_grouped_kernel(const LnReduceGroup g) {
    __shared__ float red[3][16][64];
    int i = 0;
    while (i + 1 < g.n && (int)blockIdx.x >= g.blk_end[i]) ++i;
    const LnReduceItem& it = g.it[i];
    const int chunk = blockIdx.x - (i ? g.blk_end[i - 1] : 0);
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = chunk * 64 + lane, cols = it.cols, nblocks = it.nblocks;
    const float* ws = it.ws;
    float a = 0.f, b = 0.f, d = 0.f;
    if (c < cols) {
        // eight partial rows per trip with all their loads issued before the first add (the adds keep their order: same bits as the one-row loop,
        // which walked 32 dependent round trips per thread: 20 us per launch, five launches per step)
        const bool cs = it.colsum != nullptr;
        int k = rg;
        for (; k + 16 * 7 < nblocks; k += 16 * 8) {
            float va[8], vb[8], vd[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                va[u] = ws[(size_t)(k + 16 * u) * cols + c];
                vb[u] = ws[(size_t)(nblocks + k + 16 * u) * cols + c];
                vd[u] = cs ? ws[(size_t)(2 * nblocks + k + 16 * u) * cols + c] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { a += va[u]; b += vb[u]; if (cs) d += vd[u]; }
        }
        for (; k < nblocks; k += 16) {
            a += ws[(size_t)k * cols + c]; b += ws[(size_t)(nblocks + k) * cols + c];
            if (cs) d += ws[(size_t)(2 * nblocks + k) * cols + c];
        }
    }
    red[0][rg][lane] = a; red[1][rg][lane] = b; red[2][rg][lane] = d;
    __syncthreads();
    if (rg < 3 && c < cols) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) v += red[rg][k][lane];
        float* out = rg == 0 ? it.dgamma : rg == 1 ? it.dbeta : it.colsum;
        if (out) out[c] = v;
    }
}

constexpr int BWD_BLOCKS_MAX = 1024;
int g_bwd_blocks = 512;      // workgroups of the backward kernel (4 waves each); vqa_set_layernorm_bwd_blocks

inline int nv_for(int cols) { return ceil_div(cols / 4, 64); }

}  // namespace

extern "C" {

int vqa_layernorm_fwd(const float* x, const float* add, const float* gamma, const float* beta, float* y_f32, void* y_bf16,
                      float* mean, float* rstd, int rows, int cols, float eps, float drop_p, uint64_t drop_seed,
                      uint32_t drop_stream, vqa_stream_t s) {
    if (!x || !gamma || !beta || (!y_f32 && !y_bf16) || rows < 0 || cols <= 0 || cols % 4 || cols > 4096) return VQA_ERR_ARG;
    if (rows == 0) return VQA_OK;
    const int grid = min(ceil_div(rows, WAVES), 2048);
    const float inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    const int nv = nv_for(cols);
#define LN_FWD(NV) hipLaunchKernelGGL((ln_fwd_kernel<NV>), dim3(grid), dim3(WAVES * 64), 0, (hipStream_t)s, x, add, gamma, beta, \
                                      y_f32, (h16_t*)y_bf16, mean, rstd, rows, cols, eps, drop_p, inv_keep, drop_seed, drop_stream)
    if (nv <= 1) LN_FWD(1); else if (nv <= 2) LN_FWD(2); else if (nv <= 3) LN_FWD(3); else if (nv <= 4) LN_FWD(4);
    else if (nv <= 8) LN_FWD(8); else LN_FWD(16);
#undef LN_FWD
    return (int)hipGetLastError();
}

size_t vqa_layernorm_bwd_ws_floats(int cols) { return (size_t)3 * BWD_BLOCKS_MAX * cols; }
int vqa_layernorm_bwd_blocks(int rows) { return min(ceil_div(rows, WAVES), g_bwd_blocks); }
void vqa_set_layernorm_bwd_blocks(int n) { g_bwd_blocks = n < 1 ? 1 : n > BWD_BLOCKS_MAX ? BWD_BLOCKS_MAX : n; }

static bool g_ln_defer_reduce = false;     // set around a call by vqa_layernorm_bwd_partials

int vqa_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma, const float* dres,
                      float* dx_f32, void* dx_bf16, float* dgamma, float* dbeta, float* dx_colsum, float* ws, int rows, int cols,
                      float drop_p, uint64_t drop_seed, uint32_t drop_stream, int drop_mode, vqa_stream_t s);

int vqa_layernorm_bwd_partials(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma, const float* dres,
                               float* dx_f32, void* dx_bf16, int want_colsum, float* ws, int rows, int cols,
                               float drop_p, uint64_t drop_seed, uint32_t drop_stream, int drop_mode, vqa_stream_t s) {
    if (!ws) return VQA_ERR_ARG;
    g_ln_defer_reduce = true;
    // non-null dummies select what the kernel accumulates; nothing is written through them when the reduce is deferred
    const int rc = vqa_layernorm_bwd(dy, x, mean, rstd, gamma, dres, dx_f32, dx_bf16, ws, ws, want_colsum ? ws : nullptr, ws, rows, cols,
                                     drop_p, drop_seed, drop_stream, drop_mode, s);
    g_ln_defer_reduce = false;
    return rc;
}

int vqa_layernorm_reduce_grouped(const VqaLnReduceItem* items, int n, vqa_stream_t s) {
    if (!items || n <= 0 || n > MAX_LN_GROUP) return VQA_ERR_ARG;
    LnReduceGroup g{};
    g.n = n;
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        const VqaLnReduceItem& d = items[i];
        if (!d.ws || d.nblocks <= 0 || d.cols <= 0) return VQA_ERR_ARG;
        blocks += ceil_div(d.cols, 64);
        g.blk_end[i] = blocks;
        g.it[i] = LnReduceItem{d.ws, d.dgamma, d.dbeta, d.dx_colsum, d.nblocks, d.cols};
    }
    hipLaunchKernelGGL(ln_bwd_reduce_grouped_kernel, dim3(blocks), dim3(1024), 0, (hipStream_t)s, g);
    return (int)hipGetLastError();
}

int vqa_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma, const float* dres,
                      float* dx_f32, void* dx_bf16, float* dgamma, float* dbeta, float* dx_colsum, float* ws, int rows, int cols,
                      float drop_p, uint64_t drop_seed, uint32_t drop_stream, int drop_mode, vqa_stream_t s) {
    if (drop_p <= 0.f) drop_mode = 0;
    if (!dy || !x || !mean || !rstd || !gamma || rows <= 0 || cols <= 0 || cols % 4 || cols > 4096) return VQA_ERR_ARG;
    // ws == NULL with reduction outputs requested = ACCUMULATE mode: dgamma / dbeta / dx_colsum += (fp32 atomics), the caller
    // guarantees they are initialised (the gradient arena is zero-filled once per backward)
    const bool reduce_out = dgamma || dbeta || dx_colsum;
    const bool accumulate = reduce_out && !ws;
    const int grid = min(ceil_div(rows, WAVES), g_bwd_blocks);
    const float inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    float* wsp = (reduce_out && !accumulate) ? ws : nullptr;
    float* ag = accumulate ? dgamma : nullptr; float* ab = accumulate ? dbeta : nullptr; float* ac = accumulate ? dx_colsum : nullptr;
    const size_t lds_bytes = reduce_out ? (size_t)WAVES * 3 * cols * 4 : 0;
    if (lds_bytes > 160 * 1024 - 256) return VQA_ERR_ARG;            // affine/colsum partials need 48*cols bytes of LDS: cols <= 3328
    const int nv = nv_for(cols);
#define LN_BWD(NV) hipLaunchKernelGGL((ln_bwd_kernel<NV>), dim3(grid), dim3(WAVES * 64), lds_bytes, (hipStream_t)s, dy, x, mean, rstd, \
                                      gamma, dres, dx_f32, (h16_t*)dx_bf16, wsp, rows, cols, drop_p, inv_keep, drop_seed, drop_stream, drop_mode, dx_colsum ? 1 : 0, ag, ab, ac)
    if (nv <= 1) LN_BWD(1); else if (nv <= 2) LN_BWD(2); else if (nv <= 3) LN_BWD(3); else if (nv <= 4) LN_BWD(4);
    else if (nv <= 8) {
        static bool attr8 = false;       // 8 float4/lane: 96 KiB of LDS for the three-way cross-wave combine
        if (!attr8) { (void)hipFuncSetAttribute((const void*)ln_bwd_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256); attr8 = true; }
        LN_BWD(8);
    } else {
        // 16 float4/lane: 128 KiB of LDS for the cross-wave combine exceeds the default dynamic limit
        static bool attr_set = false;
        if (!attr_set) { (void)hipFuncSetAttribute((const void*)ln_bwd_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256); attr_set = true; }
        LN_BWD(16);
    }
#undef LN_BWD
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    if (wsp && !g_ln_defer_reduce) {
        hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3(ceil_div(cols, 64)), dim3(1024), 0, (hipStream_t)s, ws, grid, cols, dgamma, dbeta, dx_colsum);
        e = hipGetLastError();
    }
    return (int)e;
}

}  // extern "C"
