// Small-sequence attention core for gfx950: softmax(Q K^T * scale + key_padding) V, forward and backward, one
// workgroup per (batch, head).  Sequences on this path are tiny (50 vision / 64 text tokens; 1-4 tokens inside
// the MoE experts), so K and V of one head live in LDS for the whole workgroup and the probabilities never
// touch HBM (recomputed in backward).  One wavefront owns one query row at a time: lane j = key j for the
// scores (row max / sum by 64-lane shuffles), lane d = feature d for the P.V product.
//
// This is the shape-generic kernel (any Sq,Skv <= 128, even Dh <= 256, optional key-padding mask and dropout on
// the probabilities).  It is <2 % of the block's FLOPs; the projections run on MFMA (gemm.hip).
#include "common.h"
#include "vqa_hip.h"

namespace {

constexpr int NT = 256, NW = 4;

struct AttnArgs {
    const h16_t *q, *k, *v, *d_o;
    h16_t *o, *dq, *dk, *dv;
    int ldq, ldk, ldv, ldo, ldd_o, lddq, lddk, lddv;
    int B, H, Sq, Skv, Dh;
    const uint8_t* mask;
    float scale, drop_p, inv_keep;
    uint64_t seed; uint32_t stream;
    // backward of shapes whose tiles do not fit the LDS together (Sq = Skv = 100, Dh = 256: the ObjectDetection expert's
    // queries): mode 1 = phase A alone (K, V in LDS; q / dO rows from global; P', dS to the workspace), mode 2 = phase B alone
    // (Q, dO in LDS; P', dS read back).  mode 0 = both phases in one launch, everything in LDS.
    int mode; float* ws;
    int causal;
};

// orders this wave's own LDS writes before its following cross-lane LDS reads (per-wave scratch rows)
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ void load_tile(h16_t* lds, const h16_t* g, int rows, int Dh, int ld, int pitch, int tid) {
    const int half = Dh / 2;
    for (int t = tid; t < rows * half; t += NT) {
        const int r = t / half, c = t % half;
        *reinterpret_cast<uint32_t*>(lds + r * pitch + 2 * c) = *reinterpret_cast<const uint32_t*>(g + (size_t)r * ld + 2 * c);
    }
}

__device__ __forceinline__ float dot_row(const float* qrow, const h16_t* krow, int Dh) {
    float acc = 0.f;
    for (int d = 0; d < Dh; d += 2) {
        const h16x2 kk = *reinterpret_cast<const h16x2*>(krow + d);
        acc += qrow[d] * (float)kk[0] + qrow[d + 1] * (float)kk[1];
    }
    return acc;
}

// scores -> probabilities for the wave's current query row; returns p (normalised, NOT dropped) for key slots
// j0 = lane, j1 = lane + 64, and the dropout keep-scales.
__device__ __forceinline__ void row_softmax(const AttnArgs& a, const float* qrow, const h16_t* Ks, int pitch, int b, int h, int qi,
                                            int lane, float& p0, float& p1, float& ks0, float& ks1) {
    const int j0 = lane, j1 = lane + 64;
    float s0 = -INFINITY, s1 = -INFINITY;
    const int jmax = a.causal ? min(a.Skv, qi + 1) : a.Skv;          // causal: query qi sees keys 0 .. qi
    if (j0 < jmax && !(a.mask && a.mask[(size_t)b * a.Skv + j0])) s0 = dot_row(qrow, Ks + j0 * pitch, a.Dh) * a.scale;
    if (j1 < jmax && !(a.mask && a.mask[(size_t)b * a.Skv + j1])) s1 = dot_row(qrow, Ks + j1 * pitch, a.Dh) * a.scale;
    const float m = wave_max(fmaxf(s0, s1));
    const float e0 = (s0 == -INFINITY) ? 0.f : __expf(s0 - m);
    const float e1 = (s1 == -INFINITY) ? 0.f : __expf(s1 - m);
    const float sum = wave_sum(e0 + e1);
    const float inv = sum > 0.f ? 1.f / sum : 0.f;         // a fully masked row yields zeros (torch would yield NaN)
    p0 = e0 * inv; p1 = e1 * inv;
    ks0 = ks1 = 1.f;
    if (a.drop_p > 0.f) {
        const uint64_t base = (((uint64_t)b * a.H + h) * a.Sq + qi) * (uint64_t)a.Skv;
        ks0 = dropout_scale(a.seed, a.stream, base + j0, a.drop_p, a.inv_keep);
        ks1 = dropout_scale(a.seed, a.stream, base + j1, a.drop_p, a.inv_keep);
    }
}

__global__ __launch_bounds__(NT) void attn_fwd_kernel(const AttnArgs a_in) {
    AttnArgs a = a_in;
    if (a.drop_p > 0.f) a.seed = resolve_seed(a.seed);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int pitch = a.Dh + 2;
    h16_t* Ks = reinterpret_cast<h16_t*>(smem);
    h16_t* Vs = Ks + a.Skv * pitch;
    float* qbuf = reinterpret_cast<float*>(Vs + a.Skv * pitch);       // [NW][Dh]
    float* pbuf = qbuf + NW * a.Dh;                                    // [NW][128]
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    load_tile(Ks, a.k + (size_t)b * a.Skv * a.ldk + h * a.Dh, a.Skv, a.Dh, a.ldk, pitch, tid);
    load_tile(Vs, a.v + (size_t)b * a.Skv * a.ldv + h * a.Dh, a.Skv, a.Dh, a.ldv, pitch, tid);
    __syncthreads();
    float* qrow = qbuf + wave * a.Dh;
    float* prow = pbuf + wave * 128;
    for (int qi = blockIdx.y * NW + wave; qi < a.Sq; qi += gridDim.y * NW) {
        const h16_t* qg = a.q + ((size_t)b * a.Sq + qi) * a.ldq + h * a.Dh;
        wave_sync();
        for (int d = lane; d < a.Dh; d += 64) qrow[d] = (float)qg[d];
        wave_sync();
        float p0, p1, ks0, ks1;
        row_softmax(a, qrow, Ks, pitch, b, h, qi, lane, p0, p1, ks0, ks1);
        prow[lane] = p0 * ks0;
        prow[lane + 64] = p1 * ks1;
        wave_sync();
        h16_t* og = a.o + ((size_t)b * a.Sq + qi) * a.ldo + h * a.Dh;
        for (int d = lane; d < a.Dh; d += 64) {
            float acc = 0.f;
            for (int j = 0; j < a.Skv; ++j) acc += prow[j] * (float)Vs[j * pitch + d];
            og[d] = (h16_t)acc;
        }
    }
}

__global__ __launch_bounds__(NT) void attn_bwd_kernel(const AttnArgs a_in) {
    AttnArgs a = a_in;
    if (a.drop_p > 0.f) a.seed = resolve_seed(a.seed);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int pitch = a.Dh + 2, pp = a.Skv + 1;
    const bool phaseA = a.mode != 2, phaseB = a.mode != 1;
    // LDS: [K, V] (phase A) [Q, dO] (phase B or mode 0) [P', dS] (mode 0 only) [per-wave rows] (phase A)
    h16_t* Ks = reinterpret_cast<h16_t*>(smem);
    h16_t* Vs = Ks + (phaseA ? a.Skv * pitch : 0);
    h16_t* Qs = Vs + (phaseA ? a.Skv * pitch : 0);
    h16_t* Gs = Qs + (a.mode != 1 ? a.Sq * pitch : 0);               // dO
    float* Ps = reinterpret_cast<float*>(Gs + (a.mode != 1 ? a.Sq * pitch : 0));     // [Sq][Skv+1] dropped probabilities
    float* Ds = Ps + (a.mode == 0 ? a.Sq * pp : 0);                    // [Sq][Skv+1] dS
    float* qbuf = Ds + (a.mode == 0 ? a.Sq * pp : 0);                  // [NW][Dh] fp32 q row
    float* gbuf = qbuf + NW * a.Dh;                                    // [NW][Dh] fp32 dO row
    float* sbuf = gbuf + NW * a.Dh;                                    // [NW][128] ds row
    if (a.mode != 0) {                                                 // P', dS live in the global workspace of this (b, h)
        Ps = a.ws + (size_t)blockIdx.x * 2 * a.Sq * pp;
        Ds = Ps + (size_t)a.Sq * pp;
    }
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const h16_t* qg0 = a.q + (size_t)b * a.Sq * a.ldq + h * a.Dh;
    const h16_t* gg0 = a.d_o + (size_t)b * a.Sq * a.ldd_o + h * a.Dh;
    if (phaseA) {
        load_tile(Ks, a.k + (size_t)b * a.Skv * a.ldk + h * a.Dh, a.Skv, a.Dh, a.ldk, pitch, tid);
        load_tile(Vs, a.v + (size_t)b * a.Skv * a.ldv + h * a.Dh, a.Skv, a.Dh, a.ldv, pitch, tid);
    }
    if (a.mode != 1) {
        load_tile(Qs, qg0, a.Sq, a.Dh, a.ldq, pitch, tid);
        load_tile(Gs, gg0, a.Sq, a.Dh, a.ldd_o, pitch, tid);
    }
    __syncthreads();
    if (phaseA) {
        float* qrow = qbuf + wave * a.Dh;
        float* grow = gbuf + wave * a.Dh;
        float* srow = sbuf + wave * 128;
        // phase A: per query row -> P', dS (LDS or workspace) and dQ (global)
        for (int qi = wave; qi < a.Sq; qi += NW) {
            wave_sync();
            if (a.mode == 0) {
                for (int d = lane; d < a.Dh; d += 64) { qrow[d] = (float)Qs[qi * pitch + d]; grow[d] = (float)Gs[qi * pitch + d]; }
            } else {
                for (int d = lane; d < a.Dh; d += 64) { qrow[d] = (float)qg0[(size_t)qi * a.ldq + d]; grow[d] = (float)gg0[(size_t)qi * a.ldd_o + d]; }
            }
            wave_sync();
            float p0, p1, ks0, ks1;
            row_softmax(a, qrow, Ks, pitch, b, h, qi, lane, p0, p1, ks0, ks1);
            const int j0 = lane, j1 = lane + 64;
            float dp0 = 0.f, dp1 = 0.f;
            if (j0 < a.Skv) dp0 = dot_row(grow, Vs + j0 * pitch, a.Dh) * ks0;
            if (j1 < a.Skv) dp1 = dot_row(grow, Vs + j1 * pitch, a.Dh) * ks1;
            const float delta = wave_sum(p0 * dp0 + p1 * dp1);
            const float ds0 = p0 * (dp0 - delta) * a.scale, ds1 = p1 * (dp1 - delta) * a.scale;
            if (j0 < a.Skv) { Ps[qi * pp + j0] = p0 * ks0; Ds[qi * pp + j0] = ds0; }
            if (j1 < a.Skv) { Ps[qi * pp + j1] = p1 * ks1; Ds[qi * pp + j1] = ds1; }
            srow[j0] = ds0; srow[j1] = ds1;
            wave_sync();
            h16_t* dqg = a.dq + ((size_t)b * a.Sq + qi) * a.lddq + h * a.Dh;
            for (int d = lane; d < a.Dh; d += 64) {
                float acc = 0.f;
                for (int j = 0; j < a.Skv; ++j) acc += srow[j] * (float)Ks[j * pitch + d];
                dqg[d] = (h16_t)acc;
            }
        }
        __syncthreads();
    }
    if (!phaseB) return;
    // phase B: per key row -> dK, dV
    for (int j = wave; j < a.Skv; j += NW) {
        h16_t* dkg = a.dk + ((size_t)b * a.Skv + j) * a.lddk + h * a.Dh;
        h16_t* dvg = a.dv + ((size_t)b * a.Skv + j) * a.lddv + h * a.Dh;
        for (int d = lane; d < a.Dh; d += 64) {
            float ak = 0.f, av = 0.f;
            for (int qi = 0; qi < a.Sq; ++qi) {
                ak += Ds[qi * pp + j] * (float)Qs[qi * pitch + d];
                av += Ps[qi * pp + j] * (float)Gs[qi * pitch + d];
            }
            dkg[d] = (h16_t)ak;
            dvg[d] = (h16_t)av;
        }
    }
}

int fill_args(const VqaAttnDesc* d, AttnArgs& a, bool bwd) {
    if (!d || !d->q || !d->k || !d->v) return VQA_ERR_ARG;
    if (d->B <= 0 || d->H <= 0 || d->Sq <= 0 || d->Skv <= 0 || d->Sq > 128 || d->Skv > 128) return VQA_ERR_ARG;
    if (d->Dh <= 0 || d->Dh > 256 || d->Dh % 2) return VQA_ERR_ARG;
    if ((d->ldq | d->ldk | d->ldv) % 2) return VQA_ERR_ARG;
    a.q = (const h16_t*)d->q; a.k = (const h16_t*)d->k; a.v = (const h16_t*)d->v; a.o = (h16_t*)d->o;
    a.ldq = d->ldq; a.ldk = d->ldk; a.ldv = d->ldv; a.ldo = d->ldo;
    a.B = d->B; a.H = d->H; a.Sq = d->Sq; a.Skv = d->Skv; a.Dh = d->Dh;
    a.mask = d->key_padding_mask;
    a.causal = d->causal;
    a.scale = d->scale != 0.f ? d->scale : 1.0f / sqrtf((float)d->Dh);
    a.drop_p = d->drop_p; a.inv_keep = d->drop_p > 0.f ? 1.f / (1.f - d->drop_p) : 1.f;
    a.seed = d->drop_seed; a.stream = d->drop_stream;
    a.d_o = (const h16_t*)d->d_o; a.ldd_o = d->ldd_o;
    a.dq = (h16_t*)d->dq; a.dk = (h16_t*)d->dk; a.dv = (h16_t*)d->dv;
    a.lddq = d->lddq; a.lddk = d->lddk; a.lddv = d->lddv;
    a.mode = 0; a.ws = nullptr;
    if (!bwd && !d->o) return VQA_ERR_ARG;
    if (bwd && (!d->d_o || !d->dq || !d->dk || !d->dv || (d->ldd_o % 2))) return VQA_ERR_ARG;
    return VQA_OK;
}

constexpr size_t LDS_MAX = 160 * 1024 - 512;

}  // namespace

int vqa_attention_mfma_fwd(const VqaAttnDesc* d, hipStream_t s);     // attention_mfma.hip; -1 = shape not covered
int vqa_attention_mfma_bwd(const VqaAttnDesc* d, hipStream_t s);
static bool g_attn_mfma = true;

extern "C" {

void vqa_set_attention_mfma(int on) { g_attn_mfma = on != 0; }

int vqa_attention_fwd(const VqaAttnDesc* d, vqa_stream_t s) {
    if (g_attn_mfma && d && d->q && d->k && d->v && d->o && d->B > 0 && d->H > 0) {
        const int rc = vqa_attention_mfma_fwd(d, (hipStream_t)s);
        if (rc >= 0) return rc;
    }
    AttnArgs a;
    int rc = fill_args(d, a, false);
    if (rc) return rc;
    const int pitch = a.Dh + 2;
    const size_t lds = (size_t)2 * a.Skv * pitch * 2 + (size_t)NW * a.Dh * 4 + (size_t)NW * 128 * 4;
    if (lds > LDS_MAX) return VQA_ERR_ARG;
    static size_t attr = 0;
    if (lds > 64 * 1024 && lds > attr) {
        hipError_t e = hipFuncSetAttribute((const void*)attn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX);
        if (e != hipSuccess) return (int)e;
        attr = LDS_MAX;
    }
    // split the query rows over gridDim.y when B*H alone cannot fill the 256 CUs
    int gy = 1;
    while (a.B * a.H * gy < 512 && gy * NW * 2 <= a.Sq) gy *= 2;
    hipLaunchKernelGGL(attn_fwd_kernel, dim3(a.B * a.H, gy), dim3(NT), lds, (hipStream_t)s, a);
    return (int)hipGetLastError();
}

size_t vqa_attention_bwd_ws_floats(int B, int H, int Sq, int Skv, int Dh) {
    const int pitch = Dh + 2, pp = Skv + 1;
    const size_t lds = (size_t)2 * (Skv + Sq) * pitch * 2 + (size_t)2 * Sq * pp * 4 + (size_t)2 * NW * Dh * 4 + (size_t)NW * 128 * 4;
    const bool mfma = g_attn_mfma && Sq <= 64 && Skv <= 64 && (Dh == 32 || Dh == 64 || Dh == 96 || Dh == 128);
    return (mfma || lds <= LDS_MAX) ? 0 : (size_t)B * H * 2 * Sq * pp;
}

int vqa_attention_bwd(const VqaAttnDesc* d, vqa_stream_t s) {
    if (g_attn_mfma && d && d->q && d->k && d->v && d->d_o && d->dq && d->dk && d->dv && d->B > 0 && d->H > 0) {
        const int rc = vqa_attention_mfma_bwd(d, (hipStream_t)s);
        if (rc >= 0) return rc;
    }
    AttnArgs a;
    int rc = fill_args(d, a, true);
    if (rc) return rc;
    const int pitch = a.Dh + 2, pp = a.Skv + 1;
    const size_t rows_b = (size_t)2 * NW * a.Dh * 4 + (size_t)NW * 128 * 4;
    const size_t lds = (size_t)2 * (a.Skv + a.Sq) * pitch * 2 + (size_t)2 * a.Sq * pp * 4 + rows_b;
    static size_t attr = 0;
    auto need = [&](size_t bytes) -> int {
        if (bytes > 64 * 1024 && bytes > attr) {
            hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX);
            if (e != hipSuccess) return (int)e;
            attr = LDS_MAX;
        }
        return 0;
    };
    if (lds <= LDS_MAX) {
        if ((rc = need(lds))) return rc;
        hipLaunchKernelGGL(attn_bwd_kernel, dim3(a.B * a.H), dim3(NT), lds, (hipStream_t)s, a);
    } else {
        // two launches with P', dS in the caller's workspace (vqa_attention_bwd_ws_floats): phase A keeps K, V in LDS, phase B Q, dO
        const size_t ldsA = (size_t)2 * a.Skv * pitch * 2 + rows_b, ldsB = (size_t)2 * a.Sq * pitch * 2 + rows_b;
        if (!d->ws || ldsA > LDS_MAX || ldsB > LDS_MAX) return VQA_ERR_ARG;
        if ((rc = need(ldsA > ldsB ? ldsA : ldsB))) return rc;
        a.ws = d->ws;
        a.mode = 1;
        hipLaunchKernelGGL(attn_bwd_kernel, dim3(a.B * a.H), dim3(NT), ldsA, (hipStream_t)s, a);
        a.mode = 2;
        hipLaunchKernelGGL(attn_bwd_kernel, dim3(a.B * a.H), dim3(NT), ldsB, (hipStream_t)s, a);
    }
    rc = (int)hipGetLastError();
    // the generic kernel does not fuse the bias-gradient column sums: separate passes (outputs are zero on entry by contract)
    const int HD = d->H * d->Dh;
    if (!rc && d->dq_colsum) rc = vqa_colsum_bf16(d->dq, d->B * d->Sq, HD, d->lddq, d->dq_colsum, s);
    if (!rc && d->dk_colsum) rc = vqa_colsum_bf16(d->dk, d->B * d->Skv, HD, d->lddk, d->dk_colsum, s);
    if (!rc && d->dv_colsum) rc = vqa_colsum_bf16(d->dv, d->B * d->Skv, HD, d->lddv, d->dv_colsum, s);
    return rc;
}

}  // extern "C"
