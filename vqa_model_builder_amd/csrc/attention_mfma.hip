// MFMA attention core for the hot shapes of the path (Sq, Skv <= 64; head dim 32/64/96/128): CLIP ViT (50x50, Dh 64),
// PhoBERT (64x64, Dh 64, key-padding mask) and the fusion block (64x64 self, 64x50 cross, Dh 96) -- and, with KT = 8 key tiles,
// up to 128 keys (the generative model's 114-token fused memory): forward for any Sq <= 128 (64-row query blocks on gridDim.y),
// backward for Sq <= 64 (its dK / dV sums run over ONE query block).
//
// One workgroup (4 wavefronts) per (batch, head); Q, K, V (and dO in backward) of the head are staged once in LDS
// (row pitch Dh*2+16 B: conflict-free ds_read_b128 row fragments, 8-B aligned transposing reads).  Wave w owns query
// rows 16w..16w+15.  All five products run on v_mfma_f32_16x16x32_bf16 with fp32 softmax in registers:
//   S^T = K Q^T  (the transposed score tile puts one query per lane column, so the row max/sum are in-lane + 2 shuffles)
//   O^T = V^T P^T   -- P^T never leaves registers: the S^T accumulator IS the B operand (k order permuted identically
//                      on the V^T side, which is produced by ds_read_b64_tr_b16)
//   backward: dP^T = V dO^T, dS = P (dP - rowsum(P dP)) scale, dQ^T = K^T dS^T (registers again);
//             dV^T = dO^T P, dK^T = Q^T dS need sums over all query rows: P^T / dS^T cross LDS once (bf16).
// Probabilities are recomputed in backward (nothing but Q,K,V,O is ever in HBM); dropout masks come from the counter
// RNG keyed by (b, h, q, kv) exactly as in the shape-generic kernel (attention.hip), which remains the fallback.
#include "common.h"
#include "vqa_hip.h"
#include "attn_core.h"

namespace {

template <int DH, int KT>
__global__ __launch_bounds__(256) void attn_mfma_fwd_kernel(const MArgs a_in) {
    MArgs a = a_in;
    if (a.drop_p > 0.f) a.seed = resolve_seed(a.seed);
    constexpr int PITCH = DH * 2 + 16, KR = 16 * KT;
    extern __shared__ __attribute__((aligned(16))) char smem[];             // Q block [64] | K [KR] | V [KR]
    char *Qs = smem, *Ks = smem + 64 * PITCH, *Vs = Ks + KR * PITCH;
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H, q0 = 64 * blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    stage_tile<DH>(Qs, a.q + ((size_t)b * a.Sq + q0) * a.ldq + h * DH, a.Sq - q0, a.ldq, tid);
    stage_tile<DH, KR>(Ks, a.k + (size_t)b * a.Skv * a.ldk + h * DH, a.Skv, a.ldk, tid);
    stage_tile<DH, KR>(Vs, a.v + (size_t)b * a.Skv * a.ldv + h * DH, a.Skv, a.ldv, tid);
    __syncthreads();
    if (q0 + 16 * w >= a.Sq) return;                 // whole wave beyond the last query row (no barrier follows)
    attn_core_fwd<DH, KT>(a, Qs, Ks, Vs, b, h, w, lane, q0);
}

template <int DH, int KT>
__global__ __launch_bounds__(256) void attn_mfma_bwd_kernel(const MArgs a_in) {
    MArgs a = a_in;
    if (a.drop_p > 0.f) a.seed = resolve_seed(a.seed);
    constexpr int PITCH = DH * 2 + 16, KR = 16 * KT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *Qs = smem, *Ks = Qs + 64 * PITCH, *Vs = Ks + KR * PITCH, *Gs = Vs + KR * PITCH;
    char *Pt = Gs + 64 * PITCH, *Dt = Pt + KR * PT;                  // [kv][q] bf16 tiles
    __shared__ __attribute__((aligned(16))) float cs_part[3][4][DH];  // bias-gradient partials: {dq, dk, dv} x wave x column
    const bool want_cs = a.dq_cs || a.dk_cs || a.dv_cs;
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, i = lane & 15;
    stage_tile<DH, KR>(Ks, a.k + (size_t)b * a.Skv * a.ldk + h * DH, a.Skv, a.ldk, tid);
    stage_tile<DH, KR>(Vs, a.v + (size_t)b * a.Skv * a.ldv + h * DH, a.Skv, a.ldv, tid);
    // Query blocks of 64 rows, one after the other (Sq <= 128: the generative model's 114-token fusion encoder).  dQ rows belong to one
    // block; dK / dV sum over the blocks: block 0 stores its bf16 partial, the next block's SAME thread reads it back, adds in fp32 and
    // stores the total (a thread's own store -> load to one address is ordered; no other thread touches it).
    const int nqb = (a.Sq + 63) / 64;
#pragma unroll 1
    for (int qb = 0; qb < nqb; ++qb) {
    const int q0 = 64 * qb, qrows = min(64, a.Sq - q0);
    const bool last_qb = qb == nqb - 1;
    if (qb > 0) __syncthreads();                                   // phase 2 of the previous block still reads Qs / Gs / Pt / Dt
    stage_tile<DH>(Qs, a.q + ((size_t)b * a.Sq + q0) * a.ldq + h * DH, qrows, a.ldq, tid);
    stage_tile<DH>(Gs, a.d_o + ((size_t)b * a.Sq + q0) * a.ldd_o + h * DH, qrows, a.ldd_o, tid);
    __syncthreads();
    // ---- phase 1: this wave's 16 query rows
    {
        f32x4 pn[KT], ks[KT], dp[KT];
        scores_softmax<DH, KT>(a, Qs, Ks, b, h, w, lane, pn, ks, q0);
#pragma unroll
        for (int t = 0; t < KT; ++t) dp[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < DH / 32; ++kk) {
            const h16x8 gf = row_frag(Gs, PITCH, 16 * w, 32 * kk, lane);
#pragma unroll
            for (int t = 0; t < KT; ++t) dp[t] = VQA_MFMA16(row_frag(Vs, PITCH, 16 * t, 32 * kk, lane), gf, dp[t]);
        }
        float delta = 0.f;
#pragma unroll
        for (int t = 0; t < KT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) { dp[t][r] *= ks[t][r]; delta += pn[t][r] * dp[t][r]; }
        delta = xor32_sum(xor16_sum(delta));
        const int ql = 16 * w + i, q = q0 + ql;                      // row inside the block (tile index) | row of the sample
        const bool qok = q < a.Sq;
#pragma unroll
        for (int t = 0; t < KT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float ds = qok ? pn[t][r] * (dp[t][r] - delta) * a.scale : 0.f;
                const float pd = qok ? pn[t][r] * ks[t][r] : 0.f;
                dp[t][r] = ds;                                         // dp now holds dS^T
                const int kv = 16 * t + 4 * g + r;
                *reinterpret_cast<h16_t*>(Pt + kv * PT + ql * 2) = (h16_t)pd;
                *reinterpret_cast<h16_t*>(Dt + kv * PT + ql * 2) = (h16_t)ds;
            }
        // dQ^T = K^T dS^T, dS^T straight from the accumulator registers
        h16x8 df[KT / 2];
#pragma unroll
        for (int u = 0; u < KT / 2; ++u) df[u] = pack8(dp[2 * u], dp[2 * u + 1]);
#pragma unroll 1
        for (int dt = 0; dt < DH / 16; ++dt) {
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < KT / 2; ++u)
                o = VQA_MFMA16(col_frag(Ks, PITCH, 32 * u + 4 * g, 32 * u + 16 + 4 * g, 16 * dt, lane), df[u], o);
            h16x4 ob;
#pragma unroll
            for (int r = 0; r < 4; ++r) ob[r] = (h16_t)o[r];
            if (qok) *reinterpret_cast<h16x4*>(a.dq + ((size_t)b * a.Sq + q) * a.lddq + h * DH + 16 * dt + 4 * g) = ob;
            if (want_cs) slab_colsum(&cs_part[0][w][16 * dt + 4 * g], ob, qok, lane, qb > 0);
        }
    }
    __syncthreads();
    // ---- phase 2: this wave's 16 key rows of every 64-key pass:  dV^T = dO^T P',  dK^T = Q^T dS   (k = q, natural order)
    // a real loop: these kernels run once per workgroup from a cold instruction cache -- measured, their run time WAS their code
    // size (fwd 1870 instructions / 8.3 us, bwd 2800 / 14 us at ~80 cycles per 64-B line); each (pass, dt) iteration is independent
#pragma unroll 1
    for (int it = 0; it < (KT / 4) * (DH / 16); ++it) {
        const int kp = it / (DH / 16), dt = it % (DH / 16);
        const int kv = 64 * kp + 16 * w + i;
        const bool kok = kv < a.Skv;
        f32x4 ov = {0.f, 0.f, 0.f, 0.f}, ok = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const h16x8 pf = row_frag(Pt, PT, 64 * kp + 16 * w, 32 * u, lane);
            const h16x8 sf = row_frag(Dt, PT, 64 * kp + 16 * w, 32 * u, lane);
            ov = VQA_MFMA16(col_frag(Gs, PITCH, 32 * u + 8 * g, 32 * u + 8 * g + 4, 16 * dt, lane), pf, ov);
            ok = VQA_MFMA16(col_frag(Qs, PITCH, 32 * u + 8 * g, 32 * u + 8 * g + 4, 16 * dt, lane), sf, ok);
        }
        h16x4 bv, bk;
        h16_t* pdv = a.dv + ((size_t)b * a.Skv + kv) * a.lddv + h * DH + 16 * dt + 4 * g;
        h16_t* pdk = a.dk + ((size_t)b * a.Skv + kv) * a.lddk + h * DH + 16 * dt + 4 * g;
        if (qb > 0 && kok) {
            const h16x4 pv = *reinterpret_cast<const h16x4*>(pdv), pk = *reinterpret_cast<const h16x4*>(pdk);
#pragma unroll
            for (int r = 0; r < 4; ++r) { ov[r] += (float)pv[r]; ok[r] += (float)pk[r]; }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { bv[r] = (h16_t)ov[r]; bk[r] = (h16_t)ok[r]; }
        if (kok) {
            *reinterpret_cast<h16x4*>(pdv) = bv;
            *reinterpret_cast<h16x4*>(pdk) = bk;
        }
        if (want_cs && last_qb) {                                   // column sums of the FINAL stored values
            slab_colsum(&cs_part[1][w][16 * dt + 4 * g], bk, kok, lane, kp > 0);
            slab_colsum(&cs_part[2][w][16 * dt + 4 * g], bv, kok, lane, kp > 0);
        }
    }
    }   // query blocks
    if (want_cs) {
        __syncthreads();
        for (int c = tid; c < 3 * DH; c += 256) {
            const int which = c / DH, col = c % DH;
            float* dst = which == 0 ? a.dq_cs : which == 1 ? a.dk_cs : a.dv_cs;
            if (dst) atomicAdd(dst + h * DH + col, cs_part[which][0][col] + cs_part[which][1][col] + cs_part[which][2][col] + cs_part[which][3][col]);
        }
    }
}

bool fill(const VqaAttnDesc* d, MArgs& a, bool bwd) {
    if (d->Sq < 1 || d->Skv < 1 || d->Skv > 128 || d->Sq > 128) return false;      // forward: query blocks over gridDim.y; backward: inside the workgroup
    if (d->Dh != 32 && d->Dh != 64 && d->Dh != 96 && d->Dh != 128) return false;
    if ((d->ldq | d->ldk | d->ldv | d->ldo) % 8) return false;
    if (((uintptr_t)d->q | (uintptr_t)d->k | (uintptr_t)d->v) & 15) return false;
    if (!bwd && ((uintptr_t)d->o & 7)) return false;
    if (bwd && ((d->ldd_o | d->lddq | d->lddk | d->lddv) % 8 || ((uintptr_t)d->d_o & 15) || (((uintptr_t)d->dq | (uintptr_t)d->dk | (uintptr_t)d->dv) & 7)))
        return false;
    a.q = (const h16_t*)d->q; a.k = (const h16_t*)d->k; a.v = (const h16_t*)d->v; a.o = (h16_t*)d->o;
    a.d_o = (const h16_t*)d->d_o; a.dq = (h16_t*)d->dq; a.dk = (h16_t*)d->dk; a.dv = (h16_t*)d->dv;
    a.ldq = d->ldq; a.ldk = d->ldk; a.ldv = d->ldv; a.ldo = d->ldo; a.ldd_o = d->ldd_o; a.lddq = d->lddq; a.lddk = d->lddk; a.lddv = d->lddv;
    a.B = d->B; a.H = d->H; a.Sq = d->Sq; a.Skv = d->Skv;
    a.mask = d->key_padding_mask;
    a.causal = d->causal;
    a.scale = d->scale != 0.f ? d->scale : 1.0f / sqrtf((float)d->Dh);
    a.drop_p = d->drop_p; a.inv_keep = d->drop_p > 0.f ? 1.f / (1.f - d->drop_p) : 1.f;
    a.seed = d->drop_seed; a.stream = d->drop_stream;
    a.dq_cs = bwd ? d->dq_colsum : nullptr; a.dk_cs = bwd ? d->dk_colsum : nullptr; a.dv_cs = bwd ? d->dv_colsum : nullptr;
    return true;
}

}  // namespace

// returns -1 when the shape is not covered (caller falls back to the generic kernel), else a hipError_t / 0
template <int DH, int KT>
static int launch_fwd(const MArgs& a, hipStream_t s) {
    constexpr size_t LDS = (64 + 2 * 16 * KT) * (DH * 2 + 16);
    static bool attr = false;
    if (LDS > 64 * 1024 && !attr) {
        hipError_t e = hipFuncSetAttribute((const void*)attn_mfma_fwd_kernel<DH, KT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    hipLaunchKernelGGL((attn_mfma_fwd_kernel<DH, KT>), dim3(a.B * a.H, (a.Sq + 63) / 64), dim3(256), LDS, s, a);
    return (int)hipGetLastError();
}

int vqa_attention_mfma_fwd(const VqaAttnDesc* d, hipStream_t s) {
    MArgs a;
    if (!fill(d, a, false)) return -1;
    const bool wide = a.Skv > 64;
    switch (d->Dh) {
        case 32: return wide ? launch_fwd<32, 8>(a, s) : launch_fwd<32, 4>(a, s);
        case 64: return wide ? launch_fwd<64, 8>(a, s) : launch_fwd<64, 4>(a, s);
        case 96: return wide ? launch_fwd<96, 8>(a, s) : launch_fwd<96, 4>(a, s);
        default: return wide ? launch_fwd<128, 8>(a, s) : launch_fwd<128, 4>(a, s);
    }
}

template <int DH, int KT>
static int launch_bwd(const MArgs& a, hipStream_t s) {
    constexpr size_t LDS = (2 * 64 + 2 * 16 * KT) * (DH * 2 + 16) + 2 * 16 * KT * PT;
    static bool attr = false;
    if (LDS > 64 * 1024 && !attr) {
        hipError_t e = hipFuncSetAttribute((const void*)attn_mfma_bwd_kernel<DH, KT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    hipLaunchKernelGGL((attn_mfma_bwd_kernel<DH, KT>), dim3(a.B * a.H), dim3(256), LDS, s, a);
    return (int)hipGetLastError();
}

int vqa_attention_mfma_bwd(const VqaAttnDesc* d, hipStream_t s) {
    MArgs a;
    if (!fill(d, a, true)) return -1;
    const bool wide = a.Skv > 64;
    switch (d->Dh) {
        case 32: return wide ? launch_bwd<32, 8>(a, s) : launch_bwd<32, 4>(a, s);
        case 64: return wide ? launch_bwd<64, 8>(a, s) : launch_bwd<64, 4>(a, s);
        case 96: return wide ? launch_bwd<96, 8>(a, s) : launch_bwd<96, 4>(a, s);
        default: return wide ? launch_bwd<128, 8>(a, s) : launch_bwd<128, 4>(a, s);
    }
}
