// MFMA attention core for the hot shapes of the path (Sq, Skv <= 64; head dim 32/64/96/128): CLIP ViT (50x50, Dh 64),
// PhoBERT (64x64, Dh 64, key-padding mask) and the fusion block (64x64 self, 64x50 cross, Dh 96).
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
#include "../vqa_model_builder_amd/csrc/common.h"
#include "vqa_hip.h"

namespace {

struct MArgs {
    const h16_t *q, *k, *v, *d_o;
    h16_t *o, *dq, *dk, *dv;
    int ldq, ldk, ldv, ldo, ldd_o, lddq, lddk, lddv;
    int B, H, Sq, Skv;
    const uint8_t* mask;
    float scale, drop_p, inv_keep;
    uint64_t seed; uint32_t stream;
    float *dq_cs, *dk_cs, *dv_cs;       // optional bias-gradient accumulators [H*Dh]
    unsigned long long* trace;
};

// column sums of one wave's 16 x 4 slab (lane (i, g) holds row i, columns 4g..4g+3 of the bf16 values it just stored):
// fold the 16 row-lanes; lanes i == 0 leave their 4 columns in the wave's LDS row.  The workgroup's four rows are summed
// at the end and added to the accumulator ONCE per workgroup: B adders per address (one per batch element), inside the
// range where float atomics keep their rate (per-wave adds -- 4 B adders -- ran the kernel 2.4x slower).
__device__ __forceinline__ void slab_colsum(float* lds_row, const h16x4& v, bool row_ok, int lane) {
    f32x4 c;
#pragma unroll
    for (int r = 0; r < 4; ++r) c[r] = row_ok ? (float)v[r] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) c[r] = row_sum16(c[r]);            // DPP rotates inside the 16-lane row: no LDS round trips
    if ((lane & 15) == 0) *reinterpret_cast<f32x4*>(lds_row) = c;
}

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
constexpr int PT = 144;          // pitch of the [kv][q] P^T / dS^T tiles (64 bf16 + 16 B)

template <int DH>
__device__ __forceinline__ void stage_tile(char* lds, const h16_t* g, int rows, int ld, int tid) {
    constexpr int PITCH = DH * 2 + 16, CPR = DH / 8;          // 16-B chunks per row
    for (int c = tid; c < 64 * CPR; c += 256) {
        const int r = c / CPR, cc = c % CPR;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (r < rows) v = *reinterpret_cast<const u32x4*>(g + (size_t)r * ld + cc * 8);
        *reinterpret_cast<u32x4*>(lds + r * PITCH + cc * 16) = v;
    }
}

// row fragment: 16 rows r0.., 8 consecutive k at k0 + 8*(lane>>4)
__device__ __forceinline__ h16x8 row_frag(const char* tile, int pitch, int r0, int k0, int lane) {
    return *reinterpret_cast<const h16x8*>(tile + (r0 + (lane & 15)) * pitch + (k0 + 8 * (lane >> 4)) * 2);
}
// column fragment through the transposing read: lane gets column c0 + (lane&15); its 8 k-slots are tile rows
// ra(g)+0..3 and rb(g)+0..3 where g = lane>>4 (the caller chooses the k order)
__device__ __forceinline__ h16x8 col_frag(const char* tile, int pitch, int ra, int rb, int c0, int lane) {
    const int i = lane & 15;
    const int off = (i >> 2) * pitch + (c0 + 4 * (i & 3)) * 2;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + ra * pitch + off));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + rb * pitch + off));
    union { struct { s16x4 a, b; } s; h16x8 v; } u;
    u.s.a = lo; u.s.b = hi;
    return u.v;
}

__device__ __forceinline__ h16x8 pack8(const f32x4& a, const f32x4& b) {
    h16x8 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) { r[j] = (h16_t)a[j]; r[4 + j] = (h16_t)b[j]; }
    return r;
}

// scores -> normalised probabilities (pn) and dropout keep-scales (ks) for this lane's query column.
// s[t][r] is S^T at kv = 16t + 4g + r, q = 16w + (lane&15).
template <int DH>
__device__ __forceinline__ void scores_softmax(const MArgs& a, const char* Qs, const char* Ks, int b, int h, int w, int lane,
                                               f32x4 (&pn)[4], f32x4 (&ks)[4]) {
    constexpr int PITCH = DH * 2 + 16;
    const int g = lane >> 4;
    f32x4 s[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) s[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < DH / 32; ++kk) {
        const h16x8 qf = row_frag(Qs, PITCH, 16 * w, 32 * kk, lane);
#pragma unroll
        for (int t = 0; t < 4; ++t) s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Ks, PITCH, 16 * t, 32 * kk, lane), qf, s[t], 0, 0, 0);
    }
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int kv = 16 * t + 4 * g + r;
            const bool ok = kv < a.Skv && !(a.mask && a.mask[(size_t)b * a.Skv + kv]);
            s[t][r] = ok ? s[t][r] * a.scale : -INFINITY;
            m = fmaxf(m, s[t][r]);
        }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float e = (s[t][r] == -INFINITY) ? 0.f : __expf(s[t][r] - m);
            pn[t][r] = e;
            sum += e;
        }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = sum > 0.f ? 1.f / sum : 0.f;
    const int q = 16 * w + (lane & 15);
    const uint64_t base = (((uint64_t)b * a.H + h) * a.Sq + q) * (uint64_t)a.Skv;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        pn[t] *= inv;
        ks[t] = (f32x4){1.f, 1.f, 1.f, 1.f};
    }
    if (a.drop_p > 0.f) {
        if ((a.Skv & 3) == 0) {                   // aligned groups of four keys: ONE counter hash per group (the hash is the cost)
#pragma unroll
            for (int t = 0; t < 4; ++t) ks[t] = dropout_scale4(a.seed, a.stream, base + 16 * t + 4 * g, a.drop_p, a.inv_keep);
        } else {
#pragma unroll 1
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) ks[t][r] = dropout_scale(a.seed, a.stream, base + 16 * t + 4 * g + r, a.drop_p, a.inv_keep);
        }
    }
}

template <int DH>
__global__ __launch_bounds__(256) void attn_mfma_fwd_kernel(const MArgs a_in) {
    MArgs a = a_in;
    if (a.drop_p > 0.f) a.seed = resolve_seed(a.seed);
    constexpr int PITCH = DH * 2 + 16;
    __shared__ __attribute__((aligned(16))) char smem[3 * 64 * PITCH];
    char *Qs = smem, *Ks = smem + 64 * PITCH, *Vs = smem + 2 * 64 * PITCH;
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4;
    stage_tile<DH>(Qs, a.q + (size_t)b * a.Sq * a.ldq + h * DH, a.Sq, a.ldq, tid);
    stage_tile<DH>(Ks, a.k + (size_t)b * a.Skv * a.ldk + h * DH, a.Skv, a.ldk, tid);
    stage_tile<DH>(Vs, a.v + (size_t)b * a.Skv * a.ldv + h * DH, a.Skv, a.ldv, tid);
    __syncthreads();
    if (16 * w >= a.Sq) return;                      // whole wave beyond the last query row (no barrier follows)
    f32x4 pn[4], ks[4];
    scores_softmax<DH>(a, Qs, Ks, b, h, w, lane, pn, ks);
    h16x8 pf[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) pf[u] = pack8(pn[2 * u] * ks[2 * u], pn[2 * u + 1] * ks[2 * u + 1]);
    const int q = 16 * w + (lane & 15);
    // a real loop: these kernels run once per workgroup from a cold instruction cache -- measured, their run time WAS their code
    // size (fwd 1870 instructions / 8.3 us, bwd 2800 / 14 us at ~80 cycles per 64-B line); each dt iteration is independent
#pragma unroll 1
    for (int dt = 0; dt < DH / 16; ++dt) {
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 2; ++u)
            o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(col_frag(Vs, PITCH, 32 * u + 4 * g, 32 * u + 16 + 4 * g, 16 * dt, lane), pf[u], o, 0, 0, 0);
        if (q < a.Sq) {
            h16x4 ob;
#pragma unroll
            for (int r = 0; r < 4; ++r) ob[r] = (h16_t)o[r];
            *reinterpret_cast<h16x4*>(a.o + ((size_t)b * a.Sq + q) * a.ldo + h * DH + 16 * dt + 4 * g) = ob;
        }
    }
}

template <int DH>
__global__ __launch_bounds__(256) void attn_mfma_bwd_kernel(const MArgs a_in) {
    MArgs a = a_in;
    if (a.drop_p > 0.f) a.seed = resolve_seed(a.seed);
    constexpr int PITCH = DH * 2 + 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *Qs = smem, *Ks = Qs + 64 * PITCH, *Vs = Ks + 64 * PITCH, *Gs = Vs + 64 * PITCH;
    char *Pt = Gs + 64 * PITCH, *Dt = Pt + 64 * PT;                  // [kv][q] bf16 tiles
    __shared__ __attribute__((aligned(16))) float cs_part[3][4][DH];  // bias-gradient partials: {dq, dk, dv} x wave x column
    const bool want_cs = a.dq_cs || a.dk_cs || a.dv_cs;
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, i = lane & 15;
    unsigned long long tr[8]; tr[0] = __builtin_readcyclecounter();
    stage_tile<DH>(Qs, a.q + (size_t)b * a.Sq * a.ldq + h * DH, a.Sq, a.ldq, tid);
    stage_tile<DH>(Ks, a.k + (size_t)b * a.Skv * a.ldk + h * DH, a.Skv, a.ldk, tid);
    stage_tile<DH>(Vs, a.v + (size_t)b * a.Skv * a.ldv + h * DH, a.Skv, a.ldv, tid);
    stage_tile<DH>(Gs, a.d_o + (size_t)b * a.Sq * a.ldd_o + h * DH, a.Sq, a.ldd_o, tid);
    __syncthreads();
    tr[1] = __builtin_readcyclecounter();
    // ---- phase 1: this wave's 16 query rows
    {
        f32x4 pn[4], ks[4], dp[4];
        scores_softmax<DH>(a, Qs, Ks, b, h, w, lane, pn, ks);
#pragma unroll
        for (int t = 0; t < 4; ++t) dp[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < DH / 32; ++kk) {
            const h16x8 gf = row_frag(Gs, PITCH, 16 * w, 32 * kk, lane);
#pragma unroll
            for (int t = 0; t < 4; ++t) dp[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Vs, PITCH, 16 * t, 32 * kk, lane), gf, dp[t], 0, 0, 0);
        }
        float delta = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) { dp[t][r] *= ks[t][r]; delta += pn[t][r] * dp[t][r]; }
        delta += __shfl_xor(delta, 16, 64);
        delta += __shfl_xor(delta, 32, 64);
        const int q = 16 * w + i;
        const bool qok = q < a.Sq;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float ds = qok ? pn[t][r] * (dp[t][r] - delta) * a.scale : 0.f;
                const float pd = qok ? pn[t][r] * ks[t][r] : 0.f;
                dp[t][r] = ds;                                         // dp now holds dS^T
                const int kv = 16 * t + 4 * g + r;
                *reinterpret_cast<h16_t*>(Pt + kv * PT + q * 2) = (h16_t)pd;
                *reinterpret_cast<h16_t*>(Dt + kv * PT + q * 2) = (h16_t)ds;
            }
        // dQ^T = K^T dS^T, dS^T straight from the accumulator registers
        h16x8 df[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) df[u] = pack8(dp[2 * u], dp[2 * u + 1]);
#pragma unroll 1
        for (int dt = 0; dt < DH / 16; ++dt) {
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 2; ++u)
                o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(col_frag(Ks, PITCH, 32 * u + 4 * g, 32 * u + 16 + 4 * g, 16 * dt, lane), df[u], o, 0, 0, 0);
            h16x4 ob;
#pragma unroll
            for (int r = 0; r < 4; ++r) ob[r] = (h16_t)o[r];
            if (qok) *reinterpret_cast<h16x4*>(a.dq + ((size_t)b * a.Sq + q) * a.lddq + h * DH + 16 * dt + 4 * g) = ob;
            if (want_cs) slab_colsum(&cs_part[0][w][16 * dt + 4 * g], ob, qok, lane);
        }
    }
    tr[2] = __builtin_readcyclecounter();
    __syncthreads();
    tr[3] = __builtin_readcyclecounter();
    // ---- phase 2: this wave's 16 key rows:  dV^T = dO^T P',  dK^T = Q^T dS   (k = q, natural order)
    const int kv = 16 * w + i;
    const bool kok = kv < a.Skv;
    // a real loop: these kernels run once per workgroup from a cold instruction cache -- measured, their run time WAS their code
    // size (fwd 1870 instructions / 8.3 us, bwd 2800 / 14 us at ~80 cycles per 64-B line); each dt iteration is independent
#pragma unroll 1
    for (int dt = 0; dt < DH / 16; ++dt) {
        f32x4 ov = {0.f, 0.f, 0.f, 0.f}, ok = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const h16x8 pf = row_frag(Pt, PT, 16 * w, 32 * u, lane);
            const h16x8 sf = row_frag(Dt, PT, 16 * w, 32 * u, lane);
            ov = __builtin_amdgcn_mfma_f32_16x16x32_bf16(col_frag(Gs, PITCH, 32 * u + 8 * g, 32 * u + 8 * g + 4, 16 * dt, lane), pf, ov, 0, 0, 0);
            ok = __builtin_amdgcn_mfma_f32_16x16x32_bf16(col_frag(Qs, PITCH, 32 * u + 8 * g, 32 * u + 8 * g + 4, 16 * dt, lane), sf, ok, 0, 0, 0);
        }
        h16x4 bv, bk;
#pragma unroll
        for (int r = 0; r < 4; ++r) { bv[r] = (h16_t)ov[r]; bk[r] = (h16_t)ok[r]; }
        if (kok) {
            *reinterpret_cast<h16x4*>(a.dv + ((size_t)b * a.Skv + kv) * a.lddv + h * DH + 16 * dt + 4 * g) = bv;
            *reinterpret_cast<h16x4*>(a.dk + ((size_t)b * a.Skv + kv) * a.lddk + h * DH + 16 * dt + 4 * g) = bk;
        }
        if (want_cs) {
            slab_colsum(&cs_part[1][w][16 * dt + 4 * g], bk, kok, lane);
            slab_colsum(&cs_part[2][w][16 * dt + 4 * g], bv, kok, lane);
        }
    }
    tr[4] = __builtin_readcyclecounter();
    if (want_cs) {
        __syncthreads();
        for (int c = tid; c < 3 * DH; c += 256) {
            const int which = c / DH, col = c % DH;
            float* dst = which == 0 ? a.dq_cs : which == 1 ? a.dk_cs : a.dv_cs;
            if (dst) atomicAdd(dst + h * DH + col, cs_part[which][0][col] + cs_part[which][1][col] + cs_part[which][2][col] + cs_part[which][3][col]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tr[5] = __builtin_readcyclecounter();
    if (a.trace && tid == 0) for (int k = 0; k < 6; ++k) a.trace[(size_t)blockIdx.x * 8 + k] = tr[k];
}

bool fill(const VqaAttnDesc* d, MArgs& a, bool bwd) {
    if (d->Sq > 64 || d->Skv > 64 || d->Sq < 1 || d->Skv < 1) return false;
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
    a.scale = d->scale != 0.f ? d->scale : 1.0f / sqrtf((float)d->Dh);
    a.drop_p = d->drop_p; a.inv_keep = d->drop_p > 0.f ? 1.f / (1.f - d->drop_p) : 1.f;
    a.seed = d->drop_seed; a.stream = d->drop_stream;
    a.trace = nullptr;
    a.dq_cs = bwd ? d->dq_colsum : nullptr; a.dk_cs = bwd ? d->dk_colsum : nullptr; a.dv_cs = bwd ? d->dv_colsum : nullptr;
    return true;
}

}  // namespace

// returns -1 when the shape is not covered (caller falls back to the generic kernel), else a hipError_t / 0
int vqa_attention_mfma_fwd(const VqaAttnDesc* d, hipStream_t s) {
    MArgs a;
    if (!fill(d, a, false)) return -1;
    dim3 grid(a.B * a.H), block(256);
    switch (d->Dh) {
        case 32: hipLaunchKernelGGL(attn_mfma_fwd_kernel<32>, grid, block, 0, s, a); break;
        case 64: hipLaunchKernelGGL(attn_mfma_fwd_kernel<64>, grid, block, 0, s, a); break;
        case 96: hipLaunchKernelGGL(attn_mfma_fwd_kernel<96>, grid, block, 0, s, a); break;
        default: hipLaunchKernelGGL(attn_mfma_fwd_kernel<128>, grid, block, 0, s, a); break;
    }
    return (int)hipGetLastError();
}

template <int DH>
static int launch_bwd(const MArgs& a, hipStream_t s) {
    constexpr size_t LDS = 4 * 64 * (DH * 2 + 16) + 2 * 64 * PT;
    static bool attr = false;
    if (LDS > 64 * 1024 && !attr) {
        hipError_t e = hipFuncSetAttribute((const void*)attn_mfma_bwd_kernel<DH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    hipLaunchKernelGGL(attn_mfma_bwd_kernel<DH>, dim3(a.B * a.H), dim3(256), LDS, s, a);
    return (int)hipGetLastError();
}

int vqa_attention_mfma_bwd(const VqaAttnDesc* d, hipStream_t s) {
    MArgs a;
    if (!fill(d, a, true)) return -1;
    switch (d->Dh) {
        case 32: return launch_bwd<32>(a, s);
        case 64: return launch_bwd<64>(a, s);
        case 96: return launch_bwd<96>(a, s);
        default: return launch_bwd<128>(a, s);
    }
}

extern "C" int lab_attn_bwd(const VqaAttnDesc* d, unsigned long long* trace, void* stream) {
    MArgs a;
    if (!fill(d, a, true)) return -1;
    a.trace = trace;
    return launch_bwd<64>(a, (hipStream_t)stream);
}
