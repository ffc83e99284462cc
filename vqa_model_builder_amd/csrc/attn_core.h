// Device-side core of the MFMA attention kernels (attention_mfma.hip) -- shared with the fused in-projection + attention
// kernel (gemm.hip: fused_inproj_attn_kernel), which leaves Q, K, V of one (sample, head) in LDS in the layout documented
// here instead of staging them from HBM.  Everything is `static` to the including translation unit.
#pragma once
#include "common.h"

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
    int causal;                         // 1: query q attends keys kv <= q only (decoder self-attention)
};

// column sums of one wave's 16 x 4 slab (lane (i, g) holds row i, columns 4g..4g+3 of the bf16 values it just stored):
// fold the 16 row-lanes; lanes i == 0 leave their 4 columns in the wave's LDS row.  The workgroup's four rows are summed
// at the end and added to the accumulator ONCE per workgroup: B adders per address (one per batch element), inside the
// range where float atomics keep their rate (per-wave adds -- 4 B adders -- ran the kernel 2.4x slower).
__device__ __forceinline__ void slab_colsum(float* lds_row, const h16x4& v, bool row_ok, int lane, bool accumulate = false) {
    f32x4 c;
#pragma unroll
    for (int r = 0; r < 4; ++r) c[r] = row_ok ? (float)v[r] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) c[r] = row_sum16(c[r]);            // DPP rotates inside the 16-lane row: no LDS round trips
    if ((lane & 15) == 0) {
        if (accumulate) c += *reinterpret_cast<f32x4*>(lds_row);   // second 64-key pass of the same wave (its own LDS row)
        *reinterpret_cast<f32x4*>(lds_row) = c;
    }
}

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
constexpr int PT = 144;          // pitch of the [kv][q] P^T / dS^T tiles (64 bf16 + 16 B)

template <int DH, int ROWS = 64>
__device__ __forceinline__ void stage_tile(char* lds, const h16_t* g, int rows, int ld, int tid) {
    constexpr int PITCH = DH * 2 + 16, CPR = DH / 8;          // 16-B chunks per row
    for (int c = tid; c < ROWS * CPR; c += 256) {
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
// s[t][r] is S^T at kv = 16t + 4g + r, q = q0 + 16w + (lane&15).  KT = key tiles of 16 staged in Ks (4: Skv <= 64, 8: <= 128);
// q0 = first query row of the 64-row block staged in Qs.
template <int DH, int KT = 4>
__device__ __forceinline__ void scores_softmax(const MArgs& a, const char* Qs, const char* Ks, int b, int h, int w, int lane,
                                               f32x4 (&pn)[KT], f32x4 (&ks)[KT], int q0 = 0) {
    constexpr int PITCH = DH * 2 + 16;
    const int g = lane >> 4;
    // key-padding mask bytes of this lane's keys (kv = 16t + 4g + r), fetched BEFORE the score MFMAs: read one by one where they are used,
    // each byte was its own dependent L2 round trip behind a branch -- 16 of them per wave (ISA of round 2: G [vmcnt(0)] x 16)
    uint32_t mw[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t) mw[t] = 0u;
    if (a.mask) {
        const unsigned char* mrow = a.mask + (size_t)b * a.Skv;
        if ((a.Skv & 3) == 0) {
#pragma unroll
            for (int t = 0; t < KT; ++t) {
                const int kv0 = 16 * t + 4 * g;
                if (kv0 < a.Skv) mw[t] = *reinterpret_cast<const uint32_t*>(mrow + kv0);
            }
        } else {
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kv = 16 * t + 4 * g + r;
                    if (kv < a.Skv) mw[t] |= (uint32_t)(mrow[kv] != 0) << (8 * r);
                }
        }
    }
    f32x4 s[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t) s[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < DH / 32; ++kk) {
        const h16x8 qf = row_frag(Qs, PITCH, 16 * w, 32 * kk, lane);
#pragma unroll
        for (int t = 0; t < KT; ++t) s[t] = VQA_MFMA16(row_frag(Ks, PITCH, 16 * t, 32 * kk, lane), qf, s[t]);
    }
    const int q = q0 + 16 * w + (lane & 15);
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int kv = 16 * t + 4 * g + r;
            const bool ok = kv < a.Skv && ((mw[t] >> (8 * r)) & 0xffu) == 0u && !(a.causal && kv > q);
            s[t][r] = ok ? s[t][r] * a.scale : -INFINITY;
            m = fmaxf(m, s[t][r]);
        }
    m = xor32_max(xor16_max(m));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float e = (s[t][r] == -INFINITY) ? 0.f : __expf(s[t][r] - m);
            pn[t][r] = e;
            sum += e;
        }
    sum = xor32_sum(xor16_sum(sum));
    const float inv = sum > 0.f ? 1.f / sum : 0.f;
    const uint64_t base = (((uint64_t)b * a.H + h) * a.Sq + q) * (uint64_t)a.Skv;
#pragma unroll
    for (int t = 0; t < KT; ++t) {
        pn[t] *= inv;
        ks[t] = (f32x4){1.f, 1.f, 1.f, 1.f};
    }
    if (a.drop_p > 0.f) {
        if ((a.Skv & 3) == 0) {                   // aligned groups of four keys: ONE counter hash per group (the hash is the cost)
#pragma unroll
            for (int t = 0; t < KT; ++t) ks[t] = dropout_scale4(a.seed, a.stream, base + 16 * t + 4 * g, a.drop_p, a.inv_keep);
        } else {
#pragma unroll 1
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) ks[t][r] = dropout_scale(a.seed, a.stream, base + 16 * t + 4 * g + r, a.drop_p, a.inv_keep);
        }
    }
}

// forward of wave w (query rows q0 + 16w .. +15) over Q (64-row block), K, V (16 KT rows) staged in LDS ([rows][DH] bf16, row pitch
// DH*2+16): scores, softmax, dropout, O = P V, stored bf16 to a.o.  No barriers inside.
template <int DH, int KT = 4>
__device__ __forceinline__ void attn_core_fwd(const MArgs& a, const char* Qs, const char* Ks, const char* Vs, int b, int h, int w, int lane, int q0 = 0) {
    constexpr int PITCH = DH * 2 + 16;
    const int g = lane >> 4;
    f32x4 pn[KT], ks[KT];
    scores_softmax<DH, KT>(a, Qs, Ks, b, h, w, lane, pn, ks, q0);
    h16x8 pf[KT / 2];
#pragma unroll
    for (int u = 0; u < KT / 2; ++u) pf[u] = pack8(pn[2 * u] * ks[2 * u], pn[2 * u + 1] * ks[2 * u + 1]);
    const int q = q0 + 16 * w + (lane & 15);
    // a real loop: these kernels run once per workgroup from a cold instruction cache -- measured, their run time WAS their code
    // size (fwd 1870 instructions / 8.3 us, bwd 2800 / 14 us at ~80 cycles per 64-B line); each dt iteration is independent
#pragma unroll 1
    for (int dt = 0; dt < DH / 16; ++dt) {
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < KT / 2; ++u)
            o = VQA_MFMA16(col_frag(Vs, PITCH, 32 * u + 4 * g, 32 * u + 16 + 4 * g, 16 * dt, lane), pf[u], o);
        if (q < a.Sq) {
            h16x4 ob;
#pragma unroll
            for (int r = 0; r < 4; ++r) ob[r] = (h16_t)o[r];
            *reinterpret_cast<h16x4*>(a.o + ((size_t)b * a.Sq + q) * a.ldo + h * DH + 16 * dt + 4 * g) = ob;
        }
    }
}

}  // namespace
