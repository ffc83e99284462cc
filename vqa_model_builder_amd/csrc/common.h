// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the VQA hot path.
// Wavefront = 64 lanes everywhere; no other target is supported.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define VQA_OK 0
#define VQA_ERR_ARG 1001      // bad dims / alignment / unsupported combination

// The 16-bit operand type of THIS build of the library.  The sources are compiled twice: libvqa_hip.so with bfloat16
// operands (torch autocast's bf16 scheme) and libvqa_hip_f16.so (-DVQA_HALF_F16) with IEEE fp16 operands -- the dtype the
// reference's main loop runs under (autocast fp16 + GradScaler, training_pipeline.py:346-347,457).  Everything else is
// identical: fp32 accumulation, fp32 LayerNorm / softmax / loss / residual stream, same tiles, same MFMA shape.  Entry points
// keep their names ("..._bf16" = "the library's 16-bit type"); vqa_half_kind() tells which build a handle is.
#ifdef VQA_HALF_F16
typedef _Float16 h16_t;
#define VQA_HALF_KIND 1
#define VQA_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)      // v_mfma_f32_16x16x32_f16
#else
typedef __bf16 h16_t;
#define VQA_HALF_KIND 0
#define VQA_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)     // v_mfma_f32_16x16x32_bf16
#endif
typedef __attribute__((ext_vector_type(8))) h16_t h16x8;
typedef __attribute__((ext_vector_type(4))) h16_t h16x4;
typedef __attribute__((ext_vector_type(2))) h16_t h16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

enum VqaAct { ACT_NONE = 0, ACT_GELU_ERF = 1, ACT_QUICK_GELU = 2, ACT_RELU = 3 };

__device__ __forceinline__ float bf2f(h16_t v) { return (float)v; }
__device__ __forceinline__ h16_t f2bf(float v) { return (h16_t)v; }   // v_cvt_pk_bf16_f32: RNE, NaN-safe

// Cross-lane reductions.  __shfl_xor compiles to ds_bpermute_b32 on gfx950 -- an LDS-crossbar round trip of ~100 cycles per
// step, and a reduction is a chain of them (measured: the bias-gradient column sums cost the attention backward 5000 cycles
// per wave).  Inside a row of 16 lanes the DPP rotate does the same exchange in the VALU: row_ror by 8, 4, 2, 1 leaves the
// row's sum (max) in all 16 lanes; the four row results of a wave are combined through v_readlane.
template <int CTRL> __device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row_sum16(float v) {          // every lane: sum over its row of 16 lanes
    v += dpp_f32<0x128>(v); v += dpp_f32<0x124>(v); v += dpp_f32<0x122>(v); v += dpp_f32<0x121>(v);
    return v;
}
__device__ __forceinline__ float row_max16(float v) {
    v = fmaxf(v, dpp_f32<0x128>(v)); v = fmaxf(v, dpp_f32<0x124>(v)); v = fmaxf(v, dpp_f32<0x122>(v)); v = fmaxf(v, dpp_f32<0x121>(v));
    return v;
}
// Across rows gfx950 has v_permlane16_swap / v_permlane32_swap (VALU).  swap(a, b) returns {r0, r1}: in the even rows (lower
// half) r0 keeps a and r1 receives the PARTNER's a; in the odd rows (upper half) r0 receives the partner's b and r1 keeps b
// (probed on the device, scratch/lane_test).  With b = 0 resp. a = 0 the two calls give the partner's value on disjoint lanes.
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
__device__ __forceinline__ float xor16_partner(float v) {
    const unsigned x = __builtin_bit_cast(unsigned, v);
    const u32x2_t p = __builtin_amdgcn_permlane16_swap(x, 0u, false, false), q = __builtin_amdgcn_permlane16_swap(0u, x, false, false);
    return __builtin_bit_cast(float, p[1] | q[0]);
}
__device__ __forceinline__ float xor32_partner(float v) {
    const unsigned x = __builtin_bit_cast(unsigned, v);
    const u32x2_t p = __builtin_amdgcn_permlane32_swap(x, 0u, false, false), q = __builtin_amdgcn_permlane32_swap(0u, x, false, false);
    return __builtin_bit_cast(float, p[1] | q[0]);
}
__device__ __forceinline__ float xor16_sum(float v) { return v + xor16_partner(v); }
__device__ __forceinline__ float xor32_sum(float v) { return v + xor32_partner(v); }
__device__ __forceinline__ float xor16_max(float v) { return fmaxf(v, xor16_partner(v)); }
__device__ __forceinline__ float xor32_max(float v) { return fmaxf(v, xor32_partner(v)); }
__device__ __forceinline__ float lane_f32(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }
__device__ __forceinline__ float wave_sum(float v) {            // every lane: sum over the 64 lanes
    v = row_sum16(v);
    return (lane_f32(v, 0) + lane_f32(v, 16)) + (lane_f32(v, 32) + lane_f32(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
    v = row_max16(v);
    return fmaxf(fmaxf(lane_f32(v, 0), lane_f32(v, 16)), fmaxf(lane_f32(v, 32), lane_f32(v, 48)));
}

// Gauss error function pieces for the exact ("erf") GELU.  Abramowitz-Stegun 7.1.26: erf(z) = 1 - poly(t) exp(-z^2),
// t = 1 / (1 + 0.3275911 z), |error| <= 1.5e-7 -- three orders below the bf16 rounding of every value these feed, and a
// third of the instructions of libm's erff (one v_exp_f32, one v_rcp_f32, six FMAs).  The SAME exp(-x^2/2) is the Gaussian
// density the derivative needs, so backward costs no second exponential.
// The arithmetic of these three functions is spelled out -- explicit FMAs, fp contraction OFF for everything else -- so that a value is a pure
// function of its argument wherever the function is inlined.  Left to the compiler's contraction the SAME source gave results one fp32 rounding
// apart in two epilogues of the same GEMM (x * (1 - h) became fma(-h, x, x) in one context and not in the other: round 3, found when the
// compile-time epilogue forms were compared bit for bit with the generic one).
struct GeluParts { float cdf, e; };      // cdf = Phi(x);  e = exp(-x*x/2)
__device__ __forceinline__ GeluParts gelu_parts(float x) {
#pragma clang fp contract(off)
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float zz = z * z;
    const float e = __expf(-zz);
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
    float poly = __builtin_fmaf(t, 1.061405429f, -1.453152027f);
    poly = __builtin_fmaf(t, poly, 1.421413741f);
    poly = __builtin_fmaf(t, poly, -0.284496736f);
    poly = __builtin_fmaf(t, poly, 0.254829592f);
    poly = t * poly;
    const float half_erfc = (0.5f * poly) * e;                  // 0.5 * (1 - erf(z))
    const float upper = 1.0f - half_erfc;
    return {x >= 0.f ? upper : half_erfc, e};
}

__device__ __forceinline__ float act_fwd(float x, int act) {
#pragma clang fp contract(off)
    switch (act) {
        case ACT_GELU_ERF: { const float c = gelu_parts(x).cdf; return x * c; }
        case ACT_QUICK_GELU: { const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * x)); return x * s; }
        case ACT_RELU: return x > 0.f ? x : 0.f;
        default: return x;
    }
}
// d act(x) / dx
__device__ __forceinline__ float act_bwd(float x, int act) {
#pragma clang fp contract(off)
    switch (act) {
        case ACT_GELU_ERF: {
            const GeluParts g = gelu_parts(x);
            return __builtin_fmaf(x * 0.39894228040143267794f, g.e, g.cdf);
        }
        case ACT_QUICK_GELU: {
            const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * x));
            const float w = (1.702f * x) * (1.0f - s);
            return s * (1.0f + w);
        }
        case ACT_RELU: return x > 0.f ? 1.f : 0.f;
        default: return 1.f;
    }
}

// Counter-based RNG for dropout / router noise: one 32-bit hash per (seed, stream, index).
// (Murmur3-style finaliser over a 64-bit key; statistically adequate for Bernoulli masks and
//  Box-Muller noise, regenerated identically in forward and backward from the same key.)
__device__ __forceinline__ uint32_t rng_u32(uint64_t seed, uint32_t stream, uint64_t idx) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (idx + 1) + ((uint64_t)stream << 40);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (uint32_t)(z >> 16);
}
__device__ __forceinline__ float rng_uniform(uint64_t seed, uint32_t stream, uint64_t idx) {
    return (rng_u32(seed, stream, idx) >> 8) * (1.0f / 16777216.0f);   // [0,1)
}
// INDIRECT seeds.  A seed with bit 63 set does not carry the key itself: bits 0..47 are the device address of a 64-bit
// epoch word and bits 48..62 a call-site salt; the key is derived from the CURRENT value of that word.  A captured HIP
// graph freezes every kernel argument, so this is how its replays still draw fresh dropout masks / router noise each
// step (the training loop bumps the epoch word inside the graph); forward and backward of one step resolve the same key.
// Kernels call this ONCE on entry (the seed is wave-uniform: one scalar load).
__device__ __forceinline__ uint64_t resolve_seed(uint64_t s) {
    if (s >> 63) {
        const uint64_t e = *reinterpret_cast<const uint64_t*>(s & 0xFFFFFFFFFFFFull);
        s = ((e + 1) * 0xD1342543DE82EF95ull) ^ (((s >> 48) & 0x7FFF) * 0x9E3779B97F4A7C15ull);
        s &= 0x7FFFFFFFFFFFFFFFull;
    }
    return s;
}
// Dropout keep-scale (inverted dropout: 0 or 1/(1-p)).  One 64-bit hash serves FOUR consecutive elements -- 16 bits
// each (p is resolved to 1/65536) -- because the hash, not the compare, is the cost (two 64-bit multiplies): the
// epilogues that own aligned groups of four call dropout_scale4; dropout_scale is the per-element form of the same map.
__device__ __forceinline__ uint64_t rng_u64(uint64_t seed, uint32_t stream, uint64_t idx) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (idx + 1) + ((uint64_t)stream << 40);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ float dropout_scale(uint64_t seed, uint32_t stream, uint64_t idx, float p, float inv_keep) {
    const uint32_t u = (uint32_t)(rng_u64(seed, stream, idx >> 2) >> (16 * (idx & 3))) & 0xFFFFu;
    return (float)u * (1.0f / 65536.0f) >= p ? inv_keep : 0.f;
}
// idx4: index of the first of four consecutive elements, a multiple of 4
__device__ __forceinline__ f32x4 dropout_scale4(uint64_t seed, uint32_t stream, uint64_t idx4, float p, float inv_keep) {
    const uint64_t h = rng_u64(seed, stream, idx4 >> 2);
    f32x4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = (float)((uint32_t)(h >> (16 * j)) & 0xFFFFu) * (1.0f / 65536.0f) >= p ? inv_keep : 0.f;
    return r;
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
