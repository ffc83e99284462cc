// bf16 MFMA GEMM for gfx950:  C[M,N] = epilogue( sum_k A(m,k) * B(n,k) )      ("NT" convention)
//
// Operand layouts (per operand, chosen at run time -> template instance):
//   k-contiguous ("KC"):  X(r,k) at x[r*ld + k]      -- activations [M,K], torch Linear weight [N,K]
//   r-contiguous ("RC"):  X(r,k) at x[k*ld + r]      -- transposed views: dX = dY * W  (B = W as RC),
//                                                       dW = dY^T * X   (A = dY as RC, B = X as RC)
// KC tiles are staged [rows][64] with a 16-B-chunk XOR swizzle and read with ds_read_b128;
// RC tiles are staged [64][rows] exactly as they lie in memory (coalesced along r) and the MFMA fragments
// are produced by the CDNA4 transposing LDS read ds_read_b64_tr_b16, so no transposed copy of any tensor ever
// exists in HBM.
//
// 256 threads = 4 waves arranged WM x WN; v_mfma_f32_16x16x32_bf16 with the operands swapped
// (MFMA "A" = B-tile rows n, MFMA "B" = A-tile rows m) so that each lane ends up with 4 CONSECUTIVE n of one
// row m: epilogue loads/stores are 8-16 B per lane and bias/residual/activation fuse without shuffles.
//
// Epilogue (all optional, fused):  + bias[n]  ->  * act'(act_grad_of[m,n])  ->  save pre-activation (bf16)
//   -> act  ->  * dropout mask (counter RNG)  ->  + residual[m,n] (fp32)  ->  store fp32 and/or bf16.
// split-K (gridDim.z > 1) accumulates fp32 partials with global_atomic_add_f32 into a pre-zeroed C.
#include "common.h"
#include "vqa_hip.h"
#include "attn_core.h"
#include <hip/hip_ext.h>
#include <vector>
#include <algorithm>

namespace {

// ---- launch + optional kernel-timestamp profiling -------------------------------------------------------------------
// With profiling on (vqa_gemm_profile), every GEMM dispatch goes through hipExtLaunchKernel with a start / stop event pair:
// the events receive the kernel's own begin / end timestamps from its dispatch packet -- the quantity rocprofv3
// --kernel-trace reports -- so bench.py's roofline numbers are rocprof-equivalent without a profiler attached (an ordinary
// hipEventRecord bracket adds ~3 us of host / queue latency per launch).  Off (default): a plain launch, nothing recorded.
struct ProfRec { hipEvent_t a, b; double flop, bytes; int tag; };      // bytes: ALGORITHMIC bytes of the launch (operands once + every output / epilogue stream once)
std::vector<ProfRec> g_prof;
std::vector<hipEvent_t> g_prof_pool;      // events created when profiling is switched on, not per launch: two hipEventCreate calls per launch left the
                                          // GPU idle between kernels, and the same kernels then measured ~12 % longer than under rocprofv3 (5.5 vs 4.8 ms of
                                          // GEMM per step on one box) -- the measurement perturbed what it measured
bool g_prof_on = false;
int g_prof_tag = 0;

template <class Kern, class Arg>
inline void vqa_launch(Kern kern, dim3 grid, dim3 block, size_t lds, hipStream_t st, const Arg& arg, double flop, double bytes = 0.0) {
    if (!g_prof_on) { hipLaunchKernelGGL(kern, grid, block, lds, st, arg); return; }
    ProfRec r{nullptr, nullptr, flop, bytes, g_prof_tag};
    if (g_prof_pool.size() >= 2) { r.a = g_prof_pool.back(); g_prof_pool.pop_back(); r.b = g_prof_pool.back(); g_prof_pool.pop_back(); }
    else if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) { hipLaunchKernelGGL(kern, grid, block, lds, st, arg); return; }
    hipExtLaunchKernelGGL(kern, grid, block, (std::uint32_t)lds, st, r.a, r.b, 0, arg);
    g_prof.push_back(r);
}

constexpr int BK = 64;
constexpr int NTHREADS = 256;

struct GemmArgs {
    const h16_t* a; const h16_t* b;
    int M, N, K, lda, ldb;
    float* c_f32; int ldc_f32;
    h16_t* c_bf16; int ldc_bf16;
    h16_t* pre_bf16; int ld_pre;          // pre-activation save
    const float* bias;                      // [N]
    const float* residual; int ld_res;      // fp32 [M,N]
    const h16_t* act_grad_of; int ld_ag;   // multiply by act'(this) (backward through an activation)
    int act;                                // activation applied in the epilogue (forward)
    int act_bwd_kind;                       // activation whose derivative is applied (backward)
    float alpha;
    float drop_p; float drop_inv_keep; unsigned long long drop_seed; unsigned int drop_stream;
    int k_per_split;                        // multiple of BK
    int tiles_n; unsigned tiles_n_magic;    // tile columns of the launch's tile shape and ceil(2^32 / tiles_n) (0 when tiles_n == 1): the
                                            // tile-id -> (row, column) split costs one multiply instead of a ~40-instruction integer division
    float* colsum;                          // optional [N] fp32, pre-zeroed: += column sums of the stored values (bias gradient)
    int tiles_m_cm; unsigned tiles_m_magic; // > 0: COLUMN-major tile order (tile rows of the launch, and ceil(2^32 / tiles_m)): with the XCD remap every XCD then owns
                                            // a range of output COLUMNS, i.e. every weight line is fetched by ONE XCD (and the activations by all eight)
    int k_rotate;                           // ring kernels: workgroups on XCD x start their k loop x/8 of the way through K (see gemm_v1_body); bits 8..10: phase
                                            // added to x (tests: another assignment of starting points = another fp32 summation order of the same products)
    int epi_dma;                            // epilogue operand staged through LDS by DMA (ring kernels): 0 none, 1 act_grad_of, 2 residual (host: alignment)
    float* sumsq;                           // optional fp32 scalar: += sum of squares of the values stored to c_f32 (weight gradients: the optimiser's global-norm
                                            // reduction rides in the GEMM that produces them instead of re-reading 4 B per parameter)
#ifdef VQA_GEMM_TRACE
    unsigned long long* trace;              // lab builds only (scratch/gemm_lab.hip): 32 s_memtime stamps per workgroup
#endif
};
// algorithmic bytes of one launch: both operands read once, every output / fused epilogue stream once (what roofline.algorithmic_bytes sums)
static inline double gemm_alg_bytes(const GemmArgs& p) {
    const double mn = (double)p.M * p.N;
    return 2.0 * ((double)p.M * p.K + (double)p.N * p.K) + (p.c_f32 ? 4.0 * mn : 0.0) + (p.c_bf16 ? 2.0 * mn : 0.0) + (p.pre_bf16 ? 2.0 * mn : 0.0) +
           (p.residual ? 4.0 * mn : 0.0) + (p.act_grad_of ? 2.0 * mn : 0.0) + (p.bias ? 4.0 * p.N : 0.0) + (p.colsum ? 4.0 * p.N : 0.0);
}
#ifdef VQA_GEMM_TRACE
#define VQA_T(i) do { if (wave == 0) tr[i] = __builtin_readcyclecounter(); } while (0)
#else
#define VQA_T(i) do { } while (0)
#endif

// ---- LDS addressing -------------------------------------------------------------------------------------------
// KC tile: [ROWS][64] bf16, 128-B rows, 8 chunks of 16 B; chunk ^= (row>>1)&7  -> ds_read_b128 conflict-free
__device__ __forceinline__ int kc_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// RC tile: [64][ROWS] bf16 (ROWS*2-byte rows); 32-B (two-chunk) blocks are kept whole, block index XORed with a
// key of the k-row so that the 8 k-rows one half-wave touches in a ds_read_b64_tr_b16 hit distinct banks.
// One half-wave of a transposing read touches 8 k-rows {q, 8 + q} (q = 0..3, + 16 / 24 for the upper half) at ONE 32-B column
// block: conflict-free iff the 8 rows land on 8 different 32-B segments of the 256-B bank row.  With S = ROWS / 16 segments per
// k-row the un-swizzled segment of row r is r * S mod 8: S = 0 mod 8 (ROWS 128, 256): all equal -> 3 key bits; S = 4 mod 8 (64,
// 192): two values -> 2 key bits; S = 2 mod 4 (32, 96, 160, 288): rows q are apart already, rows 8 + q alias them -> 1 key bit.
// The key only flips bits inside an aligned group of 8 / 8 / 4 chunks, which ROWS / 8 is a multiple of in each case.
template <int ROWS>
__device__ __forceinline__ int rc_key(int krow) {
    constexpr int S = ROWS / 16;
    static_assert(ROWS % 32 == 0, "transposed tiles come in multiples of 32 rows");
    if (S % 8 == 0) return ((krow & 3) | (((krow >> 3) & 1) << 2)) << 1;
    else if (S % 8 == 4) return (((krow >> 1) & 1) | (((krow >> 3) & 1) << 1)) << 1;
    else return ((krow >> 3) & 1) << 1;
}
template <int ROWS>
__device__ __forceinline__ int rc_off(int krow, int chunk) { return krow * (ROWS * 2) + ((chunk ^ rc_key<ROWS>(krow)) << 4); }

template <int ROWS, bool KC>
struct Stage {
    static constexpr int CHUNKS = ROWS * BK / 8;
    static constexpr int PT = (CHUNKS + NTHREADS - 1) / NTHREADS;   // chunks per thread
    u32x4 v[PT];

    // R = number of valid rows (M or N), Kend = end of this split's k range
    __device__ __forceinline__ void load(const h16_t* __restrict__ g, int ld, int row0, int R, int k0, int Kend, int tid) {
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            const int c = tid + i * NTHREADS;
            u32x4 z = {0u, 0u, 0u, 0u};
            if (CHUNKS % NTHREADS == 0 || c < CHUNKS) {
                if (KC) {
                    const int row = c >> 3, kc = c & 7;
                    const int gr = row0 + row, gk = k0 + kc * 8;
                    if (gr < R && gk < Kend) z = *reinterpret_cast<const u32x4*>(g + (size_t)gr * ld + gk);
                } else {
                    constexpr int CPR = ROWS / 8;
                    const int krow = c / CPR, rc = c % CPR;
                    const int gk = k0 + krow, gr = row0 + rc * 8;
                    if (gk < Kend && gr < R) z = *reinterpret_cast<const u32x4*>(g + (size_t)gk * ld + gr);
                }
            }
            v[i] = z;
        }
    }
    __device__ __forceinline__ void store(char* lds, int tid) const {
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            const int c = tid + i * NTHREADS;
            if (CHUNKS % NTHREADS == 0 || c < CHUNKS) {
                int off;
                if (KC) off = kc_off(c >> 3, c & 7);
                else { constexpr int CPR = ROWS / 8; off = rc_off<ROWS>(c / CPR, c % CPR); }
                *reinterpret_cast<u32x4*>(lds + off) = v[i];
            }
        }
    }
};

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

// fragment of 16 rows (r0..r0+15) x 32 k (substep s) for lane: rows on lane&15, k = 8*(lane>>4)+j
template <int ROWS, bool KC, bool USE_TR>
__device__ __forceinline__ h16x8 load_frag(const char* lds, int r0, int s, int lane) {
    const int i = lane & 15, g = lane >> 4;
    if (KC) {
        return *reinterpret_cast<const h16x8*>(lds + kc_off(r0 + i, 4 * s + g));
    } else if (USE_TR) {
        // lane 4q+p of each 16-lane group addresses k-row q, columns 4p..4p+3 of the 4(k) x 16(r) block;
        // it receives column i: element j = k-row j.
        const int q = i >> 2, p = i & 3;
        const int col = r0 + 4 * p;                       // multiple of 4 elements = 8 bytes
        const int krow = 32 * s + 8 * g + q;
        const int o0 = rc_off<ROWS>(krow, col >> 3) + ((col & 7) << 1);
        const int o1 = rc_off<ROWS>(krow + 4, col >> 3) + ((col & 7) << 1);
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + o0));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + o1));
        union { struct { s16x4 a, b; } s; h16x8 v; } u;
        u.s.a = lo; u.s.b = hi;
        return u.v;
    } else {
        // portable fallback (no transposing read): 8 scalar 16-bit LDS reads
        union { unsigned short h[8]; h16x8 v; } u;
        const int col = r0 + i;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int krow = 32 * s + 8 * g + j;
            u.h[j] = *reinterpret_cast<const unsigned short*>(lds + rc_off<ROWS>(krow, col >> 3) + ((col & 7) << 1));
        }
        return u.v;
    }
}

// blockIdx -> output tile.  XCD-aware: workgroups are dealt round-robin over the 8 XCDs, so workgroup b gets the tile id
// chunk(b % 8) + b / 8 and every XCD (private 4-MiB L2) works on one contiguous, row-major range of tile ids.  (Walking that
// range in groups of tile-rows measured 2 % slower inside the training step than plain row-major order and was removed.)
__device__ __forceinline__ int xcd_remap(int t, int ntiles) {
    const unsigned q = (unsigned)ntiles >> 3, r = (unsigned)ntiles & 7u, xcd = (unsigned)t & 7u, idx = (unsigned)t >> 3;
    return (int)(xcd * q + min(xcd, r) + idx);           // bijective: XCD x owns q (+ 1 for x < r) consecutive tile ids; branch-free
}
// t -> (t / tiles_n, t % tiles_n) by one multiply: magic = ceil(2^31 / tiles_n) (div_magic; 2^31 itself for tiles_n = 1, so no special case and
// no branch), quotient = high word of 2t * magic: exact while t * tiles_n < 2^31
__device__ __forceinline__ void tile_from_linear(int tiles_n, unsigned magic, int t, int& tm, int& tn) {
    tm = (int)__umulhi((unsigned)t << 1, magic);
    tn = t - tm * tiles_n;
}
static inline unsigned div_magic(int d) { return d <= 1 ? 0x80000000u : (unsigned)((0x80000000ull + (unsigned)d - 1) / (unsigned)d); }
template <int BM, int BN>
__device__ __forceinline__ void tile_coords(const GemmArgs& p, int& tm, int& tn) {
    const int ntiles = p.tiles_n * ((p.M + BM - 1) / BM);
    tile_from_linear(p.tiles_n, p.tiles_n_magic, xcd_remap(blockIdx.x, ntiles), tm, tn);
}

// Epilogue.  The MFMA leaves lane (r = lane&15, g = lane>>4) with C[16i + r][16j + 4g .. +3]: written straight out, every
// store/load instruction touches 16 rows x 32 B (bf16) -- quarter cache lines, and the measured cost was 30-50 % of the
// whole kernel.  Instead each wave turns its tile through a PRIVATE LDS scratch strip of 16 rows (no workgroup barrier):
// afterwards lane l owns 4 consecutive columns 4*(l % LPR) of row l / LPR, so a wave instruction covers whole
// 128/256-B row segments for every stream of the epilogue (bias, act_grad_of, pre_bf16, residual, c_f32, c_bf16).
template <int TN> struct EpiScratch {
    static constexpr int PITCH = TN * 64 + 16;           // bytes per scratch row: 16*TN fp32 + 16 B (bank spread)
    static constexpr int BYTES = 16 * PITCH;             // per wave and 16-row strip
};
// strips of a wave tile that go through the scratch together: the largest power of two that fits `avail` bytes for NW waves
template <int TM, int TN> constexpr int epi_group(int avail, int nw) {
    int g = 1;
    while (g * 2 <= TM && TM % (g * 2) == 0 && nw * g * 2 * EpiScratch<TN>::BYTES <= avail) g *= 2;
    return g;
}

// ODMA (ring kernels): the wave tile of the ONE global epilogue operand of the launch -- the saved pre-activation (act_grad_of) or the fp32
// residual; the host sets GemmArgs::epi_dma only when exactly one of them is present -- is fetched into a per-wave LDS strip ``oper``
// (epi_oper_bytes<TM, TN>()) by LDS-DMA before the C tile is turned, and the rolled loop reads it from there through inline asm.  Read from
// global inside the loop, every iteration sat out its own dependent L2 / HBM round trip AND, because loads and stores share vmcnt, the
// completion of the previous iteration's global stores: 8 such iterations per wave on a 128x64 tile were the +6 us the GELU' epilogue cost
// the fc2 input-gradient GEMM (profiles/r01/gemm_epilogue_costs.log).  Prefetching through registers needs the loop unrolled, and unrolled
// code is cold-instruction-cache time; with the operand in LDS the loop holds no global load at all, so nothing in it waits on vmcnt.
// (The two forms are separate instantiations: with both in one body the compiler waits vmcnt(0) before the asm LDS read anyway -- it
// overwrites the register the other path's global load may still be filling.)
template <int TM, int TN> constexpr int epi_oper_bytes() { return (TM * 16 * TN * 16 * 4 + 1023) / 1024 * 1024; }     // sized for fp32
template <int TM, int TN, int G, bool ODMA = false, bool NOSPLIT = false>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, f32x4 (&acc)[TM][TN], int m_base, int n_base, int lane, char* scratch, char* oper = nullptr
#ifdef VQA_GEMM_TRACE
                                              , unsigned long long* etr = nullptr
#endif
                                              ) {
#ifdef VQA_GEMM_TRACE
#define VQA_ET(i) do { if (etr) etr[i] = __builtin_readcyclecounter(); } while (0)
#else
#define VQA_ET(i) do { } while (0)
#endif
    const bool splitk = !NOSPLIT && gridDim.z > 1;
    // dropout key: resolved here, at its only use (an INDIRECT seed costs one scalar load whose latency hides behind the stores);
    // resolving it at kernel entry by patching a copy of the argument struct put the struct in scratch memory and opened every
    // launch with a scratch store + load round trip
    const uint64_t drop_seed = (p.drop_p > 0.f && !splitk) ? resolve_seed(p.drop_seed) : 0ull;
    if (splitk) {
        // fp32 partials straight into the pre-zeroed C (lane holds C[m][n..n+3], m = 16i + (lane&15), n = 16j + 4*(lane>>4))
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m_base + 16 * i + (lane & 15);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n_base + 16 * j + 4 * (lane >> 4);
                if (m < p.M && n < p.N) {
                    float* dst = p.c_f32 + (size_t)m * p.ldc_f32 + n;
#pragma unroll
                    for (int r = 0; r < 4; ++r) atomicAdd(dst + r, acc[i][j][r] * p.alpha);
                }
            }
        }
        return;
    }
    constexpr int PITCH = EpiScratch<TN>::PITCH;
    constexpr int LPR = 4 * TN;                          // lanes per row after the turn
    static_assert(LPR == 4 || LPR == 8 || LPR == 16 || LPR == 32, "column-sum fold handles these lane groups");
    constexpr int RPI = 64 / LPR;                        // rows per wave instruction
    static_assert(TM % G == 0, "strip group must divide the wave tile");
    const int wr_off = (lane & 15) * PITCH + (lane >> 4) * 16;
    const int rd_row = lane / LPR, col = 4 * (lane % LPR);
    const int n = n_base + col;
    const bool nok = n < p.N;                            // N % 4 == 0 is enforced on the host
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (p.bias && nok) bv = *reinterpret_cast<const f32x4*>(p.bias + n);
    f32x4 cs = {0.f, 0.f, 0.f, 0.f};
    float ssq = 0.f;
    VQA_ET(0);
    const int okind = ODMA ? p.epi_dma : 0;
    const int oper_rb = TN * 16 * (okind == 1 ? 2 : 4);                 // bytes per row of the wave tile in the staged operand
    if (ODMA) {
        const int esz = okind == 1 ? 2 : 4, lpr = oper_rb >> 4, rpi = 64 / lpr;       // 16-B lanes per row, rows per 1-KiB DMA instruction
        const char* gsrc = okind == 1 ? reinterpret_cast<const char*>(p.act_grad_of) : reinterpret_cast<const char*>(p.residual);
        const size_t pitch = (size_t)(okind == 1 ? p.ld_ag : p.ld_res) * esz;
        const int nmax = p.N - 16 / esz;                                 // last whole 16-B chunk of a row (host: N % 8 == 0 for 16-bit operands)
#pragma unroll 1
        for (int i = 0; i * rpi < TM * 16; ++i) {
            const int row = min(i * rpi + lane / lpr, TM * 16 - 1);
            const int m = min(m_base + row, p.M - 1), nn = max(0, min(n_base + (lane % lpr) * (16 / esz), nmax));     // clamped rows / chunks are never used
            unsigned long long addr = reinterpret_cast<unsigned long long>(gsrc + (size_t)m * pitch + (size_t)nn * esz);
            asm volatile("" : "+v"(addr));
            __builtin_amdgcn_global_load_lds((gptr_t*)addr, (lptr_t*)(oper + i * 1024), 16, 0, 0);
        }
    }
    // The groups of G strips (as many as the ring holds beside the operand strip) are walked by a ROLLED loop as well: only the
    // accumulator -> scratch writes are specialised per group (registers cannot be indexed at run time), the body exists once.
#pragma unroll 1
    for (int g = 0; g < TM / G; ++g) {
        const int i0 = g * G;
#pragma unroll
        for (int gg = 0; gg < TM / G; ++gg) {
            if (gg == g) {
#pragma unroll
                for (int i = 0; i < G; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) *reinterpret_cast<f32x4*>(scratch + i * EpiScratch<TN>::BYTES + wr_off + j * 64) = acc[gg * G + i][j] * p.alpha;
            }
        }
        if (ODMA && g == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the staged operand landed (under the turn's LDS writes)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (g == 0) VQA_ET(1);
        // ROLLED (the body carries every fused option; unrolled TM x 16/RPI times it was ~100 KB of once-executed code that ran at
        // instruction-fetch speed -- and even a 4x unrolled chunk measured 6 % slower over the step than this form: at one cold
        // launch per kernel, code size is time).
        constexpr int NQ = G * 16 / RPI;
#pragma unroll 1
        for (int q = 0; q < NQ; ++q) {
            const int row = q * RPI + rd_row;            // 0 .. 16 G - 1 (scratch rows are contiguous across the G strips)
            const int trow = 16 * i0 + row, m = m_base + trow;
            if (m < p.M && nok) {
                h16x4 pv; f32x4 rv;
                if (ODMA) {
                    // inline asm: behind a plain LDS load the compiler waits for every outstanding global STORE of the loop (vmcnt
                    // counts them too) on the grounds that the load may alias an LDS-DMA
                    const unsigned oaddr = (unsigned)(uintptr_t)oper + trow * oper_rb;
                    if (okind == 1) asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(pv) : "v"(oaddr + col * 2));
                    else asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(rv) : "v"(oaddr + col * 4));
                } else {
                    if (p.act_grad_of) pv = *reinterpret_cast<const h16x4*>(p.act_grad_of + (size_t)m * p.ld_ag + n);
                    if (p.residual) rv = *reinterpret_cast<const f32x4*>(p.residual + (size_t)m * p.ld_res + n);
                }
                f32x4 v = *reinterpret_cast<const f32x4*>(scratch + row * PITCH + col * 4) + bv;
                if (ODMA ? okind == 1 : p.act_grad_of != nullptr) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] *= act_bwd((float)pv[r], p.act_bwd_kind);
                }
                if (p.pre_bf16) {
                    h16x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (h16_t)v[r];
                    *reinterpret_cast<h16x4*>(p.pre_bf16 + (size_t)m * p.ld_pre + n) = o;
                }
                if (p.act != ACT_NONE) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = act_fwd(v[r], p.act);
                }
                if (p.drop_p > 0.f) v *= dropout_scale4(drop_seed, p.drop_stream, (uint64_t)m * p.N + n, p.drop_p, p.drop_inv_keep);
                cs += v;                                  // column sums of the stored values BEFORE the residual (bias gradient)
                if (ODMA ? okind == 2 : p.residual != nullptr) v += rv;
                ssq += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
                if (p.c_f32) *reinterpret_cast<f32x4*>(p.c_f32 + (size_t)m * p.ldc_f32 + n) = v;
                if (p.c_bf16) {
                    h16x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (h16_t)v[r];
                    *reinterpret_cast<h16x4*>(p.c_bf16 + (size_t)m * p.ldc_bf16 + n) = o;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();                 // LDS executes a wave's accesses in order: the next group's writes follow these reads
        if (g == 0) VQA_ET(2);
    }
    VQA_ET(3);
    if (p.sumsq) {
        ssq = wave_sum(ssq);
        if (lane == 0) atomicAdd(p.sumsq + ((blockIdx.x & (VQA_SUMSQ_SLOTS - 1)) * VQA_SUMSQ_STRIDE), ssq);      // slotted: see include/vqa_hip.h
    }
    if (p.colsum) {
        // lanes l, l + LPR, l + 2 LPR ... hold the same 4 columns: fold them, then ONE lane per column group adds
#pragma unroll
        for (int o = LPR; o < 64; o <<= 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r)             // VALU lane exchanges (no ds_bpermute): rotations inside a 16-lane row keep l % LPR
                cs[r] = o == 4 ? cs[r] + dpp_f32<0x124>(cs[r]) : o == 8 ? cs[r] + dpp_f32<0x128>(cs[r]) : o == 16 ? xor16_sum(cs[r]) : xor32_sum(cs[r]);
        }
        if (lane < LPR && nok) {
#pragma unroll
            for (int r = 0; r < 4; ++r) atomicAdd(p.colsum + n + r, cs[r]);
        }
    }
}

// ---- SPECIALISED epilogues -------------------------------------------------------------------------------------------------------------
// The generic epilogue above carries every fused option behind run-time flags, so its row loop must stay rolled (unrolled it is ~100 KB of
// once-executed code) -- and a rolled body is one long dependent chain per iteration: LDS read -> wait -> a dozen skipped option branches ->
// convert -> store, ~600 - 700 cycles per 8 rows with 1.5 waves per SIMD left to hide anything (in-kernel stamps, scratch/gemm_lab.hip: the
// epilogue of a 64 x 64 tile is 4 such iterations + 1700 cycles of cold set-up = 5000 of the workgroup's 19 000 cycles at K = 768).  The step's
// Linear layers use NINE option sets on two tiles (scratch/gemm_census.py); for those the option set is a template argument (EPI): the flags
// fold at compile time, the body shrinks to the options in use, and all rows of a strip group go through as ONE batch -- every LDS read issued,
// one wait, the arithmetic of the rows interleaved, the stores back to back.  Same operations in the same order per element as the generic
// form (fp contraction off here: `v * keep + residual` sits in one basic block in this form and must not become an FMA the generic form
// cannot form), so the two agree to the bit (tests/test_kernels_gpu.py::test_specialised_epilogues_equal_the_generic_form_bit_for_bit).
enum : unsigned { E_S = 1u, E_BIAS = 2u, E_PRE = 4u, E_DROP = 8u, E_RES = 16u, E_F32 = 32u, E_B16 = 64u, E_COLSUM = 128u, E_ACT_SHIFT = 8, E_ACTB_SHIFT = 11, E_SUMSQ = 1u << 14 };
constexpr unsigned epi_make(bool bias, int act, bool pre, int actb, bool drop, bool res, bool f32, bool b16, bool colsum, bool sumsq = false) {
    return E_S | (bias ? E_BIAS : 0u) | (pre ? E_PRE : 0u) | (drop ? E_DROP : 0u) | (res ? E_RES : 0u) | (f32 ? E_F32 : 0u) | (b16 ? E_B16 : 0u) | (colsum ? E_COLSUM : 0u) |
           ((unsigned)act << E_ACT_SHIFT) | ((unsigned)actb << E_ACTB_SHIFT) | (sumsq ? E_SUMSQ : 0u);
}
// option set of a launch, or 0 when it has something no specialisation carries (alpha)
static inline unsigned epi_code(const GemmArgs& p) {
    if (p.alpha != 1.0f) return 0u;
    return epi_make(p.bias != nullptr, p.act, p.pre_bf16 != nullptr, p.act_grad_of ? p.act_bwd_kind : 0, p.drop_p > 0.f, p.residual != nullptr, p.c_f32 != nullptr,
                    p.c_bf16 != nullptr, p.colsum != nullptr, p.sumsq != nullptr);
}

template <int TM, int TN, int G, bool ODMA, unsigned EPI>
__device__ __forceinline__ void gemm_epilogue_s(const GemmArgs& p, f32x4 (&acc)[TM][TN], int m_base, int n_base, int lane, char* scratch, char* oper) {
#pragma clang fp contract(off)
    constexpr bool BIAS = (EPI & E_BIAS) != 0, PRE = (EPI & E_PRE) != 0, DROP = (EPI & E_DROP) != 0, RES = (EPI & E_RES) != 0, F32 = (EPI & E_F32) != 0,
                   B16 = (EPI & E_B16) != 0, CS = (EPI & E_COLSUM) != 0, SSQ = (EPI & E_SUMSQ) != 0;
    constexpr int ACT = (EPI >> E_ACT_SHIFT) & 7, ACTB = (EPI >> E_ACTB_SHIFT) & 7;
    static_assert(!ODMA || (RES != (ACTB != 0)), "the staged operand is the residual or the saved pre-activation, exactly one");
    static_assert(ODMA || (!RES && ACTB == 0), "a global epilogue operand is always staged in this form");
    constexpr int PITCH = EpiScratch<TN>::PITCH, LPR = 4 * TN, RPI = 64 / LPR, NQ = G * 16 / RPI;
    constexpr int OESZ = ACTB ? 2 : 4, OPER_RB = TN * 16 * OESZ;              // staged operand: element size, bytes per row of the wave tile
    const int wr_off = (lane & 15) * PITCH + (lane >> 4) * 16;
    const int rd_row = lane / LPR, col = 4 * (lane % LPR);
    const int n = n_base + col;
    const bool nok = n < p.N;                                                  // N % 4 == 0 (host); with a staged operand whole tiles in N as well
    if (ODMA) {
        constexpr int OLPR = OPER_RB >> 4, ORPI = 64 / OLPR;                   // 16-B lanes per row, rows per 1-KiB DMA instruction
        const char* gsrc = ACTB ? reinterpret_cast<const char*>(p.act_grad_of) : reinterpret_cast<const char*>(p.residual);
        const size_t pitch = (size_t)(ACTB ? p.ld_ag : p.ld_res) * OESZ;
#pragma unroll
        for (int i = 0; i * ORPI < TM * 16; ++i) {
            const int row = i * ORPI + lane / OLPR;
            const int m = min(m_base + row, p.M - 1);                          // clamped rows are never used
            unsigned long long addr = reinterpret_cast<unsigned long long>(gsrc + (size_t)m * pitch + (size_t)(n_base + (lane % OLPR) * (16 / OESZ)) * OESZ);
            asm volatile("" : "+v"(addr));
            __builtin_amdgcn_global_load_lds((gptr_t*)addr, (lptr_t*)(oper + i * 1024), 16, 0, 0);
        }
    }
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (BIAS && nok) bv = *reinterpret_cast<const f32x4*>(p.bias + n);
    const uint64_t drop_seed = DROP ? resolve_seed(p.drop_seed) : 0ull;
    f32x4 cs = {0.f, 0.f, 0.f, 0.f};
    float ssq = 0.f;
    const unsigned sbase = (unsigned)(uintptr_t)scratch + rd_row * PITCH + col * 4;
    const unsigned obase = ODMA ? (unsigned)(uintptr_t)oper + rd_row * OPER_RB + col * OESZ : 0u;
#pragma unroll
    for (int g = 0; g < TM / G; ++g) {
#pragma unroll
        for (int i = 0; i < G; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) *reinterpret_cast<f32x4*>(scratch + i * EpiScratch<TN>::BYTES + wr_off + j * 64) = acc[g * G + i][j];
        if (ODMA && g == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the staged operand landed (under the turn's LDS writes)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        f32x4 v[NQ], rv[NQ];
        h16x4 pv[NQ];
        // every LDS read of the batch through inline asm, one wait behind them (plain loads would be waited for one by one, and behind a plain
        // load of the staged operand the compiler also waits for the previous group's global stores: they share vmcnt with the DMA)
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            asm volatile("ds_read_b128 %0, %1" : "=v"(v[q]) : "v"(sbase + q * RPI * PITCH));
            if (ODMA) {
                const unsigned oa = obase + (16 * g * G + q * RPI) * OPER_RB;
                if (ACTB) asm volatile("ds_read_b64 %0, %1" : "=v"(pv[q]) : "v"(oa));
                else asm volatile("ds_read_b128 %0, %1" : "=v"(rv[q]) : "v"(oa));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            asm volatile("" : "+v"(v[q]));
            if (ODMA && ACTB) asm volatile("" : "+v"(pv[q]));
            if (ODMA && !ACTB) asm volatile("" : "+v"(rv[q]));
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int m = m_base + 16 * g * G + q * RPI + rd_row;
            if (m < p.M && nok) {
                f32x4 x = v[q] + bv;
                if (ACTB) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) x[r] *= act_bwd((float)pv[q][r], ACTB);
                }
                // (a 16-bit store takes a MATERIALISED fp32 value, as in the generic form: in the fp16 library the compiler otherwise folds the last
                // multiply into the conversion -- v_fma_mixlo_f16, one rounding instead of two -- and the two forms differ by an ulp now and then)
                if (PRE) {
                    asm volatile("" : "+v"(x));
                    h16x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (h16_t)x[r];
                    *reinterpret_cast<h16x4*>(p.pre_bf16 + (size_t)m * p.ld_pre + n) = o;
                }
                if (ACT) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) x[r] = act_fwd(x[r], ACT);
                }
                if (DROP) x *= dropout_scale4(drop_seed, p.drop_stream, (uint64_t)m * p.N + n, p.drop_p, p.drop_inv_keep);
                if (CS) cs += x;
                if (RES) x += rv[q];
                if (SSQ) ssq += (x[0] * x[0] + x[1] * x[1]) + (x[2] * x[2] + x[3] * x[3]);
                if (F32) *reinterpret_cast<f32x4*>(p.c_f32 + (size_t)m * p.ldc_f32 + n) = x;
                if (B16) {
                    asm volatile("" : "+v"(x));
                    h16x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (h16_t)x[r];
                    *reinterpret_cast<h16x4*>(p.c_bf16 + (size_t)m * p.ldc_bf16 + n) = o;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();                 // LDS executes a wave's accesses in order: the next group's writes follow these reads
    }
    if (SSQ) {
        ssq = wave_sum(ssq);
        if (lane == 0) atomicAdd(p.sumsq + ((blockIdx.x & (VQA_SUMSQ_SLOTS - 1)) * VQA_SUMSQ_STRIDE), ssq);      // slotted: see include/vqa_hip.h
    }
    if (CS) {
#pragma unroll
        for (int o = LPR; o < 64; o <<= 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                cs[r] = o == 4 ? cs[r] + dpp_f32<0x124>(cs[r]) : o == 8 ? cs[r] + dpp_f32<0x128>(cs[r]) : o == 16 ? xor16_sum(cs[r]) : xor32_sum(cs[r]);
        }
        if (lane < LPR && nok) {
#pragma unroll
            for (int r = 0; r < 4; ++r) atomicAdd(p.colsum + n + r, cs[r]);
        }
    }
}

template <int BM, int BN, int WM, int WN, bool A_KC, bool B_KC, bool USE_TR>
__global__ __launch_bounds__(NTHREADS) void gemm_kernel(const GemmArgs p) {
    constexpr int WTM = BM / WM, WTN = BN / WN;          // wave tile
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
    __shared__ __attribute__((aligned(16))) char smem[2 * (A_BYTES + B_BYTES)];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    // XCD-aware tile order: consecutive tiles of one M-row-panel share the A panel -> keep them on one XCD's L2
    int tm, tn;
    tile_coords<BM, BN>(p, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = blockIdx.z * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);
    const int nk = (kend - kbeg + BK - 1) / BK;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    Stage<BM, A_KC> sa;
    Stage<BN, B_KC> sb;
    constexpr int BUF_BYTES = A_BYTES + B_BYTES;           // buffer i: A tile at i*BUF_BYTES, B tile right behind it

    if (nk > 0) {
        sa.load(p.a, p.lda, m0, p.M, kbeg, kend, tid);
        sb.load(p.b, p.ldb, n0, p.N, kbeg, kend, tid);
        sa.store(smem, tid);
        sb.store(smem + A_BYTES, tid);
    }
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nk;
        if (more) {
            sa.load(p.a, p.lda, m0, p.M, kbeg + (kt + 1) * BK, kend, tid);
            sb.load(p.b, p.ldb, n0, p.N, kbeg + (kt + 1) * BK, kend, tid);
        }
        const char* la = smem + cur * BUF_BYTES;
        const char* lb = la + A_BYTES;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            h16x8 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = load_frag<BM, A_KC, USE_TR>(la, wm * WTM + 16 * i, s, lane);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = load_frag<BN, B_KC, USE_TR>(lb, wn * WTN + 16 * j, s, lane);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = VQA_MFMA16(fb[j], fa[i], acc[i][j]);
        }
        if (more) {
            sa.store(smem + (cur ^ 1) * BUF_BYTES, tid);
            sb.store(smem + (cur ^ 1) * BUF_BYTES + A_BYTES, tid);
        }
        __syncthreads();
    }

    static_assert(4 * EpiScratch<TN>::BYTES <= 2 * (A_BYTES + B_BYTES), "epilogue scratch does not fit the stage buffers");
    constexpr int EG = epi_group<TM, TN>(2 * (A_BYTES + B_BYTES), 4);
    gemm_epilogue<TM, TN, EG>(p, acc, m0 + wm * WTM, n0 + wn * WTN, lane, smem + wave * EG * EpiScratch<TN>::BYTES);
}

// ================================================================================================================
// v1: LDS-DMA pipeline.  Same tiles / fragments / epilogue, but operand tiles travel HBM -> LDS with
// global_load_lds_dwordx4 (no VGPR staging), 4 LDS stages of BK = 32, three tiles in flight behind COUNTED
// s_waitcnt vmcnt(N) and raw s_barrier (never __syncthreads(): it would drain the DMA queue).  The LDS image of a DMA is
// lane-linear (wave-uniform base + lane*16 B), so the bank-conflict swizzle is applied to each lane's SOURCE address
// and undone by the same XOR on the fragment reads.  Ragged edges: rows are clamped (their results are never stored),
// k beyond K reads a zero page.
// ================================================================================================================
__device__ __attribute__((aligned(256))) unsigned int g_zero_page[64];

// KC tile of the BK=32 ring, [ROWS][32] bf16: 64-B rows = 4 chunks, 4 rows per 256-B bank row.  ds_read_b128 is served in
// the lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31} (+32): one group reads rows {0-3,12-15} at chunk c and rows {4-11}
// at chunk c^1, so the XOR key must separate (row>>2) in {0,3} from {1,2}: key = ((row>>3)&1)<<1  (PMC: SQ_LDS_BANK_CONFLICT = 0).
// The BK=64 ring uses the [ROWS][64] layout of the register-staged kernel (kc_off): 128-B rows = whole cache lines per row.
__device__ __forceinline__ int kc1_key(int row) { return ((row >> 3) & 1) << 1; }
template <int BKT> __device__ __forceinline__ int kcT_key(int row) { return BKT == 64 ? ((row >> 1) & 7) : kc1_key(row); }
template <int BKT> __device__ __forceinline__ int kcT_off(int row, int chunk) { return row * (BKT * 2) + ((chunk ^ kcT_key<BKT>(row)) << 4); }


// Per-lane DMA descriptor of one 1-KiB wave-instruction of an operand tile: everything that does not depend on the
// k-step is computed ONCE (row clamp, swizzled chunk, zero-page redirection); the k loop only adds `step` to `ptr`.
struct DmaLane {
    unsigned long long ptr;      // source address of this lane's 16 B for the NEXT tile to issue
    unsigned long long step;     // bytes per k-tile (0 for lanes parked on the zero page)
    int kmax;                    // lane is inside K while tile_k0 < kmax (only consulted for the ragged last tile)
};

template <int ROWS, bool KC, int BKT, int NW>
__device__ __forceinline__ void dma_init(DmaLane (&d)[ROWS * (BKT / 8) / 64 / NW], const h16_t* __restrict__ g, int ld, int row0, int R,
                                         int kbeg, int kend, int wave, int lane) {
    constexpr int PER_WAVE = ROWS * (BKT / 8) / 64 / NW;
    constexpr int CPK = BKT / 8;                                  // 16-B chunks per k-contiguous row
    const unsigned long long zero = reinterpret_cast<unsigned long long>(g_zero_page);
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
        const int pos = (wave * PER_WAVE + i) * 64 + lane;       // 16-B chunk index in LDS order
        if (KC) {
            const int r = pos / CPK, c = (pos % CPK) ^ kcT_key<BKT>(r);
            const int gr = min(row0 + r, R - 1);
            d[i].ptr = reinterpret_cast<unsigned long long>(g + (size_t)gr * ld + kbeg + c * 8);
            d[i].step = BKT * 2;
            d[i].kmax = kend - c * 8;
        } else {
            constexpr int CPR = ROWS / 8;
            const int krow = pos / CPR, c = (pos % CPR) ^ rc_key<ROWS>(krow);
            const int gr = row0 + c * 8;
            if (gr < R) {
                d[i].ptr = reinterpret_cast<unsigned long long>(g + (size_t)(kbeg + krow) * ld + gr);
                d[i].step = (unsigned long long)BKT * ld * 2;
                d[i].kmax = kend - krow;
            } else { d[i].ptr = zero; d[i].step = 0; d[i].kmax = 0x7fffffff; }
        }
    }
}

template <int N_, bool CHECKED>
__device__ __forceinline__ void dma_issue(DmaLane (&d)[N_], char* lds_tile, int k0, int wave) {
#pragma unroll
    for (int i = 0; i < N_; ++i) {
        unsigned long long addr = d[i].ptr;
        if (CHECKED) addr = k0 < d[i].kmax ? addr : reinterpret_cast<unsigned long long>(g_zero_page);
        asm volatile("" : "+v"(addr));          // one opaque address -> exactly ONE DMA instruction (counted vmcnt waits)
        __builtin_amdgcn_global_load_lds((gptr_t*)addr, (lptr_t*)(lds_tile + (wave * N_ + i) * 1024), 16, 0, 0);
        d[i].ptr += d[i].step;
    }
}

// per-lane LDS byte offsets of the fragments of a wave tile for k-substep 0, computed once.  KC: o1 = offset of substep 1
// (BK=64 only: chunk index ^ 4, i.e. byte offset ^ 64).  RC: o0/o1 = the two transposing reads (k-rows q and q+4) of substep
// 0; substep 1 lies 32 k-rows further (an immediate, the swizzle key ignores bit 5 of the k-row).
template <int ROWS, bool KC, int BKT, int NT>
__device__ __forceinline__ void frag_offsets(int (&o0)[NT], int (&o1)[NT], int r_base, int lane) {
    const int i = lane & 15, g = lane >> 4;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int r0 = r_base + 16 * t;
        if (KC) { o0[t] = kcT_off<BKT>(r0 + i, g); o1[t] = o0[t] ^ 64; }
        else {
            const int q = i >> 2, pp = i & 3, col = r0 + 4 * pp, krow = 8 * g + q;
            o0[t] = rc_off<ROWS>(krow, col >> 3) + ((col & 7) << 1);
            o1[t] = rc_off<ROWS>(krow + 4, col >> 3) + ((col & 7) << 1);
        }
    }
}

// Fragment reads of the LDS-DMA ring.  KC: a plain ds_read_b128 (the compiler tracks its lgkmcnt).  RC: the transposing read
// is issued through INLINE ASM: behind the builtin, the compiler cannot prove that the read does not alias the LDS
// writes of the global_load_lds still in flight (other ring stages) and drains the whole DMA queue first
// (s_waitcnt vmcnt(0) before every fragment group -- measured: the ring ran with no overlap at all, 1430 cycles per k-step
// whatever its depth).  The asm reads are invisible to the compiler's counters, so the caller closes each group with
// frag_fence(): s_waitcnt lgkmcnt(0) plus a register dependency that keeps the MFMAs behind it.
template <int ROWS, bool KC>
__device__ __forceinline__ h16x8 load_frag1(const char* lds, int o0, int o1, int s) {
    if (KC) return *reinterpret_cast<const h16x8*>(lds + (s ? o1 : o0));
    constexpr int SUB = 32 * ROWS * 2;
    const unsigned base = (unsigned)(uintptr_t)lds + s * SUB;       // low 32 bits of a flat LDS address = LDS byte offset
    s16x4 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(base + o0));
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(base + o1));
    union { struct { s16x4 a, b; } s; h16x8 v; } u;
    u.s.a = lo; u.s.b = hi;
    return u.v;
}
// gemm_v1_body (the waves that read fragments are the waves that keep DMAs in flight): the k-contiguous read goes through inline asm as
// well.  Behind the plain C++ load the compiler placed an s_waitcnt vmcnt(0) in the MIDDLE of the fragment reads of some instances (64x64
// tiles on the 3-stage ring, both operands k-contiguous = every Linear forward with N < 1536: ISA of round 2, found when the fc2-forward shape
// 2048 x 768 x 3072 measured 33 us whatever the ring depth while its transposed-B twin took 20 us) -- the same may-alias-the-DMA reasoning as
// for the transposing read, applied per LDS offset: the ring was drained in two of three k-steps.
template <int ROWS, bool KC>
__device__ __forceinline__ h16x8 load_frag1_asm(const char* lds, int o0, int o1, int s) {
#ifdef VQA_KC_PLAIN_READS                  // A/B builds only (scratch/ab_build.sh): the round-1 form
    return load_frag1<ROWS, KC>(lds, o0, o1, s);
#endif
    if (!KC) return load_frag1<ROWS, KC>(lds, o0, o1, s);
    const unsigned base = (unsigned)(uintptr_t)lds;                 // low 32 bits of a flat LDS address = LDS byte offset
    h16x8 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(base + (s ? o1 : o0)));
    return v;
}
template <bool ANY_RC, int NA, int NB>
__device__ __forceinline__ void frag_fence(h16x8 (&fa)[NA], h16x8 (&fb)[NB]) {
    if (!ANY_RC) return;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < NA; ++i) asm volatile("" : "+v"(fa[i]));
#pragma unroll
    for (int j = 0; j < NB; ++j) asm volatile("" : "+v"(fb[j]));
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// WM_ x WN_ waves, each a (BM/WM_) x (BN/WN_) wave tile; BKT = 32 or 64 elements of k per ring stage.
// FAST (host: K % BKT == 0, no split-K, every 16-B chunk of a row-contiguous operand inside its row): the k-tile index is all the state the ring's
// refills carry.  Lane descriptors are bare base pointers; tile kk of an operand lies kk * (uniform byte step) behind them -- one scalar multiply per
// operand and one v_lshl_add_u64 per DMA -- so there is no per-lane step / limit, no zero page, no checked twin of every issue and no pointer
// rewind at the k-rotation's wrap.  What this buys is CODE SIZE in front of the first DMA: every dispatch starts with cold instruction and scalar
// caches (the packet's acquire invalidates them), and the first workgroup of a launch on a CU walks the set-up at ~13 cycles per instruction instead
// of ~4.5 (in-kernel stamps, scratch/gemm_lab.hip: entry -> ring primed 4100 cycles at the median against 1100 for a workgroup that finds the
// code resident).
template <int BM, int BN, int WM_, int WN_, int BKT, int STAGES1, bool A_KC, bool B_KC, bool FAST = false, unsigned EPI = 0>
__device__ __forceinline__ void gemm_v1_body(const GemmArgs& p, const int tile_linear) {
    constexpr int NW = WM_ * WN_;
    constexpr int WTM = BM / WM_, WTN = BN / WN_, TM = WTM / 16, TN = WTN / 16;
    constexpr int A_BYTES = BM * BKT * 2, B_BYTES = BN * BKT * 2, STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int PA = BM * (BKT / 8) / 64 / NW, PB = BN * (BKT / 8) / 64 / NW, GL = PA + PB;   // DMA instructions per wave per tile
    static_assert(PA >= 1 && PB >= 1, "tile too small for the wave count");
    static_assert(STAGES1 >= 2 && STAGES1 <= 6 && (STAGES1 - 2) * GL <= 63, "ring depth: vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(1024))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN_, wn = wave % WN_;
    int tm, tn;
    if (!FAST && p.tiles_m_cm > 0) tile_from_linear(p.tiles_m_cm, p.tiles_m_magic, tile_linear, tn, tm);      // column-major tile order: lab (vqa_set_gemm_tile_order(2))
    else tile_from_linear(p.tiles_n, p.tiles_n_magic, tile_linear, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = FAST ? 0 : blockIdx.z * p.k_per_split;
    const int kend = FAST ? p.K : min(p.K, kbeg + p.k_per_split);
    const int nk = FAST ? p.K / BKT : (kend - kbeg + BKT - 1) / BKT;
    const int nk_full = FAST ? nk : (kend - kbeg) / BKT;     // tiles entirely inside K: no per-lane k check needed
#ifdef VQA_GEMM_TRACE
    unsigned long long tr[32];
    for (int i = 0; i < 32; ++i) tr[i] = 0;
#endif
    VQA_T(0);

    // The ring is primed FIRST: the first tiles' flight from (cold) L2 / HBM -- ~2300 cycles on the in-kernel timeline -- runs
    // under the rest of the once-executed set-up code instead of after it.
    DmaLane da[PA], db[PB];
    unsigned long long fa_ptr[PA], fb_ptr[PB];               // FAST: this lane's 16 B of k tile 0
    // k ROTATION: the workgroups of XCD x (workgroups are dealt round-robin over the XCDs: x = blockIdx.x % 8) walk the k tiles starting x/8 of
    // the way through K and wrap around.  Every XCD reads the whole weight operand; started together at k = 0 all eight miss on the same lines
    // at the same time -- in the step the weights always come from HBM (profiles/r02/gemm_cold_weights.log: +2 - 4 us per launch) -- rotated, a
    // line is fetched from HBM for one XCD and found in the memory-side cache by the other seven.  fp32 accumulation order changes, nothing else.
    const int rot = (p.k_rotate && nk >= 8) ? (int)(((unsigned)((blockIdx.x + (p.k_rotate >> 8)) & 7) * (unsigned)nk) >> 3) : 0;
    int kk_next = rot;                                       // FAST: k tile the next issue fetches
    if constexpr (FAST) {
        constexpr int CPK = BKT / 8;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int pos = (wave * PA + i) * 64 + lane;
            if (A_KC) { const int r = pos / CPK, c = (pos % CPK) ^ kcT_key<BKT>(r); fa_ptr[i] = reinterpret_cast<unsigned long long>(p.a + (size_t)min(m0 + r, p.M - 1) * p.lda + c * 8); }
            else { const int krow = pos / (BM / 8), c = (pos % (BM / 8)) ^ rc_key<BM>(krow); fa_ptr[i] = reinterpret_cast<unsigned long long>(p.a + (size_t)krow * p.lda + m0 + c * 8); }
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int pos = (wave * PB + i) * 64 + lane;
            if (B_KC) { const int r = pos / CPK, c = (pos % CPK) ^ kcT_key<BKT>(r); fb_ptr[i] = reinterpret_cast<unsigned long long>(p.b + (size_t)min(n0 + r, p.N - 1) * p.ldb + c * 8); }
            else { const int krow = pos / (BN / 8), c = (pos % (BN / 8)) ^ rc_key<BN>(krow); fb_ptr[i] = reinterpret_cast<unsigned long long>(p.b + (size_t)krow * p.ldb + n0 + c * 8); }
        }
    } else {
        dma_init<BM, A_KC, BKT, NW>(da, p.a, p.lda, m0, p.M, kbeg, kend, wave, lane);
        dma_init<BN, B_KC, BKT, NW>(db, p.b, p.ldb, n0, p.N, kbeg, kend, wave, lane);
        if (rot) {
#pragma unroll
            for (int i = 0; i < PA; ++i) da[i].ptr += (unsigned long long)rot * da[i].step;
#pragma unroll
            for (int i = 0; i < PB; ++i) db[i].ptr += (unsigned long long)rot * db[i].step;
        }
    }
    const unsigned long long fa_step = A_KC ? (unsigned long long)(BKT * 2) : (unsigned long long)BKT * 2 * (unsigned)p.lda;
    const unsigned long long fb_step = B_KC ? (unsigned long long)(BKT * 2) : (unsigned long long)BKT * 2 * (unsigned)p.ldb;
    VQA_T(31);
    auto issue = [&](int t, int stage) {
        char* st = smem + stage * STAGE_BYTES;
        if constexpr (FAST) {
            const unsigned long long oa = (unsigned long long)(unsigned)kk_next * fa_step, ob = (unsigned long long)(unsigned)kk_next * fb_step;
#pragma unroll
            for (int i = 0; i < PA; ++i) {
                unsigned long long addr = fa_ptr[i] + oa;
                asm volatile("" : "+v"(addr));
                __builtin_amdgcn_global_load_lds((gptr_t*)addr, (lptr_t*)(st + (wave * PA + i) * 1024), 16, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                unsigned long long addr = fb_ptr[i] + ob;
                asm volatile("" : "+v"(addr));
                __builtin_amdgcn_global_load_lds((gptr_t*)addr, (lptr_t*)(st + A_BYTES + (wave * PB + i) * 1024), 16, 0, 0);
            }
            kk_next = kk_next + 1 == nk ? 0 : kk_next + 1;
        } else {
        int kk = t + rot;
        if (kk >= nk) kk -= nk;
        if (rot && kk == 0) {                                // wrap: this issue is k tile 0
#pragma unroll
            for (int i = 0; i < PA; ++i) da[i].ptr -= (unsigned long long)nk * da[i].step;
#pragma unroll
            for (int i = 0; i < PB; ++i) db[i].ptr -= (unsigned long long)nk * db[i].step;
        }
        if (kk < nk_full) { dma_issue<PA, false>(da, st, 0, wave); dma_issue<PB, false>(db, st + A_BYTES, 0, wave); }
        else { dma_issue<PA, true>(da, st, kbeg + kk * BKT, wave); dma_issue<PB, true>(db, st + A_BYTES, kbeg + kk * BKT, wave); }
        }
    };
#pragma unroll
    for (int t = 0; t < STAGES1 - 1; ++t)
        if (t < nk) issue(t, t);
    __builtin_amdgcn_sched_barrier(0);
    VQA_T(1);
    int ao0[TM], ao1[TM], bo0[TN], bo1[TN];
    frag_offsets<BM, A_KC, BKT, TM>(ao0, ao1, wm * WTM, lane);
    frag_offsets<BN, B_KC, BKT, TN>(bo0, bo1, wn * WTN, lane);

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

#ifndef VQA_GEMM_SWP        // default.  -DVQA_GEMM_SWP (scratch/ab_build.sh) builds the software-pipelined loop below instead
    for (int kt0 = 0; kt0 < nk; kt0 += STAGES1) {
#pragma unroll
        for (int s = 0; s < STAGES1; ++s) {                  // compile-time stage index: LDS addresses fold to immediates
            const int kt = kt0 + s;
            if (kt < nk) {
                const int rem = min(nk - 1 - kt, STAGES1 - 2);   // younger tiles that may stay in flight
                if (STAGES1 >= 6 && rem >= 4) wait_vmcnt<4 * GL>();
                else if (STAGES1 >= 5 && rem >= 3) wait_vmcnt<3 * GL>();
                else if (STAGES1 >= 4 && rem >= 2) wait_vmcnt<2 * GL>();
                else if (STAGES1 >= 3 && rem >= 1) wait_vmcnt<GL>();
                else wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();                // tile kt landed for every wave; stage (s-1) is free again
#ifdef VQA_GEMM_TRACE
                if (kt < 24) VQA_T(2 + kt);
#endif
#ifndef VQA_DMA_AFTER_READS
                if (kt + STAGES1 - 1 < nk) issue(kt + STAGES1 - 1, (s + STAGES1 - 1) % STAGES1);
#endif
                const char* la = smem + s * STAGE_BYTES;
                const char* lb = la + A_BYTES;
#pragma unroll
                for (int ks = 0; ks < BKT / 32; ++ks) {
                    h16x8 fa[TM], fb[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) fa[i] = load_frag1_asm<BM, A_KC>(la, ao0[i], ao1[i], ks);
#pragma unroll
                    for (int j = 0; j < TN; ++j) fb[j] = load_frag1_asm<BN, B_KC>(lb, bo0[j], bo1[j], ks);
#ifdef VQA_DMA_AFTER_READS      // A/B build: the refill of the ring is issued under the LDS round trip of the first fragment reads
                    if (ks == 0 && kt + STAGES1 - 1 < nk) issue(kt + STAGES1 - 1, (s + STAGES1 - 1) % STAGES1);
#endif
#ifdef VQA_KC_PLAIN_READS
                    frag_fence<!A_KC || !B_KC>(fa, fb);
#else
                    frag_fence<true>(fa, fb);
#endif
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = VQA_MFMA16(fb[j], fa[i], acc[i][j]);
                }
            }
        }
    }
#else
    // Software-pipelined k loop (BKT = 64 = two 32-deep substeps per tile): a wave's fragment reads of substep 1 are issued BEFORE the
    // MFMAs of substep 0, and the reads of the NEXT tile's substep 0 before the MFMAs of substep 1; tile kt+1 is awaited (counted vmcnt +
    // s_barrier) between the two MFMA batches of tile kt, when every wave has finished READING tile kt, and its stage refilled there.
    // MEASURED, NOT ADOPTED (round 2, same-box A/B, gpurun_out s2_ab_swp: bit-exact on every layout test): single launches 0 - 3 %
    // faster (three waves per SIMD already hide the LDS round trip), the grouped 128 x 128 weight-gradient launch SLOWER (293 -> 451 us:
    // the second fragment set takes the kernel from 136 to 324 registers, i.e. from two workgroups per CU to one), the step 7.25 ->
    // 7.72 ms.  And at one workgroup per CU the pipelined loop equals the serial one (23.4 vs 23.6 % of peak): the LDS round trip is
    // not what bounds that loop -- the waves' own DMA issue (~60 cycles per 1-KiB global_load_lds, 8 per wave and k-step on a 128 x 128
    // tile = as long as the k-step's 32 MFMAs) is, which is the L1 -> LDS path's 64 B/clk seen from the issuing wave.
    static_assert(BKT == 64, "two substeps per ring stage");
    auto wait_tile = [&](int t) {                            // tile t landed for this wave; younger tiles may stay in flight
        const int rem = min(nk - 1 - t, STAGES1 - 2);
        if (STAGES1 >= 6 && rem >= 4) wait_vmcnt<4 * GL>();
        else if (STAGES1 >= 5 && rem >= 3) wait_vmcnt<3 * GL>();
        else if (STAGES1 >= 4 && rem >= 2) wait_vmcnt<2 * GL>();
        else if (STAGES1 >= 3 && rem >= 1) wait_vmcnt<GL>();
        else wait_vmcnt<0>();
    };
    h16x8 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
    if (nk > 0) {
        wait_tile(0);
        __builtin_amdgcn_s_barrier();
        if (STAGES1 - 1 < nk) issue(STAGES1 - 1, STAGES1 - 1);
#pragma unroll
        for (int i = 0; i < TM; ++i) fa0[i] = load_frag1_asm<BM, A_KC>(smem, ao0[i], ao1[i], 0);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb0[j] = load_frag1_asm<BN, B_KC>(smem + A_BYTES, bo0[j], bo1[j], 0);
    }
    for (int kt0 = 0; kt0 < nk; kt0 += STAGES1) {
#pragma unroll
        for (int s = 0; s < STAGES1; ++s) {                  // compile-time stage index: LDS addresses fold to immediates
            const int kt = kt0 + s;
            if (kt < nk) {
                const char* la = smem + s * STAGE_BYTES;
                const char* lb = la + A_BYTES;
                frag_fence<true>(fa0, fb0);                  // substep 0 of tile kt is in registers (issued one MFMA batch ago)
#pragma unroll
                for (int i = 0; i < TM; ++i) fa1[i] = load_frag1_asm<BM, A_KC>(la, ao0[i], ao1[i], 1);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb1[j] = load_frag1_asm<BN, B_KC>(lb, bo0[j], bo1[j], 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = VQA_MFMA16(fb0[j], fa0[i], acc[i][j]);
                __builtin_amdgcn_sched_barrier(0);
                frag_fence<true>(fa1, fb1);                  // every read of tile kt by this wave is complete
                if (kt + 1 < nk) {
                    wait_tile(kt + 1);
                    __builtin_amdgcn_s_barrier();            // tile kt+1 landed for every wave; every wave is done reading stage s
#ifdef VQA_GEMM_TRACE
                    if (kt + 1 < 24) VQA_T(2 + kt + 1);
#endif
                    if (kt + STAGES1 < nk) issue(kt + STAGES1, s);
                    const char* na = smem + ((s + 1) % STAGES1) * STAGE_BYTES;
#pragma unroll
                    for (int i = 0; i < TM; ++i) fa0[i] = load_frag1_asm<BM, A_KC>(na, ao0[i], ao1[i], 0);
#pragma unroll
                    for (int j = 0; j < TN; ++j) fb0[j] = load_frag1_asm<BN, B_KC>(na + A_BYTES, bo0[j], bo1[j], 0);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = VQA_MFMA16(fb1[j], fa1[i], acc[i][j]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
#endif
    static_assert(NW * EpiScratch<TN>::BYTES <= STAGES1 * STAGE_BYTES, "epilogue scratch does not fit the ring");
    __syncthreads();                                         // every wave is done with the ring: it becomes epilogue scratch
    VQA_T(26);
    // ring layout in the epilogue: [NW x EG scratch strips][NW x operand strip] when both fit, else scratch only (operands read from global)
    constexpr int OPER = epi_oper_bytes<TM, TN>();
    constexpr bool OPER_OK = NW * (EpiScratch<TN>::BYTES + OPER) <= STAGES1 * STAGE_BYTES;
    constexpr int EG = epi_group<TM, TN>(STAGES1 * STAGE_BYTES, NW);
    constexpr int EGD = OPER_OK ? epi_group<TM, TN>(STAGES1 * STAGE_BYTES - NW * OPER, NW) : 1;
    if constexpr (EPI != 0) {
        constexpr bool S_ODMA = ((EPI & E_RES) != 0) != (((EPI >> E_ACTB_SHIFT) & 7) != 0);
        static_assert(!S_ODMA || OPER_OK, "the staged operand must fit the ring");
        if constexpr (S_ODMA)
            gemm_epilogue_s<TM, TN, EGD, true, EPI>(p, acc, m0 + wm * WTM, n0 + wn * WTN, lane, smem + wave * EGD * EpiScratch<TN>::BYTES,
                                                    smem + NW * EGD * EpiScratch<TN>::BYTES + wave * OPER);
        else
            gemm_epilogue_s<TM, TN, EG, false, EPI>(p, acc, m0 + wm * WTM, n0 + wn * WTN, lane, smem + wave * EG * EpiScratch<TN>::BYTES, nullptr);
    } else
    if (OPER_OK && p.epi_dma)
        gemm_epilogue<TM, TN, EGD, true, FAST>(p, acc, m0 + wm * WTM, n0 + wn * WTN, lane, smem + wave * EGD * EpiScratch<TN>::BYTES,
                                               smem + NW * EGD * EpiScratch<TN>::BYTES + wave * OPER);
    else
        gemm_epilogue<TM, TN, EG, false, FAST>(p, acc, m0 + wm * WTM, n0 + wn * WTN, lane, smem + wave * EG * EpiScratch<TN>::BYTES
#ifdef VQA_GEMM_TRACE
                                               , nullptr, wave == 0 ? tr + 20 : nullptr
#endif
                                               );
#ifdef VQA_GEMM_TRACE
    VQA_T(27);
    wait_vmcnt<0>();
    VQA_T(28);
    if (p.trace && wave == 0 && lane == 0) {
        tr[29] = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // XCC_ID
        tr[30] = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_ID
        for (int i = 0; i < 32; ++i) p.trace[(size_t)blockIdx.x * 32 + i] = tr[i];
    }
#endif
}

// waves per SIMD the ring's LDS footprint allows (workgroups per CU x waves per workgroup / 4 SIMDs, at most 4): given to the register
// allocator as the occupancy to keep -- the 128x64 forward kernel sat 4 registers over the 168 that three workgroups per CU need
constexpr int ring_waves_per_simd(int lds_bytes, int nw) {
    int wg = 160 * 1024 / lds_bytes, w = wg * nw / 4;
    return w < 1 ? 1 : w > 4 ? 4 : w;
}
template <int BM, int BN, int WM_, int WN_, int BKT, int STAGES1, bool A_KC, bool B_KC, bool FAST = false, unsigned EPI = 0>
__global__ __launch_bounds__(WM_ * WN_ * 64) __attribute__((amdgpu_waves_per_eu(ring_waves_per_simd(STAGES1 * (BM + BN) * BKT * 2, WM_ * WN_))))
void gemm_v1_kernel(const GemmArgs p) {
    const int ntiles = p.tiles_n * ((p.M + BM - 1) / BM);
#ifdef VQA_GEMM_PERSIST     // A/B builds only (scratch/ab_build.sh): the rounds-1/2 form
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        gemm_v1_body<BM, BN, WM_, WN_, BKT, STAGES1, A_KC, B_KC>(p, xcd_remap(t, ntiles));
        __syncthreads();
    }
#else
    // ONE tile per workgroup, no loop.  Through round 2 this was a persistent loop `for (t = blockIdx.x; t < ntiles; t += gridDim.x)` for a grid
    // cap that measured slower and was never switched on -- and the loop cost every launch: everything loop-invariant in the body (the epilogue's
    // ~40 argument words, its flag tests, the dropout / GELU constants, 60 SGPR spills into VGPR lanes) was hoisted by LICM into the loop
    // pre-header, i.e. IN FRONT of the first global_load_lds the body so carefully issues first: ~290 instructions and six kernarg round trips
    // before the ring was primed (ISA of round 3).
    // Every argument word the set-up reads, fetched as ONE batch of scalar loads: left alone, the compiler loads each word in the basic block that
    // first needs it -- three DEPENDENT kernarg round trips before the first DMA (~1000 cycles each for the first workgroup of a launch on a CU:
    // a launch's argument block is fresh memory, not in any cache).
#ifndef VQA_ARGS_LAZY
    asm volatile("" ::"s"(p.a), "s"(p.b), "s"(p.M), "s"(p.N), "s"(p.K), "s"(p.lda), "s"(p.ldb), "s"(p.k_per_split), "s"(p.tiles_n), "s"(p.tiles_n_magic),
                 "s"(p.tiles_m_cm), "s"(p.tiles_m_magic), "s"(p.k_rotate), "s"(p.bias), "s"(p.sumsq));     // + one word of each remaining 64-B line
#endif
    gemm_v1_body<BM, BN, WM_, WN_, BKT, STAGES1, A_KC, B_KC, FAST, EPI>(p, xcd_remap(blockIdx.x, ntiles));
#endif
}

// GROUPED launch: up to MAX_GROUP independent fp32-output GEMMs of one operand layout in ONE grid (the weight-gradient GEMMs
// of a backward pass: nothing waits for them, so they are queued and issued together).  One launch pays one cold start and
// one tail for all of them, and thousands of equal-cost tiles balance over the CUs where a single 768 x 768 output has 144.
constexpr int MAX_GROUP = 32;
struct GroupItem { const h16_t* a; const h16_t* b; float* c; int M, N, K, lda, ldb, ldc, tiles_n; unsigned tiles_n_magic; };
struct GroupArgs { int n; int k_rotate; float* sumsq; int tile_end[MAX_GROUP]; GroupItem it[MAX_GROUP]; };

template <int BM, int BN, int WM_, int WN_, int BKT, int STAGES1, bool A_KC, bool B_KC, unsigned EPI = 0>
__global__ __launch_bounds__(WM_ * WN_ * 64) void gemm_v1_grouped_kernel(const GroupArgs g) {
    // one tile per workgroup (the persistent form `for (tt = blockIdx.x; tt < total; tt += gridDim.x)` of rounds 1-2 is an A/B build only: see
    // gemm_v1_kernel for what the loop cost every launch)
    const int total = g.tile_end[g.n - 1];
    int i = 0;
#ifdef VQA_GEMM_PERSIST
    for (int tt = blockIdx.x; tt < total; tt += gridDim.x) {
#else
    {
        const int tt = blockIdx.x;
#endif
        const int t = xcd_remap(tt, total);
        if (t < (i ? g.tile_end[i - 1] : 0)) i = 0;
        while (i + 1 < g.n && t >= g.tile_end[i]) ++i;
        const GroupItem& it = g.it[i];
        GemmArgs p{};
        p.a = it.a; p.b = it.b; p.M = it.M; p.N = it.N; p.K = it.K; p.lda = it.lda; p.ldb = it.ldb;
        p.c_f32 = it.c; p.ldc_f32 = it.ldc;
        p.alpha = 1.f; p.drop_inv_keep = 1.f; p.tiles_n = it.tiles_n; p.tiles_n_magic = it.tiles_n_magic;
        p.k_per_split = (it.K + BKT - 1) / BKT * BKT;
        p.k_rotate = g.k_rotate;
        p.sumsq = g.sumsq;
        gemm_v1_body<BM, BN, WM_, WN_, BKT, STAGES1, A_KC, B_KC, false, EPI>(p, t - (i ? g.tile_end[i - 1] : 0));
#ifdef VQA_GEMM_PERSIST
        __syncthreads();                                     // the ring (epilogue scratch) is free again
#endif
    }
}

// ================================================================================================================
// ws: ONE right-sized tile per CU, waves specialised into loaders and MFMA consumers.
//
// Why.  At 32 samples per GPU the path's GEMMs have M = 2048 / 1600 rows and N in {768, 1536, 2304, 3072}: with 128x64 / 64x64
// tiles (what it takes to put >= 2 workgroups on every CU) each CU pulls (BM + BN) * 128 B through its 64 B/clk L1->LDS path for
// only 2 * BM * BN * 64 FLOP -- the k loop of gemm_v1 measured 50 B/clk/CU = L2-bound at ~53 % MFMA utilisation -- and the 1.5 - 3
// tiles per CU leave a tail.  Here the tile is chosen per (M, N) so that the grid is one wave of <= 256 workgroups, one per CU
// (2048 x 2304 -> 64 x 288, 2048 x 3072 -> 128 x 192, 1600 x 2304 -> 160 x 96 ...): 1.3 - 1.8x the FLOP per L2 byte, no tail.
// One workgroup per CU cannot hide its own DMA issue (~60 cycles per 1-KiB global_load_lds, 8 - 11 of them per wave and k-step)
// behind a neighbour, so the work is split by ROLE: waves 4-7 (one per SIMD) only issue the LDS-DMA ring and wait for it,
// waves 0-3 (one per SIMD, 2 x 2 over the tile) only read fragments and issue MFMAs; one raw s_barrier per k-step is the
// whole hand-shake (data: the loaders' counted vmcnt before the barrier; ring slot reuse: the consumers have finished slot
// kt-1 when they arrive at barrier kt).  Same LDS images, swizzles, fragment reads and fused epilogue options as gemm_v1.
// ================================================================================================================
template <int TM, int TN>
__device__ __forceinline__ void gemm_epilogue_ws(const GemmArgs& p, f32x4 (&acc)[TM][TN], int m_base, int n_base, int lane, char* scratch) {
    // any TN: the wave's (16 TM) x (16 TN) tile is turned through its LDS scratch once; afterwards the wave walks it as a flat
    // array of float4 (row-major), so every load / store instruction covers whole contiguous row segments of 64 TN bytes.
    constexpr int PITCH = TN * 64 + 16, C4 = 4 * TN;
    const uint64_t drop_seed = p.drop_p > 0.f ? resolve_seed(p.drop_seed) : 0ull;
    float* cs_lds = reinterpret_cast<float*>(scratch + 16 * TM * PITCH);       // [16 TN] column sums of this wave (bias gradient)
    if (p.colsum) for (int c = lane; c < 16 * TN; c += 64) cs_lds[c] = 0.f;
    const int wr_off = (lane & 15) * PITCH + (lane >> 4) * 16;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) *reinterpret_cast<f32x4*>(scratch + i * 16 * PITCH + wr_off + j * 64) = acc[i][j] * p.alpha;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    constexpr int NQ = TM * TN, NB = NQ % 3 == 0 ? 3 : NQ % 2 == 0 ? 2 : 1;      // chunked like gemm_epilogue: loads first, then use
#pragma unroll 1
    for (int q0 = 0; q0 < NQ; q0 += NB) {
        h16x4 pv[NB]; f32x4 rv[NB], vv[NB], bb[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int f = (q0 + u) * 64 + lane, row = f / C4, c4 = f - row * C4;
            const int m = m_base + row, n = n_base + 4 * c4;
            const bool ok = m < p.M && n < p.N;              // N % 4 == 0 is enforced on the host
            vv[u] = *reinterpret_cast<const f32x4*>(scratch + row * PITCH + c4 * 16);
            bb[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (p.bias && ok) bb[u] = *reinterpret_cast<const f32x4*>(p.bias + n);
            if (p.act_grad_of && ok) pv[u] = *reinterpret_cast<const h16x4*>(p.act_grad_of + (size_t)m * p.ld_ag + n);
            if (p.residual && ok) rv[u] = *reinterpret_cast<const f32x4*>(p.residual + (size_t)m * p.ld_res + n);
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int f = (q0 + u) * 64 + lane, row = f / C4, c4 = f - row * C4;
            const int m = m_base + row, n = n_base + 4 * c4;
            if (m >= p.M || n >= p.N) continue;
            f32x4 v = vv[u] + bb[u];
            if (p.act_grad_of) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] *= act_bwd((float)pv[u][r], p.act_bwd_kind);
            }
            if (p.pre_bf16) {
                h16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (h16_t)v[r];
                *reinterpret_cast<h16x4*>(p.pre_bf16 + (size_t)m * p.ld_pre + n) = o;
            }
            if (p.act != ACT_NONE) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = act_fwd(v[r], p.act);
            }
            if (p.drop_p > 0.f) v *= dropout_scale4(drop_seed, p.drop_stream, (uint64_t)m * p.N + n, p.drop_p, p.drop_inv_keep);
            if (p.colsum) {
#pragma unroll
                for (int r = 0; r < 4; ++r) atomicAdd(cs_lds + 4 * c4 + r, v[r]);      // ds_add_f32: <= 64 / C4 + 1 lanes per address
            }
            if (p.residual) v += rv[u];
            if (p.c_f32) *reinterpret_cast<f32x4*>(p.c_f32 + (size_t)m * p.ldc_f32 + n) = v;
            if (p.c_bf16) {
                h16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (h16_t)v[r];
                *reinterpret_cast<h16x4*>(p.c_bf16 + (size_t)m * p.ldc_bf16 + n) = o;
            }
        }
    }
    if (p.colsum) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int c = lane; c < 16 * TN; c += 64)
            if (n_base + c < p.N) atomicAdd(p.colsum + n_base + c, cs_lds[c]);
    }
}

template <int BM, int BN, int STAGES, bool A_KC, bool B_KC>
__global__ __launch_bounds__(512) void gemm_ws_kernel(const GemmArgs p) {
    constexpr int BKT = 64;
    constexpr int WTM = BM / 2, WTN = BN / 2, TM = WTM / 16, TN = WTN / 16;
    static_assert(BM % 32 == 0 && BN % 32 == 0, "2 x 2 consumer waves of 16-row MFMA tiles; 4 loader waves of 8-row DMA pieces");
    constexpr int A_BYTES = BM * BKT * 2, B_BYTES = BN * BKT * 2, STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int PA = BM / 32, PB = BN / 32, GL = PA + PB;        // DMA instructions per loader wave and k-tile
    static_assert(STAGES >= 3 && STAGES <= 5 && (STAGES - 2) * GL <= 63, "ring depth: vmcnt is a 6-bit counter");
    static_assert(4 * (16 * TM * (TN * 64 + 16) + 64 * TN) <= STAGES * STAGE_BYTES, "epilogue scratch does not fit the ring");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntiles = p.tiles_n * ((p.M + BM - 1) / BM);
    int tm, tn;
    tile_from_linear(p.tiles_n, p.tiles_n_magic, xcd_remap(blockIdx.x, ntiles), tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int nk = (p.K + BKT - 1) / BKT, nk_full = p.K / BKT;

    if (wave >= 4) {
        // ---------------------------------------------------------------- loaders: nothing but the ring
        const int lw = wave - 4;
#ifdef VQA_GEMM_TRACE
        unsigned long long trl[32];
        for (int i = 0; i < 32; ++i) trl[i] = 0;
        trl[0] = __builtin_readcyclecounter();
#endif
        DmaLane da[PA], db[PB];
        dma_init<BM, A_KC, BKT, 4>(da, p.a, p.lda, m0, p.M, 0, p.K, lw, lane);
        dma_init<BN, B_KC, BKT, 4>(db, p.b, p.ldb, n0, p.N, 0, p.K, lw, lane);
        auto issue = [&](int t, int stage) {
            char* st = smem + stage * STAGE_BYTES;
            if (t < nk_full) { dma_issue<PA, false>(da, st, 0, lw); dma_issue<PB, false>(db, st + A_BYTES, 0, lw); }
            else { dma_issue<PA, true>(da, st, t * BKT, lw); dma_issue<PB, true>(db, st + A_BYTES, t * BKT, lw); }
        };
#pragma unroll
        for (int t = 0; t < STAGES - 1; ++t)
            if (t < nk) issue(t, t);
#ifdef VQA_GEMM_TRACE
        trl[1] = __builtin_readcyclecounter();
#endif
        for (int kt0 = 0; kt0 < nk; kt0 += STAGES) {
#pragma unroll
            for (int s = 0; s < STAGES; ++s) {
                const int kt = kt0 + s;
                if (kt < nk) {
                    const int rem = min(nk - 1 - kt, STAGES - 2);    // younger tiles that may stay in flight
                    if (STAGES >= 5 && rem >= 3) wait_vmcnt<3 * GL>();
                    else if (STAGES >= 4 && rem >= 2) wait_vmcnt<2 * GL>();
                    else if (rem >= 1) wait_vmcnt<GL>();
                    else wait_vmcnt<0>();
#ifdef VQA_GEMM_TRACE
                    if (kt < 24) trl[2 + kt] = __builtin_readcyclecounter();         // tile kt landed (this loader's share)
#endif
                    __builtin_amdgcn_s_barrier();                    // tile kt is in LDS for everybody; slot (s-1) is free again
                    if (kt + STAGES - 1 < nk) issue(kt + STAGES - 1, (s + STAGES - 1) % STAGES);
                }
            }
        }
        __builtin_amdgcn_s_barrier();                                // pairs with the consumers' "ring becomes scratch" barrier
#ifdef VQA_GEMM_TRACE
        if (p.trace && wave == 4 && lane == 0) for (int i = 0; i < 32; ++i) p.trace[(size_t)blockIdx.x * 64 + 32 + i] = trl[i];
#endif
        return;
    }
#ifdef VQA_GEMM_TRACE
    unsigned long long tr[32];
    for (int i = 0; i < 32; ++i) tr[i] = 0;
#endif
    VQA_T(0);
    // -------------------------------------------------------------------- consumers: fragments + MFMA + epilogue
    const int wm = wave >> 1, wn = wave & 1;
    int ao0[TM], ao1[TM], bo0[TN], bo1[TN];
    frag_offsets<BM, A_KC, BKT, TM>(ao0, ao1, wm * WTM, lane);
    frag_offsets<BN, B_KC, BKT, TN>(bo0, bo1, wn * WTN, lane);
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    VQA_T(1);
    for (int kt0 = 0; kt0 < nk; kt0 += STAGES) {
#pragma unroll
        for (int s = 0; s < STAGES; ++s) {
            if (kt0 + s < nk) {
                __builtin_amdgcn_s_barrier();
#ifdef VQA_GEMM_TRACE
                if (kt0 + s < 24) VQA_T(2 + kt0 + s);
#endif
                const char* la = smem + s * STAGE_BYTES;
                const char* lb = la + A_BYTES;
#pragma unroll
                for (int ks = 0; ks < BKT / 32; ++ks) {
                    h16x8 fa[TM], fb[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) fa[i] = load_frag1<BM, A_KC>(la, ao0[i], ao1[i], ks);
#pragma unroll
                    for (int j = 0; j < TN; ++j) fb[j] = load_frag1<BN, B_KC>(lb, bo0[j], bo1[j], ks);
                    frag_fence<!A_KC || !B_KC>(fa, fb);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) acc[i][j] = VQA_MFMA16(fb[j], fa[i], acc[i][j]);
                }
            }
        }
    }
    __builtin_amdgcn_s_barrier();                                    // every consumer is done with the ring: it becomes epilogue scratch
    VQA_T(26);
    gemm_epilogue_ws<TM, TN>(p, acc, m0 + wm * WTM, n0 + wn * WTN, lane, smem + wave * (16 * TM * (TN * 64 + 16) + 64 * TN));
#ifdef VQA_GEMM_TRACE
    VQA_T(27);
    wait_vmcnt<0>();
    VQA_T(28);
    if (p.trace && wave == 0 && lane == 0) {
        tr[29] = __builtin_amdgcn_s_getreg((31 << 11) | 20);          // XCC_ID
        for (int i = 0; i < 32; ++i) p.trace[(size_t)blockIdx.x * 64 + i] = tr[i];
    }
#endif
}


// tiles of the ws kernel (BM, BN, ring stages); ws_pick() chooses per (M, N, K)
struct WsTile { int bm, bn, st; };
constexpr WsTile WS_TILES[] = {{64, 288, 3}, {128, 192, 3}, {64, 96, 4}, {160, 96, 4}, {160, 128, 3}, {160, 64, 4}, {128, 128, 3}};
constexpr int N_WS_TILES = sizeof(WS_TILES) / sizeof(WS_TILES[0]);
int g_ws_mode = 0;          // 0: off (default); 1: auto (ws_pick); 2 + i: force WS_TILES[i] for every eligible launch (lab sweeps)
                            // Measured (profiles/r02/gemm_ws.md): back-to-back launches of one shape run 5 - 25 % faster on ws tiles,
                            // the training step runs SLOWER with them (8.05 vs 7.31 ms): a 135-KB-LDS workgroup owns its CU, so the
                            // other encoder branch of the captured graph can no longer fill the gaps of this one's launches
unsigned g_ws_mask = 0xffffffffu;   // auto mode: bit i allows WS_TILES[i]

int g_force_cfg = -1, g_force_stages = 2;
int g_k_rotate = 0;        // k rotation per XCD in the ring kernels (vqa_set_gemm_k_rotate; see gemm_v1_body).  The models switch it ON for train()-mode steps
                           // (hip/kernels.py: set_training_numerics) and OFF otherwise: rotated, a row's fp32 summation order depends on the XCD that
                           // computes its tile, i.e. a sample's result depends on its batch position at the 16-bit rounding level -- harmless under
                           // dropout, unwanted for inference and for the bitwise properties the eval-mode tests hold.  One box, cfg2 train step:
                           // 7.16 -> 6.94 ms, GEMM sum 5.61 -> 5.36 ms; cfg3 9.58 -> 9.32 ms (profiles/r02/gemm_k_rotate.log)
int g_k_rotate_grouped = 0;  // k rotation in the grouped weight-gradient launch too (vqa_set_gemm_k_rotate(2)): lab
int g_k_rotate_phase = 0;    // 0..7, added to the XCD index before the starting point is derived (vqa_set_gemm_k_rotate(1 | phase << 8)): tests
int g_tile_order = 0;      // 0 / 1: row-major tile ids (default: an XCD owns rows of the output: activations fetched once, weights by every XCD); 2: column-major (lab, see vqa_gemm_bf16)
int g_grid_cap = 0;        // > 0: persistent LDS-DMA GEMMs on at most this many workgroups (vqa_set_gemm_grid_cap)
bool g_use_v1 = false;     // diagnostics: tile_hint launches use the LDS-DMA kernel when set
bool g_force_dma = false;
bool g_v1_epi = true;      // specialised epilogues (gemm_epilogue_s) where a launch's option set has one (vqa_set_gemm_v1_fast(1 | 2): bit 1 = 0 switches them off)
bool g_v1_fast = true;     // FAST instantiations of the ring kernel where the shape allows (vqa_set_gemm_v1_fast(0): A/B and tests of the general form)
int g_v1_stages = 2;       // measured: occupancy (32-KiB workgroups) beats deeper DMA rings at K <= 3072


template <int BM, int BN, int WM_, int WN_, int BKT, int ST, bool AK, bool BKC>
int launch_v1k(const GemmArgs& p, int splits, hipStream_t st) {
    constexpr int LDS = ST * (BM + BN) * BKT * 2;
    static bool attr_set = false;
    auto kern = gemm_v1_kernel<BM, BN, WM_, WN_, BKT, ST, AK, BKC>;
    // FAST instantiation (see gemm_v1_body) for the tiles the step's Linear layers run on
    constexpr bool HAS_FAST = BM * BN <= 128 * 64;
    if constexpr (HAS_FAST) {
        if (g_v1_fast && splits == 1 && p.tiles_m_cm == 0 && p.K % BKT == 0 && (AK || p.M % BM == 0) && (BKC || p.N % BN == 0)) {
            kern = gemm_v1_kernel<BM, BN, WM_, WN_, BKT, ST, AK, BKC, true>;
            // the step's nine Linear-layer option sets on their two tiles (scratch/gemm_census.py; gemm_epilogue_s): a global epilogue operand must be
            // the staged one (epi_dma: alignment checked by the caller)
            const unsigned code = (g_v1_epi && p.N % BN == 0) ? epi_code(p) : 0u;
            const bool staged_ok = (p.residual == nullptr && p.act_grad_of == nullptr) || p.epi_dma != 0;
#define VQA_EPI(...) if (code == epi_make(__VA_ARGS__)) kern = gemm_v1_kernel<BM, BN, WM_, WN_, BKT, ST, AK, BKC, true, epi_make(__VA_ARGS__)>
            if (code && staged_ok) {
                //                          bias   act             pre    act'            drop   res    f32    b16    colsum
                if constexpr (BM == 64 && BN == 64 && ST == 3 && AK && BKC) {
                    VQA_EPI(true, ACT_NONE, false, ACT_NONE, true, true, true, false, false);          // out_proj / fc2 forward, text tower (hidden dropout)
                    VQA_EPI(true, ACT_NONE, false, ACT_NONE, false, true, true, false, false);         // out_proj / fc2 forward, vision tower
                    VQA_EPI(true, ACT_NONE, false, ACT_NONE, false, false, false, true, false);        // projections to 16-bit (generative decoder: cross-attention q)
                    VQA_EPI(false, ACT_NONE, false, ACT_NONE, false, false, true, false, false);       // patch embedding
                }
                if constexpr (BM == 64 && BN == 64 && ST == 3 && AK && !BKC) {
                    VQA_EPI(false, ACT_NONE, false, ACT_NONE, false, true, true, false, false);        // input gradient + the residual stream's gradient
                    VQA_EPI(false, ACT_NONE, false, ACT_NONE, false, false, true, false, false);       // input gradient, fp32
                    VQA_EPI(false, ACT_NONE, false, ACT_NONE, false, false, false, true, false);       // input gradient, 16-bit
                }
                if constexpr (BM == 128 && BN == 64 && ST == 2 && AK && BKC) {
                    VQA_EPI(true, ACT_GELU_ERF, true, ACT_NONE, false, false, false, true, false);     // fc1 forward, text tower
                    VQA_EPI(true, ACT_QUICK_GELU, true, ACT_NONE, false, false, false, true, false);   // fc1 forward, vision tower
                    VQA_EPI(true, ACT_GELU_ERF, true, ACT_NONE, true, false, false, true, false);      // generative fusion / decoder: linear1 + GELU + dropout
                    VQA_EPI(true, ACT_NONE, false, ACT_NONE, false, false, false, true, false);        // packed in-projections to 16-bit
                    VQA_EPI(false, ACT_NONE, false, ACT_NONE, false, false, true, false, false);       // the 64 000-way output projection
                }
                if constexpr (BM == 128 && BN == 64 && ST == 2 && AK && !BKC) {
                    VQA_EPI(false, ACT_NONE, false, ACT_GELU_ERF, false, false, false, true, true);    // fc2 input gradient x GELU'(z), + fc1's bias gradient
                    VQA_EPI(false, ACT_NONE, false, ACT_QUICK_GELU, false, false, false, true, true);
                    VQA_EPI(false, ACT_NONE, false, ACT_GELU_ERF, true, false, false, true, true);     // ... through the dropout of the generative layers
                }
                // one-token-per-sample launches (experts, answer head): 32 x 32 tiles
                if constexpr (BM == 32 && BN == 32 && ST == 3 && AK && BKC) {
                    VQA_EPI(true, ACT_NONE, false, ACT_NONE, true, true, true, false, false);
                    VQA_EPI(true, ACT_NONE, false, ACT_NONE, false, false, false, true, false);
                    VQA_EPI(true, ACT_GELU_ERF, true, ACT_NONE, true, false, false, true, false);
                    VQA_EPI(true, ACT_NONE, false, ACT_NONE, false, false, true, false, false);
                }
                if constexpr (BM == 32 && BN == 32 && ST == 3 && AK && !BKC) {
                    VQA_EPI(false, ACT_NONE, false, ACT_NONE, false, true, true, false, false);
                    VQA_EPI(false, ACT_NONE, false, ACT_NONE, false, false, true, false, false);
                    VQA_EPI(false, ACT_NONE, false, ACT_NONE, false, false, false, true, false);
                    VQA_EPI(false, ACT_NONE, false, ACT_GELU_ERF, true, false, false, true, true);
                }
            }
#undef VQA_EPI
        }
    }
    if (!attr_set && LDS > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_v1_kernel<BM, BN, WM_, WN_, BKT, ST, AK, BKC>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        if constexpr (HAS_FAST) {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_v1_kernel<BM, BN, WM_, WN_, BKT, ST, AK, BKC, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
            if (e != hipSuccess) return (int)e;
        }
        attr_set = true;
    }
    const int tiles = ceil_div(p.M, BM) * ceil_div(p.N, BN);
    int grid = tiles;
#ifdef VQA_GEMM_PERSIST
    if (g_grid_cap > 0 && splits == 1 && tiles > g_grid_cap) grid = g_grid_cap / 8 * 8;
#endif
    vqa_launch(kern, dim3(grid, 1, splits), dim3(WM_ * WN_ * 64), LDS, st, p, 2.0 * p.M * p.N * p.K, gemm_alg_bytes(p));
    return (int)hipGetLastError();
}
template <int BM, int BN, int WM_, int WN_, int BKT, int ST>
int launch_v1s(const GemmArgs& p, int a_kc, int b_kc, int splits, hipStream_t st) {
    if (a_kc && b_kc) return launch_v1k<BM, BN, WM_, WN_, BKT, ST, true, true>(p, splits, st);
    if (a_kc && !b_kc) return launch_v1k<BM, BN, WM_, WN_, BKT, ST, true, false>(p, splits, st);
    if (!a_kc && !b_kc) return launch_v1k<BM, BN, WM_, WN_, BKT, ST, false, false>(p, splits, st);
    return launch_v1k<BM, BN, WM_, WN_, BKT, ST, false, true>(p, splits, st);
}
#ifdef VQA_GEMM_LAB
}  // namespace
#else
// LDS-DMA rings in use: BK = 64 (whole 128-B lines per k-contiguous row), 2 or 3 stages.  (The BK = 32 rings of the first
// version measured equal or slower on every shape of the path -- profiles/r01/gemm_tiles.log -- and are no longer built.)
template <int BM, int BN, int WM_, int WN_>
int launch_v1(const GemmArgs& p, int a_kc, int b_kc, int splits, int stages, hipStream_t st) {
    // deeper rings for the long-k shapes (K = 3072 with N = 768: A alone is 12.6 MB, nothing of the operands stays in an XCD's 4-MiB L2, every
    // k-step's DMA pays the memory-side latency -- the loop is bound by bytes in flight, not by L2 -> LDS bandwidth)
    if constexpr (BM == 64 && BN == 64) {
        if (stages >= 4) return launch_v1s<BM, BN, WM_, WN_, 64, 4>(p, a_kc, b_kc, splits, st);
    }
    if constexpr ((BM + BN) * 64 * 2 * 3 <= 160 * 1024 && BM * BN <= 128 * 64) {
        if (stages >= 3) return launch_v1s<BM, BN, WM_, WN_, 64, 3>(p, a_kc, b_kc, splits, st);
    }
    return launch_v1s<BM, BN, WM_, WN_, 64, 2>(p, a_kc, b_kc, splits, st);
}

template <int BM, int BN, int ST, bool AK, bool BKC>
int launch_wsk(const GemmArgs& p, hipStream_t st) {
    constexpr int LDS = ST * (BM + BN) * 64 * 2;
    static bool attr_set = false;
    auto kern = gemm_ws_kernel<BM, BN, ST, AK, BKC>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int tiles = ceil_div(p.M, BM) * ceil_div(p.N, BN);
    vqa_launch(kern, dim3(tiles), dim3(512), LDS, st, p, 2.0 * p.M * p.N * p.K, gemm_alg_bytes(p));
    return (int)hipGetLastError();
}
template <int BM, int BN, int ST>
int launch_ws(GemmArgs& p, int b_kc, hipStream_t st) {
    p.tiles_n = ceil_div(p.N, BN); p.tiles_n_magic = div_magic(p.tiles_n);
    return b_kc ? launch_wsk<BM, BN, ST, true, true>(p, st) : launch_wsk<BM, BN, ST, true, false>(p, st);
}
int launch_ws_tile(int i, GemmArgs& p, int b_kc, hipStream_t st) {
    switch (i) {
        case 0: return launch_ws<64, 288, 3>(p, b_kc, st);
        case 1: return launch_ws<128, 192, 3>(p, b_kc, st);
        case 2: return launch_ws<64, 96, 4>(p, b_kc, st);
        case 3: return launch_ws<160, 96, 4>(p, b_kc, st);
        case 4: return launch_ws<160, 128, 3>(p, b_kc, st);
        case 5: return launch_ws<160, 64, 4>(p, b_kc, st);
        default: return launch_ws<128, 128, 3>(p, b_kc, st);
    }
}
// Cycle model of one workgroup of tile t on a CU of its own: per k-step the slower of the consumers' MFMA issue (16 cycles per
// 16x16x32) and the ring's L2 -> LDS transfer (~55 B/clk/CU sustained), plus a fixed prologue / epilogue; x the number of
// rounds the grid needs over the 256 CUs.  Returns the best tile, or -1 when no ws tile covers the output in <= 2 rounds with
// >= 70 % of the CUs busy (small or odd outputs stay on gemm_v1).
int ws_pick(int M, int N, int K) {
    int best = -1; double best_t = 1e30;
    for (int i = 0; i < N_WS_TILES; ++i) {
        const WsTile& t = WS_TILES[i];
        if (!((g_ws_mask >> i) & 1u)) continue;
        const long tiles = (long)ceil_div(M, t.bm) * ceil_div(N, t.bn);
        const long rounds = (tiles + 255) / 256;
        if (rounds > 2 || tiles < 180 * rounds) continue;
        if ((double)M * N / ((double)tiles * t.bm * t.bn) < 0.85) continue;      // ragged edges would waste the tile
        const double mfma = (t.bm / 32) * (t.bn / 32) * 2 * 16.0, dma = (t.bm + t.bn) * 128 / 55.0;
        const double cyc = rounds * (6000.0 + ceil_div(K, 64) * (mfma > dma ? mfma : dma) * 1.1);
        if (cyc < best_t) { best_t = cyc; best = i; }
    }
    return best;
}


#include "fused_attn.h"      // fused in-projection + attention forward (uses the ring / fragment helpers above)
#include "gemm_dw256.h"      // 256 x 256 weight-gradient tiles (8 waves, one workgroup per CU)

bool g_use_tr = true;

template <int BM, int BN, int WM, int WN>
int launch_cfg(const GemmArgs& p, int a_kc, int b_kc, int splits, hipStream_t st) {
    const int tiles = ceil_div(p.M, BM) * ceil_div(p.N, BN);
    dim3 grid(tiles, 1, splits), block(NTHREADS);
#define VQA_LAUNCH(AK, BKC, TR) vqa_launch(gemm_kernel<BM, BN, WM, WN, AK, BKC, TR>, grid, block, 0, st, p, 2.0 * p.M * p.N * p.K, gemm_alg_bytes(p))
    if (a_kc && b_kc) VQA_LAUNCH(true, true, true);
    else if (a_kc && !b_kc) { if (g_use_tr) VQA_LAUNCH(true, false, true); else VQA_LAUNCH(true, false, false); }
    else if (!a_kc && !b_kc) { if (g_use_tr) VQA_LAUNCH(false, false, true); else VQA_LAUNCH(false, false, false); }
    else { if (g_use_tr) VQA_LAUNCH(false, true, true); else VQA_LAUNCH(false, true, false); }
#undef VQA_LAUNCH
    return (int)hipGetLastError();
}

}  // namespace

extern "C" int vqa_fused_inproj_attention_fwd(const VqaFusedAttnDesc* d, vqa_stream_t stream_) {
    if (!d || !d->xq || !d->xkv || !d->w_in || !d->o || d->B <= 0 || d->H <= 0) return VQA_ERR_ARG;
    if (d->D % 64 || d->D % d->H) return VQA_ERR_ARG;
    const int dh = d->D / d->H;
    if ((dh != 64 && dh != 96) || d->Sq < 1 || d->Sq > 64 || d->Skv < 1 || d->Skv > 64) return VQA_ERR_ARG;
    if ((d->ldxq | d->ldxkv | d->ldw | d->ldo) % 8 || (((uintptr_t)d->xq | (uintptr_t)d->xkv | (uintptr_t)d->w_in) & 15) || ((uintptr_t)d->o & 7)) return VQA_ERR_ARG;
    if (d->b_in && ((uintptr_t)d->b_in & 15)) return VQA_ERR_ARG;
    if ((d->q && (d->ldq % 8 || ((uintptr_t)d->q & 7))) || (d->k && (d->ldk % 8 || ((uintptr_t)d->k & 7))) || (d->v && (d->ldv % 8 || ((uintptr_t)d->v & 7)))) return VQA_ERR_ARG;
    FusedArgs p{};
    p.xq = (const h16_t*)d->xq; p.xkv = (const h16_t*)d->xkv; p.w = (const h16_t*)d->w_in; p.bias = d->b_in;
    p.q = (h16_t*)d->q; p.k = (h16_t*)d->k; p.v = (h16_t*)d->v;
    p.ldxq = d->ldxq; p.ldxkv = d->ldxkv; p.ldw = d->ldw; p.ldq = d->ldq; p.ldk = d->ldk; p.ldv = d->ldv; p.D = d->D;
    MArgs& a = p.a;
    a.o = (h16_t*)d->o; a.ldo = d->ldo; a.B = d->B; a.H = d->H; a.Sq = d->Sq; a.Skv = d->Skv;
    a.mask = d->key_padding_mask;
    a.causal = d->causal;
    a.scale = d->scale != 0.f ? d->scale : 1.0f / sqrtf((float)dh);
    a.drop_p = d->drop_p; a.inv_keep = d->drop_p > 0.f ? 1.f / (1.f - d->drop_p) : 1.f;
    a.seed = d->drop_seed; a.stream = d->drop_stream;
    hipStream_t st = (hipStream_t)stream_;
    return dh == 96 ? launch_fused_attn<96>(p, st) : launch_fused_attn<64>(p, st);
}

extern "C" void vqa_gemm_profile(int on, int tag) {
    g_prof_on = on != 0; g_prof_tag = tag;
    if (g_prof_on && g_prof_pool.size() < 2048) {
        while (g_prof_pool.size() < 4096) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) break; g_prof_pool.push_back(e); }
    }
}
extern "C" int vqa_gemm_profile_collect2(int ntags, double* flop, double* ms, int* launches, double* bytes) {
    for (int t = 0; t < ntags; ++t) { flop[t] = 0.0; ms[t] = 0.0; launches[t] = 0; if (bytes) bytes[t] = 0.0; }
    int rc = 0;
    for (ProfRec& r : g_prof) {
        float e = 0.f;
        hipError_t err = hipEventSynchronize(r.b);
        if (err == hipSuccess) err = hipEventElapsedTime(&e, r.a, r.b);
        if (err != hipSuccess) rc = (int)err;
        else if (r.tag >= 0 && r.tag < ntags) { flop[r.tag] += r.flop; ms[r.tag] += e; launches[r.tag] += 1; if (bytes) bytes[r.tag] += r.bytes; }
        g_prof_pool.push_back(r.a); g_prof_pool.push_back(r.b);        // back into the pool
    }
    g_prof.clear();
    return rc;
}
extern "C" int vqa_gemm_profile_collect(int ntags, double* flop, double* ms, int* launches) { return vqa_gemm_profile_collect2(ntags, flop, ms, launches, nullptr); }
extern "C" void vqa_set_gemm_ws(int mode) { if (mode >= 0x100) { g_ws_mode = 1; g_ws_mask = (unsigned)(mode >> 8); } else { g_ws_mode = mode; g_ws_mask = 0xffffffffu; } }
extern "C" void vqa_set_gemm_use_tr(int on) { g_use_tr = on != 0; }
extern "C" void vqa_set_gemm_v1_fast(int on) { g_v1_fast = (on & 1) != 0; g_v1_epi = on == 1 || (on & 2) != 0; }      // 0: general form; 1: default; 5: FAST without the specialised epilogues
#ifdef VQA_GEMM_PERSIST
extern "C" void vqa_set_gemm_grid_cap(int cap) { g_grid_cap = cap; }
#endif
extern "C" void vqa_set_gemm_force(int cfg, int stages) { g_force_cfg = cfg; g_force_stages = stages; }
extern "C" void vqa_set_gemm_tile_order(int order) { g_tile_order = order; }
extern "C" void vqa_set_gemm_k_rotate(int on) { g_k_rotate = (on & 0xff) != 0; g_k_rotate_grouped = (on & 0xff) >= 2; g_k_rotate_phase = (on >> 8) & 7; }
extern "C" void vqa_set_gemm_pipeline(int v1) {
    // diagnostics for tile_hint launches.  0: register-staged double buffer; 2 / 3: LDS-DMA ring with that many stages
    g_use_v1 = v1 != 0; g_force_dma = v1 != 0;
    g_v1_stages = v1 >= 3 ? 3 : 2;
}

extern "C" int vqa_gemm_bf16(const VqaGemmDesc* d, vqa_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!d || !d->a || !d->b || d->M <= 0 || d->N <= 0 || d->K <= 0) return VQA_ERR_ARG;
    // 16-byte vector access requirements
    if (d->lda % 8 || d->ldb % 8 || d->N % 4) return VQA_ERR_ARG;
    if (d->a_kc ? (d->K % 8) : (d->M % 8)) return VQA_ERR_ARG;
    if (d->b_kc ? (d->K % 8) : (d->N % 8)) return VQA_ERR_ARG;
    if (((uintptr_t)d->a | (uintptr_t)d->b) & 15) return VQA_ERR_ARG;
    if (d->c_f32 && (d->ldc_f32 % 4 || ((uintptr_t)d->c_f32 & 15))) return VQA_ERR_ARG;
    if (d->c_bf16 && (d->ldc_bf16 % 4 || ((uintptr_t)d->c_bf16 & 7))) return VQA_ERR_ARG;
    if (d->pre_bf16 && d->ld_pre % 4) return VQA_ERR_ARG;
    if (d->residual && d->ld_res % 4) return VQA_ERR_ARG;
    if (d->act_grad_of && d->ld_ag % 4) return VQA_ERR_ARG;
    if (!d->c_f32 && !d->c_bf16 && !d->pre_bf16) return VQA_ERR_ARG;

    GemmArgs p;
    p.a = (const h16_t*)d->a; p.b = (const h16_t*)d->b;
    p.M = d->M; p.N = d->N; p.K = d->K; p.lda = d->lda; p.ldb = d->ldb;
    p.c_f32 = d->c_f32; p.ldc_f32 = d->ldc_f32;
    p.c_bf16 = (h16_t*)d->c_bf16; p.ldc_bf16 = d->ldc_bf16;
    p.pre_bf16 = (h16_t*)d->pre_bf16; p.ld_pre = d->ld_pre;
    p.bias = d->bias; p.residual = d->residual; p.ld_res = d->ld_res;
    p.act_grad_of = (const h16_t*)d->act_grad_of; p.ld_ag = d->ld_ag;
    p.act = d->act; p.act_bwd_kind = d->act_bwd;
    p.alpha = d->alpha == 0.f ? 1.f : d->alpha;
    p.drop_p = d->drop_p; p.drop_inv_keep = d->drop_p > 0.f ? 1.f / (1.f - d->drop_p) : 1.f;
    p.drop_seed = d->drop_seed; p.drop_stream = d->drop_stream;
    p.colsum = d->colsum;
    p.sumsq = nullptr;
    // the launch's ONE global epilogue operand (saved pre-activation or residual, not both) is staged through LDS by DMA
    // (16-byte chunks: row starts and the row end must be 16-byte aligned); see gemm_epilogue
    p.epi_dma = 0;
#ifndef VQA_EPI_NO_DMA
    if (d->act_grad_of && !d->residual && d->ld_ag % 8 == 0 && d->N % 8 == 0 && !((uintptr_t)d->act_grad_of & 15)) p.epi_dma = 1;
    else if (d->residual && !d->act_grad_of && !((uintptr_t)d->residual & 15)) p.epi_dma = 2;          // ld_res % 4 and N % 4 are enforced above
#endif

    // tile choice: fill >= ~256 workgroups where the shape allows it
    int cfg;   // 0: 128x128, 1: 64x64, 2: 32x128 (skinny M), 3: 128x32 (skinny N), 4: 128x64, 5: 64x128, 6: 256x128 (8 waves), 7: 32x32, 8: 32x64
    bool dma = g_use_v1;     // LDS-DMA pipeline vs register-staged double buffer
    int stages = g_v1_stages;
    if (d->tile_hint > 0) cfg = d->tile_hint - 1;
    else if (d->M <= 32 && d->N >= 64) { cfg = 7; dma = true; stages = 3; }     // skinny M (answer head, experts at one token per sample):
                                                                                 // 32x32 tiles for workgroups, LDS-DMA ring for the long k (9 vs 18 us at 2048^2)
    else if (d->M <= 32) cfg = 2;
    else if (d->N <= 32) cfg = 3;
    else {
        // Measured on MI355X (profiles/r01/gemm_tiles.log, in-kernel timelines in profiles/r01/gemm_pmc.md).  The k loop
        // of the LDS-DMA ring runs near the CU's L2->LDS rate; what is left is per-launch cost (cold start, C stores), so
        // the choice is about workgroups per CU and bytes per FLOP:
        //   k-contiguous A and a wide N (>= 1536)  -> 128x64 tiles, 2-stage ring (3 workgroups per CU)
        //   everything else with >= 256 rows       -> 64x64 tiles, 3-stage ring (back-to-back microbenchmarks prefer 2 stages
        //                                             for short k / big grids; inside the step, beside the other encoder's
        //                                             launches, 3 stages everywhere measured 2 % faster end to end)
        //   32 < M < 256 (the Segmentation expert's 4 mask tokens x 32 samples; k-contiguous A)
        //                                          -> 32x32 tiles through a 3-stage ring, as for M <= 32: these launches stream a
        //                                             2048 x 2048 .. 6144 weight past a handful of rows, and what they need is
        //                                             workgroups (profiles/r02/gemm_mid_m.log: 128 x 2048 x 4096 40.5 -> 16.6 us,
        //                                             dX 128 x 2048 x 6144 60.5 -> 24.4 us against the register-staged 64x64 kernel)
        cfg = 1; dma = false;
        if (d->a_kc && d->M >= 512 && d->N >= 1536) { cfg = 4; dma = true; stages = 2; }
        else if (d->M >= 256 && d->N >= 64) {
            cfg = 1; dma = true; stages = 3;
        } else if (d->a_kc && d->N >= 64) { cfg = 7; dma = true; stages = 3; }
        if (g_force_cfg >= 0 && dma) { cfg = g_force_cfg; stages = g_force_stages; }     // diagnostics (vqa_set_gemm_force)
    }
    const int bm = cfg == 0 ? 128 : cfg == 1 ? 64 : cfg == 2 ? 32 : cfg == 3 ? 128 : cfg == 4 ? 128 : cfg == 6 ? 256 : cfg >= 7 ? 32 : 64;
    const int bn = cfg == 0 ? 128 : cfg == 1 ? 64 : cfg == 2 ? 128 : cfg == 3 ? 32 : cfg == 4 ? 64 : cfg == 7 ? 32 : cfg == 8 ? 64 : 128;
    const long tiles = (long)ceil_div(d->M, bm) * ceil_div(d->N, bn);
    p.tiles_n = ceil_div(d->N, bn); p.tiles_n_magic = div_magic(p.tiles_n);
    p.tiles_m_cm = 0; p.tiles_m_magic = div_magic(1);
    // only the encoder / fusion GEMMs over all tokens: their weights are what eight XCDs miss on together; the one-token-per-sample launches of the
    // experts and the head (M <= 128) have one or two tile rows, and their parity margins against the reference are the tightest of the path
    p.k_rotate = (g_k_rotate && d->M >= 256) ? (1 | (g_k_rotate_phase << 8)) : 0;
    // Column-major tile ids (vqa_set_gemm_tile_order(2)): under the XCD remap every XCD then owns a range of output COLUMNS, so every line of
    // the weight operand is fetched from HBM by one XCD instead of missing in eight L2s at once.  MEASURED, NOT ADOPTED (profiles/r02/gemm_tile_order.log):
    // launch by launch with weights streamed from HBM it wins on every shape (2048x2304x768 19.8 -> 16.1 us, dX 2048x3072x768 21.7 -> 18.3) -- but
    // that microbenchmark re-uses ONE activation buffer, warm in all eight L2s; in the step the activations were just written by the previous
    // kernel and every XCD now pulls all of them through the fabric: cfg2 7.21 -> 7.43 ms, GEMM sum 5.63 -> 5.92 ms, cfg3 9.72 -> 9.87 (one box).
    if (g_tile_order == 2 && ceil_div(d->M, bm) > 1) { p.tiles_m_cm = ceil_div(d->M, bm); p.tiles_m_magic = div_magic(p.tiles_m_cm); }

    int splits = d->split_k;
    const bool can_split = d->c_f32 && !d->c_bf16 && !d->pre_bf16 && !d->bias && !d->residual && !d->act_grad_of &&
                           d->act == ACT_NONE && d->drop_p == 0.f && !d->colsum;
    if (splits <= 0) {
        splits = 1;
        if (can_split && d->allow_split_k) {
            // few tiles: fill the chip; a very long reduction (the 64 000-way head's dX: 192 tiles x 1000 k-steps) is also cut while
            // every part keeps >= 64 k-steps, up to ~4 workgroups per CU
            while (((tiles * splits < 192 && d->K / (splits * 2) >= 4 * BK) || (tiles * splits < 1024 && d->K / (splits * 2) >= 64 * BK)) && splits < 16)
                splits *= 2;
        }
    }
    if (splits > 1 && !can_split) return VQA_ERR_ARG;
    int kps = ceil_div(ceil_div(d->K, splits), BK) * BK;
    splits = ceil_div(d->K, kps);
    p.k_per_split = kps;
    if (splits > 1 && !d->c_prezeroed) {
        hipError_t e = hipMemset2DAsync(d->c_f32, (size_t)d->ldc_f32 * 4, 0, (size_t)d->N * 4, d->M, stream);
        if (e != hipSuccess) return (int)e;
    }
    // one right-sized tile per CU (gemm_ws): k-contiguous A (forward and dX GEMMs), no split-K, K a multiple of 8
    if (g_ws_mode && g_use_tr && d->a_kc && splits == 1 && d->tile_hint == 0 && d->M >= 256) {
        const int wt = g_ws_mode >= 2 ? (g_ws_mode - 2 < N_WS_TILES ? g_ws_mode - 2 : -1) : ws_pick(d->M, d->N, d->K);
        if (wt >= 0 && (d->b_kc || d->N % 8 == 0)) return launch_ws_tile(wt, p, d->b_kc, stream);
    }
    if (cfg == 6) return launch_v1<256, 128, 4, 2>(p, d->a_kc, d->b_kc, splits, 2, stream);
    if ((dma || g_force_dma) && g_use_tr && cfg != 2 && cfg != 3) {
        switch (cfg) {
            case 7: return launch_v1<32, 32, 2, 2>(p, d->a_kc, d->b_kc, splits, stages, stream);
            case 8: return launch_v1<32, 64, 2, 2>(p, d->a_kc, d->b_kc, splits, stages, stream);
            case 0: return launch_v1<128, 128, 2, 2>(p, d->a_kc, d->b_kc, splits, stages, stream);
            case 1: return launch_v1<64, 64, 2, 2>(p, d->a_kc, d->b_kc, splits, stages, stream);
            case 4: return launch_v1<128, 64, 2, 2>(p, d->a_kc, d->b_kc, splits, stages, stream);
            default: return launch_v1<64, 128, 2, 2>(p, d->a_kc, d->b_kc, splits, stages, stream);
        }
    }
    switch (cfg) {
        case 0: return launch_cfg<128, 128, 2, 2>(p, d->a_kc, d->b_kc, splits, stream);
        case 1: return launch_cfg<64, 64, 2, 2>(p, d->a_kc, d->b_kc, splits, stream);
        case 2: return launch_cfg<32, 128, 1, 4>(p, d->a_kc, d->b_kc, splits, stream);
        case 3: return launch_cfg<128, 32, 4, 1>(p, d->a_kc, d->b_kc, splits, stream);
        case 4: return launch_cfg<128, 64, 2, 2>(p, d->a_kc, d->b_kc, splits, stream);
        case 7: return launch_cfg<32, 32, 2, 2>(p, d->a_kc, d->b_kc, splits, stream);
        case 8: return launch_cfg<32, 64, 1, 4>(p, d->a_kc, d->b_kc, splits, stream);
        default: return launch_cfg<64, 128, 2, 2>(p, d->a_kc, d->b_kc, splits, stream);
    }
}
int g_group_persistent = 0; // > 0: grouped launches run persistent on at most this many workgroups (vqa_set_gemm_group_persistent)
template <int BM, int BN, int ST, bool AK, bool BKC, int WM_ = 2, int WN_ = 2>
static int launch_grouped(const GroupArgs& g, hipStream_t st) {
    constexpr int LDS = ST * (BM + BN) * 64 * 2;
    // a grouped launch stores fp32 outputs and nothing else (+ the fused sum of squares): always the compile-time epilogue (gemm_epilogue_s) -- with
    // the experts' 32-token reductions the epilogue IS the kernel (one k-step against 16 row batches of a 128 x 128 tile)
    constexpr unsigned E0 = epi_make(false, ACT_NONE, false, ACT_NONE, false, false, true, false, false, false), E1 = E0 | E_SUMSQ;
    auto kern = g.sumsq ? gemm_v1_grouped_kernel<BM, BN, WM_, WN_, 64, ST, AK, BKC, E1> : gemm_v1_grouped_kernel<BM, BN, WM_, WN_, 64, ST, AK, BKC, E0>;
    static bool attr_set = false;
    if (!attr_set && LDS > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_v1_grouped_kernel<BM, BN, WM_, WN_, 64, ST, AK, BKC, E0>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_v1_grouped_kernel<BM, BN, WM_, WN_, 64, ST, AK, BKC, E1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    double flop = 0.0, bytes = 0.0;
    for (int i = 0; i < g.n; ++i) {
        flop += 2.0 * g.it[i].M * g.it[i].N * g.it[i].K;
        bytes += 2.0 * ((double)g.it[i].M + g.it[i].N) * g.it[i].K + 4.0 * (double)g.it[i].M * g.it[i].N;
    }
    int grid = g.tile_end[g.n - 1];
#ifdef VQA_GEMM_PERSIST
    if (g_group_persistent > 0 && grid > g_group_persistent) grid = g_group_persistent / 8 * 8;
#endif
    vqa_launch(kern, dim3(grid), dim3(WM_ * WN_ * 64), LDS, st, g, flop, bytes);
    return (int)hipGetLastError();
}

int g_group_tile = 0;      // 0: heuristic; 1: 64x64; 2: 128x64; 3: 128x128 (diagnostics: vqa_set_gemm_group_tile)
extern "C" void vqa_set_gemm_group_tile(int t) { g_group_tile = t; }
#ifdef VQA_GEMM_PERSIST
extern "C" void vqa_set_gemm_group_persistent(int n) { g_group_persistent = n; }
#endif

int g_dw256 = 1;            // 1 (default): weight-gradient items with 256-aligned outputs and 64-aligned token counts run on 256 x 256 tiles (gemm_dw256.h)
extern "C" void vqa_set_gemm_dw256(int on) { g_dw256 = on; }

static int grouped_v1(const VqaGemmGroupItem* items, int n, int a_kc, int b_kc, float* sumsq, hipStream_t st) {
    long kmin = 1 << 30, t64 = 0;
    for (int i = 0; i < n; ++i) {
        const VqaGemmGroupItem& d = items[i];
        t64 += (long)ceil_div(d.M, 64) * ceil_div(d.N, 64);
        kmin = d.K < kmin ? d.K : kmin;
    }
    // Tile: with thousands of tiles in the grid the per-workgroup DMA rate no longer decides (occupancy is free), bytes per
    // FLOP does: 128x128 once there are >= 3 rounds of them over the 256 CUs, else the 64x64 tile of the single launches.
    int tile = g_group_tile;
    if (tile == 0) tile = t64 / 4 >= 3 * 256 ? 3 : 1;
    // lab tiles (vqa_set_gemm_group_tile; scratch/group_dw_bench.py, none beats 128x128 / 2 stages at two workgroups per CU: 32 % of peak):
    // 4: 256x128 8 waves 2 stages (29 %); 5: 256x128 8 waves 3 stages (32 %); 7: 128x128 3 stages, one workgroup per CU (22 %)
    if (tile == 6 || tile > 7) return VQA_ERR_ARG;
    const int bm = tile == 1 ? 64 : (tile == 4 || tile == 5) ? 256 : 128, bn = (tile == 1 || tile == 2) ? 64 : 128;
    GroupArgs g{};
    g.n = n;
    g.k_rotate = g_k_rotate_grouped ? (1 | (g_k_rotate_phase << 8)) : 0;
    g.sumsq = sumsq;
    long tiles = 0;
    for (int i = 0; i < n; ++i) {
        const VqaGemmGroupItem& d = items[i];
        tiles += (long)ceil_div(d.M, bm) * ceil_div(d.N, bn);
        if (tiles > 0x3fffffff) return VQA_ERR_ARG;
        g.tile_end[i] = (int)tiles;
        g.it[i] = GroupItem{(const h16_t*)d.a, (const h16_t*)d.b, d.c_f32, d.M, d.N, d.K, d.lda, d.ldb, d.ldc, ceil_div(d.N, bn), div_magic(ceil_div(d.N, bn))};
    }
    // ring depth as for single launches: the third stage pays when k is long and the grid is under two workgroups per CU
    const bool deep = tile == 1 && kmin >= 2048 && tiles < 512;
#define VQA_G(AK, BKC) (tile == 4 ? launch_grouped<256, 128, 2, AK, BKC, 4, 2>(g, st) : tile == 5 ? launch_grouped<256, 128, 3, AK, BKC, 4, 2>(g, st) \
                        : tile == 7 ? launch_grouped<128, 128, 3, AK, BKC>(g, st) \
                        : tile == 3 ? launch_grouped<128, 128, 2, AK, BKC>(g, st) : tile == 2 ? launch_grouped<128, 64, 2, AK, BKC>(g, st) \
                        : deep ? launch_grouped<64, 64, 3, AK, BKC>(g, st) : launch_grouped<64, 64, 2, AK, BKC>(g, st))
    if (!a_kc && !b_kc) return VQA_G(false, false);
    if (a_kc && b_kc) return VQA_G(true, true);
    return VQA_ERR_ARG;          // mixed layouts: not instantiated (no caller)
#undef VQA_G
}

extern "C" int vqa_gemm_bf16_grouped2(const VqaGemmGroupItem* items, int n, int a_kc, int b_kc, float* sumsq, vqa_stream_t stream_) {
    if (!items || n <= 0 || n > 2 * DW_MAX_ITEMS) return VQA_ERR_ARG;
    for (int i = 0; i < n; ++i) {
        const VqaGemmGroupItem& d = items[i];
        if (!d.a || !d.b || !d.c_f32 || d.M <= 0 || d.N <= 0 || d.K <= 0) return VQA_ERR_ARG;
        if (d.lda % 8 || d.ldb % 8 || d.N % 4 || d.ldc % 4) return VQA_ERR_ARG;
        if (a_kc ? (d.K % 8) : (d.M % 8)) return VQA_ERR_ARG;
        if (b_kc ? (d.K % 8) : (d.N % 8)) return VQA_ERR_ARG;
        if (((uintptr_t)d.a | (uintptr_t)d.b | (uintptr_t)d.c_f32) & 15) return VQA_ERR_ARG;
    }
    hipStream_t st = (hipStream_t)stream_;
    // weight gradients (both operands token-major): the big regular outputs on 256 x 256 tiles (<= 64 items per launch), everything else -- and
    // every other layout -- through the ring kernel in launches of <= 32
    const VqaGemmGroupItem* big[2 * DW_MAX_ITEMS];
    VqaGemmGroupItem rest[2 * DW_MAX_ITEMS];
    int nbig = 0, nrest = 0;
    for (int i = 0; i < n; ++i) {
        if (!a_kc && !b_kc && g_dw256 && g_use_tr && dw256_eligible(items[i])) big[nbig++] = &items[i];
        else rest[nrest++] = items[i];
    }
    for (int i0 = 0; i0 < nbig; i0 += DW_MAX_ITEMS) {
        const int rc = launch_dw256(big + i0, nbig - i0 < DW_MAX_ITEMS ? nbig - i0 : DW_MAX_ITEMS, sumsq, st);
        if (rc) return rc;
    }
    for (int i0 = 0; i0 < nrest; i0 += MAX_GROUP) {
        const int rc = grouped_v1(rest + i0, nrest - i0 < MAX_GROUP ? nrest - i0 : MAX_GROUP, a_kc, b_kc, sumsq, st);
        if (rc) return rc;
    }
    return 0;
}
extern "C" int vqa_gemm_bf16_grouped(const VqaGemmGroupItem* items, int n, int a_kc, int b_kc, vqa_stream_t stream_) {
    if (n > MAX_GROUP) return VQA_ERR_ARG;
    return vqa_gemm_bf16_grouped2(items, n, a_kc, b_kc, nullptr, stream_);
}
#endif  // VQA_GEMM_LAB
