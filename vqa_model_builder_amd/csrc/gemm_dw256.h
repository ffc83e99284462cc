// Weight-gradient GEMM on 256 x 256 output tiles: dW[N_out, K_in] = dY^T X, both operands "r-contiguous" (the activations as they lie in HBM,
// [tokens][features]: the reduction index -- the token -- is the SLOW one), fp32 output, plain store.  Grouped like gemm_v1_grouped_kernel.
//
// Included by gemm.hip inside its anonymous namespace (shares the swizzle / DMA / epilogue helpers and vqa_launch).
//
// Why a second kernel.  The weight gradients are a third of the step's FLOPs (674 of 2023 GFLOP at batch 32) and the only GEMMs of the path
// that are BIG: ~110 outputs of 768 .. 3072 rows / columns with 1600 - 2048 tokens to reduce over, issued together at the end of the step with
// the chip to themselves.  On 128 x 128 tiles (4 waves, two workgroups per CU) they ran at 32 % of the MFMA peak whatever the ring depth
// (profiles/r02/group_dw_tiles.log): per k-step every wave issues 8 LDS-DMA instructions (~60 - 100 cycles each, in order, in front of its own
// 32 MFMAs = 512 cycles) and the CU pulls 64 KB through L2 -> LDS per 4.2 MFLOP.  Here ONE workgroup of 8 waves owns the CU with a 256 x 256
// tile: a wave tile of 128 x 64 makes it 8 DMA instructions per 64 MFMAs (1024 cycles) and 64 KB per 8.4 MFLOP -- half the issue cost and half
// the L2 bytes per FLOP -- and the accumulators (128 registers) leave room for ONE quadrant's fragments at a time (48 registers):
//
//   LDS (128 KB): A tile t&1 | A tile t&1^1 | B tile ... : [64 tokens][256 rows] bf16 images, 512-B rows, 32-B blocks XOR-swizzled (rc_off<256>);
//   per 64-token k-tile t:  the 8 DMA instructions of tile t+1 (into the other buffer) ride in front of phases 0 and 1;
//                           4 phases = the wave's 4 quadrants (64 rows x 32 columns, 16 MFMAs each) in the order (A0,B0) (A0,B1) (A1,B1) (A1,B0):
//                           every phase re-reads only the operand half that changed (ds_read_b64_tr_b16: the transposing LDS read turns the
//                           token-major image into k-contiguous MFMA fragments -- no transposed copy of any activation exists anywhere);
//                           s_waitcnt vmcnt(0) + ONE s_barrier per k-tile (tile t+1 landed for everybody; everybody is done reading tile t).
//   Fragment addresses: 8 + 4 VGPRs (one per 16-row block of the wave's operand half; the swizzle key is a lane constant, so block b of a
//   lane sits at base ^ (b << 5)); the k-substep (+16384), the second transposing read (+2048) and the buffer (+32768) are immediates.
//
// Eligible items: rows and columns multiples of 256, tokens a multiple of 64 (every Linear of the encoders / fusion block at 768 / 2304 / 3072
// features and 1600 / 2048 tokens); the grouped entry point sends the rest through gemm_v1_grouped_kernel as before.

constexpr int DW_BM = 256, DW_BN = 256, DW_LDS = 2 * (DW_BM + DW_BN) * 64 * 2;       // 131072
constexpr int DW_A_OFF = 0, DW_B_OFF = 65536, DW_BUF = 32768;                          // A0 | A1 | B0 | B1

template <int OFF>
__device__ __forceinline__ h16x8 dw_frag(unsigned addr) {
    s16x4 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(addr), "n"(OFF));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "n"(OFF + 2048));
    union { struct { s16x4 a, b; } s; h16x8 v; } u;
    u.s.a = lo; u.s.b = hi;
    return u.v;
}

struct DwTile {
    const h16_t* a; const h16_t* b; float* c;
    int lda, ldb, ldc, nk;          // nk: 64-token k-tiles
};

// One 256 x 256 output tile.  smem: the 128-KB ring; wave / lane as usual.
__device__ __forceinline__ void gemm_dw256_tile(const DwTile& T, int m0, int n0, float* sumsq, char* smem, int wave, int lane) {
    const int wm = wave >> 2, wn = wave & 3;
    // ---- DMA descriptors: instruction i of this wave covers LDS chunks (wave * 4 + i) * 64 .. + 63 of an operand image = 2 token rows
    unsigned long long pa[4], pb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pos = (wave * 4 + i) * 64 + lane, krow = pos >> 5, c = (pos & 31) ^ rc_key<256>(krow);
        pa[i] = reinterpret_cast<unsigned long long>(T.a + (size_t)krow * T.lda + m0 + c * 8);
        pb[i] = reinterpret_cast<unsigned long long>(T.b + (size_t)krow * T.ldb + n0 + c * 8);
    }
    const unsigned long long step_a = (unsigned long long)T.lda * 128, step_b = (unsigned long long)T.ldb * 128;      // 64 tokens x 2 B
    char* const dst_a = smem + DW_A_OFF + wave * 4096;
    char* const dst_b = smem + DW_B_OFF + wave * 4096;
    unsigned dbuf = DW_BUF;                                 // LDS offset of the buffer the NEXT tile is staged into (toggles 32768 <-> 0)
    auto issue_a = [&](int i0) {
#ifdef DW_LAB_NO_DMA
        return;
#endif
#pragma unroll
        for (int i = i0; i < i0 + 2; ++i) {
            unsigned long long addr = pa[i];
            asm volatile("" : "+v"(addr));
            __builtin_amdgcn_global_load_lds((gptr_t*)addr, (lptr_t*)(dst_a + dbuf + i * 1024), 16, 0, 0);
            pa[i] += step_a;
        }
    };
    auto issue_b = [&](int i0) {
#ifdef DW_LAB_NO_DMA
        return;
#endif
#pragma unroll
        for (int i = i0; i < i0 + 2; ++i) {
            unsigned long long addr = pb[i];
            asm volatile("" : "+v"(addr));
            __builtin_amdgcn_global_load_lds((gptr_t*)addr, (lptr_t*)(dst_b + dbuf + i * 1024), 16, 0, 0);
            pb[i] += step_b;
        }
    };
    // ---- fragment addresses (k-substep 0, first transposing read) in the buffer being READ; toggled by XOR 32768 after every k-tile
    unsigned fa_addr[8], fb_addr[4];
    {
        const int i16 = lane & 15, g = lane >> 4, q = i16 >> 2, pp = i16 & 3, krow = 8 * g + q;
        const unsigned base = (unsigned)(uintptr_t)smem;      // ring base: 1024-byte aligned, so XOR 32768 == +/- 32768 on every address below
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int col = wm * 128 + 16 * j + 4 * pp;
            fa_addr[j] = base + DW_A_OFF + rc_off<256>(krow, col >> 3) + ((col & 7) << 1);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = wn * 64 + 16 * j + 4 * pp;
            fb_addr[j] = base + DW_B_OFF + rc_off<256>(krow, col >> 3) + ((col & 7) << 1);
        }
    }
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // prologue: tile 0 into buffer 0
    dbuf = 0;
    issue_a(0); issue_a(2); issue_b(0); issue_b(2);
    dbuf = DW_BUF;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    h16x8 fa[4][2] = {}, fb[2][2] = {};
#ifdef DW_LAB_NO_READS      // lab (scratch/dw_lab.sh): the loop without its LDS fragment reads (fragments = whatever the registers hold)
#define DW_LOAD_A(AH) do { _Pragma("unroll") for (int i = 0; i < 4; ++i) { asm volatile("" : "+v"(fa[i][0])); asm volatile("" : "+v"(fa[i][1])); } } while (0)
#define DW_LOAD_B(BH) do { _Pragma("unroll") for (int j = 0; j < 2; ++j) { asm volatile("" : "+v"(fb[j][0])); asm volatile("" : "+v"(fb[j][1])); } } while (0)
#else
#define DW_LOAD_A(AH) do { _Pragma("unroll") for (int i = 0; i < 4; ++i) { fa[i][0] = dw_frag<0>(fa_addr[(AH) * 4 + i]); fa[i][1] = dw_frag<16384>(fa_addr[(AH) * 4 + i]); } } while (0)
#define DW_LOAD_B(BH) do { _Pragma("unroll") for (int j = 0; j < 2; ++j) { fb[j][0] = dw_frag<0>(fb_addr[(BH) * 2 + j]); fb[j][1] = dw_frag<16384>(fb_addr[(BH) * 2 + j]); } } while (0)
#endif
#define DW_FENCE() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) { asm volatile("" : "+v"(fa[i][0])); asm volatile("" : "+v"(fa[i][1])); } \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) { asm volatile("" : "+v"(fb[j][0])); asm volatile("" : "+v"(fb[j][1])); } \
        __builtin_amdgcn_sched_barrier(0); } while (0)
#ifdef DW_LAB_NO_MMA        // lab: the loop without its MFMAs (the fragments stay live)
#define DW_MMA(AH, BH) do { __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define DW_MMA(AH, BH) do { __builtin_amdgcn_s_setprio(1); \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) \
            acc[(AH) * 4 + i][(BH) * 2 + j] = VQA_MFMA16(fb[j][ks], fa[i][ks], acc[(AH) * 4 + i][(BH) * 2 + j]); \
        __builtin_amdgcn_s_setprio(0); __builtin_amdgcn_sched_barrier(0); } while (0)
#endif
    // One k-tile per iteration, ONE loop body.  The staging of the next tile is unconditional: behind the LAST tile the eight instructions
    // fetch one k-tile beyond the reduction (clamped to the last tile: nobody reads that buffer again) -- a branch-free body is what lets the
    // register allocator keep the 128 accumulator registers in place.
    const long long back_a = -(long long)step_a, back_b = -(long long)step_b;
    for (int t = 0; t < T.nk; ++t) {
        if (t + 1 == T.nk) {                                  // wave-uniform: point the staging at the last tile again
#pragma unroll
            for (int i = 0; i < 4; ++i) { pa[i] += back_a; pb[i] += back_b; }
        }
        issue_a(0); issue_a(2);
        DW_LOAD_A(0); DW_LOAD_B(0); DW_FENCE(); DW_MMA(0, 0);
        issue_b(0); issue_b(2);
        DW_LOAD_B(1); DW_FENCE(); DW_MMA(0, 1);
        DW_LOAD_A(1); DW_FENCE(); DW_MMA(1, 1);
        DW_LOAD_B(0); DW_FENCE(); DW_MMA(1, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        dbuf ^= DW_BUF;
#pragma unroll
        for (int j = 0; j < 8; ++j) fa_addr[j] ^= DW_BUF;
#pragma unroll
        for (int j = 0; j < 4; ++j) fb_addr[j] ^= DW_BUF;
    }
#undef DW_MMA
#undef DW_FENCE
#undef DW_LOAD_B
#undef DW_LOAD_A
    // ---- epilogue: the ring becomes the waves' turn-around scratch (the last barrier of the loop ordered every read before this)
    GemmArgs p{};
    p.M = m0 + 256; p.N = n0 + 256;                       // bounds of THIS tile: eligible items have no ragged edge
    p.c_f32 = T.c; p.ldc_f32 = T.ldc; p.alpha = 1.f; p.drop_inv_keep = 1.f;
    p.sumsq = sumsq;
    constexpr int EG = epi_group<8, 4>(DW_LDS, 8);
    gemm_epilogue<8, 4, EG>(p, acc, m0 + wm * 128, n0 + wn * 64, lane, smem + wave * EG * EpiScratch<4>::BYTES);
}

struct Dw256Args { GroupArgs g; float* sumsq; };

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_dw256_kernel(const Dw256Args A) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const GroupArgs& g = A.g;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int total = g.tile_end[g.n - 1];
    int i = 0;
    for (int tt = blockIdx.x; tt < total; tt += gridDim.x) {
        const int t = xcd_remap(tt, total);
        if (t < (i ? g.tile_end[i - 1] : 0)) i = 0;
        while (i + 1 < g.n && t >= g.tile_end[i]) ++i;
        const GroupItem& it = g.it[i];
        int tm, tn;
        tile_from_linear(it.tiles_n, it.tiles_n_magic, t - (i ? g.tile_end[i - 1] : 0), tm, tn);
        DwTile T{it.a, it.b, it.c, it.lda, it.ldb, it.ldc, it.K >> 6};
        gemm_dw256_tile(T, tm * 256, tn * 256, A.sumsq, smem, wave, lane);
        __syncthreads();                                     // the ring (epilogue scratch) is free again
    }
}

static bool dw256_eligible(const VqaGemmGroupItem& d) { return d.M % 256 == 0 && d.N % 256 == 0 && d.K % 64 == 0 && d.K >= 64; }

static int launch_dw256(const GroupArgs& g, float* sumsq, hipStream_t st) {
    static bool attr_set = false;
    auto kern = gemm_dw256_kernel;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, DW_LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    double flop = 0.0, bytes = 0.0;
    for (int i = 0; i < g.n; ++i) {
        flop += 2.0 * g.it[i].M * g.it[i].N * g.it[i].K;
        bytes += 2.0 * ((double)g.it[i].M + g.it[i].N) * g.it[i].K + 4.0 * (double)g.it[i].M * g.it[i].N;
    }
    int grid = g.tile_end[g.n - 1];
    if (grid > 256) grid = 256;                              // persistent: one workgroup per CU walks tiles b, b + 256, ... (a multiple of 8: the XCD remap stays a bijection)
    Dw256Args A{g, sumsq};
    vqa_launch(kern, dim3(grid), dim3(512), DW_LDS, st, A, flop, bytes);
    return (int)hipGetLastError();
}
