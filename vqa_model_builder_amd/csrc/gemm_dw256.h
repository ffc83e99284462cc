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
//   LDS (128 KB + a ticket word): A buffer 0 | A buffer 1 | B buffer 0 | B buffer 1: [64 tokens][256 rows] bf16 images, 512-B rows, 32-B blocks
//   XOR-swizzled (rc_off<256>);
//   per 64-token k-tile t:  the 8 DMA instructions of tile t+1 (into the other buffer) ride behind the fragment reads of phases 0 and 1;
//                           4 phases = the wave's 4 quadrants (64 rows x 32 columns, 16 MFMAs each) in the order (A0,B0) (A0,B1) (A1,B1) (A1,B0):
//                           every phase re-reads only the operand half that changed (ds_read_b64_tr_b16: the transposing LDS read turns the
//                           token-major image into k-contiguous MFMA fragments -- no transposed copy of any activation exists anywhere), and
//                           starts its first eight MFMAs when the fragments of the first 32 tokens have arrived (counted lgkmcnt);
//                           s_waitcnt vmcnt(0) + ONE s_barrier per k-tile (tile t+1 landed for everybody; everybody is done reading tile t);
//                           waves 4 - 7 run one phase behind waves 0 - 3 (STAGGER, below).
//   Fragment addresses: 8 + 4 VGPRs (one per 16-row block of the wave's operand half), toggled between the buffers by XOR 32768; the k-substep
//   (+16384) and the second transposing read (+2048) are immediates.
//
// Measured (one MI355X, random data, scratch/dw256_bench.py = the 96 weight gradients of a cfg2 step, scratch/dw_trace.py = in-kernel stamps):
//   128 x 128 ring kernel 0.85 ms = 29 % of the 2.5 PFLOP/s peak  ->  this kernel 0.68 ms = 36 %; cfg2 step 6.95 -> 6.77 ms (same box).
//   k-tile: 2200 cycles of work for the leading wave (2 x 1024 cycles of MFMA per SIMD is the floor) + ~1100 cycles until its SIMD partner
//   arrives = ~3300 - 3500 at 1.9 - 2.0 GHz (the clock the chip holds here; all-zero operands run 5 % faster: not power-bound); the epilogue
//   5.5 k cycles and ~2.5 k of prologue per 100 - 130 k-cycle tile.  What did NOT move it: an L2 prefetch two tiles ahead (slower), all eight
//   DMA instructions at the top of the k-tile, the k-substep split of the phases (neutral); what did: dynamic per-XCD tickets instead of
//   static rounds (-7 %), 96 instead of 32 items per launch, the unrolled epilogue (13 k -> 5.5 k cycles per tile), the stagger (-14 % cycles
//   per k-tile where every panel is shared by 12 tiles).  Without any DMA the loop still takes ~2900 cycles per k-tile: the bound is the lockstep
//   of eight waves around one barrier with a full LDS round trip in front of every 16 MFMAs, not a memory or matrix-pipe rate.
//
// Eligible items: rows and columns multiples of 256, tokens a multiple of 64 (every Linear of the encoders / fusion block at 768 / 2304 / 3072
// features and 1600 / 2048 tokens); the grouped entry point sends the rest through gemm_v1_grouped_kernel as before.

constexpr int DW_BM = 256, DW_BN = 256, DW_LDS = 2 * (DW_BM + DW_BN) * 64 * 2;       // 131072
constexpr int DW_SINK = DW_LDS + 64, DW_LDS_TOTAL = DW_SINK + 8 * 256;                 // ticket word | 8 x 256-B sinks of the L2 prefetches
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

#ifdef DW_TRACE             // lab (scratch/dw_trace.py): s_memtime stamps of the first tile of every workgroup
__device__ unsigned long long g_dw_trace[256 * 16];
#define DW_STAMP(i) do { if (trace && wave == 0 && lane == 0) trace[i] = __builtin_readcyclecounter(); } while (0)
#else
#define DW_STAMP(i) do { } while (0)
#endif

struct DwTile {
    const h16_t* a; const h16_t* b; float* c;
    int lda, ldb, ldc, nk;          // nk: 64-token k-tiles
};

// One 256 x 256 output tile.  smem: the 128-KB ring; wave / lane as usual.
__device__ __forceinline__ void gemm_dw256_tile(const DwTile& T, int m0, int n0, float* sumsq, char* smem, int wave, int lane, unsigned long long* trace = nullptr) {
    DW_STAMP(0);
#ifdef DW_TRACE
    if (trace && wave == 0 && lane == 0) trace[15] = __builtin_amdgcn_s_memrealtime();
#endif
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
    // L2 PREFETCH two tiles ahead -- LAB ONLY (-DDW_PREFETCH), MEASURED SLOWER.  Idea: the lead wave spends ~1100 cycles of every k-tile at the closing
    // wait; if that were the latency of the next tile's DMA under load (two 64-KB buffers cannot request a tile more than one tile ahead), touching
    // the lines of tile t + 2 with ONE extra LDS-DMA instruction per wave and k-tile (one dword of each of 64 cache lines, into a sink nobody reads)
    // would let the real DMA find them in L2.  Result (one MI355X, scratch/dw_trace.py): the closing wait GREW to 1650 - 1750 cycles and the step-
    // equivalent of weight gradients from 0.684 to 0.749 ms -- the 512 extra line requests per CU and k-tile cost more than they hide; issuing the
    // eight real DMA instructions at the very top of the k-tile did not move the wait either (3570 cycles per k-tile).  The wait is the SIMD partner
    // finishing, not memory: see the STAGGER note below.
    unsigned long long pf;
    {
        const int line = (wave & 3) * 64 + lane, tok = line >> 2, seg = line & 3;
        const h16_t* base = wave < 4 ? T.a + (size_t)tok * T.lda + m0 : T.b + (size_t)tok * T.ldb + n0;
        pf = reinterpret_cast<unsigned long long>(base + seg * 64) + (T.nk > 2 ? 2 : T.nk - 1) * (wave < 4 ? step_a : step_b);      // tile 2 (or the last one)
    }
    const unsigned long long pf_step = wave < 4 ? step_a : step_b;
    char* const sink = smem + DW_SINK + wave * 256;
    int pf_left = T.nk - 3;                                 // advances left: the pointer walks tiles 2 .. nk - 1 and stays on the last one
    auto prefetch = [&]() {
#ifdef DW_PREFETCH
        unsigned long long addr = pf;
        asm volatile("" : "+v"(addr));
        __builtin_amdgcn_global_load_lds((gptr_t*)addr, (lptr_t*)sink, 4, 0, 0);
        if (pf_left > 0) pf += pf_step;
        --pf_left;
#endif
    };
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
    DW_STAMP(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    DW_STAMP(2);

    h16x8 fa[4][2] = {}, fb[2][2] = {};
    // Fragment reads per k-substep (ks = 0: tokens 0..31 of the tile, ks = 1: +16384 B), so that a phase can start its first eight MFMAs when the
    // ks = 0 fragments have arrived (counted lgkmcnt: LDS reads return in order) while the ks = 1 reads are still in flight.
#define DW_LOAD_A(AH, KS) do { _Pragma("unroll") for (int i = 0; i < 4; ++i) fa[i][KS] = dw_frag<(KS) * 16384>(fa_addr[(AH) * 4 + i]); } while (0)
#define DW_LOAD_B(BH, KS) do { _Pragma("unroll") for (int j = 0; j < 2; ++j) fb[j][KS] = dw_frag<(KS) * 16384>(fb_addr[(BH) * 2 + j]); } while (0)
    // wait until at most N LDS reads are outstanding, then pin the fragments of substep KS behind the wait (rule: the MFMAs must not be hoisted over it)
#define DW_WAIT(N, KS) do { asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(N) : "memory"); \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(fa[i][KS])); \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(fb[j][KS])); \
        __builtin_amdgcn_sched_barrier(0); } while (0)
#define DW_MMA_KS(AH, BH, KS) do { __builtin_amdgcn_s_setprio(1); \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) \
            acc[(AH) * 4 + i][(BH) * 2 + j] = VQA_MFMA16(fb[j][KS], fa[i][KS], acc[(AH) * 4 + i][(BH) * 2 + j]); \
        __builtin_amdgcn_s_setprio(0); __builtin_amdgcn_sched_barrier(0); } while (0)
#define DW_MMA(AH, BH) do { DW_MMA_KS(AH, BH, 0); DW_MMA_KS(AH, BH, 1); } while (0)
    // the four phases of a k-tile; the DMA instructions of the next tile ride behind the reads of phases 0 and 1
#define DW_PHASE_AB(AH, BH, ISSUE) do { DW_LOAD_A(AH, 0); DW_LOAD_B(BH, 0); DW_LOAD_A(AH, 1); DW_LOAD_B(BH, 1); ISSUE; \
        DW_WAIT(12, 0); DW_MMA_KS(AH, BH, 0); DW_WAIT(0, 1); DW_MMA_KS(AH, BH, 1); } while (0)
#define DW_PHASE_B(AH, BH, ISSUE) do { DW_LOAD_B(BH, 0); DW_LOAD_B(BH, 1); ISSUE; \
        DW_WAIT(4, 0); DW_MMA_KS(AH, BH, 0); DW_WAIT(0, 1); DW_MMA_KS(AH, BH, 1); } while (0)
#define DW_PHASE_A(AH, BH) do { DW_LOAD_A(AH, 0); DW_LOAD_A(AH, 1); \
        DW_WAIT(8, 0); DW_MMA_KS(AH, BH, 0); DW_WAIT(0, 1); DW_MMA_KS(AH, BH, 1); } while (0)
    // One k-tile per iteration, ONE loop body per wave group.  The staging of the next tile is unconditional: behind the LAST tile the eight
    // instructions fetch the last tile once more (nobody reads that buffer again) -- a branch-free body is what lets the register allocator keep
    // the 128 accumulator registers in place.  Each operand's DMA instructions are issued BEHIND the phase's fragment reads (their issue time,
    // ~60 - 100 cycles apiece, passes while the LDS reads are in flight) and one full phase before the wait that needs them.
    //
    // STAGGER.  The two waves of a SIMD (w and w + 4) run the same program between the same barriers: left alone they wait for their fragments
    // at the same time (the matrix pipe idles) and want the pipe at the same time.  In-kernel stamps (scratch/dw_trace.py) put the k-tile at
    // ~3200 cycles for 2 x 1024 cycles of MFMA per SIMD, ~950 of them one wave waiting at the barrier for its SIMD partner.  Waves 4 - 7
    // therefore run ONE PHASE behind: they finish the fragment reads of a tile's last quadrant before the barrier (so the barrier still means
    // "nobody reads this buffer any more") but issue its 16 MFMAs AFTER it, from registers, while waves 0 - 3 are issuing the next tile's
    // DMA instructions and waiting for its first fragments -- and their own reads then fall under the MFMAs of waves 0 - 3.
    const long long back_a = -(long long)step_a, back_b = -(long long)step_b;
#ifdef DW_PREFETCH
#define DW_VMWAIT "s_waitcnt vmcnt(1)"                        // the next tile's eight DMA instructions landed; the prefetch (issued last) may stay in flight
#else
#define DW_VMWAIT "s_waitcnt vmcnt(0)"
#endif
#define DW_TILE_END() do { \
        asm volatile(DW_VMWAIT ::: "memory"); \
        __builtin_amdgcn_s_barrier(); \
        dbuf ^= DW_BUF; \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) fa_addr[j] ^= DW_BUF; \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) fb_addr[j] ^= DW_BUF; } while (0)
#define DW_LAST_TILE_FIXUP() do { if (t + 1 == T.nk) { _Pragma("unroll") for (int i = 0; i < 4; ++i) { pa[i] += back_a; pb[i] += back_b; } } } while (0)
#ifdef DW_NO_STAGGER
    const bool lead = true;
#else
    const bool lead = wave < 4;
#endif
    if (lead) {
        for (int t = 0; t < T.nk; ++t) {
            DW_LAST_TILE_FIXUP();
            DW_PHASE_AB(0, 0, (issue_a(0), issue_a(2)));
            DW_PHASE_B(0, 1, (issue_b(0), issue_b(2), prefetch()));
            DW_PHASE_A(1, 1);
            DW_PHASE_B(1, 0, (void)0);
#ifdef DW_TRACE
            if (t < 4) DW_STAMP(3 + 2 * t);
#endif
            DW_TILE_END();
#ifdef DW_TRACE
            if (t < 4) DW_STAMP(4 + 2 * t);
#endif
        }
    } else {
        for (int t = 0; t < T.nk; ++t) {
            DW_LAST_TILE_FIXUP();
            DW_MMA(1, 0);                                  // the previous tile's last quadrant (t == 0: zero fragments, adds nothing)
            DW_PHASE_AB(0, 0, (issue_a(0), issue_a(2)));
            DW_PHASE_B(0, 1, (issue_b(0), issue_b(2), prefetch()));
            DW_PHASE_A(1, 1);
            DW_LOAD_B(0, 0); DW_LOAD_B(0, 1); DW_WAIT(0, 0); DW_WAIT(0, 1);      // every LDS read of this tile is complete before the barrier
            DW_TILE_END();
        }
        DW_MMA(1, 0);
    }
#undef DW_LAST_TILE_FIXUP
#undef DW_TILE_END
#undef DW_PHASE_A
#undef DW_PHASE_B
#undef DW_PHASE_AB
#undef DW_MMA
#undef DW_MMA_KS
#undef DW_WAIT
#undef DW_LOAD_B
#undef DW_LOAD_A
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the last prefetch (into the sink) has landed before the wave moves on
    DW_STAMP(11);
    // ---- epilogue: the ring becomes the waves' turn-around scratch (the last barrier of the loop ordered every read before this).
    // Plain fp32 store of a 128 x 64 wave tile, two 16-row strips at a time through a per-wave LDS scratch so that every store instruction
    // covers whole 256-byte row segments (straight from the accumulators an instruction would touch 16 rows x 64 B).  UNROLLED: eight
    // ds_read_b128 in flight, then eight stores -- the rolled loop of gemm_epilogue (one read, one wait, one store per trip; sized for the cold
    // instruction cache of a 20-us kernel) took 13 k cycles per tile here, 12 % of the tile, and this kernel runs ten tiles per workgroup.
    {
        constexpr int PITCH = EpiScratch<4>::PITCH;                  // 272 B per scratch row
        char* scratch = smem + wave * 2 * EpiScratch<4>::BYTES;
        const int wr_off = (lane & 15) * PITCH + (lane >> 4) * 16;
        const int rd_row = lane >> 4, col = 4 * (lane & 15);        // after the turn: lane owns 4 consecutive columns of row (4 q + lane / 16)
        float* const cbase = T.c + (size_t)(m0 + wm * 128 + rd_row) * T.ldc + n0 + wn * 64 + col;
        float ssq = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(scratch + i * EpiScratch<4>::BYTES + wr_off + j * 64) = acc[2 * g + i][j];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            f32x4 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = *reinterpret_cast<const f32x4*>(scratch + (4 * q + rd_row) * PITCH + col * 4);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                ssq += v[q][0] * v[q][0] + v[q][1] * v[q][1] + v[q][2] * v[q][2] + v[q][3] * v[q][3];
                *reinterpret_cast<f32x4*>(cbase + (size_t)(32 * g + 4 * q) * T.ldc) = v[q];
            }
            __builtin_amdgcn_wave_barrier();                         // LDS executes a wave's accesses in order: the next group's writes follow these reads
        }
        if (sumsq) {
            ssq = wave_sum(ssq);
            if (lane == 0) atomicAdd(sumsq + ((blockIdx.x & (VQA_SUMSQ_SLOTS - 1)) * VQA_SUMSQ_STRIDE), ssq);
        }
    }
    DW_STAMP(12);
#ifdef DW_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DW_STAMP(13);
    if (trace && wave == 0 && lane == 0) { trace[14] = T.nk; trace[15] = __builtin_amdgcn_s_memrealtime() - trace[15]; }
#endif
}

// Items of one launch: up to 96 (the kernel-argument block is 4 KB).  Every dimension of an eligible item is a multiple of 64 and below 2^22, so it
// travels as a count of 64s in 16 bits.
constexpr int DW_MAX_ITEMS = 96;
struct Dw256Args {
    int n; float* sumsq; unsigned* ticket;               // ticket: 8 counters, one per XCD
    unsigned short chunk_end[8];                         // tiles [chunk_end[x - 1], chunk_end[x]) belong to XCD x (equal shares of the k-tiles)
    const void* ptr[DW_MAX_ITEMS][3];                    // a, b, c
    unsigned short dim[DW_MAX_ITEMS][5];                 // lda / 64, ldb / 64, ldc / 64, k-tiles, tile columns
    unsigned short tile_end[DW_MAX_ITEMS];
};
static_assert(sizeof(Dw256Args) <= 4096, "kernel arguments are limited to 4 KB");

// Tickets of the launches in flight: a launch takes the next slot, the stream zeroes it in front of the kernel (a memset node under capture), and the
// workgroups of the launch draw their tile numbers from it.
__device__ __attribute__((aligned(64))) unsigned g_dw_tickets[32 * 16];      // 32 slots of 64 bytes: 8 counters each (the memset node stays a multiple of 16 bytes)
int g_dw_slot = 0;

// Tiles are handed out DYNAMICALLY and PER XCD.  The items are sorted by reduction length (longest first) and the tile sequence is cut into eight
// contiguous chunks of equal work; the workgroups of XCD x (workgroups are dealt round-robin over the XCDs: x = blockIdx.x % 8) start on the first
// tiles of chunk x and draw every further tile from the chunk's ticket counter -- so the 32 CUs of an XCD work on ~32 CONSECUTIVE tiles, which share
// their operand panels through that XCD's L2 (a 3072 x 768 output: 11 + 3 panels for 32 tiles instead of 64) --; an XCD that runs dry takes tiles
// from the chunks of the others.  History (scratch/dw_trace.py, in-kernel stamps): static rounds (tile b, b + 256, ...) ran 4 rounds for 3.4 rounds
// of work per launch; one global ticket counter removed the rounds but spread neighbouring tiles over all eight L2s: the k-tile of the layer-shaped
// mix took 3600 - 3800 cycles (1100 of them waiting for DMA) against 2970 where every panel is shared by 12 tiles.
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_dw256_kernel(const Dw256Args A) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int* next = reinterpret_cast<int*>(smem + DW_LDS);        // one word behind the ring (same dynamic allocation: no second __shared__ object)
    const int x = blockIdx.x & 7, wpx = (gridDim.x - x + 7) >> 3;          // my XCD label and the number of workgroups that share it
    const int c0 = x ? A.chunk_end[x - 1] : 0, c1 = A.chunk_end[x];
    int t = c0 + (int)(blockIdx.x >> 3), i = 0;
    if (t >= c1) t = -1;
    for (;;) {
        if (threadIdx.x == 0) {
            // the NEXT tile's number travels while this one computes: own chunk first, then the other XCDs' (in ring order)
            int nt = -1;
            for (int d = 0; d < 8 && nt < 0; ++d) {
                const int y = (x + d) & 7, y0 = y ? A.chunk_end[y - 1] : 0, y1 = A.chunk_end[y];
                const int wpy = (gridDim.x - y + 7) >> 3;
                if (y0 + wpy >= y1) continue;                 // that chunk has no ticketed tiles at all
                const int cand = y0 + wpy + (int)atomicAdd(A.ticket + y, 1u);
                if (cand < y1) nt = cand;
            }
            *next = nt;
        }
        if (t >= 0) {
            if (t < (i ? A.tile_end[i - 1] : 0)) i = 0;       // a stolen tile may lie before the current item
            while (i + 1 < A.n && t >= A.tile_end[i]) ++i;
            const int local = t - (i ? A.tile_end[i - 1] : 0);
            const int tiles_n = A.dim[i][4];
            const int tm = local / tiles_n, tn = local - tm * tiles_n;
            DwTile T{(const h16_t*)A.ptr[i][0], (const h16_t*)A.ptr[i][1], (float*)A.ptr[i][2], A.dim[i][0] * 64, A.dim[i][1] * 64, A.dim[i][2] * 64, A.dim[i][3]};
#ifdef DW_TRACE
#ifdef DW_TRACE_LAST
            gemm_dw256_tile(T, tm * 256, tn * 256, A.sumsq, smem, wave, lane, g_dw_trace + blockIdx.x * 16);          // every tile: the LAST one's stamps remain
#else
            gemm_dw256_tile(T, tm * 256, tn * 256, A.sumsq, smem, wave, lane, t == c0 + (int)(blockIdx.x >> 3) ? g_dw_trace + blockIdx.x * 16 : nullptr);
#endif
#else
            gemm_dw256_tile(T, tm * 256, tn * 256, A.sumsq, smem, wave, lane);
#endif
        }
        __syncthreads();                                     // the ring (epilogue scratch) is free again; thread 0's ticket (written long ago) is visible
        t = __builtin_amdgcn_readfirstlane(*next);
        __syncthreads();                                     // everybody has read it before thread 0 draws again
        if (t < 0) break;
    }
}

static bool dw256_eligible(const VqaGemmGroupItem& d) {
    return d.M % 256 == 0 && d.N % 256 == 0 && d.K % 64 == 0 && d.K >= 64 && d.lda % 64 == 0 && d.ldb % 64 == 0 && d.ldc % 64 == 0 &&
           d.lda < (1 << 22) && d.ldb < (1 << 22) && d.ldc < (1 << 22) && d.K < (1 << 22) && d.N / 256 < 65536;
}

// items: eligible ones, n <= DW_MAX_ITEMS
static int launch_dw256(const VqaGemmGroupItem* const* items, int n, float* sumsq, hipStream_t st) {
    static bool attr_set = false;
    auto kern = gemm_dw256_kernel;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, DW_LDS_TOTAL);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    // longest reduction first (stable: equal lengths keep the caller's order, i.e. tiles that share an operand panel stay neighbours)
    const VqaGemmGroupItem* order[DW_MAX_ITEMS];
    for (int i = 0; i < n; ++i) order[i] = items[i];
    std::stable_sort(order, order + n, [](const VqaGemmGroupItem* x, const VqaGemmGroupItem* y) { return x->K > y->K; });
    Dw256Args A{};
    A.n = n; A.sumsq = sumsq;
    double flop = 0.0, bytes = 0.0;
    long tiles = 0;
    for (int i = 0; i < n; ++i) {
        const VqaGemmGroupItem& d = *order[i];
        tiles += (long)(d.M / 256) * (d.N / 256);
        if (tiles > 65535) return VQA_ERR_ARG;
        A.tile_end[i] = (unsigned short)tiles;
        A.ptr[i][0] = d.a; A.ptr[i][1] = d.b; A.ptr[i][2] = d.c_f32;
        A.dim[i][0] = (unsigned short)(d.lda / 64); A.dim[i][1] = (unsigned short)(d.ldb / 64); A.dim[i][2] = (unsigned short)(d.ldc / 64);
        A.dim[i][3] = (unsigned short)(d.K / 64); A.dim[i][4] = (unsigned short)(d.N / 256);
        flop += 2.0 * d.M * d.N * d.K;
        bytes += 2.0 * ((double)d.M + d.N) * d.K + 4.0 * (double)d.M * d.N;
    }
    // eight chunks of equal work (k-tiles), cut at tile granularity
    {
        double work = 0.0;
        for (int i = 0; i < n; ++i) work += (double)(A.tile_end[i] - (i ? A.tile_end[i - 1] : 0)) * A.dim[i][3];
        int i = 0;
        long t = 0;
        double done = 0.0;
        for (int x = 0; x < 8; ++x) {
            const double want = work * (x + 1) / 8.0;
            while (t < tiles && done + 0.5 * A.dim[i][3] <= want) {
                done += A.dim[i][3];
                if (++t >= A.tile_end[i] && i + 1 < n) ++i;
            }
            A.chunk_end[x] = (unsigned short)(x == 7 ? tiles : t);
        }
    }
    static unsigned* tickets = nullptr;                       // resolved once (the first launch is an eager warm-up step, never a stream capture)
    if (!tickets) {
        hipError_t es = hipGetSymbolAddress(reinterpret_cast<void**>(&tickets), HIP_SYMBOL(g_dw_tickets));
        if (es != hipSuccess) { tickets = nullptr; return (int)es; }
    }
    A.ticket = tickets + 16 * (g_dw_slot++ & 31);
    hipError_t e = hipMemsetAsync(A.ticket, 0, 64, st);
    if (e != hipSuccess) return (int)e;
    const int grid = tiles < 256 ? (int)((tiles + 7) / 8 * 8) : 256;      // one workgroup per CU; a multiple of 8 (an XCD's workgroups: every 8th)
    vqa_launch(kern, dim3(grid), dim3(512), DW_LDS_TOTAL, st, A, flop, bytes);
    return (int)hipGetLastError();
}

#ifdef DW_TRACE
extern "C" int vqa_dw_trace_read(unsigned long long* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_dw_trace), sizeof(unsigned long long) * 256 * 16); }
#endif
