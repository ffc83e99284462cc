// Fused in-projection + attention forward: ONE workgroup per (sample, head) computes that head's Q, K, V from the block's
// input rows (MFMA, LDS-DMA ring, loader / consumer wave roles -- the gemm_ws machinery) and runs the attention core on them
// while they are still in LDS.  Replaces the pair  {packed in_proj GEMM  ->  attention kernel}  of nn.MultiheadAttention
// (reference vqa_model.py:300,304 for the fusion block; the encoders' self-attention has the same shape): one launch instead of
// two, and the [B*S, 3D] projection tensor is only WRITTEN (for backward), never read back in forward.
//
// Included by gemm.hip inside its anonymous namespace (shares the ring, the swizzles, the fragment reads and vqa_launch).
//
// Work of workgroup (b, h), 512 threads:
//   phase 0:  Q_h  [64 x DH]   = Xq_b  [64 x D] * Wq_h^T          (rows h*DH..    of the packed in_proj weight)
//   phase 1:  K_h | V_h [64 x 2DH] = Xkv_b [64 x D] * [Wk_h ; Wv_h]^T   (rows D + h*DH.., 2D + h*DH..)
//   both phases walk k in 64-wide steps through ONE ring of 3 stages {A tile 64 x 64, B tile 2DH x 64}; the unified sequence of
//   2 * D/64 k-tiles keeps the ring full across the phase change (a phase-0 tile fills only the first DH rows of the B slot and
//   is PB/2 fewer DMA instructions: the loaders' counted vmcnt wait allows exactly the instructions of the NEXT tile in flight).
//   waves 4-7: loaders (DMA issue + wait + barrier);  waves 0-3: consumers 2 x 2 over the tile (phase 1: wave column 0 = K, 1 = V).
//   epilogue: + bias, -> bf16, written to the LDS tiles the attention core reads (pitch DH*2+16) and, optionally, to HBM for
//   backward; then wave w runs attn_core_fwd on query rows 16w..16w+15.
// Rows beyond Sq / Skv are clamped duplicates of the last valid row (finite; the core masks keys >= Skv and never stores
// queries >= Sq).

struct FusedArgs {
    const h16_t *xq, *xkv, *w;
    const float* bias;
    h16_t *q, *k, *v;
    int ldxq, ldxkv, ldw, ldq, ldk, ldv, D;
    MArgs a;
};

// k-contiguous operand rows through a row map: tile row r reads global row map(r), or the zero page when map(r) < 0
template <int ROWS, int NW, class Map>
__device__ __forceinline__ void dma_init_map(DmaLane (&d)[ROWS * 8 / 64 / NW], const h16_t* __restrict__ g, int ld, int kend, int wave, int lane, Map map) {
    constexpr int PER_WAVE = ROWS * 8 / 64 / NW;
    const unsigned long long zero = reinterpret_cast<unsigned long long>(g_zero_page);
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
        const int pos = (wave * PER_WAVE + i) * 64 + lane;
        const int r = pos >> 3, c = (pos & 7) ^ kcT_key<64>(r);
        const int gr = map(r);
        if (gr >= 0) {
            d[i].ptr = reinterpret_cast<unsigned long long>(g + (size_t)gr * ld + c * 8);
            d[i].step = 128;
            d[i].kmax = kend - c * 8;
        } else { d[i].ptr = zero; d[i].step = 0; d[i].kmax = 0x7fffffff; }
    }
}

template <int TN, int TOTAL>
__device__ __forceinline__ void fused_phase(const char* smem, int stage_bytes, int& stage, int nk, int (&ao0)[2], int (&ao1)[2], int b_row0, int lane,
                                            f32x4 (&acc)[2][TN]) {
    int bo0[TN], bo1[TN];
    frag_offsets<TOTAL, true, 64, TN>(bo0, bo1, b_row0, lane);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < nk; ++t) {
        __builtin_amdgcn_s_barrier();
        const char* la = smem + stage * stage_bytes;
        const char* lb = la + 64 * 128;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            h16x8 fa[2], fb[TN];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = load_frag1<64, true>(la, ao0[i], ao1[i], ks);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = load_frag1<TOTAL, true>(lb, bo0[j], bo1[j], ks);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = VQA_MFMA16(fb[j], fa[i], acc[i][j]);
        }
        stage = stage == 2 ? 0 : stage + 1;
    }
}

// + bias, -> bf16, into the LDS tile (pitch DH*2+16) and optionally HBM.  acc[i][j][r] is row 16 i + (lane & 15), column
// 16 j + 4 (lane >> 4) + r of the wave's tile (rows from r_base, columns from c_base of the head's DH columns).
template <int DH, int TN>
__device__ __forceinline__ void fused_store(f32x4 (&acc)[2][TN], char* tile, int r_base, int c_base, const float* bias, h16_t* gdst, int ld, int rows,
                                            size_t grow0, int gcol0, int lane) {
    constexpr int PITCH = DH * 2 + 16;
    const int g = lane >> 4, li = lane & 15;
    // every bias vector first: loaded inside the store loop, each load's wait (vmcnt counts stores too) also sat out the previous columns'
    // global stores -- TN store round trips in a row
    f32x4 bbs[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        bbs[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (bias) bbs[j] = *reinterpret_cast<const f32x4*>(bias + gcol0 + c_base + 16 * j + 4 * g);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int c = c_base + 16 * j + 4 * g;
        const f32x4 bb = bbs[j];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = r_base + 16 * i + li;
            const f32x4 v = acc[i][j] + bb;
            h16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (h16_t)v[r];
            *reinterpret_cast<h16x4*>(tile + row * PITCH + c * 2) = o;
            if (gdst && row < rows) *reinterpret_cast<h16x4*>(gdst + (grow0 + row) * ld + gcol0 + c) = o;
        }
    }
}

template <int DH>
__global__ __launch_bounds__(512) void fused_inproj_attn_kernel(const FusedArgs p) {
    constexpr int BN = 2 * DH, STAGES = 3;
    constexpr int STAGE_BYTES = (64 + BN) * 128;
    constexpr int PA = 2, PB = BN / 32, GL = PA + PB;
    constexpr int PITCH = DH * 2 + 16;
    static_assert(3 * 64 * PITCH <= STAGES * STAGE_BYTES, "Q | K | V tiles reuse the ring");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x / p.a.H, h = blockIdx.x % p.a.H;
    const int nk = p.D / 64;                                  // host: D % 64 == 0
    const int Sq = p.a.Sq, Skv = p.a.Skv;

    if (wave >= 4) {
        // ---------------------------------------------------------------- loaders
        const int lw = wave - 4;
        constexpr int PQ = PB / 2, GL0 = PA + PQ;               // phase 0: Q rows only
        DmaLane qa[PA], qb[PQ], ka[PA], kb[PB];
        const int D = p.D;
        dma_init_map<64, 4>(qa, p.xq + (size_t)b * Sq * p.ldxq, p.ldxq, D, lw, lane, [=](int r) { return min(r, Sq - 1); });
        dma_init_map<DH, 4>(qb, p.w, p.ldw, D, lw, lane, [=](int r) { return h * DH + r; });
        dma_init_map<64, 4>(ka, p.xkv + (size_t)b * Skv * p.ldxkv, p.ldxkv, D, lw, lane, [=](int r) { return min(r, Skv - 1); });
        dma_init_map<BN, 4>(kb, p.w, p.ldw, D, lw, lane, [=](int r) { return (r < DH ? D : 2 * D - DH) + h * DH + r; });
        const int nt = 2 * nk;
        auto issue = [&](int t, int stage) {
            char* st = smem + stage * STAGE_BYTES;
            if (t < nk) { dma_issue<PA, false>(qa, st, 0, lw); dma_issue<PQ, false>(qb, st + 64 * 128, 0, lw); }
            else { dma_issue<PA, false>(ka, st, 0, lw); dma_issue<PB, false>(kb, st + 64 * 128, 0, lw); }
        };
        issue(0, 0);
        issue(1, 1);                                            // nt >= 2 always
        int slot = 2;                                           // ring slot of tile t + 2
        for (int t = 0; t < nt; ++t) {
            if (t + 1 >= nt) wait_vmcnt<0>();                   // tile t has landed when at most tile t + 1's instructions are in flight
            else if (t + 1 < nk) wait_vmcnt<GL0>();
            else wait_vmcnt<GL>();
            __builtin_amdgcn_s_barrier();                       // tile t is in LDS for everybody; the slot of tile t - 1 is free again
            if (t + 2 < nt) issue(t + 2, slot);
            slot = slot == 2 ? 0 : slot + 1;
        }
        __builtin_amdgcn_s_barrier();                           // ring -> Q | K | V tiles
        __syncthreads();                                        // tiles written
        return;
    }
    // -------------------------------------------------------------------- consumers
    const int wm = wave >> 1, wn = wave & 1;
    int ao0[2], ao1[2];
    frag_offsets<64, true, 64, 2>(ao0, ao1, wm * 32, lane);
    constexpr int TNQ = DH / 32, TNK = DH / 16;
    f32x4 accq[2][TNQ], acck[2][TNK];
    int stage = 0;
    fused_phase<TNQ, BN>(smem, STAGE_BYTES, stage, nk, ao0, ao1, wn * (DH / 2), lane, accq);
    fused_phase<TNK, BN>(smem, STAGE_BYTES, stage, nk, ao0, ao1, wn * DH, lane, acck);
    __builtin_amdgcn_s_barrier();                               // every consumer is done with the ring
    char *Qs = smem, *Ks = smem + 64 * PITCH, *Vs = smem + 2 * 64 * PITCH;
    const int D = p.D;
    fused_store<DH, TNQ>(accq, Qs, wm * 32, wn * (DH / 2), p.bias, p.q, p.ldq, Sq, (size_t)b * Sq, h * DH, lane);
    if (wn == 0) fused_store<DH, TNK>(acck, Ks, wm * 32, 0, p.bias ? p.bias + D : nullptr, p.k, p.ldk, Skv, (size_t)b * Skv, h * DH, lane);
    else fused_store<DH, TNK>(acck, Vs, wm * 32, 0, p.bias ? p.bias + 2 * D : nullptr, p.v, p.ldv, Skv, (size_t)b * Skv, h * DH, lane);
    __syncthreads();
    if (16 * wave >= Sq) return;
    MArgs a = p.a;
    if (a.drop_p > 0.f) a.seed = resolve_seed(a.seed);
    attn_core_fwd<DH>(a, Qs, Ks, Vs, b, h, wave, lane);
}

template <int DH>
int launch_fused_attn(const FusedArgs& p, hipStream_t st) {
    constexpr int LDS = 3 * (64 + 2 * DH) * 128;
    static bool attr_set = false;
    auto kern = fused_inproj_attn_kernel<DH>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const double flop = 2.0 * p.a.B * ((double)p.a.Sq * p.D * p.D + 2.0 * p.a.Skv * p.D * p.D) + 4.0 * p.a.B * p.a.H * (double)p.a.Sq * p.a.Skv * DH;
    const double rows_q = (double)p.a.B * p.a.Sq, rows_kv = (double)p.a.B * p.a.Skv;
    const double bytes = 2.0 * (rows_q + (p.xkv != p.xq ? rows_kv : 0.0)) * p.D + 2.0 * 3.0 * p.D * p.D + (p.bias ? 12.0 * p.D : 0.0) +
                         2.0 * p.D * ((p.q ? rows_q : 0.0) + (p.k ? rows_kv : 0.0) + (p.v ? rows_kv : 0.0)) + 2.0 * rows_q * p.D;
    vqa_launch(kern, dim3(p.a.B * p.a.H), dim3(512), LDS, st, p, flop, bytes);
    return (int)hipGetLastError();
}
