// How fast can a GEMM epilogue's C tile leave the chip?  Each workgroup writes a BM x BN tile of a [M][N] matrix.
//   mode 0: 8 B per lane (bf16x4), 16 lanes per 128-B row segment       (the epilogue's bf16 stream)
//   mode 1: 16 B per lane, 8 lanes per 128-B row segment
//   mode 2: 16 B per lane, 16 lanes per 256-B row segment (fp32x4 stream of a 64-column wave tile)
//   mode 3: 16 B per lane, whole 64-lane wave on one 1-KiB row segment
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int MODE>
__global__ __launch_bounds__(256) void k(char* c, int N_bytes, int BMr, int BN_bytes, int tiles_n) {
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    char* base = c + (size_t)tm * BMr * N_bytes + (size_t)tn * BN_bytes;
    constexpr int LB = MODE == 0 ? 8 : 16;                      // bytes per lane
    constexpr int LPR = MODE == 0 ? 16 : MODE == 1 ? 8 : MODE == 2 ? 16 : 64;
    constexpr int SEG = LB * LPR, RPI = 64 / LPR;
    const int segs = BN_bytes / SEG;
    for (int r0 = wave * RPI; r0 < BMr; r0 += nw * RPI) {
        const int row = r0 + lane / LPR;
        for (int s = 0; s < segs; ++s) {
            char* p = base + (size_t)row * N_bytes + s * SEG + (lane % LPR) * LB;
            if (LB == 8) *reinterpret_cast<uint2*>(p) = make_uint2(lane, s);
            else *reinterpret_cast<uint4*>(p) = make_uint4(lane, s, row, 1);
        }
    }
}
int main(int argc, char** argv) {
    const int M = 2048, Nb = argc > 1 ? atoi(argv[1]) : 3072 * 2;     // row bytes
    char* c; hipMalloc(&c, (size_t)M * Nb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct { int bm, bnb, thr; } cfgs[] = {{256, 256, 512}, {128, 128, 256}, {64, 128, 256}, {128, 256, 256}, {256, 1024, 512}};
    for (auto cf : cfgs) {
        if (Nb % cf.bnb) continue;
        const int tiles_n = Nb / cf.bnb, tiles = (M / cf.bm) * tiles_n;
        for (int mode = 0; mode < 4; ++mode) {
            if ((mode == 2 && cf.bnb % 256) || (mode == 3 && cf.bnb % 1024)) continue;
            auto launch = [&]() {
                switch (mode) {
                    case 0: hipLaunchKernelGGL(k<0>, dim3(tiles), dim3(cf.thr), 0, 0, c, Nb, cf.bm, cf.bnb, tiles_n); break;
                    case 1: hipLaunchKernelGGL(k<1>, dim3(tiles), dim3(cf.thr), 0, 0, c, Nb, cf.bm, cf.bnb, tiles_n); break;
                    case 2: hipLaunchKernelGGL(k<2>, dim3(tiles), dim3(cf.thr), 0, 0, c, Nb, cf.bm, cf.bnb, tiles_n); break;
                    default: hipLaunchKernelGGL(k<3>, dim3(tiles), dim3(cf.thr), 0, 0, c, Nb, cf.bm, cf.bnb, tiles_n); break;
                }
            };
            for (int i = 0; i < 5; ++i) launch();
            hipEventRecord(e0);
            for (int i = 0; i < 50; ++i) launch();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double us = ms / 50 * 1e3, mb = (double)M * Nb / 1e6;
            printf("rowbytes %5d tile %3d rows x %4d B (%4d wgs x %d thr) mode %d: %6.1f us  %5.2f TB/s\n", Nb, cf.bm, cf.bnb, tiles, cf.thr, mode, us, mb / us / 1e6 * 1e0);
        }
    }
    return 0;
}
