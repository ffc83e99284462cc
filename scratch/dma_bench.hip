// How fast does one workgroup pull L2-resident bytes into LDS with global_load_lds_dwordx4?  (The GEMM k loop's bound.)
// Each wave issues N DMA instructions (1 KiB each), then s_waitcnt vmcnt(0); repeated; s_memtime around the loop.
//   pattern 0: each instruction reads 1 KiB contiguous          pattern 1: 8 rows x 128 B, rows `stride` bytes apart
//   pattern 2: as 1 with the 16-B chunks of a row permuted (the GEMM's bank swizzle)
//   mode R   : register-staged instead (global_load_dwordx4 + ds_write_b128)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;
template <int N, int PATTERN, bool REG>
__global__ __launch_bounds__(256) void k(const char* src, size_t region, int stride, int iters, unsigned long long* out) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    // every workgroup walks its own 512-KiB window of the (L2-resident) region
    const char* base = src + ((size_t)blockIdx.x * (512 << 10)) % region;
    size_t lane_off;
    if (PATTERN == 0) lane_off = (size_t)lane * 16;
    else { const int row = lane >> 3, c = lane & 7; lane_off = (size_t)row * stride + ((PATTERN == 2 ? c ^ (row & 7) : c) << 4); }
    unsigned long long t0 = 0, t1 = 0;
    uint4 acc = {0, 0, 0, 0};
    for (int it = -2; it < iters; ++it) {
        if (it == 0) t0 = __builtin_readcyclecounter();
        const char* p = base + (size_t)((it & 15) * nw + wave) * (PATTERN == 0 ? N * 1024 : 128) % (256 << 10);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const char* a = p + lane_off + (PATTERN == 0 ? (size_t)i * 1024 : (size_t)i * 8 * stride);
            if (REG) {
                const uint4 v = *reinterpret_cast<const uint4*>(a);
                *reinterpret_cast<uint4*>(lds + (wave * N + i) * 1024 + lane * 16) = v;
            } else {
                unsigned long long addr = (unsigned long long)a;
                asm volatile("" : "+v"(addr));
                __builtin_amdgcn_global_load_lds((gptr_t*)addr, (lptr_t*)(lds + (wave * N + i) * 1024), 16, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (lds[threadIdx.x] == 77 && acc.x) out[0] = 1;
}
template <int N, int P, bool REG>
void run(const char* src, size_t region, int stride, int waves, int wgs, unsigned long long* out, const char* tag) {
    const int iters = 64;
    hipLaunchKernelGGL((k<N, P, REG>), dim3(wgs), dim3(waves * 64), waves * N * 1024, 0, src, region, stride, iters, out);
    hipLaunchKernelGGL((k<N, P, REG>), dim3(wgs), dim3(waves * 64), waves * N * 1024, 0, src, region, stride, iters, out);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(wgs);
    hipMemcpy(h.data(), out, wgs * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += v;
    const double cyc = s / wgs / iters;
    printf("%-4s N=%2d pattern %d stride %5d waves %d wgs %4d: %7.0f cycles/iter  %6.2f B/clk/WG  (%5.1f cycles per instr per wave)\n", tag, N, P, stride,
           waves, wgs, cyc, (double)waves * N * 1024 / cyc, cyc / N);
}
int main() {
    const size_t region = 24 << 20;
    char* src; hipMalloc(&src, region + (4 << 20)); hipMemset(src, 1, region + (4 << 20));
    unsigned long long* out; hipMalloc(&out, 4096 * 8);
    for (int wgs : {256, 512, 1024}) {
        run<1, 0, false>(src, region, 0, 4, wgs, out, "dma"); run<4, 0, false>(src, region, 0, 4, wgs, out, "dma");
        run<8, 0, false>(src, region, 0, 4, wgs, out, "dma"); run<16, 0, false>(src, region, 0, 4, wgs, out, "dma");
        run<4, 1, false>(src, region, 1536, 4, wgs, out, "dma"); run<8, 1, false>(src, region, 1536, 4, wgs, out, "dma");
        run<4, 2, false>(src, region, 1536, 4, wgs, out, "dma"); run<4, 1, false>(src, region, 6144, 4, wgs, out, "dma");
        run<4, 1, false>(src, region, 4096, 4, wgs, out, "dma");
        run<4, 0, true>(src, region, 0, 4, wgs, out, "reg"); run<8, 0, true>(src, region, 0, 4, wgs, out, "reg");
        run<4, 1, true>(src, region, 1536, 4, wgs, out, "reg");
    }
    run<4, 0, false>(src, region, 0, 8, 256, out, "dma"); run<8, 0, false>(src, region, 0, 8, 256, out, "dma");
    run<4, 1, false>(src, region, 1536, 8, 256, out, "dma");
    return 0;
}
