// Lab build of ONE gemm_ws instance with the in-kernel timeline compiled in (never part of libvqa_hip.so).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -I include -I vqa_model_builder_amd/csrc \
//         -DLAB_BM=64 -DLAB_BN=288 -DLAB_ST=3 scratch/gemm_lab_ws.hip -o scratch/labws_64x288.so
#define VQA_GEMM_TRACE 1
#define VQA_GEMM_LAB 1
#include "../vqa_model_builder_amd/csrc/gemm.hip"

extern "C" int lab_gemm(const void* a, const void* b, void* c_bf16, float* c_f32, int M, int N, int K, int lda, int ldb, int a_kc, int b_kc,
                        int unused, unsigned long long* trace, void* stream) {
    GemmArgs p{};
    p.a = (const h16_t*)a; p.b = (const h16_t*)b; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb;
    p.c_bf16 = (h16_t*)c_bf16; p.ldc_bf16 = N; p.c_f32 = c_f32; p.ldc_f32 = N;
    p.alpha = 1.f; p.drop_inv_keep = 1.f; p.trace = trace;
    p.k_per_split = (K + 63) / 64 * 64;
    p.tiles_n = (N + LAB_BN - 1) / LAB_BN; p.tiles_n_magic = div_magic(p.tiles_n);
    constexpr int LDS = LAB_ST * (LAB_BM + LAB_BN) * 128;
    const int tiles = ((M + LAB_BM - 1) / LAB_BM) * p.tiles_n;
    if (b_kc) {
        auto kern = gemm_ws_kernel<LAB_BM, LAB_BN, LAB_ST, true, true>;
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        hipLaunchKernelGGL(kern, dim3(tiles), dim3(512), LDS, (hipStream_t)stream, p);
    } else {
        auto kern = gemm_ws_kernel<LAB_BM, LAB_BN, LAB_ST, true, false>;
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        hipLaunchKernelGGL(kern, dim3(tiles), dim3(512), LDS, (hipStream_t)stream, p);
    }
    return (int)hipGetLastError();
}
