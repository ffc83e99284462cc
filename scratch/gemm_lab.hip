// Lab build of ONE LDS-DMA GEMM instance with the in-kernel s_memtime trace compiled in (never part of libvqa_hip.so).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -I include -I vqa_model_builder_amd/csrc \
//         -DLAB_BM=128 -DLAB_BN=128 -DLAB_WM=2 -DLAB_WN=2 -DLAB_BK=64 -DLAB_ST=2 scratch/gemm_lab.hip -o scratch/lab_x.so
#define VQA_GEMM_TRACE 1
#define VQA_GEMM_LAB 1
#include "../vqa_model_builder_amd/csrc/gemm.hip"

extern "C" int lab_gemm(const void* a, const void* b, void* c_bf16, float* c_f32, int M, int N, int K, int lda, int ldb, int a_kc, int b_kc,
                        int group_m, unsigned long long* trace, void* stream) {
    GemmArgs p{};
    p.a = (const h16_t*)a; p.b = (const h16_t*)b; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb;
    p.c_bf16 = (h16_t*)c_bf16; p.ldc_bf16 = N; p.c_f32 = c_f32; p.ldc_f32 = N;
    p.alpha = 1.f; p.drop_inv_keep = 1.f; p.trace = trace; (void)group_m;       // (tile order is fixed since round 2: group_m kept in the signature for lab_run.py)
    p.tiles_n = (N + LAB_BN - 1) / LAB_BN; p.tiles_n_magic = div_magic(p.tiles_n);
    p.k_per_split = (K + LAB_BK - 1) / LAB_BK * LAB_BK;
    return launch_v1s<LAB_BM, LAB_BN, LAB_WM, LAB_WN, LAB_BK, LAB_ST>(p, a_kc, b_kc, 1, (hipStream_t)stream);
}

extern "C" void lab_set_fast(int on) { g_v1_fast = on != 0; }
