#include <hip/hip_runtime.h>
#include <cstdio>
#include "../vqa_model_builder_amd/csrc/common.h"
__global__ void k(float* o) {
    const float v = (float)(threadIdx.x * threadIdx.x + 1);
    o[threadIdx.x] = xor16_sum(v);
    o[64 + threadIdx.x] = xor32_sum(v);
    o[128 + threadIdx.x] = row_sum16(v);
    o[192 + threadIdx.x] = wave_sum(v);
    o[256 + threadIdx.x] = v + dpp_f32<0x128>(v);
}
int main() {
    float* d; hipMalloc(&d, 320 * 4); k<<<1, 64>>>(d); float h[320]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    auto f = [](int l) { return (float)(l * l + 1); };
    int bad[5] = {0, 0, 0, 0, 0};
    float tot = 0; for (int l = 0; l < 64; ++l) tot += f(l);
    for (int l = 0; l < 64; ++l) {
        float rs = 0; for (int j = 0; j < 16; ++j) rs += f((l & ~15) + j);
        bad[0] += h[l] != f(l) + f(l ^ 16); bad[1] += h[64 + l] != f(l) + f(l ^ 32); bad[2] += h[128 + l] != rs; bad[3] += h[192 + l] != tot;
        bad[4] += h[256 + l] != f(l) + f(l ^ 8);
    }
    printf("mismatches xor16 %d xor32 %d row16 %d wave %d ror8 %d   (lane 5: xor16 %g want %g)\n", bad[0], bad[1], bad[2], bad[3], bad[4], h[5], f(5) + f(21));
    return 0;
}
