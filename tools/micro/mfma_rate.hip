// Sustained MFMA rate on register-resident random operands: v_mfma_f32_16x16x32_bf16 vs v_mfma_f32_32x32x16_bf16.
// (Is the 1x1-conv GEMM's ~1.15 PFLOP/s on real data a clock/power limit, and does the wider tile draw less?)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_rate tools/micro/mfma_rate.hip && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ void __launch_bounds__(512) k(const bf16x8* in, float* out, int iters) {
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = in[threadIdx.x * 8 + i]; b[i] = in[threadIdx.x * 8 + 4 + i]; }
    if (MODE == 0) {
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[i >> 2], acc[i], 0, 0, 0);
        float s = 0;
        for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    } else {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i & 1) * 2 + r], b[(i >> 1) * 2 + r], acc[i], 0, 0, 0);
        float s = 0;
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    }
}

int main() {
    const int blocks = 256 * 2, iters = 20000;
    bf16x8* in; float* out;
    hipMalloc(&in, 512 * 8 * sizeof(bf16x8)); hipMalloc(&out, blocks * 512 * 4);
    unsigned short* h = (unsigned short*)malloc(512 * 8 * 16);
    for (int zero = 0; zero < 2; ++zero) {
        srand(1);
        for (int i = 0; i < 512 * 8 * 8; ++i) h[i] = zero ? 0 : (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));   // ~ +-[0.008, 0.03)
        hipMemcpy(in, h, 512 * 8 * 16, hipMemcpyHostToDevice);
        for (int mode = 0; mode < 2; ++mode) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(512), 0, 0, in, out, iters);
                else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(512), 0, 0, in, out, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // per wave per iteration: mode 0: 16 MFMA x 16*16*32*2 flop; mode 1: 8 MFMA x 32*32*16*2 flop -> both 262144 flop
            const double fl = (double)blocks * 8 * iters * 262144.0;
            printf("%s operands, %s: %.2f ms  %.0f TFLOP/s\n", zero ? "zero  " : "random", mode ? "32x32x16" : "16x16x32", ms, fl / ms / 1e9);
        }
    }
    return 0;
}
