// Can the units of a CU work at the same time when different WAVES of one workgroup use them?  (in the fused
// depthwise + pointwise kernels the phases of a K-step add up.)  One 512-lane workgroup per CU, no barriers, register-resident
// operands; per iteration a wave runs a block of 32 MFMAs (16x16x32 f16), a block of 256 v_dot2c (8 independent chains) and / or a
// block of 24 ds_read_b128, in the order its role says:
//   mode 0  all waves: MFMA block only              mode 1  all waves: VALU block only            mode 2  all waves: LDS block only
//   mode 3  all waves: MFMA block, then VALU block (lock-step)
//   mode 4  waves 0-3: MFMA then VALU; waves 4-7: VALU then MFMA (de-phased: same work as mode 3)
//   mode 5  waves 0-3: two MFMA blocks; waves 4-7: two VALU blocks (specialised: same work as mode 3)
//   mode 6  all waves: MFMA, VALU, LDS blocks       mode 7  the same, each wave starting at block (wave mod 3)
//   mode 8  odd / even waves instead of 0-3 / 4-7 for the de-phasing of mode 4
//   hipcc --offload-arch=gfx950 -O3 [-DVK=2] -o /tmp/overlap tools/micro/overlap.hip && /tmp/overlap      (profiles/r05/overlap_micro.log)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));

struct State {
    f16x8 a[4], b[4];
    f32x4 acc[8];
    float o[8];
    f16x2 x[8], w[8];
    v4i l[4];
};

__device__ __forceinline__ void mfma_block(State& s) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) s.acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(s.a[i & 3], s.b[(i + r) & 3], s.acc[i], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
}
// VK: the vector instruction of the VALU block -- 0 v_dot2c_f32_f16, 2 v_fma_mix_f32 (f16 operands, f32 sum)
#ifndef VK
#define VK 0
#endif
__device__ __forceinline__ void valu_block(State& s) {
#pragma unroll
    for (int r = 0; r < 32; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (VK == 0) s.o[i] = __builtin_amdgcn_fdot2(s.x[(i + r) & 7], s.w[i], s.o[i], false);
            if (VK == 2) s.o[i] = __builtin_fmaf((float)s.x[(i + r) & 7][0], (float)s.w[i][1], s.o[i]);
        }
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void lds_block(State& s, const char* lds, int off) {
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        v4i t[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) t[i] = *reinterpret_cast<const v4i*>(lds + ((off + (r * 4 + i) * 1024) & 0xffff));
#pragma unroll
        for (int i = 0; i < 4; ++i) s.l[i] ^= t[i];
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <int MODE>
__global__ void __launch_bounds__(512) k(const f16x8* in, float* out, int iters, long long* clk) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    State s;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = 0; i < 4; ++i) { s.a[i] = in[tid * 8 + i]; s.b[i] = in[tid * 8 + 4 + i]; }
    for (int i = 0; i < 8; ++i) {
        s.acc[i] = f32x4{0, 0, 0, 0};
        s.o[i] = 0.f;
        s.x[i] = f16x2{s.a[i & 3][i], s.a[i & 3][(i + 1) & 7]};
        s.w[i] = f16x2{s.b[i & 3][i], s.b[i & 3][(i + 3) & 7]};
    }
    for (int i = 0; i < 4; ++i) s.l[i] = v4i{0, 0, 0, 0};
    for (int i = tid; i < 65536 / 16; i += 512) reinterpret_cast<v4i*>(lds)[i] = v4i{i, i + 1, i + 2, i + 3};
    __syncthreads();
    const int off = (tid & 63) * 16 + wave * 4096;          // conflict-free: a wave reads 1 KB contiguous
    const bool second = MODE == 8 ? (wave & 1) : wave >= 4;
    const long long c0 = clock64(), w0 = wall_clock64();          // shader-clock counter / constant 100 MHz counter
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) mfma_block(s);
        if (MODE == 1) valu_block(s);
        if (MODE == 2) lds_block(s, lds, off);
        if (MODE == 3) { mfma_block(s); valu_block(s); }
        if (MODE == 4 || MODE == 8) {
            if (!second) { mfma_block(s); valu_block(s); } else { valu_block(s); mfma_block(s); }
        }
        if (MODE == 5) {
            if (!second) { mfma_block(s); mfma_block(s); } else { valu_block(s); valu_block(s); }
        }
        if (MODE == 6) { mfma_block(s); valu_block(s); lds_block(s, lds, off); }
        if (MODE == 7) {
            const int ph = wave % 3;
            if (ph == 0) { mfma_block(s); valu_block(s); lds_block(s, lds, off); }
            else if (ph == 1) { valu_block(s); lds_block(s, lds, off); mfma_block(s); }
            else { lds_block(s, lds, off); mfma_block(s); valu_block(s); }
        }
    }
    if (blockIdx.x == 7 && tid == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
    float r = 0;
    for (int i = 0; i < 8; ++i) r += s.acc[i][0] + s.acc[i][3] + s.o[i];
    for (int i = 0; i < 4; ++i) r += (float)(s.l[i][0] ^ s.l[i][3]);
    out[blockIdx.x * 512 + tid] = r;
}

template <int MODE>
static float run(const f16x8* in, float* out, int blocks, int iters, long long* clk, double* mhz) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 100 * 1024, 0, in, out, iters, clk);       // 100 KB: one workgroup per CU
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    long long h[2];
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    *mhz = 100.0 * (double)h[0] / (double)h[1];
    return ms;
}

int main() {
    const int blocks = 256, iters = 4000;
    f16x8* in;
    float* out;
    hipMalloc(&in, 512 * 8 * sizeof(f16x8));
    hipMalloc(&out, blocks * 512 * 4);
    unsigned short* h = (unsigned short*)malloc(512 * 8 * 16);
    srand(1);
    for (int i = 0; i < 512 * 8 * 8; ++i) h[i] = (unsigned short)(0x2c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));
    hipMemcpy(in, h, 512 * 8 * 16, hipMemcpyHostToDevice);
    printf("VALU kind %d\n", VK);
    const char* names[9] = {"MFMA only", "VALU only", "LDS only", "MFMA, VALU lock-step", "de-phased (waves 0-3 / 4-7)", "specialised (MFMA waves / VALU waves)",
                            "MFMA, VALU, LDS lock-step", "three-way de-phased", "de-phased (even / odd waves)"};
    float ms[9];
    double mhz[9];
    long long* clk;
    hipMalloc(&clk, 16);
    ms[0] = run<0>(in, out, blocks, iters, clk, mhz + 0); ms[1] = run<1>(in, out, blocks, iters, clk, mhz + 1); ms[2] = run<2>(in, out, blocks, iters, clk, mhz + 2);
    ms[3] = run<3>(in, out, blocks, iters, clk, mhz + 3); ms[4] = run<4>(in, out, blocks, iters, clk, mhz + 4); ms[5] = run<5>(in, out, blocks, iters, clk, mhz + 5);
    ms[6] = run<6>(in, out, blocks, iters, clk, mhz + 6); ms[7] = run<7>(in, out, blocks, iters, clk, mhz + 7); ms[8] = run<8>(in, out, blocks, iters, clk, mhz + 8);
    for (int m = 0; m < 9; ++m)      // cycles per iteration at 2.4 GHz (a block: 32 MFMAs = 512 matrix-pipe cycles per wave; 256 dot2c; 24 b128 reads)
        printf("mode %d  %-40s %8.3f ms  %7.0f cycles per iteration at 2.4 GHz; clock64 / wall_clock64: %6.0f MHz\n", m, names[m], ms[m], ms[m] * 1e-3 * 2.4e9 / iters, mhz[m]);
    return 0;
}
