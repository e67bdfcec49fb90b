// What does one CU's LDS-DMA path deliver from L2, and does the SOURCE ADDRESS PATTERN of a 1 KB instruction matter?
//   hipcc --offload-arch=gfx950 -O2 -o bin/dma_rate dma_rate.hip && ./bin/dma_rate
// One 512-thread workgroup per CU (256 of them), every wave sends global_load_lds_dwordx4 instructions back to back into a 64 KB LDS
// ring (no MFMA, no LDS reads), draining with vmcnt(0) every 8 instructions; the 2 MB source region is shared by all workgroups (L2
// hits).  Patterns for the 64 lanes x 16 B of one instruction:
//   0  8 rows x 128 B, row pitch 4096 B (what a GEMM tile's rows look like: K = 2048 f16)      8 cache lines in 8 pages
//   1  8 rows x 128 B, row pitch 2048 B                                                          8 lines in 4 pages
//   2  1 KB contiguous (a pre-tiled operand)                                                     8 lines in 1 page
//   3  8 rows x 128 B, row pitch 4096 B, chunks XOR-swizzled inside the row (the real pattern)
// Prints cycles per instruction per CU (s_memtime) and GB/s per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ void glds16(const void* sbase, unsigned voff, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte_addr) : "memory");
}

typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void blds16(v4i rsrc, unsigned voff, unsigned soff, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_byte_addr) : "memory");
}

// PAT 4: pattern 0 through a buffer descriptor (buffer_load_dwordx4 ... offen lds); nwaves: only waves < nwaves send
template <int PAT>
__global__ void __launch_bounds__(512) k(const char* src, unsigned long long* out, int iters, int nwaves) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const unsigned lds0 = (unsigned)(size_t)(const __attribute__((address_space(3))) void*)lds;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned voff;
    if (PAT == 0) voff = (lane >> 3) * 4096 + (lane & 7) * 16;
    else if (PAT == 1) voff = (lane >> 3) * 2048 + (lane & 7) * 16;
    else if (PAT == 2) voff = lane * 16;
    else if (PAT == 3) voff = (lane >> 3) * 4096 + (((lane & 7) ^ (lane >> 3)) * 16);
    else voff = (lane >> 3) * 4096 + (lane & 7) * 16;
    // buffer descriptor by hand: base (48 bits), stride 0, num_records, flags 0x00020000 (raw buffer, gfx9 data format)
    const unsigned long long ba = (unsigned long long)src;
    const v4i rsrc = {(int)__builtin_amdgcn_readfirstlane((unsigned)ba), (int)__builtin_amdgcn_readfirstlane((unsigned)(ba >> 32) & 0xffffu), 8 << 20, 0x00020000};
    __syncthreads();
    if (wave >= nwaves) { if (lane == 0) out[blockIdx.x * 8 + wave] = 0; return; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            // instruction (it, i) of this wave: a different 1 KB piece every time, all inside a 2 MB window
            const unsigned piece = (unsigned)((it * 8 + i) * 8 + wave) & 63u;
            const char* sb = PAT == 2 ? src + piece * 1024 + (it & 31) * 65536
                                      : src + (piece & 7) * 128 + (piece >> 3) * (PAT == 1 ? 16384 : 32768) + (it & 7) * 262144;
            if (PAT == 4) blds16(rsrc, voff, (unsigned)(sb - src), lds0 + (unsigned)(((i * 8 + wave) & 63) * 1024));
            else glds16(sb, voff, lds0 + (unsigned)(((i * 8 + wave) & 63) * 1024));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

int main() {
    char* src; unsigned long long* out;
    (void)hipMalloc(&src, 8 << 20); (void)hipMemset(src, 1, 8 << 20);
    (void)hipHostMalloc(&out, 256 * 8 * 8, 0);
    const int iters = 2000;
    const int cfgs[][2] = {{0, 8}, {0, 4}, {0, 2}, {0, 1}, {2, 8}, {2, 4}, {3, 8}, {4, 8}, {4, 4}, {4, 1}};
    for (auto& c : cfgs) {
        const int pat = c[0], nw = c[1];
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0, 0);
#define L(P) { (void)hipFuncSetAttribute((const void*)k<P>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536); hipLaunchKernelGGL(k<P>, dim3(256), dim3(512), 65536, 0, src, out, iters, nw); }
            if (pat == 0) L(0) else if (pat == 1) L(1) else if (pat == 2) L(2) else if (pat == 3) L(3) else L(4)
            (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
            float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep == 0) continue;
            double cyc = 0; for (int i = 0; i < 256 * 8; ++i) cyc += (double)out[i];
            cyc /= 256 * nw;
            const double instr_per_cu = (double)iters * 8 * nw;
            printf("pattern %d, %d waves sending: %.3f ms, %.1f cycles per DMA instruction per CU, %.0f per instruction and WAVE, %.1f GB/s per CU\n", pat, nw, ms,
                   cyc / instr_per_cu, cyc / ((double)iters * 8), instr_per_cu * 1024 / (ms * 1e6));
        }
    }
    return 0;
}
