// Do vector-memory loads of one wave return in issue order on gfx950?  Each lane issues load A (cold: a 2 GB buffer walked
// once -> HBM miss) and then load B (hot: one cache line everybody reads), waits with `s_waitcnt vmcnt(1)` -- which is
// enough for A if and only if loads complete in order -- and checks A's destination register against a poison value.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/load_order tools/micro/load_order.hip && /tmp/load_order
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: global_load A, global_load B;  1: buffer_load A, buffer_load B;  2: buffer_load A, global_load B
__global__ void k(const int* cold, const int* hot, unsigned long long n_cold, unsigned* bad, int round) {
    const unsigned long long i = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) * 4 + (unsigned long long)round * 64;
    const int* pa = cold + (i * 1021) % (n_cold - 4);          // scattered 4-byte reads over 2 GB
    const int* pb = hot + (threadIdx.x & 15);
    int a = 0x5a5a5a5a, b = 0;
    v4i res;
    res[0] = (int)(unsigned)(unsigned long long)cold; res[1] = (int)((unsigned long long)cold >> 32) & 0xffff; res[2] = 0x7ffffff0; res[3] = 0x00020000;
    res[0] = __builtin_amdgcn_readfirstlane(res[0]); res[1] = __builtin_amdgcn_readfirstlane(res[1]);
    const unsigned offa = (unsigned)((const char*)pa - (const char*)cold);
    if (MODE == 0) {
        asm volatile("global_load_dword %0, %2, off\n\tglobal_load_dword %1, %3, off\n\ts_waitcnt vmcnt(1)" : "+v"(a), "=v"(b) : "v"(pa), "v"(pb) : "memory");
    } else if (MODE == 1) {
        v4i rh;
        rh[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)hot); rh[1] = __builtin_amdgcn_readfirstlane((int)((unsigned long long)hot >> 32) & 0xffff);
        rh[2] = 4096; rh[3] = 0x00020000;
        const unsigned offb = (threadIdx.x & 15) * 4;
        asm volatile("buffer_load_dword %0, %2, %4, 0 offen\n\tbuffer_load_dword %1, %3, %5, 0 offen\n\ts_waitcnt vmcnt(1)"
                     : "+v"(a), "=v"(b) : "v"(offa), "v"(offb), "s"(res), "s"(rh) : "memory");
    } else {
        asm volatile("buffer_load_dword %0, %2, %4, 0 offen\n\tglobal_load_dword %1, %3, off\n\ts_waitcnt vmcnt(1)"
                     : "+v"(a), "=v"(b) : "v"(offa), "v"(pb), "s"(res) : "memory");
    }
    const int got = a;                       // read A's register right after the counted wait
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(b)::"memory");
    if (got == 0x5a5a5a5a) atomicAdd(bad, 1u);          // the cold buffer holds no poison values
    if (b == 0x12345678) atomicAdd(bad + 1, 1u);
}

int main() {
    const unsigned long long n_cold = 512ull << 20;       // 2 GB of ints
    int *cold, *hot; unsigned* bad;
    hipMalloc(&cold, n_cold * 4); hipMalloc(&hot, 4096); hipMalloc(&bad, 8);
    hipMemset(cold, 1, n_cold * 4); hipMemset(hot, 2, 4096);
    for (int mode = 0; mode < 3; ++mode) {
        hipMemset(bad, 0, 8);
        for (int r = 0; r < 20; ++r) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(65536), dim3(256), 0, 0, cold, hot, n_cold, bad, r);
            else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(65536), dim3(256), 0, 0, cold, hot, n_cold, bad, r);
            else hipLaunchKernelGGL(k<2>, dim3(65536), dim3(256), 0, 0, cold, hot, n_cold, bad, r);
        }
        unsigned h[2]; hipMemcpy(h, bad, 8, hipMemcpyDeviceToHost);
        printf("mode %d (%s): %u of %llu lanes read A before it arrived\n", mode,
               mode == 0 ? "global, global" : mode == 1 ? "buffer, buffer" : "buffer, global", h[0], 20ull * 65536 * 256);
    }
    return 0;
}
