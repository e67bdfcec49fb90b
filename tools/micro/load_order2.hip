// Deeper version of load_order.hip, shaped like k_dwpw's tap wait: 9 cold 16-byte buffer loads, then 13 hot ones,
// `s_waitcnt vmcnt(13)`, then every cold destination register is checked against its poison value.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(512) k(const int* cold, const int* hot, unsigned long long n_cold, unsigned* bad, int round) {
    v4i res, rh;
    res[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)cold);
    res[1] = __builtin_amdgcn_readfirstlane((int)((unsigned long long)cold >> 32) & 0xffff);
    res[2] = 0x7ffffff0; res[3] = 0x00020000;
    rh[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)hot);
    rh[1] = __builtin_amdgcn_readfirstlane((int)((unsigned long long)hot >> 32) & 0xffff);
    rh[2] = 65536; rh[3] = 0x00020000;
    const unsigned long long i = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) + (unsigned long long)round * 977;
    unsigned offa[9];
    for (int t = 0; t < 9; ++t) offa[t] = (unsigned)(((i * 9 + t) * 4099ull * 16ull) % 0x7ff00000ull) & ~15u;
    v4i a[9], b[13];
    for (int t = 0; t < 9; ++t) a[t] = v4i{0x5a5a5a5a, 0x5a5a5a5a, 0x5a5a5a5a, 0x5a5a5a5a};
    for (int t = 0; t < 9; ++t) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "+v"(a[t]) : "v"(offa[t]), "s"(res) : "memory");
    for (int t = 0; t < 13; ++t) {
        const unsigned offb = ((threadIdx.x & 7) + t * 8) * 16;
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(b[t]) : "v"(offb), "s"(rh) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(13)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8])::"memory");
    int stale = 0;
    for (int t = 0; t < 9; ++t) stale += (a[t][0] == 0x5a5a5a5a) + (a[t][3] == 0x5a5a5a5a);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int sum = 0;
    for (int t = 0; t < 13; ++t) sum += b[t][0];
    if (stale) atomicAdd(bad, (unsigned)stale);
    if (sum == 0x12345678) atomicAdd(bad + 1, 1u);
}

int main() {
    const unsigned long long n_cold = 512ull << 20;
    int *cold, *hot; unsigned* bad;
    hipMalloc(&cold, n_cold * 4); hipMalloc(&hot, 65536); hipMalloc(&bad, 8);
    hipMemset(cold, 1, n_cold * 4); hipMemset(hot, 2, 65536); hipMemset(bad, 0, 8);
    for (int r = 0; r < 40; ++r) hipLaunchKernelGGL(k, dim3(8192), dim3(512), 0, 0, cold, hot, n_cold, bad, r);
    unsigned h[2]; hipMemcpy(h, bad, 8, hipMemcpyDeviceToHost);
    printf("9 cold + 13 hot dwordx4 buffer loads, vmcnt(13): %u stale registers in %llu lanes\n", h[0], 40ull * 8192 * 512);
    return 0;
}
