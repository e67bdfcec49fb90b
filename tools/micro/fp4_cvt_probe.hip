// What v_cvt_scalef32_pk_fp4_f32 (gfx950) does: nibble order, meaning of the scale operand, rounding, saturation.
//   hipcc --offload-arch=gfx950 -O2 -o fp4_cvt_probe fp4_cvt_probe.hip && ./fp4_cvt_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* a, const float* b, const float* sc, unsigned* out, int n) {
    const int i = threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(0xAABBCC00u, a[i], b[i], sc[i], 0);
}
int main() {
    const float A[] = {1.f, 0.5f, 6.f, 7.f, 100.f, -1.5f, 0.24f, 0.26f, 0.75f, 1.25f, 1.75f, 2.5f, 3.5f, 5.f, 2.f, 2.f, 2.f, 1e-30f, 8.f, 3.f};
    const float B[] = {2.f, 3.f, 4.f, -7.f, -100.f, 0.f, 0.25f, 0.74f, 0.76f, 1.26f, 1.74f, 2.49f, 3.51f, 5.01f, 2.f, 2.f, 2.f, 0.f, 8.f, 3.f};
    const float S[] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 2.f, 0.5f, 3.f, 1.f, 4.f, 1.5f};
    const int n = sizeof(A) / 4;
    float *da, *db, *ds; unsigned* dout; unsigned h[64];
    (void)hipMalloc(&da, 256); (void)hipMalloc(&db, 256); (void)hipMalloc(&ds, 256); (void)hipMalloc(&dout, 256);
    (void)hipMemcpy(da, A, sizeof(A), hipMemcpyHostToDevice); (void)hipMemcpy(db, B, sizeof(B), hipMemcpyHostToDevice); (void)hipMemcpy(ds, S, sizeof(S), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, ds, dout, n);
    (void)hipMemcpy(h, dout, n * 4, hipMemcpyDeviceToHost);
    const float V[8] = {0, .5f, 1, 1.5f, 2, 3, 4, 6};
    for (int i = 0; i < n; ++i) {
        const unsigned lo = h[i] & 15, hi = (h[i] >> 4) & 15;
        printf("src0 %8g src1 %8g scale %4g -> word %08x  low nibble %x = %5g   high nibble %x = %5g\n", A[i], B[i], S[i], h[i], lo,
               (lo & 8 ? -1 : 1) * V[lo & 7], hi, (hi & 8 ? -1 : 1) * V[hi & 7]);
    }
    return 0;
}
