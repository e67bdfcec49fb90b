// Operand-layout probe for v_mfma_scale_f32_16x16x128_f8f6f4 with FP4 (e2m1) operands on gfx950.
// The kernel is layout-agnostic: lane l feeds the 32 bytes a[l], b[l] and the scale words sa[l], sb[l] it is given; the
// host packs A (16 x 128), B (128 x 16) and the E8M0 block scales under a HYPOTHESIS about the lane / nibble / scale-byte
// mapping and checks D against a double reference.   hipcc --offload-arch=gfx950 -O2 -o mx_probe mx_probe.hip && ./mx_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int OPA, int OPB>
__global__ void k(const v8i* a, const v8i* b, const int* sa, const int* sb, v4f* d) {
    const int l = threadIdx.x;
    v4f c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], c, 4, 4, OPA, sa[l], OPB, sb[l]);
    d[l] = c;
}

static const float FP4[8] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f};
static float fp4v(int code) { return (code & 8 ? -1.f : 1.f) * FP4[code & 7]; }

int main() {
    srand(1);
    int A[16][128], B[128][16];            // fp4 codes
    int SA[16][4], SB[16][4];              // E8M0 exponents per (row, 32-block) / (col, 32-block)
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 128; ++k) { A[i][k] = rand() & 15; B[k][i] = rand() & 15; }
    for (int i = 0; i < 16; ++i) for (int q = 0; q < 4; ++q) { SA[i][q] = 124 + rand() % 7; SB[i][q] = 124 + rand() % 7; }
    double ref[16][16];
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double s = 0;
        for (int k = 0; k < 128; ++k) s += (double)fp4v(A[i][k]) * ldexp(1.0, SA[i][k / 32] - 127) * (double)fp4v(B[k][j]) * ldexp(1.0, SB[j][k / 32] - 127);
        ref[i][j] = s;
    }
    v8i *da, *db; int *dsa, *dsb; v4f* dd;
    hipMalloc(&da, 64 * 32); hipMalloc(&db, 64 * 32); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dd, 64 * 16);
    // hypotheses: kmap 0: lane l -> row l%16, k = 32*(l/16) + j ; kmap 1: k = 8*(l/16) + 32*(j/8) + j%8 (interleaved like 4 x 16x16x32)
    //             nib 0: element 2i in the LOW nibble of byte i ; nib 1: in the high nibble
    //             scale byte position p (0..3) of the per-lane scale word, opsel fixed at compile time (variants 0 and 1 below)
    for (int opsel = 0; opsel < 2; ++opsel)
    for (int kmap = 0; kmap < 2; ++kmap) for (int nib = 0; nib < 2; ++nib) for (int sp = 0; sp < 4; ++sp) {
        uint8_t ha[64][32] = {}, hb[64][32] = {};
        int hsa[64], hsb[64];
        for (int l = 0; l < 64; ++l) {
            const int r = l % 16, g = l / 16;
            for (int j = 0; j < 32; ++j) {
                const int kk = kmap == 0 ? 32 * g + j : 8 * g + 32 * (j / 8) + j % 8;
                const int ca = A[r][kk], cb = B[kk][r];
                const int byte = j / 2, hi = (j & 1) ^ nib;
                ha[l][byte] |= (uint8_t)(ca << (hi ? 4 : 0));
                hb[l][byte] |= (uint8_t)(cb << (hi ? 4 : 0));
            }
            // the scale of the lane's own 32-block (kmap 0) -- for kmap 1 a lane spans all four blocks, so only uniform scales could work
            hsa[l] = 0x7f7f7f7f; hsb[l] = 0x7f7f7f7f;
            hsa[l] = (hsa[l] & ~(0xff << (8 * sp))) | (SA[r][g] << (8 * sp));
            hsb[l] = (hsb[l] & ~(0xff << (8 * sp))) | (SB[r][g] << (8 * sp));
        }
        hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
        hipMemcpy(dsa, hsa, sizeof(hsa), hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, sizeof(hsb), hipMemcpyHostToDevice);
        if (opsel == 0) hipLaunchKernelGGL((k<0, 0>), dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
        else hipLaunchKernelGGL((k<1, 1>), dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
        float hd[64][4];
        hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost);
        // C/D: col = lane & 15, row = (lane >> 4) * 4 + reg   (D[i][j], i = A row, j = B col)
        int bad = 0, badT = 0;
        double maxerr = 0;
        for (int l = 0; l < 64; ++l) for (int rg = 0; rg < 4; ++rg) {
            const int col = l & 15, row = (l >> 4) * 4 + rg;
            const double e = fabs(hd[l][rg] - ref[row][col]);
            if (e > 1e-3 * (1 + fabs(ref[row][col]))) ++bad;
            if (fabs(hd[l][rg] - ref[col][row]) > 1e-3 * (1 + fabs(ref[col][row]))) ++badT;
            if (e > maxerr) maxerr = e;
        }
        printf("opsel %d kmap %d nib %d scalebyte %d: mismatches %3d (transposed D: %3d) max err %.3g\n", opsel, kmap, nib, sp, bad, badT, maxerr);
    }
    return 0;
}
