// 1x1 convolution = GEMM on the matrix cores (gfx950).
//
//   out[m][n] = act( sum_k in[m][k] * w[n][k] + bias[n] (+ res[m][n]) )
//
// in  : NHWC activations, one row per pixel (row stride lda), K = input channels contiguous
// w   : [n][k] (PyTorch's [Cout][Cin] with BatchNorm folded in), K contiguous
// 94 % of the network's MACs run here, so this is the MFMA-bound kernel.
//
// Structure (one workgroup = WM x WN waves, each wave a 64 x 64 output tile, BK = 128 bytes of K):
//   * both operand tiles go HBM/L2 -> LDS with global_load_lds_dwordx4 (no VGPR staging), two LDS
//     buffers, the load of K-step t+1 in flight while step t is multiplied; one barrier per step;
//   * LDS rows are 128 B; the 16-byte chunk index is XOR-ed with (row & 7).  The DMA writes LDS
//     linearly, so the swizzle is applied to the per-lane SOURCE address and again on the read;
//   * the product is computed transposed (D^T = W . in^T): the MFMA "A" operand is the weight
//     fragment, "B" the activation fragment, so a lane ends up holding 16 CONSECUTIVE channels of one
//     pixel (weight rows are permuted inside the 64-wide wave tile to make that so) and the NHWC
//     store / residual load are 32-byte (bf16) or 64-byte (fp32) contiguous per lane;
//   * workgroup ids are remapped so that the tiles sharing an activation panel run on one XCD (L2).
//   * T = bf16 : v_mfma_f32_16x16x32_bf16, fp32 accumulate.
//     T = float: v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain) -- the reference-precision mode.
#include <cstdlib>

#include "seg_types.h"

namespace avl {
namespace {

struct GemmArgs {
    const void* A;
    const void* W;
    const float* bias;
    const void* R;
    void* C;
    int lda, ldr, ldc;
    int M, N, K;
    int relu, out_f32;
    int ntiles;
    // "mixed" precision (f16): nsub = MFMA passes per 64-wide K block.  1: plain.  2: weights split (hi, lo), A used twice.
    // 3: A split as well: (A hi, W hi), (A hi, W lo), (A lo, W hi); a_lo_delta = byte distance from A's hi plane to its lo plane.
    // W rows are K * nsub long and already interleaved in that order by the host.  R_lo / C_lo: low planes of the
    // residual / of the output (NULL = single plane).
    int nsub;
    long long a_lo_delta;
    const void* R_lo;
    void* C_lo;
};

// which sub-steps of a K block bring a NEW activation tile (the others reuse the tile already in LDS)
template <int NSUB> __device__ __forceinline__ bool sub_needs_a(int j) { return NSUB == 1 || j == 0 || (NSUB == 3 && j == 2); }

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <typename T, int WM, int WN>
__global__ void __launch_bounds__(WM* WN * 64) k_gemm(GemmArgs p) {
    constexpr int NW = WM * WN, BM = WM * 64, BN = WN * 64;
    constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128, BUF = A_BYTES + W_BYTES;
    constexpr int A_INSTR = BM / 8 / NW, W_INSTR = BN / 8 / NW;
    static_assert(A_INSTR >= 1 && W_INSTR >= 1, "tile too small for the wave count");
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // XCD-aware, bijective remap: workgroups b and b+8 share an XCD (and its L2); give each XCD a
    // contiguous run of logical tiles so the WN.. tiles of one activation panel hit the same L2.
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, xcd = bid & 7, local = bid >> 3, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    }
    const int nt = bid % p.ntiles, mt = bid / p.ntiles;
    const int m0 = mt * BM, n0 = nt * BN;

    // per-lane DMA sources: lane -> (row = lane>>3 of an 8-row group, physical chunk = lane&7)
    const char* a_src[A_INSTR];
    const char* w_src[W_INSTR];
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i) {
        const int r = (i * NW + wave) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ (r & 7);
        a_src[i] = static_cast<const char*>(p.A) + (long long)(m0 + r) * p.lda * (int)sizeof(T) + c * 16;
    }
#pragma unroll
    for (int i = 0; i < W_INSTR; ++i) {
        const int r = (i * NW + wave) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ (r & 7);
        w_src[i] = static_cast<const char*>(p.W) + (long long)(n0 + r) * p.K * p.nsub * (int)sizeof(T) + c * 16;
    }
    auto stage = [&](int buf, int kt) {
        char* base = lds + buf * BUF;
        const int kb = kt / p.nsub, j = kt - kb * p.nsub;                    // K block, pass inside it (see GemmArgs::nsub)
        const long long aoff = (long long)kb * 128 + (j == 2 ? p.a_lo_delta : 0);
#pragma unroll
        for (int i = 0; i < A_INSTR; ++i) glds16(a_src[i] + aoff, base + (i * NW + wave) * 1024);
#pragma unroll
        for (int i = 0; i < W_INSTR; ++i) glds16(w_src[i] + (long long)kt * 128, base + A_BYTES + (i * NW + wave) * 1024);
    };

    // fragment addresses (constant over the K loop)
    const int fr = lane & 15, kq = lane >> 4;
    int a_row[4], w_row[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) a_row[mi] = wm * 64 + mi * 16 + fr;
#pragma unroll
    for (int nj = 0; nj < 4; ++nj) w_row[nj] = wn * 64 + (fr >> 2) * 16 + nj * 4 + (fr & 3);

    f32x4 acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int nj = 0; nj < 4; ++nj) acc[mi][nj] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K * (int)sizeof(T) / 128 * p.nsub;
    stage(0, 0);
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of tile kt has landed
        __syncthreads();                                     // everyone's has; buffer cur^1 is free again
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        const char* ab = lds + cur * BUF;
        const char* wb = ab + A_BYTES;
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int chunk = kk * 4 + kq;
                typedef typename Half16<T>::v8 v8;
                v8 af[4], wf[4];
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    af[mi] = *reinterpret_cast<const v8*>(ab + a_row[mi] * 128 + ((chunk ^ (a_row[mi] & 7)) << 4));
#pragma unroll
                for (int nj = 0; nj < 4; ++nj)
                    wf[nj] = *reinterpret_cast<const v8*>(wb + w_row[nj] * 128 + ((chunk ^ (w_row[nj] & 7)) << 4));
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int nj = 0; nj < 4; ++nj) acc[mi][nj] = Half16<T>::mfma(wf[nj], af[mi], acc[mi][nj]);
            }
        } else {
            // fp32: lane kq owns k = 8*kq .. 8*kq+7 of the 32-wide step; MFMA step s sums k-set {s, 8+s, 16+s, 24+s}
            f32x4 af[4][2], wf[4][2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int chunk = kq * 2 + h;
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    af[mi][h] = *reinterpret_cast<const f32x4*>(ab + a_row[mi] * 128 + ((chunk ^ (a_row[mi] & 7)) << 4));
#pragma unroll
                for (int nj = 0; nj < 4; ++nj)
                    wf[nj][h] = *reinterpret_cast<const f32x4*>(wb + w_row[nj] * 128 + ((chunk ^ (w_row[nj] & 7)) << 4));
            }
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int nj = 0; nj < 4; ++nj)
                        acc[mi][nj] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[nj][s >> 2][s & 3], af[mi][s >> 2][s & 3],
                                                                           acc[mi][nj], 0, 0, 0);
        }
        cur ^= 1;
    }

    // epilogue: lane = (pixel fr of each 16-pixel sub-tile, channel block kq): 16 consecutive channels
    const int nbase = n0 + wn * 64 + kq * 16;
    float bias[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float4 b = *reinterpret_cast<const float4*>(p.bias + nbase + 4 * j);
        bias[4 * j] = b.x; bias[4 * j + 1] = b.y; bias[4 * j + 2] = b.z; bias[4 * j + 3] = b.w;
    }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int m = m0 + wm * 64 + mi * 16 + fr;
        if (m >= p.M || nbase >= p.N) continue;
        float v[16];
#pragma unroll
        for (int nj = 0; nj < 4; ++nj)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[nj * 4 + r] = acc[mi][nj][r] + bias[nj * 4 + r];
        const bool full = nbase + 16 <= p.N;
        if (p.R) {
            const T* rp = static_cast<const T*>(p.R) + (long long)m * p.ldr + nbase;
            if (full) {
                float r0[8], r1[8];
                Vec8<T>::load(rp, r0);
                Vec8<T>::load(rp + 8, r1);
#pragma unroll
                for (int i = 0; i < 8; ++i) { v[i] += r0[i]; v[8 + i] += r1[i]; }
            } else {
                for (int i = 0; i < 16; ++i)
                    if (nbase + i < p.N) v[i] += to_f32(rp[i]);
            }
        }
        if (p.relu) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = fmaxf(v[i], 0.f);
        }
        if (p.out_f32) {
            float* cp = static_cast<float*>(p.C) + (long long)m * p.ldc + nbase;
            for (int i = 0; i < 16; ++i)
                if (nbase + i < p.N) cp[i] = v[i];
        } else {
            T* cp = static_cast<T*>(p.C) + (long long)m * p.ldc + nbase;
            if (full) {
                float lo[8], hi[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) { lo[i] = v[i]; hi[i] = v[8 + i]; }
                Vec8<T>::store(cp, lo);
                Vec8<T>::store(cp + 8, hi);
            } else {
                for (int i = 0; i < 16; ++i)
                    if (nbase + i < p.N) cp[i] = from_f32<T>(v[i]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// v2 (16-bit types): persistent workgroups, 256 x 128 or 256 x 256 tile (8 waves = 2 per SIMD), LDS ring.
//   * one workgroup per CU walks tiles t = vb, vb + grid, ...; the K-steps of all its tiles form one
//     stream, so the DMA for the NEXT tile's first steps is already in flight during the epilogue;
//   * the ring keeps STAGES - 1 steps ahead in flight: counted `s_waitcnt vmcnt(N)` + raw s_barrier
//     (a __syncthreads() would drain the LDS-DMA queue with vmcnt(0));
//   * weight-tile swizzle key is built from the row bits the permuted fragment rows actually vary in
//     (conflict-free ds_read_b128 for both operands).
//
// The vmcnt invariant: the only inline-asm memory operations are LDS-DMAs (no register outputs) and they
// retire in issue order.  At the top of step g the wave needs every DMA of steps <= g landed; the DMAs it may leave in
// flight are exactly those of step g+1 (STAGES = 3; none for STAGES = 2), so it waits for vmcnt(N) with N = the number
// of DMA instructions this wave issued for step g+1.  Compiler-visible loads/stores (bias in init_acc, the residual
// loads and the output stores of the epilogue) share the counter; they are all YOUNGER than step g+1's DMAs only when
// issued after them, and an s_waitcnt vmcnt(N) with extra younger operations outstanding merely waits for more than
// it needs (in-order retirement: "all but the N youngest" then still covers every DMA of steps <= g).  hipcc's own
// waits for those loads are vmcnt(0)-style and only over-wait as well.  So the count can over-wait, never under-wait.
//
// NSUB ("mixed" precision, f16): MFMA passes per 64-wide K block, see GemmArgs::nsub.  The activation ring and the weight
// ring advance independently: a sub-step always brings a weight tile, an activation tile only when it changes
// (sub_needs_a), so W hi/lo cost 1.5x the L2->LDS bytes of the plain GEMM for 2x the MFMAs.
// IO: the epilogue may read a split residual (R + R_lo) and write a split result (C + C_lo), both optional at run time.
// STAGES = weight tiles in the ring (the producer runs STAGES - 1 sub-steps ahead); AST = activation tiles in the ring:
// = STAGES for the plain GEMM; 2 is enough for NSUB = 2 at any depth because an activation tile lives for two sub-steps
// (256 x 256 tile: 2 x 32 KB + 3 x 32 KB = the whole 160 KB of LDS, two sub-steps of DMA in flight instead of one).
template <typename H, int WM, int WN, int MI, int STAGES, int PROBE = 0, int NSUB = 1, int IO = 0, int AST = STAGES>
__global__ void __launch_bounds__(WM* WN * 64) k_gemm_ring(GemmArgs p, int mtiles) {
    typedef typename Half16<H>::v8 v8;
    constexpr int NW = WM * WN, BM = WM * MI * 16, BN = WN * 64;
    constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128;
    constexpr int W_REGION = AST * A_BYTES;                    // LDS: [AST activation tiles][STAGES weight tiles]
    static_assert(AST == STAGES || (NSUB == 2 && AST == 2), "activation ring too short for this pass pattern");
    constexpr int A_INSTR = BM / 8 / NW, W_INSTR = BN / 8 / NW;
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int nwg = gridDim.x;
    int vb;
    {
        const int bid = blockIdx.x, xcd = bid & 7, local = bid >> 3, q = nwg >> 3, r = nwg & 7;
        vb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    }
    const int total = mtiles * p.ntiles;
    const int nkb = p.K / 64;                  // K blocks per tile; NSUB sub-steps each
    const unsigned lds_base = lds_addr(lds);

    // ---- producer side: per-lane DMA sources of the tile whose stages are being issued
    const int srow = lane >> 3, schunk = lane & 7;
    const char* a_src[A_INSTR];
    const char* w_src[W_INSTR];
    int pt = vb, pk = 0;                       // tile and K block being issued next
    int issued = 0, issued_a = 0;              // sub-steps / activation tiles issued so far (ring positions)
    auto set_tile = [&](int t) {
        const int nt = t % p.ntiles, mt = t / p.ntiles;
#pragma unroll
        for (int i = 0; i < A_INSTR; ++i) {
            const int r = (i * NW + wave) * 8 + srow;
            a_src[i] = static_cast<const char*>(p.A) + (long long)(mt * BM + r) * p.lda * 2 + ((schunk ^ (r & 7)) << 4);
        }
#pragma unroll
        for (int i = 0; i < W_INSTR; ++i) {
            const int r = (i * NW + wave) * 8 + srow;
            const int key = ((r >> 1) & 1) | (((r >> 4) & 3) << 1);
            w_src[i] = static_cast<const char*>(p.W) + (long long)(nt * BN + r) * p.K * (2 * NSUB) + ((schunk ^ key) << 4);
        }
    };
    // one sub-step = W_INSTR (+ A_INSTR when the activation tile changes) DMA instructions per wave; issue_part(h, NP, jj)
    // sends the h-th of NP equal shares of sub-step jj of K block pk (so the instructions can be spread between MFMA groups
    // instead of queueing up in front of them).  jj is wave-uniform: the producer runs STAGES - 1 sub-steps ahead of the consumer.
    auto issue_part = [&](int h, int np, int jj) {
        const bool na = sub_needs_a<NSUB>(jj);
        if (na) {
            const unsigned abase = lds_base + (issued_a % AST) * A_BYTES + wave * 1024;
            const long long aoff = (long long)pk * 128 + ((NSUB == 3 && jj == 2) ? p.a_lo_delta : 0);
#pragma unroll
            for (int i = 0; i < A_INSTR; ++i)
                if (i * np / A_INSTR == h) glds16_asm(a_src[i] + aoff, abase + i * NW * 1024);
        }
        const unsigned wbase = lds_base + W_REGION + (issued % STAGES) * W_BYTES + wave * 1024;
        const long long woff = (long long)(pk * NSUB + jj) * 128;
#pragma unroll
        for (int i = 0; i < W_INSTR; ++i)
            if (i * np / W_INSTR == h) glds16_asm(w_src[i] + woff, wbase + i * NW * 1024);
        if (h == np - 1) {
            ++issued;
            if (na) ++issued_a;
            if (jj == NSUB - 1) {
                if (++pk == nkb) {
                    pk = 0;
                    pt += nwg;
                    if (pt < total) set_tile(pt);
                }
            }
        }
    };
    if (pt < total) set_tile(pt);
#pragma unroll
    for (int i = 0; i < STAGES - 1; ++i)
        if (pt < total) issue_part(0, 1, i % NSUB);

    // ---- consumer side
    const int fr = lane & 15, kq = lane >> 4;
    int a_off[MI], w_off[4], a_key[MI], w_key[4];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int row = wm * (MI * 16) + mi * 16 + fr;
        a_off[mi] = row * 128;
        a_key[mi] = row & 7;
    }
#pragma unroll
    for (int nj = 0; nj < 4; ++nj) {
        const int row = wn * 64 + (fr >> 2) * 16 + nj * 4 + (fr & 3);
        w_off[nj] = W_REGION + row * 128;
        w_key[nj] = ((row >> 1) & 1) | (((row >> 4) & 3) << 1);
    }
    // The accumulators start from the bias, so the epilogue issues no load whose result could still be
    // pending when the K loop resumes (hipcc would then put a vmcnt(0) in front of every K-step's ds_reads).
    f32x4 acc[MI][4];
    auto init_acc = [&](int t) {
        const int nb = (t % p.ntiles) * BN + wn * 64 + kq * 16;
#pragma unroll
        for (int nj = 0; nj < 4; ++nj) {
            const float4 b = *reinterpret_cast<const float4*>(p.bias + nb + 4 * nj);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) acc[mi][nj] = f32x4{b.x, b.y, b.z, b.w};
        }
    };
    if (vb < total) init_acc(vb);

    int g = 0;      // sub-steps consumed so far
    int ca = -1;    // activation tiles consumed so far - 1 = ring position of the current one
    for (int t = vb; t < total; t += nwg) {
        const int nt = t % p.ntiles, mt = t / p.ntiles;
        // one loop over the tile's sub-steps; j = sub-step inside the K block (a scalar: every branch on it is uniform)
        for (int q = 0, j = 0; q < nkb * NSUB; ++q, ++g, j = (NSUB == 1 || j + 1 == NSUB) ? 0 : j + 1) {
            if (sub_needs_a<NSUB>(j)) ++ca;
            const int jn = (j + STAGES - 1) % NSUB;           // the sub-step that is issued during this one
            // sub-step g has landed (this wave's share); sub-step g+1 may stay in flight
            if (STAGES >= 3 && issued > g + 1) {
                if (sub_needs_a<NSUB>((j + 1) % NSUB)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_INSTR + W_INSTR) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W_INSTR) : "memory");
            } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const bool feed = pt < total;                  // sub-step g+STAGES-1 -> the slots consumed in step g-1
            const char* abase = lds + (ca % AST) * A_BYTES;
            const char* wbase = lds + (g % STAGES) * W_BYTES;
            if (PROBE == 1) { if (feed) issue_part(0, 1, jn); continue; }      // DMA only
            auto rd_w = [&](int kk, int nj) { return *reinterpret_cast<const v8*>(wbase + w_off[nj] + (((kk * 4 + kq) ^ w_key[nj]) << 4)); };
            auto rd_a = [&](int kk, int mi) { return *reinterpret_cast<const v8*>(abase + a_off[mi] + (((kk * 4 + kq) ^ a_key[mi]) << 4)); };
            // The step is cut into 4 MFMA groups; the next stage's DMA instructions and the second K-half's fragment
            // reads are placed BETWEEN the groups (a DMA burst issued in one go in front of the MFMAs was measured to
            // serialise with them: DMA-only 75 us + MFMA-only 65 us = 131 us for layer4.conv1).
            constexpr int HALF = MI / 2;
            // Waves w and w + NW/2 share a SIMD.  The second half issues its DMA at the head of the step, the first
            // half in the middle, so that on every SIMD one wave is in an MFMA group while its partner pays the
            // (100+ cycle per instruction) LDS-DMA issue cost, instead of both doing the same thing at the same time.
            const bool early = MI == 8 && NW >= 8 && wave >= NW / 2;      // measured: +4-7 % on 256x256 tiles, -4 % on 256x128
            if (feed && early) { issue_part(0, 2, jn); issue_part(1, 2, jn); }
            v8 wa[4], wb[4], af[MI];
#pragma unroll
            for (int nj = 0; nj < 4; ++nj) wa[nj] = rd_w(0, nj);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) af[mi] = rd_a(0, mi);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mi = 0; mi < HALF; ++mi)
#pragma unroll
                for (int nj = 0; nj < 4; ++nj) acc[mi][nj] = Half16<H>::mfma(wa[nj], af[mi], acc[mi][nj]);
            __builtin_amdgcn_sched_barrier(0);
            if (feed && !early) issue_part(0, 2, jn);
#pragma unroll
            for (int nj = 0; nj < 4; ++nj) wb[nj] = rd_w(1, nj);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mi = HALF; mi < MI; ++mi)
#pragma unroll
                for (int nj = 0; nj < 4; ++nj) acc[mi][nj] = Half16<H>::mfma(wa[nj], af[mi], acc[mi][nj]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mi = 0; mi < HALF; ++mi) af[mi] = rd_a(1, mi);
            if (feed && !early) issue_part(1, 2, jn);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mi = 0; mi < HALF; ++mi)
#pragma unroll
                for (int nj = 0; nj < 4; ++nj) acc[mi][nj] = Half16<H>::mfma(wb[nj], af[mi], acc[mi][nj]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mi = HALF; mi < MI; ++mi) af[mi] = rd_a(1, mi);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mi = HALF; mi < MI; ++mi)
#pragma unroll
                for (int nj = 0; nj < 4; ++nj) acc[mi][nj] = Half16<H>::mfma(wb[nj], af[mi], acc[mi][nj]);
        }
        // ---- epilogue of tile t (the next tile's first stages are already in flight).  All residual loads
        // are issued before the first one is consumed, so their latency is paid once per tile, not per sub-tile.
        const int nbase = nt * BN + wn * 64 + kq * 16;
        const bool ncol_ok = nbase + 16 <= p.N;
        constexpr int EB = IO ? 2 : 4;              // sub-tiles per epilogue batch (residual registers: 8 (16 split) per sub-tile)
        const bool rsplit = IO && p.R_lo != nullptr, csplit = IO && p.C_lo != nullptr;
#pragma unroll
        for (int b0 = 0; b0 < MI; b0 += EB) {
            v8 res[EB][2], resl[IO ? EB : 1][2];
            if (p.R) {
#pragma unroll
                for (int e = 0; e < EB; ++e) {
                    const int m = mt * BM + wm * (MI * 16) + (b0 + e) * 16 + fr;
                    if (m < p.M && ncol_ok) {
                        const H* rp = static_cast<const H*>(p.R) + (long long)m * p.ldr + nbase;
                        res[e][0] = *reinterpret_cast<const v8*>(rp);
                        res[e][1] = *reinterpret_cast<const v8*>(rp + 8);
                        if constexpr (IO != 0) {
                            if (rsplit) {
                                const H* rl = static_cast<const H*>(p.R_lo) + (long long)m * p.ldr + nbase;
                                resl[e][0] = *reinterpret_cast<const v8*>(rl);
                                resl[e][1] = *reinterpret_cast<const v8*>(rl + 8);
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < EB; ++e) {
                const int mi = b0 + e;
                const int m = mt * BM + wm * (MI * 16) + mi * 16 + fr;
                if (m < p.M && ncol_ok) {
                    float v[16];
#pragma unroll
                    for (int nj = 0; nj < 4; ++nj)
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[nj * 4 + r] = acc[mi][nj][r];
                    if (p.R) {
                        if constexpr (IO != 0) {
                            if (rsplit) {          // (hi + lo) first: exact in fp32 (lo is below hi's last bit)
#pragma unroll
                                for (int i = 0; i < 8; ++i) {
                                    v[i] += (float)res[e][0][i] + (float)resl[e][0][i];
                                    v[8 + i] += (float)res[e][1][i] + (float)resl[e][1][i];
                                }
                            } else {
#pragma unroll
                                for (int i = 0; i < 8; ++i) { v[i] += (float)res[e][0][i]; v[8 + i] += (float)res[e][1][i]; }
                            }
                        } else {
#pragma unroll
                            for (int i = 0; i < 8; ++i) { v[i] += (float)res[e][0][i]; v[8 + i] += (float)res[e][1][i]; }
                        }
                    }
                    if (p.relu) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) v[i] = fmaxf(v[i], 0.f);
                    }
                    H* cp = static_cast<H*>(p.C) + (long long)m * p.ldc + nbase;
                    float lo[8], hi[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) { lo[i] = v[i]; hi[i] = v[8 + i]; }
                    Vec8<H>::store(cp, lo);
                    Vec8<H>::store(cp + 8, hi);
                    if constexpr (IO != 0) {
                        if (csplit) {              // low plane: what the rounding to the 16-bit type dropped
#pragma unroll
                            for (int i = 0; i < 8; ++i) { lo[i] = v[i] - (float)(H)v[i]; hi[i] = v[8 + i] - (float)(H)v[8 + i]; }
                            H* cl = static_cast<H*>(p.C_lo) + (long long)m * p.ldc + nbase;
                            Vec8<H>::store(cl, lo);
                            Vec8<H>::store(cl + 8, hi);
                        }
                    }
                }
            }
        }
        if (t + nwg < total) init_acc(t + nwg);
    }
}

template <typename H, int WM, int WN, int MI, int STAGES, int NSUB = 1, int IO = 0, int AST = STAGES>
int launch_ring(const GemmArgs& a0, int M, hipStream_t s) {
    constexpr int BM = WM * MI * 16, BN = WN * 64, LDS = (AST * BM + STAGES * BN) * 128;
    static_assert(LDS <= 160 * 1024, "ring does not fit LDS");
    // set on every launch: the attribute is per device and this may be called from several threads / for several devices
    AVL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_ring<H, WM, WN, MI, STAGES, 0, NSUB, IO, AST>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    GemmArgs a = a0;
    a.ntiles = (a.N + BN - 1) / BN;
    const int mtiles = (M + BM - 1) / BM;
    const int total = mtiles * a.ntiles;
    const int grid = total < 256 ? total : 256;
    static const int probe = getenv("AVL_GEMM_PROBE") ? atoi(getenv("AVL_GEMM_PROBE")) : 0;   // timing experiments only
    if (probe == 1 && NSUB == 1 && IO == 0) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_ring<H, WM, WN, MI, STAGES, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        hipLaunchKernelGGL((k_gemm_ring<H, WM, WN, MI, STAGES, 1>), dim3(grid), dim3(WM * WN * 64), LDS, s, a, mtiles);
    } else
        hipLaunchKernelGGL((k_gemm_ring<H, WM, WN, MI, STAGES, 0, NSUB, IO, AST>), dim3(grid), dim3(WM * WN * 64), LDS, s, a, mtiles);
    AVL_LAUNCH_CHECK();
    return AVL_OK;
}

template <typename T, int WM, int WN>
int launch_cfg(const GemmArgs& a, int mtiles, hipStream_t s) {
    constexpr int LDS = 2 * (WM * 64 + WN * 64) * 128;
    AVL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm<T, WM, WN>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    hipLaunchKernelGGL((k_gemm<T, WM, WN>), dim3(mtiles * a.ntiles), dim3(WM * WN * 64), LDS, s, a);
    AVL_LAUNCH_CHECK();
    return AVL_OK;
}

struct TileCfg { int bm, bn; };
inline TileCfg pick_tile(const avl_seg_op& op);
inline bool ring_eligible(const avl_seg_op& op) {
    const int M = op.out_h * op.out_w;
    return is_half(op.dtype) && op.w_layout != 1 && op.out_c > 64 && !op.out_f32 && op.out_c % 128 == 0 && op.in_rows >= (M + 255) / 256 * 256;
}
inline TileCfg pick_tile(const avl_seg_op& op) {
    if (op.out_c <= 64) return {256, 64};
    return {128, 128};
}

}  // namespace

int validate_gemm(const avl_seg_op& op) {
    const int es = elem_size(op.dtype);
    AVL_REQUIRE(is_half(op.dtype) || op.dtype == AVL_F32, "GEMM dtype %d", op.dtype);
    AVL_REQUIRE(op.in && op.out && op.weight && op.bias, "GEMM has NULL buffers");
    const int M = op.out_h * op.out_w, K = op.in_c, N = op.out_c;
    AVL_REQUIRE(M > 0 && N > 0 && K > 0, "GEMM M/N/K = %d/%d/%d", M, N, K);
    AVL_REQUIRE(op.in_h * op.in_w == M, "GEMM in/out pixel counts differ (%d vs %d)", op.in_h * op.in_w, M);
    AVL_REQUIRE((K * es) % 128 == 0, "GEMM K = %d is not a multiple of %d", K, 128 / es);
    AVL_REQUIRE(op.in_ld >= K && (op.in_ld * es) % 16 == 0, "GEMM in_ld %d", op.in_ld);
    AVL_REQUIRE(op.out_ld >= N, "GEMM out_ld %d < N %d", op.out_ld, N);
    const TileCfg t = pick_tile(op);
    const int mtiles = (M + t.bm - 1) / t.bm, ntiles = (N + t.bn - 1) / t.bn;
    AVL_REQUIRE(op.in_rows >= mtiles * t.bm, "GEMM reads %d rows, input has %d allocated", mtiles * t.bm, op.in_rows);
    AVL_REQUIRE(op.w_rows >= ntiles * t.bn, "GEMM reads %d weight rows, %d allocated", ntiles * t.bn, op.w_rows);
    AVL_REQUIRE(op.out_rows >= M, "GEMM writes %d rows, output has %d", M, op.out_rows);
    if (!op.out_f32) AVL_REQUIRE((op.out_ld * es) % 16 == 0 && N % 16 == 0, "GEMM out_ld %d / N %d not 16-aligned", op.out_ld, N);
    if (op.in2) AVL_REQUIRE(op.in2_ld >= N && (op.in2_ld * es) % 16 == 0, "GEMM residual ld %d", op.in2_ld);
    if (op.w_split || op.in_lo || op.in2_lo || op.out_lo) {
        AVL_REQUIRE(op.dtype == AVL_F16 && op.w_split == 1, "split (hi + lo) operands need AVL_F16 activations and w_split = 1");
        AVL_REQUIRE(!op.in2_lo || op.in2, "in2_lo without in2");
        AVL_REQUIRE((reinterpret_cast<uintptr_t>(op.in_lo) | reinterpret_cast<uintptr_t>(op.in2_lo) | reinterpret_cast<uintptr_t>(op.out_lo)) % 16 == 0,
                    "GEMM low planes must be 16-byte aligned");
        if (op.in2_lo || op.out_lo)
            AVL_REQUIRE(ring_eligible(op), "split residual / output need the ring GEMM (N %% 128 == 0, 16-bit output, rows padded to 256)");
    }
    AVL_REQUIRE((reinterpret_cast<uintptr_t>(op.in) | reinterpret_cast<uintptr_t>(op.weight) | reinterpret_cast<uintptr_t>(op.out) |
                 reinterpret_cast<uintptr_t>(op.bias) | reinterpret_cast<uintptr_t>(op.in2)) % 16 == 0,
                "GEMM buffers must be 16-byte aligned");
    return AVL_OK;
}

int launch_gemm(const avl_seg_op& op, hipStream_t s) {
    GemmArgs a;
    a.A = op.in; a.W = op.weight; a.bias = op.bias; a.R = op.in2; a.C = op.out;
    a.lda = op.in_ld; a.ldr = op.in2_ld; a.ldc = op.out_ld;
    a.M = op.out_h * op.out_w; a.N = op.out_c; a.K = op.in_c;
    a.relu = op.relu; a.out_f32 = op.out_f32;
    a.nsub = op.w_split ? (op.in_lo ? 3 : 2) : 1;
    a.a_lo_delta = op.in_lo ? static_cast<const char*>(op.in_lo) - static_cast<const char*>(op.in) : 0;
    a.R_lo = op.in2_lo; a.C_lo = op.out_lo;
    const TileCfg t = pick_tile(op);
    const int mtiles = (a.M + t.bm - 1) / t.bm;
    a.ntiles = (a.N + t.bn - 1) / t.bn;
    // 16-bit variants.  w_layout: 0 = pick by shape, 1 = v1 (128x128, 2 LDS buffers, 2 workgroups/CU),
    // 2 = ring 256x128 x3 stages, 3 = ring 256x256 x2 stages, 4 = ring 256x128 (4 waves) x3 stages.
    // 256x256 halves the L2->LDS bytes per flop (the measured limiter) but needs >= ~200 tiles to fill 256 CUs.
    if (ring_eligible(op)) {
        const bool can256 = a.N % 256 == 0 && op.w_rows % 256 == 0;
        int v = op.w_layout;
        if (v == 0) v = (can256 && ((a.M + 255) / 256) * (a.N / 256) >= 192) ? 3 : 2;
        if (a.nsub == 2) {
            static const int deep = getenv("AVL_GEMM_DEEP") ? atoi(getenv("AVL_GEMM_DEEP")) : 1;      // A/B: 0 = 2 + 2 tiles, one sub-step ahead
            if (v == 3 && can256) return deep ? launch_ring<f16, 2, 4, 8, 3, 2, 1, 2>(a, a.M, s) : launch_ring<f16, 2, 4, 8, 2, 2, 1>(a, a.M, s);
            return launch_ring<f16, 4, 2, 4, 3, 2, 1>(a, a.M, s);
        }
        if (a.nsub == 3) {
            if (v == 3 && can256) return launch_ring<f16, 2, 4, 8, 2, 3, 1>(a, a.M, s);
            return launch_ring<f16, 4, 2, 4, 3, 3, 1>(a, a.M, s);
        }
        const bool bf = op.dtype == AVL_BF16;
        if (v == 3 && can256) return bf ? launch_ring<bf16, 2, 4, 8, 2>(a, a.M, s) : launch_ring<f16, 2, 4, 8, 2>(a, a.M, s);
        if (v == 4) return bf ? launch_ring<bf16, 2, 2, 8, 3>(a, a.M, s) : launch_ring<f16, 2, 2, 8, 3>(a, a.M, s);
        return bf ? launch_ring<bf16, 4, 2, 4, 3>(a, a.M, s) : launch_ring<f16, 4, 2, 4, 3>(a, a.M, s);
    }
    if (op.dtype == AVL_BF16) {
        if (t.bn == 64) return launch_cfg<bf16, 4, 1>(a, mtiles, s);
        return launch_cfg<bf16, 2, 2>(a, mtiles, s);
    }
    if (op.dtype == AVL_F16) {
        if (t.bn == 64) return launch_cfg<f16, 4, 1>(a, mtiles, s);
        return launch_cfg<f16, 2, 2>(a, mtiles, s);
    }
    if (t.bn == 64) return launch_cfg<float, 4, 1>(a, mtiles, s);
    return launch_cfg<float, 2, 2>(a, mtiles, s);
}

}  // namespace avl
