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
#include <cstring>
#include <type_traits>

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
    // k_gemm with out_f32 and N <= 32 (the classifier) has no low plane: C_lo is then the uint8 label map, torch.argmax over the row's N
    // values (semantic_segmentation.py:56: first maximal index wins, a NaN counts as maximal) written as one byte per row from the very
    // values stored as logits; NULL = not wanted.  (Not a field of its own: MxArgs embeds this struct and k_gemm_mx_pipe keeps every
    // kernel argument in SGPRs.)
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
    unsigned char* const labels = p.out_f32 ? static_cast<unsigned char*>(p.C_lo) : nullptr;
    if (labels) {
        // the classifier's arg-max (N <= 32: channel blocks kq = 0 and 1 of a row sit 16 lanes apart), on the values stored below; every lane
        // takes part in the exchange, so this runs in front of the loop whose lanes drop out individually
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int m = m0 + wm * 64 + mi * 16 + fr;
            float best = acc[mi][0][0] + bias[0];
            int bi = nbase;
#pragma unroll
            for (int i = 1; i < 16; ++i) {
                const float val = acc[mi][i >> 2][i & 3] + bias[i];
                if (nbase + i < p.N && (val > best || (val != val && best == best))) { best = val; bi = nbase + i; }
            }
            const float ob = __shfl_down(best, 16);
            const int oi = __shfl_down(bi, 16);
            if (nbase + 16 < p.N && (ob > best || (ob != ob && best == best))) { best = ob; bi = oi; }
            if (kq == 0 && m < p.M) labels[m] = (unsigned char)bi;
        }
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
        }
        if (!p.out_f32) {
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
            // PROBE 2: LDS reads + MFMAs, no DMA;  PROBE 3: DMA + LDS reads, no MFMAs (timing experiments, results are garbage)
            auto mm = [&](const v8& w_, const v8& a_, const f32x4& c_) -> f32x4 {
                if constexpr (PROBE == 3) { asm volatile("" ::"v"(w_), "v"(a_)); return c_; }
                else return Half16<H>::mfma(w_, a_, c_);
            };
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
            if (feed && early && PROBE != 2) { issue_part(0, 2, jn); issue_part(1, 2, jn); }
            v8 wa[4], wb[4], af[MI];
#pragma unroll
            for (int nj = 0; nj < 4; ++nj) wa[nj] = rd_w(0, nj);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) af[mi] = rd_a(0, mi);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mi = 0; mi < HALF; ++mi)
#pragma unroll
                for (int nj = 0; nj < 4; ++nj) acc[mi][nj] = mm(wa[nj], af[mi], acc[mi][nj]);
            __builtin_amdgcn_sched_barrier(0);
            if (feed && !early && PROBE != 2) issue_part(0, 2, jn);
#pragma unroll
            for (int nj = 0; nj < 4; ++nj) wb[nj] = rd_w(1, nj);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mi = HALF; mi < MI; ++mi)
#pragma unroll
                for (int nj = 0; nj < 4; ++nj) acc[mi][nj] = mm(wa[nj], af[mi], acc[mi][nj]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mi = 0; mi < HALF; ++mi) af[mi] = rd_a(1, mi);
            if (feed && !early && PROBE != 2) issue_part(1, 2, jn);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mi = 0; mi < HALF; ++mi)
#pragma unroll
                for (int nj = 0; nj < 4; ++nj) acc[mi][nj] = mm(wb[nj], af[mi], acc[mi][nj]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mi = HALF; mi < MI; ++mi) af[mi] = rd_a(1, mi);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mi = HALF; mi < MI; ++mi)
#pragma unroll
                for (int nj = 0; nj < 4; ++nj) acc[mi][nj] = mm(wb[nj], af[mi], acc[mi][nj]);
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
#ifdef AVL_EXPERIMENTS
    const int probe = AVL_EXP_INT("AVL_GEMM_PROBE", 0);     // timing experiments only: the results are garbage
    if (probe >= 1 && probe <= 3 && NSUB == 1 && IO == 0) {
#define AVL_PROBE_LAUNCH(P)                                                                                                                     \
    do {                                                                                                                                        \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_ring<H, WM, WN, MI, STAGES, P>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS); \
        hipLaunchKernelGGL((k_gemm_ring<H, WM, WN, MI, STAGES, P>), dim3(grid), dim3(WM * WN * 64), LDS, s, a, mtiles);                         \
    } while (0)
        if (probe == 1) AVL_PROBE_LAUNCH(1);
        else if (probe == 2) AVL_PROBE_LAUNCH(2);
        else AVL_PROBE_LAUNCH(3);
#undef AVL_PROBE_LAUNCH
    } else
#endif
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

// ---------------------------------------------------------------------------------------------
// MX variant of the 256 x 256 ring GEMM ("mixed" precision, layer3 / layer4 / ASPP 1x1 convs).
//
//   out = W . x   with  W = Wh + Wl,  x = xh (+ xl)   (f16 pairs)
//       ~ Wh.xh                                  main pass:   v_mfma_f32_16x16x32_f16, K blocks of 64
//       + Q4(Wl).Q4(xh)  [+ Q4(Wh).Q4(xl)]       corrections: v_mfma_scale_f32_16x16x128_f8f6f4 on MX-FP4 operands
//
// The correction terms are 2^-11 of the result, so 2-3 significant bits are plenty for them (tools/precision_study.py:
// the logits error does not move): FP4 (e2m1) with one power-of-two scale per 32 values along K, which the CDNA4 matrix
// cores multiply at 4x the f16 rate -- a correction pass costs a quarter of an f16 pass instead of a whole one.
// The scaled MFMA has the same C/D register layout as the f16 one, so all passes accumulate into the same registers.
//
// Data: an FP4 plane is [rows][K/2] bytes (element 2i in the low nibble of byte i); its scales are E8M0 bytes laid out
// [K/256][rows][8] so that a 256-row tile's scales for one sub-step are 2 KB contiguous.  In LDS an FP4 tile is 256 rows
// x 128 B = 256 K-values per row: exactly the geometry (DMA, XOR swizzle, fragment addressing) of an f16 tile with 64.
// One K macro-block of 256 therefore is 4 f16 sub-steps + 1 or 2 FP4 sub-steps, every one of them "DMA a 32 KB
// activation tile and a 32 KB weight tile, 64 MFMAs per wave".  Two ring stages, vmcnt(0) waits (no hand-counted
// immediates here: the scale DMAs make the per-wave instruction count non-uniform).
struct MxArgs {
    GemmArgs g;                 // A = x hi plane (f16), W = W hi [N][K] f16 (plain rows), R / R_lo, C / C_lo as usual
    const char* Aq[2];          // FP4 planes of the input: Q4(x hi), Q4(x lo) (second one only when nmx == 2)
    const char* As[2];          // their scales
    const char* Wq[2];          // FP4 weights paired with them: Q4(W lo), Q4(W hi)
    const char* Ws[2];
    int ldaq;                   // bytes per row of the input FP4 planes
    long long a_srows, w_srows; // rows per K macro-block in the scale arrays (allocated rows)
    char* Cq[2];                // FP4 planes of the OUTPUT (hi part, lo part) and their scales, or NULL
    char* Cs[2];
    int ldcq;
    long long c_srows;
    int nmx;                    // correction passes: 1 (weights only) or 2 (input split as well)
    const char* Rq;             // FP4 plane + scales of the residual's lo part (instead of the f16 plane g.R_lo), or NULL
    const char* Rs;
    int ldrq;
    long long r_srows;
    // second input appended along K (conv3 + downsample in one product): macro-blocks >= nmb1 read it
    int nmb1;
    const void* A2;
    int lda2, ldaq2;
    const char* Aq2[2];
    const char* As2[2];
    long long a_srows2;
    int stagger;                // 1: waves 0-3 issue their LDS-DMAs after the first K half of a sub-step (A/B switch)
    unsigned long long* dbg;    // AVL_MX_PROBE=3 only: per-wave cycle sums (host-visible memory), else NULL
    int same_tile;              // AVL_MX_PROBE=3 + AVL_MX_SAMETILE=1 (experiments): all workgroups load tile (0, 0): what would an all-hits K loop take?
};

// 16 values of one lane + the 16 of its partner (lane ^ 16) form one MX block: shared E8M0 scale, e2m1 elements.
// Returns the scale byte; q[0..1] = the lane's 16 values packed (element 2i in the low nibble of byte i).
__device__ __forceinline__ unsigned quantize_fp4_block(const float (&v)[16], unsigned (&q)[2]) {
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) amax = fmaxf(amax, fabsf(v[i]));
    {
        // partner = lane ^ 16, i.e. the neighbouring 16-lane row: v_permlane16_swap exchanges odd and even rows at VALU speed
        // ([0] holds rows (0,0,2,2), [1] rows (1,1,3,3)); non-negative floats order as unsigned integers
        const unsigned ab = __float_as_uint(amax);
        const auto r16 = __builtin_amdgcn_permlane16_swap(ab, ab, false, false);
        amax = __uint_as_float(max(r16[0], r16[1]));
    }
    const unsigned sbyte = mx_fp4_scale_byte(amax);
    const float scale = __uint_as_float(sbyte << 23);
    q[0] = q[1] = 0u;
#define AVL_FP4_PAIR(i)                                                                                  \
    q[0] = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(q[0], v[2 * (i)], v[2 * (i) + 1], scale, (i));        \
    q[1] = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(q[1], v[8 + 2 * (i)], v[8 + 2 * (i) + 1], scale, (i))
    AVL_FP4_PAIR(0); AVL_FP4_PAIR(1); AVL_FP4_PAIR(2); AVL_FP4_PAIR(3);
#undef AVL_FP4_PAIR
    return sbyte;
}

// The epilogue of an MX GEMM tile (shared by k_gemm_ring_mx and k_gemm_mx_pipe): bias is in the accumulators; residual (one or two
// planes), ReLU, f16 hi [+ lo] planes, and the FP4 planes + scales the next MX GEMM reads.
template <int IO, int MI>
__device__ __forceinline__ void mx_epilogue(const MxArgs& q, f32x4 (&acc)[MI][4], int nt, int mt, int wm, int wn, int fr, int kq) {
    typedef f16 H;
    typedef typename Half16<H>::v8 v8;
    constexpr int BM = 2 * MI * 16, BN = 256;
    const GemmArgs& p = q.g;
    const int nbase = nt * BN + wn * 64 + kq * 16;
    const bool ncol_ok = nbase + 16 <= p.N;
    const bool rsplit = p.R_lo != nullptr, csplit = p.C_lo != nullptr;
    // The epilogue is VALU-issue bound (eight rounds of ~250 vector instructions per wave while the matrix pipe idles), so:
    // every address is a 32-bit offset from a uniform plane base (validate_gemm bounds the planes to 2 GB), the result is
    // converted to f16 ONCE (the packed vector is stored and read back as hi), every lane stores its own 8 FP4 bytes.
    const unsigned m0 = (unsigned)(mt * BM + wm * (MI * 16) + fr);
    const unsigned sc_r = (unsigned)(nbase >> 8) * (unsigned)q.r_srows * 8u + (unsigned)((nbase >> 5) & 7);
    const unsigned sc_c = (unsigned)(nbase >> 8) * (unsigned)q.c_srows * 8u + (unsigned)((nbase >> 5) & 7);
    // the residual of row mi + 1 is requested before row mi is worked on (its loads could otherwise not move above row
    // mi's stores, and eight dependent load -> compute -> store rounds cost more than the K loop of a short-K tile)
    v8 nr0 = {}, nr1 = {};
    uint2 npk = make_uint2(0u, 0u);
    unsigned nsb = 127u;
    auto load_res = [&](int mi) {
        const unsigned m = m0 + mi * 16;
        if (p.R && (int)m < p.M && ncol_ok) {
            const H* rp = static_cast<const H*>(p.R) + (m * (unsigned)p.ldr + (unsigned)nbase);
            nr0 = *reinterpret_cast<const v8*>(rp);
            nr1 = *reinterpret_cast<const v8*>(rp + 8);
            if (q.Rq) {       // the lo part as FP4: 16 values = 8 bytes, one scale byte for the lane pair's 32-block
                npk = *reinterpret_cast<const uint2*>(q.Rq + (m * (unsigned)q.ldrq + (unsigned)(nbase / 2)));
                nsb = (unsigned char)q.Rs[sc_r + m * 8u];
            }
        }
    };
    load_res(0);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const unsigned m = m0 + mi * 16;
        const bool live = (int)m < p.M && ncol_ok;
        const v8 r0 = nr0, r1 = nr1;
        const uint2 pk = npk;
        const unsigned sb = nsb;
        if (mi + 1 < MI) load_res(mi + 1);
        float v[16];
#pragma unroll
        for (int nj = 0; nj < 4; ++nj)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[nj * 4 + r] = acc[mi][nj][r];
        if (p.R && live) {
            float rl[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) rl[i] = 0.f;
            if (rsplit) {
                const H* rlp = static_cast<const H*>(p.R_lo) + (m * (unsigned)p.ldr + (unsigned)nbase);
                const v8 l0 = *reinterpret_cast<const v8*>(rlp), l1 = *reinterpret_cast<const v8*>(rlp + 8);
#pragma unroll
                for (int i = 0; i < 8; ++i) { rl[i] = (float)l0[i]; rl[8 + i] = (float)l1[i]; }
            } else if (q.Rq) {
                typedef float v2f __attribute__((ext_vector_type(2)));
                const float sc = __uint_as_float(sb << 23);
#define AVL_FP4_DEC(w, sel, o) { const v2f d = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(w, sc, sel); rl[o] = d.x; rl[o + 1] = d.y; }
                AVL_FP4_DEC(pk.x, 0, 0) AVL_FP4_DEC(pk.x, 1, 2) AVL_FP4_DEC(pk.x, 2, 4) AVL_FP4_DEC(pk.x, 3, 6)
                AVL_FP4_DEC(pk.y, 0, 8) AVL_FP4_DEC(pk.y, 1, 10) AVL_FP4_DEC(pk.y, 2, 12) AVL_FP4_DEC(pk.y, 3, 14)
#undef AVL_FP4_DEC
            }
            // hi + lo first (exact in fp32), then the accumulator: same order as k_gemm_ring
#pragma unroll
            for (int i = 0; i < 8; ++i) { v[i] += (float)r0[i] + rl[i]; v[8 + i] += (float)r1[i] + rl[8 + i]; }
        }
        if (p.relu) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = fmaxf(v[i], 0.f);
        }
        v8 h0, h1;
#pragma unroll
        for (int i = 0; i < 8; ++i) { h0[i] = (H)v[i]; h1[i] = (H)v[8 + i]; }
        float hi[16], lo[16];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            hi[i] = (float)h0[i];
            hi[8 + i] = (float)h1[i];
            lo[i] = v[i] - hi[i];
            lo[8 + i] = v[8 + i] - hi[8 + i];
        }
        const unsigned coff = m * (unsigned)p.ldc + (unsigned)nbase;
        if (live) {
            H* cp = static_cast<H*>(p.C) + coff;
            *reinterpret_cast<v8*>(cp) = h0;
            *reinterpret_cast<v8*>(cp + 8) = h1;
        }
        if (csplit) {          // a stored lo plane is f16: its FP4 copy is taken from what it holds
            v8 l0, l1;
#pragma unroll
            for (int i = 0; i < 8; ++i) { l0[i] = (H)lo[i]; l1[i] = (H)lo[8 + i]; }
            if (live) {
                H* cl = static_cast<H*>(p.C_lo) + coff;
                *reinterpret_cast<v8*>(cl) = l0;
                *reinterpret_cast<v8*>(cl + 8) = l1;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) { lo[i] = (float)l0[i]; lo[8 + i] = (float)l1[i]; }
        }
        if constexpr (IO != 0) {
            // every lane takes part in the block exchange; rows past M quantise whatever they hold and store nothing
            // (the partner lane is the same row, so it is dead as well)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                if (q.Cq[pl] == nullptr) continue;
                unsigned pq[2];
                const unsigned sq = quantize_fp4_block(pl == 0 ? hi : lo, pq);
                if (live) {
                    *reinterpret_cast<uint2*>(q.Cq[pl] + (m * (unsigned)q.ldcq + (unsigned)(nbase / 2))) = make_uint2(pq[0], pq[1]);
                    if ((kq & 1) == 0) q.Cs[pl][sc_c + m * 8u] = (char)sq;
                }
            }
        }
    }
}

// MI = 8: 256 x 256 tiles; MI = 4: 128 x 256 tiles for shapes that would otherwise leave half the CUs without a tile.
template <int IO, int MI>
__global__ void __launch_bounds__(512) k_gemm_ring_mx(MxArgs q, int mtiles) {
    typedef f16 H;
    typedef typename Half16<H>::v8 v8;
    typedef int v8i __attribute__((ext_vector_type(8)));
    constexpr int WM = 2, WN = 4, STAGES = 2;
    constexpr int NW = WM * WN, BM = WM * MI * 16, BN = WN * 64;
    constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128;
    constexpr int W_REGION = STAGES * A_BYTES, S_REGION = W_REGION + STAGES * W_BYTES;      // [A tiles][W tiles][scale blocks]
    constexpr int S_BYTES = 4096;                                                            // A scales 2 KB | W scales 2 KB
    constexpr int A_INSTR = BM / 8 / NW, W_INSTR = BN / 8 / NW;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const GemmArgs& p = q.g;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int nwg = gridDim.x;
    int vb;
    {
        const int bid = blockIdx.x, xcd = bid & 7, local = bid >> 3, qq = nwg >> 3, r = nwg & 7;
        vb = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + local;
    }
    const int total = mtiles * p.ntiles;
    const int nmb = p.K / 256;                 // K macro-blocks per tile
    const int nsub = 4 + q.nmx;                // sub-steps per macro-block: 4 f16 + the FP4 corrections
    const unsigned lds_base = lds_addr(lds);

    // ---- producer: SGPR tile base + VGPR offset addressing.  DMA instruction i of this wave covers rows (i * NW + wave) * 8 + srow:
    // the (i, wave) part is uniform and goes into the scalar base, the lane keeps srow * row_bytes + its swizzled chunk
    // ((r & 7) == srow); the weight swizzle key mixes uniform and lane bits, so those offsets stay per instruction.
    const int srow = lane >> 3, schunk = lane & 7;
    const unsigned ct = (unsigned)((schunk ^ srow) << 4);
    const unsigned a_lane16 = (unsigned)srow * (unsigned)(p.lda * 2) + ct, a_laneq = (unsigned)srow * (unsigned)q.ldaq + ct;
    const unsigned a_lane16_2 = (unsigned)srow * (unsigned)(q.lda2 * 2) + ct, a_laneq_2 = (unsigned)srow * (unsigned)q.ldaq2 + ct;
    unsigned w_off16[W_INSTR], w_offq[W_INSTR];
#pragma unroll
    for (int i = 0; i < W_INSTR; ++i) {
        const int r = (i * NW + wave) * 8 + srow;
        const int key = ((r >> 1) & 1) | (((r >> 4) & 3) << 1);
        w_off16[i] = (unsigned)r * (unsigned)(p.K * 2) + ((schunk ^ key) << 4);
        w_offq[i] = (unsigned)r * (unsigned)(p.K / 2) + ((schunk ^ key) << 4);
    }
    int pt = vb, pmb = 0, pj = 0, issued = 0;
    auto issue = [&]() {
        const int nt = pt % p.ntiles, mt = pt / p.ntiles;
        const unsigned abase = lds_base + (issued % STAGES) * A_BYTES + wave * 1024;
        const unsigned wbase = lds_base + W_REGION + (issued % STAGES) * W_BYTES + wave * 1024;
        const bool second = pmb >= q.nmb1;                     // which input this K macro-block comes from
        const int mbl = second ? pmb - q.nmb1 : pmb;
        if (pj < 4) {
            const long long rowb = second ? (long long)q.lda2 * 2 : (long long)p.lda * 2;
            const char* sa = static_cast<const char*>(second ? q.A2 : p.A) + ((long long)mt * BM + wave * 8) * rowb + (long long)(mbl * 4 + pj) * 128;
            const char* sw = static_cast<const char*>(p.W) + (long long)nt * BN * p.K * 2 + (long long)(pmb * 4 + pj) * 128;
            const unsigned al = second ? a_lane16_2 : a_lane16;
#pragma unroll
            for (int i = 0; i < A_INSTR; ++i) glds16_saddr(sa + (long long)i * (NW * 8) * rowb, al, abase + i * NW * 1024);
#pragma unroll
            for (int i = 0; i < W_INSTR; ++i) glds16_saddr(sw, w_off16[i], wbase + i * NW * 1024);
        } else {
            const int t = pj - 4;
            const long long rowb = second ? q.ldaq2 : q.ldaq;
            const char* sa = (second ? q.Aq2[t] : q.Aq[t]) + ((long long)mt * BM + wave * 8) * rowb + (long long)mbl * 128;
            const char* sw = q.Wq[t] + (long long)nt * BN * (p.K / 2) + (long long)pmb * 128;
            const unsigned al = second ? a_laneq_2 : a_laneq;
#pragma unroll
            for (int i = 0; i < A_INSTR; ++i) glds16_saddr(sa + (long long)i * (NW * 8) * rowb, al, abase + i * NW * 1024);
#pragma unroll
            for (int i = 0; i < W_INSTR; ++i) glds16_saddr(sw, w_offq[i], wbase + i * NW * 1024);
            // scales: BM x 8 bytes for the activation rows (waves 0, 1: one KB each), 2 KB for the weight rows (waves 2, 3)
            if (wave < 4 && (wave >= 2 || wave * 128 < BM)) {
                const char* as = second ? q.As2[t] + ((long long)mbl * q.a_srows2 + (long long)mt * BM) * 8
                                        : q.As[t] + ((long long)mbl * q.a_srows + (long long)mt * BM) * 8;
                const char* ss = wave < 2 ? as + wave * 1024
                                          : q.Ws[t] + ((long long)pmb * q.w_srows + (long long)nt * BN) * 8 + (wave - 2) * 1024;
                glds16_saddr(ss, (unsigned)lane * 16u, lds_base + S_REGION + (issued % STAGES) * S_BYTES + wave * 1024);
            }
        }
        ++issued;
        if (++pj == nsub) {
            pj = 0;
            if (++pmb == nmb) { pmb = 0; pt += nwg; }
        }
    };
    if (pt < total) issue();

    // ---- consumer
    // fragment addressing: row r of a tile sits at r * 128 with its 16-byte chunks XOR-swizzled by key(r); the keys and the
    // scale addresses (r * 8) are recomputed from the row offsets instead of being kept in registers (the 128
    // accumulators leave little room: every array here is sized to stay clear of spills)
    const int fr = lane & 15, kq = lane >> 4;
    const int a_row0 = wm * (MI * 16) + fr;                       // + 16 mi
    int w_off[4], w_key[4];
#pragma unroll
    for (int nj = 0; nj < 4; ++nj) {
        const int row = wn * 64 + (fr >> 2) * 16 + nj * 4 + (fr & 3);
        w_off[nj] = W_REGION + row * 128;
        w_key[nj] = ((row >> 1) & 1) | (((row >> 4) & 3) << 1);
    }
    const int a_key = a_row0 & 7;                                  // (a_row0 + 16 mi) & 7 is the same for every mi
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

    int g = 0;
    static_assert(NW == 8, "the DMA stagger assumes waves w and w + 4 on one SIMD");
    const int late = (wave >= 4 || q.stagger == 0) ? 0 : 1;       // the K half in front of which this wave issues its DMAs
    for (int t = vb; t < total; t += nwg) {
        const int nt = t % p.ntiles, mt = t / p.ntiles;
        // two plain loops per macro-block (f16 sub-steps, then FP4 sub-steps) rather than one loop with an if / else on the
        // sub-step kind: the accumulators then have ONE definition chain (an if / else made hipcc keep two copies of them)
        for (int mb = 0; mb < nmb; ++mb) {
            for (int j = 0; j < 4; ++j, ++g) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // sub-step g has landed (this wave's share)
                __builtin_amdgcn_s_barrier();
                // sub-step g+1 -> the slots read during step g-1.  Waves w and w + 4 share a SIMD: one issues its LDS-DMAs (100+
                // cycles of issue each) at the head of the step, the other after its first K half, so that the SIMD's matrix pipe
                // has one wave's MFMAs to run while the other is busy issuing (GEMMs 3.53 -> 3.40 ms per frame; spreading the
                // instructions two at a time between MFMA groups instead: 4.0 ms)
                const char* abase = lds + (g % STAGES) * A_BYTES + a_row0 * 128;
                const char* wbase = lds + (g % STAGES) * W_BYTES;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    if (kk == late && pt < total) issue();
                    v8 wf[4];
#pragma unroll
                    for (int nj = 0; nj < 4; ++nj) wf[nj] = *reinterpret_cast<const v8*>(wbase + w_off[nj] + (((kk * 4 + kq) ^ w_key[nj]) << 4));
                    const int achunk = ((kk * 4 + kq) ^ a_key) << 4;
                    v8 af = *reinterpret_cast<const v8*>(abase + achunk);
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        const v8 cur = af;
                        if (mi + 1 < MI) af = *reinterpret_cast<const v8*>(abase + (mi + 1) * 2048 + achunk);
#pragma unroll
                        for (int nj = 0; nj < 4; ++nj) acc[mi][nj] = Half16<H>::mfma(wf[nj], cur, acc[mi][nj]);
                    }
                }
            }
            for (int u = 0; u < q.nmx; ++u, ++g) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                const char* abase = lds + (g % STAGES) * A_BYTES + a_row0 * 128;
                const char* wbase = lds + (g % STAGES) * W_BYTES;
                const char* sbase = lds + (g % STAGES) * S_BYTES + S_REGION;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    if (kk == late && pt < total) issue();
                    int4 wf[4];
                    int ws[4];
#pragma unroll
                    for (int nj = 0; nj < 4; ++nj) {
                        wf[nj] = *reinterpret_cast<const int4*>(wbase + w_off[nj] + (((kk * 4 + kq) ^ w_key[nj]) << 4));
                        {   // weight scales as network.permute_w_scales lays them out: [16-row block][row & 3][kq][nj][kk]
                            const uint2 w8 = *reinterpret_cast<const uint2*>(sbase + 2048 + (wn * 4 + (fr >> 2)) * 128 + ((fr & 3) * 4 + kq) * 8);
                            ws[nj] = (int)(((nj < 2 ? w8.x : w8.y) >> (8 * ((nj & 1) * 2 + kk))) & 0xffu);
                        }
                    }
                    const int achunk = ((kk * 4 + kq) ^ a_key) << 4;
                    int4 af = *reinterpret_cast<const int4*>(abase + achunk);
                    unsigned asw = *reinterpret_cast<const unsigned*>(sbase + a_row0 * 8 + kk * 4);
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        const int4 cur = af;
                        const int as = (int)((asw >> (8 * kq)) & 0xffu);
                        if (mi + 1 < MI) {
                            af = *reinterpret_cast<const int4*>(abase + (mi + 1) * 2048 + achunk);
                            asw = *reinterpret_cast<const unsigned*>(sbase + (a_row0 + (mi + 1) * 16) * 8 + kk * 4);
                        }
                        const v8i xa = {cur.x, cur.y, cur.z, cur.w, 0, 0, 0, 0};
#pragma unroll
                        for (int nj = 0; nj < 4; ++nj) {
                            const v8i wa = {wf[nj].x, wf[nj].y, wf[nj].z, wf[nj].w, 0, 0, 0, 0};
                            acc[mi][nj] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wa, xa, acc[mi][nj], 4, 4, 0, ws[nj], 0, as);
                        }
                    }
                }
            }
        }
        mx_epilogue<IO, MI>(q, acc, nt, mt, wm, wn, fr, kq);
        if (t + nwg < total) init_acc(t + nwg);
    }
}

// ---------------------------------------------------------------------------------------------
// k_gemm_mx_pipe (round 3): the same tile, ring (two stages), LDS image, DMA geometry and epilogue as k_gemm_ring_mx, but the K
// loop is ONE software-pipelined instruction stream per wave instead of "barrier, read, multiply" rounds.
//
// What was wrong with the rounds (ISA of k_gemm_ring_mx, round 2): after the barrier every wave read 6 fragments, waited, issued
// 8 MFMAs, read 2 more, waited, 8 MFMAs ...: a read -> wait -> multiply chain with the LDS latency exposed every 128 cycles and
// the whole read burst of all 8 waves exposed at every step boundary.
//
// The stream (MI = 8: 16 "steps" of 4 MFMAs per sub-step; step s multiplies activation fragment mi = s % 8 of K half kk = s / 8):
//   * fragments are read D = 3 steps before their MFMAs, one activation fragment per step (a rolling window of four), two sets of
//     weight fragments (the set of the second K half is read during steps 0-3 of the first);
//   * the step boundaries are moved INTO the stream, and there are two per sub-step, one per LDS slot kind ("events"):
//       EW, in front of step 5:  s_waitcnt lgkmcnt(1|2); s_barrier -- every wave has read its last WEIGHT fragment of slot g % 2:
//                                the weight tile of sub-step g + 2 may overwrite it;
//       EV, in front of step 13: s_waitcnt vmcnt(4) lgkmcnt(0); s_barrier -- every wave's DMAs for sub-step g + 1 have landed (only
//                                the 4 instructions of a weight burst sent since may still fly; vmcnt(0) when none was sent) and
//                                every wave has read its last ACTIVATION fragment of slot g % 2.  Right behind it the wave reads
//                                the first fragments of sub-step g + 1 from the other slot -- while it still has the 12 MFMAs of
//                                steps 13 .. 15 in registers to issue: the matrix pipe never waits for a barrier + a cold LDS burst;
//   * one sub-step's DMA is therefore two bursts of 4 instructions per wave (weights; activations + the scale blocks of an FP4
//     sub-step), and the bursts of the two waves of a SIMD are COMPLEMENTARY: waves 4-7 send theirs right behind EW / EV, waves 0-3
//     one event later (right in front of EV / of the next EW), so that one wave of every SIMD issues MFMAs while the other pays
//     the ~150 cycles per LDS-DMA instruction (s_memtime stamps, AVL_MX_PROBE=3: a 4-instruction burst takes ~600 cycles: the 16
//     instructions of the four waves that burst together queue in the CU's one vector-memory path).
// What bounds it now (stamps): 64 KB per sub-step through that path at ~27 B/clk = ~2400 cycles, against 2048 cycles of MFMA
// issue per SIMD: the 256 x 256 tile is L2 -> LDS bound on this chip; the stream runs at ~3500 cycles per sub-step (was ~4000).
// RAW / WAR on LDS (MI355X_MICROARCH.md, "Read a staged buffer one phase AFTER the wait that retires it"): a slot is read only
// behind the barrier that follows every wave's vmcnt wait for it; it is overwritten only by DMAs issued behind the barrier that
// follows every wave's lgkmcnt wait with all its reads of that slot already issued.  The two hand counts: vmcnt(4) -- younger
// operations (compiler-visible loads / stores, scratch) only make it wait for more; lgkmcnt(1|2) -- at least that many LDS reads
// are issued behind the last weight read (step 4: one fragment, plus its scale byte in an FP4 sub-step).
// Sub-step kinds (f16 / FP4) alternate inside the stream; the fragments carried across a boundary are plain 128-bit values.
// MI = 4 (128 x 256 tiles): one event (both slots) in front of step 5 of 8, vmcnt(0).
// PROBE (timing experiments, results are garbage): 1 = no DMA after the prologue, 2 = no MFMAs, 3 = s_memtime stamps around the
// events and the DMA bursts, per-wave sums -> q.dbg (each stamp drains the LDS queue: read the SHARES, not the totals),
// 4 = no weight-fragment LDS reads after the first two sub-steps (a third of the LDS read bytes), 5 = every other activation
// fragment not read (another third), 6 = every other DMA instruction not sent (half the staging bytes: throughput- or latency-bound?)
template <int IO, int MI, int LATE, int PROBE = 0, int SPREAD = 0>
__global__ void __launch_bounds__(512) k_gemm_mx_pipe(MxArgs q, int mtiles) {
    typedef f16 H;
    typedef typename Half16<H>::v8 v8;
    typedef int v8i __attribute__((ext_vector_type(8)));
    constexpr int WM = 2, WN = 4, STAGES = 2;
    constexpr int NW = WM * WN, BM = WM * MI * 16, BN = WN * 64;
    constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128;
    constexpr int W_REGION = STAGES * A_BYTES, S_REGION = W_REGION + STAGES * W_BYTES;
    constexpr int S_BYTES = 4096;
    constexpr int A_INSTR = BM / 8 / NW, W_INSTR = BN / 8 / NW;
    constexpr int NS = 2 * MI, D = 3, EV = NS - D;              // steps per sub-step; fragments are read D steps ahead; the (main) event sits in front of step EV
    constexpr int EW = (MI == 8 && SPREAD != 5) ? 5 : EV;        // ... and the weight-slot event (MI = 4: one event for both slots; SPREAD = 5: experiment, one event for MI = 8 too)
    constexpr bool SPR = SPREAD == 1 || SPREAD == 2, PIN = SPREAD != 0;     // experiments: DMA schedules 1 / 2; 3 = the release schedule with pinned MFMAs
    constexpr bool ONE = SPREAD == 5;       // experiment: ONE event per sub-step (step 13, both slots); waves 4-7 send both tiles right behind it, waves 0-3 at step 5
    constexpr bool MID = SPREAD == 4;       // experiment: waves 0-3 send their bursts in the MIDDLE between two events (steps 1 / 9) instead of in front of the next one
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const GemmArgs& p = q.g;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int nwg = gridDim.x;
    int vb;
    {
        const int bid = blockIdx.x, xcd = bid & 7, local = bid >> 3, qq = nwg >> 3, r = nwg & 7;
        vb = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + local;
    }
    const int ntiles = p.ntiles, nmx = q.nmx;
    const int total = mtiles * ntiles;
    const int nmb = p.K / 256;
    const int nsub = 4 + nmx;
    const unsigned lds_base = lds_addr(lds);

    // ---- producer: SGPR base + VGPR lane offset addressing.  DMA instruction i of this wave covers rows (i * 8 + wave) * 8 + srow
    // of a tile: the (i, wave) part is uniform and goes into the scalar base, the lane keeps srow * row_bytes + its swizzled chunk
    // ((r & 7) == srow for activation rows).  For weight rows the swizzle key ((r >> 1) & 1) | (((r >> 4) & 3) << 1) does not depend
    // on i either (i moves r by 64): ONE lane offset per plane kind.
    // The hot path of issue() works on a handful of scalars (cur_*: the planes of the input the current K macro-block comes from, at
    // this tile and wave) that are re-derived from the argument struct only where they change -- at a tile switch and where the K
    // loop passes from the first input to the second (conv3 + downsample): no scalar (kernarg) loads inside the stream, where
    // their out-of-order lgkmcnt would force `s_waitcnt lgkmcnt(0)` in front of every LDS fragment use.
    // The lane offsets are NOT kept in registers (the K loop has none to spare: hipcc spilled them to scratch, and a scratch reload is a
    // vector-memory operation in the middle of the hand-counted DMA stream): every burst re-derives them from the lane id, ~8 VALU
    // instructions beside 64 MFMAs.  The lane id comes out of a volatile asm so that the arithmetic is not hoisted back out.
    auto lane_now = [&]() __attribute__((always_inline)) {
        unsigned l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return l;
    };
    int pt = vb, pmb = 0, pj = 0, issued = 0, pmbl = 0;        // tile, K macro-block (global / inside its input), sub-step being issued next
    long long cur_rowb16 = 0, cur_rowbq = 0, cur_asrows = 0;
    const char *cur_a16 = nullptr, *cur_aq = nullptr, *cur_as = nullptr, *cur_w16 = nullptr, *cur_wq = nullptr, *cur_ws = nullptr;
    long long cur_aps = 0, cur_wps = 0;                          // distance from correction plane 0 to plane 1 (activations / weights)
    int p_mt = 0;
    auto set_input = [&](bool second) __attribute__((always_inline)) {     // rare: tile switch, first -> second input
        const long long rows0 = (long long)p_mt * BM + wave * 8;
        if (!second) {
            cur_rowb16 = (long long)p.lda * 2; cur_rowbq = q.ldaq; cur_asrows = q.a_srows;
            cur_a16 = static_cast<const char*>(p.A) + rows0 * cur_rowb16;
            cur_aq = q.Aq[0] + rows0 * cur_rowbq;
            cur_as = q.As[0] + (long long)p_mt * BM * 8;
            cur_aps = q.Aq[1] - q.Aq[0];
        } else {
            cur_rowb16 = (long long)q.lda2 * 2; cur_rowbq = q.ldaq2; cur_asrows = q.a_srows2;
            cur_a16 = static_cast<const char*>(q.A2) + rows0 * cur_rowb16;
            cur_aq = q.Aq2[0] + rows0 * cur_rowbq;
            cur_as = q.As2[0] + (long long)p_mt * BM * 8;
            cur_aps = q.Aq2[1] - q.Aq2[0];
        }
        pmbl = 0;
    };
    auto set_tile = [&]() __attribute__((always_inline)) {                  // rare: once per tile
        int nt_ = pt % ntiles;
        p_mt = pt / ntiles;
        if (PROBE == 3 && q.same_tile) { nt_ = 0; p_mt = 0; }   // probe (results garbage): every workgroup streams THE SAME tiles = all L2 hits
        cur_w16 = static_cast<const char*>(p.W) + (long long)nt_ * BN * p.K * 2;
        cur_wq = q.Wq[0] + (long long)nt_ * BN * (p.K / 2);
        cur_ws = q.Ws[0] + (long long)nt_ * BN * 8;
        cur_wps = q.Wq[1] - q.Wq[0];
        set_input(false);
    };
    // One sub-step's DMA is issued in two bursts (see the events below): the weight tile, then the activation tile (+ the scale
    // blocks of an FP4 sub-step, so that a weight burst is W_INSTR instructions for EVERY wave: the hand count at the event).
    // piece i of a burst = its i-th DMA instruction (1 KB per wave); `i < 0`: the whole burst
    auto issue_w = [&](int only = -1) __attribute__((always_inline)) {
        const unsigned wbase = lds_base + W_REGION + (issued & 1) * W_BYTES + wave * 1024;
        // weight row of instruction i: r = (i * 8 + wave) * 8 + srow; its swizzle key ((r >> 1) & 1) | (((r >> 4) & 3) << 1) =
        // ((srow >> 1) & 1) | (((wave >> 1) & 3) << 1) does not depend on i
        const unsigned l = lane_now(), srow = l >> 3, cw = ((l & 7u) ^ (((srow >> 1) & 1u) | (((unsigned)(wave >> 1) & 3u) << 1))) << 4;
        const unsigned wr0 = (unsigned)wave * 8u + srow;
        if (pj < 4) {
            const char* sw = cur_w16 + (long long)(pmb * 4 + pj) * 128;
            const unsigned wl = wr0 * (unsigned)(p.K * 2) + cw;
#pragma unroll
            for (int i = 0; i < W_INSTR; ++i)
                if ((only < 0 || only == i) && !(PROBE == 6 && (i & 1))) glds16_saddr(sw + (long long)i * (NW * 8) * p.K * 2, wl, wbase + i * NW * 1024);
        } else {
            const char* sw = cur_wq + (pj == 4 ? 0 : cur_wps) + (long long)pmb * 128;
            const unsigned wl = wr0 * (unsigned)(p.K / 2) + cw;
#pragma unroll
            for (int i = 0; i < W_INSTR; ++i)
                if ((only < 0 || only == i) && !(PROBE == 6 && (i & 1))) glds16_saddr(sw + (long long)i * (NW * 8) * (p.K / 2), wl, wbase + i * NW * 1024);
        }
    };
    // (the last piece also brings the scale blocks of an FP4 sub-step and moves the producer on to the next sub-step)
    auto issue_a = [&](int only = -1) __attribute__((always_inline)) {
        const unsigned abase = lds_base + (issued & 1) * A_BYTES + wave * 1024;
        const unsigned l = lane_now(), srow = l >> 3, ct = ((l & 7u) ^ srow) << 4;        // activation rows: key = r & 7 = srow
        if (pj < 4) {
            const char* sa = cur_a16 + (long long)(pmbl * 4 + pj) * 128;
            const unsigned al = srow * (unsigned)cur_rowb16 + ct;
#pragma unroll
            for (int i = 0; i < A_INSTR; ++i)
                if ((only < 0 || only == i) && !(PROBE == 6 && (i & 1))) glds16_saddr(sa + (long long)i * (NW * 8) * cur_rowb16, al, abase + i * NW * 1024);
        } else {
            const long long ta = pj == 4 ? 0 : cur_aps, tw = pj == 4 ? 0 : cur_wps;     // which correction pass
            const char* sa = cur_aq + ta + (long long)pmbl * 128;
            const unsigned a_lq = srow * (unsigned)cur_rowbq + ct;
#pragma unroll
            for (int i = 0; i < A_INSTR; ++i)
                if ((only < 0 || only == i) && !(PROBE == 6 && (i & 1))) glds16_saddr(sa + (long long)i * (NW * 8) * cur_rowbq, a_lq, abase + i * NW * 1024);
            // scales: BM x 8 bytes for the activation rows (waves 0, 1: one KB each), 2 KB for the weight rows (waves 2, 3)
            if ((only < 0 || only == A_INSTR - 1) && wave < 4 && (wave >= 2 || wave * 128 < BM)) {
                const char* ss = wave < 2 ? cur_as + ta + (long long)pmbl * cur_asrows * 8 + wave * 1024
                                          : cur_ws + tw + (long long)pmb * q.w_srows * 8 + (wave - 2) * 1024;
                glds16_saddr(ss, l * 16u, lds_base + S_REGION + (issued & 1) * S_BYTES + wave * 1024);
            }
        }
        if (only >= 0 && only != A_INSTR - 1) return;
        ++issued;
        if (++pj == nsub) {
            pj = 0;
            ++pmbl;
            if (++pmb == nmb) {
                pmb = 0;
                pt += nwg;
                if (pt < total) set_tile();
            } else if (pmb == q.nmb1) set_input(true);
        }
    };
    if (pt < total) set_tile();
    static_assert(NW == 8, "the DMA stagger assumes waves w and w + 4 on one SIMD");
    const bool early = wave >= 4 || q.stagger == 0;             // issues its bursts right behind the events; the others LATE steps later
    if (pt < total) { issue_w(); issue_a(); }                    // sub-step 0 -> slots 0
    if (pt < total) { issue_w(); issue_a(); }                    // sub-step 1 -> slots 1

    // ---- consumer addressing: byte offsets inside a slot (the slot base is uniform and added per sub-step)
    const int fr = lane & 15, kq = lane >> 4;
    const int a_row0 = wm * (MI * 16) + fr;
    // weight row of n-tile nj: wrow0 + 4 nj; its swizzle key ((row >> 1) & 1) | (((row >> 4) & 3) << 1) does not depend on nj, and the
    // second K half is chunk ^ 4: one base register per operand, the rest is immediate offsets and one XOR
    const int wrow0 = wn * 64 + (fr >> 2) * 16 + (fr & 3);
    const int wkey_r = ((wrow0 >> 1) & 1) | (((wrow0 >> 4) & 3) << 1);
    const unsigned a_v0 = (unsigned)(a_row0 * 128 + ((kq ^ (a_row0 & 7)) << 4));
    const unsigned w_v0 = (unsigned)(W_REGION + wrow0 * 128 + ((kq ^ wkey_r) << 4));
    const unsigned sa_v = (unsigned)(S_REGION + a_row0 * 8 + kq);
    const unsigned sw_v = (unsigned)(S_REGION + 2048 + (wn * 4 + (fr >> 2)) * 128 + ((fr & 3) * 4 + kq) * 8);

    f32x4 acc[MI][4];
    auto init_acc = [&](int t) {
        const int nb = (t % ntiles) * BN + wn * 64 + kq * 16;
#pragma unroll
        for (int nj = 0; nj < 4; ++nj) {
            const float4 b = *reinterpret_cast<const float4*>(p.bias + nb + 4 * nj);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) acc[mi][nj] = f32x4{b.x, b.y, b.z, b.w};
        }
    };
    if (vb < total) init_acc(vb);

    // fragments: step s of a sub-step uses Af[s] (and its scale byte As[s]) with the weight set Wf[s / MI]
    int4 Af[NS], Wf[2][4];
    unsigned As[NS];
    uint2 Wsc = make_uint2(0u, 0u), Wsn = make_uint2(0u, 0u);    // weight scales of this / the next FP4 sub-step: byte 2 nj + kk
    auto rdA = [&](const char* slot, int kk, int mi) { return *reinterpret_cast<const int4*>(slot + (a_v0 ^ (unsigned)(kk * 64)) + mi * 2048); };
    auto rdW = [&](const char* slot, int kk, int nj) { return *reinterpret_cast<const int4*>(slot + (w_v0 ^ (unsigned)(kk * 64)) + nj * 512); };
    // activation scales: the array keeps [row][8 K-blocks]; this lane's block of a 128-wide MFMA is kq + 4 kk: a byte read
    auto rdAs = [&](const char* slot, int kk, int mi) { return (unsigned)*reinterpret_cast<const unsigned char*>(slot + sa_v + mi * 128 + kk * 4); };
    // weight scales: the host lays them out per 16-row block as [row & 3][kq][n-tile nj][kk] (network.permute_w_scales), so the
    // eight bytes a lane needs in a sub-step are ONE 8-byte read and the MFMA picks its byte with op_sel: no VALU, two registers
    auto rdWs = [&](const char* slot) { return *reinterpret_cast<const uint2*>(slot + sw_v); };

    int g = 0;                                                   // sub-steps consumed so far (ring position)
    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;      // PROBE 3: cycles in [0] sub-steps [1] EW wait [2] W burst [3] EV wait [4] A burst
    auto stamp = [&]() __attribute__((always_inline)) {
        unsigned long long t;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        return t;
    };
    bool w_sent = false;                                         // this wave has issued the weight burst of the current sub-step
    // KIND: 0 = f16 sub-step (K = 64), 1 = FP4 sub-step (K = 256, block scales); next_q: the sub-step that follows is an FP4 one
    // (run-time, uniform: it only decides whether scales are read ahead with the fragments -- TWO loop bodies in all, as in
    // k_gemm_ring_mx: with a body per (kind, next kind) pair hipcc moved the accumulators between two register sets).
    // One step = 4 MFMAs; S is a compile-time constant (op_sel immediates, static register indices).
    auto step = [&](auto kind_c, auto s_c, const bool next_q) __attribute__((always_inline)) {
        constexpr int KIND = decltype(kind_c)::value, S = decltype(s_c)::value;
        constexpr int kk = S / MI, mi = S % MI;
        const char* ab = lds + (g & 1) * A_BYTES;                // a_v0 / w_v0 / s*_v carry the region offsets
        const char* wb = lds + (g & 1) * W_BYTES;
        const char* sb = lds + (g & 1) * S_BYTES;
        const char* abn = lds + ((g + 1) & 1) * A_BYTES;
        const char* wbn = lds + ((g + 1) & 1) * W_BYTES;
        const char* sbn = lds + ((g + 1) & 1) * S_BYTES;
        auto readA = [&](int t) {                                // fragment (and scale byte) of step t of THIS sub-step
            Af[t] = rdA(ab, t / MI, t % MI);
            if (KIND == 1) As[t] = rdAs(sb, t / MI, t % MI);
        };
        auto readAn = [&](int t) {                               // ... of the NEXT sub-step (its slot, its kind)
            Af[t] = rdA(abn, t / MI, t % MI);
            As[t] = rdAs(sbn, t / MI, t % MI);                   // (unconditional: see Wsn below)
        };
        if (PROBE == 3 && S == 0) { const unsigned long long t = stamp(); if (tlast) tsum[0] += t - tlast; tlast = t; tsum[5] += 1; }
        if (EW != EV && S == EW) {
            // complementary bursts: waves 0-3 send the ACTIVATION tile of the sub-step after next here, one event after waves 4-7
            // did (its slot was released at the previous main event; not in the very first sub-step: that tile is already in flight)
            if (!SPR && !MID && LATE == 0 && !early && g > 0 && PROBE != 1 && pt < total) {
                unsigned long long t1 = 0;
                if (PROBE == 3) t1 = stamp();
                issue_a();
                if (PROBE == 3) tsum[4] += stamp() - t1;
            }
            unsigned long long t0 = 0;
            if (PROBE == 3) t0 = stamp();
            // weight-slot event: the last weight fragment of this slot was read at step 3, in front of at least one younger LDS
            // read (step 4: one fragment, and its scale byte in an FP4 sub-step), so lgkmcnt(1 / 2) retires every weight read
            if (KIND == 0) asm volatile("s_waitcnt lgkmcnt(1)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(2)\n\ts_barrier" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            unsigned long long t1 = 0;
            if (PROBE == 3) { t1 = stamp(); tsum[1] += t1 - t0; }
            if (!SPR && PROBE != 1 && early && pt < total) { issue_w(); w_sent = true; }
            if (PROBE == 3) tsum[2] += stamp() - t1;
        }
        if (!SPR && EW != EV && LATE != 0 && !early && S == EW + LATE && PROBE != 1 && pt < total) {
            unsigned long long t1 = 0;
            if (PROBE == 3) t1 = stamp();
            issue_w();
            w_sent = true;
            if (PROBE == 3) tsum[2] += stamp() - t1;
        }
        if (S == EV) {
            // every read of this slot was issued >= 2 steps ago.  DMA: the only operations this wave may still have in flight
            // are the W_INSTR of the weight burst it issued a few steps ago (for the sub-step after next); everything older
            // -- both tiles of the next sub-step, epilogue stores -- is waited for
            if (!SPR && !MID && EW != EV && LATE == 0 && !early && PROBE != 1 && pt < total) {
                unsigned long long t1 = 0;
                if (PROBE == 3) t1 = stamp();
                issue_w();
                w_sent = true;
                if (PROBE == 3) tsum[2] += stamp() - t1;
            }
            // (no weight burst once the work has run out: then nothing may stay in flight)
            unsigned long long t0 = 0;
            if (PROBE == 3) t0 = stamp();
            if (EW != EV && w_sent) asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(PROBE == 6 ? W_INSTR / 2 : W_INSTR) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            w_sent = false;
            __builtin_amdgcn_sched_barrier(0);
            unsigned long long t1 = 0;
            if (PROBE == 3) { t1 = stamp(); tsum[3] += t1 - t0; }
            if (!SPR && PROBE != 1 && early && pt < total) { if (EW == EV) issue_w(); issue_a(); }
            if (PROBE == 3) tsum[4] += stamp() - t1;
            if (PROBE != 4 || g < 2) {
#pragma unroll
                for (int nj = 0; nj < 4; ++nj) Wf[0][nj] = rdW(wbn, 0, nj);
            }
            // Scales are read ahead whether or not the next sub-step is an FP4 one (then they are whatever the slot holds and nothing
            // uses them): a read under the run-time flag would keep Wsn / Wsc / As[0 .. 2] LIVE through every f16 sub-step (seven
            // registers hipcc spilled to scratch and reloaded inside the hand-counted stream); four LDS reads per sub-step are free.
            Wsn = rdWs(sbn);
            readAn(0);
        }
        if (!SPR && !ONE && PROBE != 1 && !early && (EW == EV || LATE != 0) && S == EV + LATE && pt < total) {
            unsigned long long t1 = 0;
            if (PROBE == 3) t1 = stamp();
            if (EW == EV) issue_w();
            issue_a();
            if (PROBE == 3) tsum[4] += stamp() - t1;
        }
        if constexpr (ONE) {
            if constexpr (S == 5) { if (!early && g > 0 && pt < total) { issue_w(); issue_a(); } }
        }
        if constexpr (MID) {
            // whole bursts, four steps behind those of waves 4-7 (which send right behind the events at steps 5 / 13): the weight tile of
            // sub-step g + 2 at step 9 (its slot was released at EW, step 5), the activation tile of sub-step g + 1 at step 1 (slot
            // released at the previous EV).  Same order per wave (weights, activations), so the counts at the events stand.
            static_assert(MI == 8 && EV == 13, "MID: 16-step sub-steps only");
            if constexpr (S == 9) { if (!early && pt < total) { issue_w(); w_sent = true; } }
            if constexpr (S == 1) { if (!early && g > 0 && pt < total) issue_a(); }
        }
        if constexpr (SPR) {
            // SPREAD: no bursts.  Every step, one wave of each SIMD sends ONE DMA instruction: waves 4-7 on the odd steps, waves 0-3
            // on the even ones -- the weight tile of sub-step g + 2 in steps 5 .. 12 (behind EW, which releases its slot), its
            // activation tile in steps 13 .. 15 (behind EV) and 0 .. 4 of the next sub-step.  Four 1 KB instructions per step and CU
            // instead of sixteen at an event: the CU's vector-memory path takes one every ~37 cycles, so a burst made each of its
            // waves wait ~600 cycles at the issue (stamps) while this keeps the queue short.  The counts at the events stand: at EV the
            // four youngest operations of a wave are exactly its weight pieces of steps 5 .. 12, everything older (the activation
            // pieces of the tile that is certified there included, the last of them issued at step 3 / 4) is waited for.
            static_assert(MI == 8 && EW == 5 && EV == 13, "SPREAD: 16-step sub-steps only");
            if constexpr (SPREAD == 1) {
                if constexpr (S >= 5 && S <= 12) {
                    if (early == ((S & 1) == 1) && pt < total) { issue_w((S - 5) / 2); w_sent = true; }
                }
                constexpr int pa_e = S == 13 ? 0 : S == 15 ? 1 : S == 1 ? 2 : S == 3 ? 3 : -1;
                constexpr int pa_l = S == 14 ? 0 : S == 0 ? 1 : S == 2 ? 2 : S == 4 ? 3 : -1;
                // (g == 0: the prologue has sent sub-steps 0 and 1 whole)
                if constexpr (pa_e >= 0) { if (early && pt < total && (S >= 13 || g > 0)) issue_a(pa_e); }
                if constexpr (pa_l >= 0) { if (!early && pt < total && (S >= 13 || g > 0)) issue_a(pa_l); }
            } else {
                // SPREAD = 2: half bursts (two instructions) at four points of the sub-step; waves 4-7 at steps 5 / 9 (weights) and
                // 13 / 1 (activations), waves 0-3 two steps later
                constexpr int pw_e = S == 5 ? 0 : S == 9 ? 2 : -1, pw_l = S == 7 ? 0 : S == 11 ? 2 : -1;
                constexpr int pa_e = S == 13 ? 0 : S == 1 ? 2 : -1, pa_l = S == 15 ? 0 : S == 3 ? 2 : -1;
                if constexpr (pw_e >= 0) { if (early && pt < total) { issue_w(pw_e); issue_w(pw_e + 1); w_sent = true; } }
                if constexpr (pw_l >= 0) { if (!early && pt < total) { issue_w(pw_l); issue_w(pw_l + 1); w_sent = true; } }
                if constexpr (pa_e >= 0) { if (early && pt < total && (S >= 13 || g > 0)) { issue_a(pa_e); issue_a(pa_e + 1); } }
                if constexpr (pa_l >= 0) { if (!early && pt < total && (S >= 13 || g > 0)) { issue_a(pa_l); issue_a(pa_l + 1); } }
            }
        }
        // activation fragments: D steps ahead, one per step -- the last one of the slot at step EV - 1, the first of the next slot
        // right behind the event
        if (S + D < NS && (PROBE != 5 || g < 2 || ((S + D) & 1) == 0)) readA(S + D);
        if (S >= EV && S + 1 < NS) readAn(S - EV + 1);
        // weight fragments of the second K half
        if constexpr (MI == 8 && S < 4) if (PROBE != 4 || g < 2) Wf[1][S] = rdW(wb, 1, S);
        if constexpr (MI == 4 && S < 2) { Wf[1][2 * S] = rdW(wb, 1, 2 * S); Wf[1][2 * S + 1] = rdW(wb, 1, 2 * S + 1); }
        if (KIND == 0) {
            const v8 af = __builtin_bit_cast(v8, Af[S]);
#pragma unroll
            for (int nj = 0; nj < 4; ++nj) {
                if (PROBE == 2) asm volatile("" ::"v"(Wf[kk][nj].x), "v"(Wf[kk][nj].w), "v"(Af[S].x), "v"(Af[S].w));
                else acc[mi][nj] = Half16<H>::mfma(__builtin_bit_cast(v8, Wf[kk][nj]), af, acc[mi][nj]);
            }
        } else {
            const v8i xa = {Af[S].x, Af[S].y, Af[S].z, Af[S].w, 0, 0, 0, 0};
            const int as = (int)As[S];
#define AVL_QMFMA(NJ)                                                                                                              \
    do {                                                                                                                           \
        const v8i wa = {Wf[kk][NJ].x, Wf[kk][NJ].y, Wf[kk][NJ].z, Wf[kk][NJ].w, 0, 0, 0, 0};                                       \
        const int ws = (int)((NJ) < 2 ? Wsc.x : Wsc.y);                                                                            \
        if (PROBE == 2) asm volatile("" ::"v"(Wf[kk][NJ].x), "v"(Wf[kk][NJ].w), "v"(Af[S].x), "v"(Af[S].w), "v"(ws), "v"(as));     \
        else acc[mi][NJ] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wa, xa, acc[mi][NJ], 4, 4, ((NJ) & 1) * 2 + kk, ws, 0, as); \
    } while (0)
            AVL_QMFMA(0); AVL_QMFMA(1); AVL_QMFMA(2); AVL_QMFMA(3);
#undef AVL_QMFMA
        }
        if constexpr (PIN) {
            // with a (uniform) branch in every step hipcc sinks the MFMAs of all sixteen steps below the last branch of the
            // sub-step (one block of 64 MFMAs, every fragment live: 45 spilled registers): an empty volatile asm that "touches" the
            // step's accumulators keeps them in their step
#pragma unroll
            for (int nj = 0; nj < 4; ++nj) asm volatile("" : "+v"(acc[mi][nj]));
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto substep = [&](auto kind_c, const bool next_q) __attribute__((always_inline)) {
#define AVL_STEP(N) step(kind_c, std::integral_constant<int, N>(), next_q)
        AVL_STEP(0); AVL_STEP(1); AVL_STEP(2); AVL_STEP(3); AVL_STEP(4); AVL_STEP(5); AVL_STEP(6); AVL_STEP(7);
        if constexpr (NS == 16) { AVL_STEP(8); AVL_STEP(9); AVL_STEP(10); AVL_STEP(11); AVL_STEP(12); AVL_STEP(13); AVL_STEP(14); AVL_STEP(15); }
#undef AVL_STEP
        Wsc = Wsn;                                               // (a move only where the next sub-step is an FP4 one)
        ++g;
    };
    typedef std::integral_constant<int, 0> F16;
    typedef std::integral_constant<int, 1> FP4;

    // prologue: this wave's share of sub-step 0 (and 1) has landed, then everyone's
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int t = vb; t < total; t += nwg) {
        const int nt = t % ntiles, mt = t / ntiles;
        {   // cold start of the tile: the first fragments of its first sub-step (an f16 one; its slot was certified by the previous
            // event, or by the prologue barrier).  The fragments the last sub-step of the previous tile read ahead are not kept
            // across the epilogue (registers), they are simply read again.
            const char* ab = lds + (g & 1) * A_BYTES;
            const char* wb = lds + (g & 1) * W_BYTES;
#pragma unroll
            for (int nj = 0; nj < 4; ++nj) Wf[0][nj] = rdW(wb, 0, nj);
#pragma unroll
            for (int s = 0; s < D; ++s) Af[s] = rdA(ab, 0, s);
        }
        for (int mb = 0; mb < nmb; ++mb) {
            for (int j = 0; j < 4; ++j) substep(F16(), j == 3);
            for (int u = 0; u < nmx; ++u) substep(FP4(), u + 1 < nmx);
        }
        unsigned long long te0 = 0;
        if (PROBE == 3) te0 = stamp();
        mx_epilogue<IO, MI>(q, acc, nt, mt, wm, wn, fr, kq);
        if (t + nwg < total) init_acc(t + nwg);
        if (PROBE == 3) { tsum[6] += stamp() - te0; tsum[7] += 1; tlast = 0; }      // (the epilogue is not part of the sub-step sums: [6] / [7] = cycles per epilogue)
    }
    if (PROBE == 3 && q.dbg && lane == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) q.dbg[((size_t)blockIdx.x * 8 + wave) * 8 + i] = tsum[i];
    }
}

#ifdef AVL_EXPERIMENTS
#include "experiments/seg_gemm_mx_pp.inc"      // k_gemm_mx_pp (ping-pong schedule experiment): not part of the release translation unit
#endif  // AVL_EXPERIMENTS

template <int IO, int MI>
int launch_ring_mx_t(const MxArgs& a0, hipStream_t s) {
    constexpr int BM = 2 * MI * 16, LDS = 2 * (BM + 256) * 128 + 2 * 4096;
    MxArgs a = a0;
    a.stagger = AVL_EXP_INT("AVL_MX_STAGGER", 1);
    a.same_tile = 0;
    a.g.ntiles = a.g.N / 256;
    const int mtiles = (a.g.M + BM - 1) / BM;
    const int total = mtiles * a.g.ntiles;
    int grid = total < 256 ? total : 256;
#ifdef AVL_EXPERIMENTS
    { const int gl = AVL_EXP_INT("AVL_MX_GRID", 0); if (gl > 0 && gl < grid) grid = gl; }      // fewer workgroups: is a phase chip- or CU-bound?
#endif
    // Which main loop: the software-pipelined stream wins where the K loop is long (K >= 1024: -2...-9 % on layer4 / layer3 conv1
    // and conv3 + downsample), the round-2 kernel (two plain barriers per sub-step, a cheaper tile switch) where a tile is only 1-2 K
    // macro-blocks (K = 256 / 512: layer2, layer3 conv3, the decoder's pointwise convs: the stream is +2...+19 % there);
    // profiles/r03/gemm_mx_pipe_vs_ring_ab.log.  AVL_MX_PIPE = 0 / 1 forces one of them (A/B experiments).
#ifdef AVL_EXPERIMENTS
    if (MI == 8) {
        const int pp = AVL_EXP_INT("AVL_MX_PP", 0);        // 1: ping-pong kernel for every 256-row-tile MX GEMM, 2: only where K >= 1024
        if (pp == 1 || (pp == 2 && a.g.K >= 1024)) {
            AVL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_mx_pp<IO>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
            hipLaunchKernelGGL((k_gemm_mx_pp<IO>), dim3(grid), dim3(512), LDS, s, a, mtiles);
            AVL_LAUNCH_CHECK();
            return AVL_OK;
        }
    }
#endif
    const int pipe_env = AVL_EXP_INT("AVL_MX_PIPE", -1);
    const bool pipe = pipe_env >= 0 ? pipe_env != 0 : a.g.K >= 1024;
    if (pipe) {
        // LATE: how many steps behind the event(s) waves 0-3 issue their bursts (waves 4-7: right behind them)
        constexpr int L0 = MI == 8 ? 0 : 1;
#define AVL_PIPE_LAUNCH(...)                                                                                                                   \
    do {                                                                                                                                       \
        AVL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_mx_pipe<__VA_ARGS__>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS)); \
        hipLaunchKernelGGL((k_gemm_mx_pipe<__VA_ARGS__>), dim3(grid), dim3(512), LDS, s, a, mtiles);                                           \
    } while (0)
#ifdef AVL_EXPERIMENTS
        constexpr int L1 = 2;
        const int late_env = AVL_EXP_INT("AVL_MX_LATE", -1);
        const int probe = AVL_EXP_INT("AVL_MX_PROBE", 0);       // timing experiments only (256-row tiles): 1, 2, 4, 5 give garbage results
        if (MI == 8 && probe == 1) AVL_PIPE_LAUNCH(IO, 8, 0, 1);
        else if (MI == 8 && probe == 2) AVL_PIPE_LAUNCH(IO, 8, 0, 2);
        else if (MI == 8 && probe == 4) AVL_PIPE_LAUNCH(IO, 8, 0, 4);
        else if (MI == 8 && probe == 5) AVL_PIPE_LAUNCH(IO, 8, 0, 5);
        else if (MI == 8 && probe == 6) AVL_PIPE_LAUNCH(IO, 8, 0, 6);
        else if (MI == 8 && probe == 3) {
            hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
            AVL_HIP_CHECK(hipStreamIsCapturing(s, &cap));
            AVL_REQUIRE(cap == hipStreamCaptureStatusNone, "AVL_MX_PROBE=3 synchronises the stream: not while it is being captured (MODEL.HIP_GRAPH = False)");
            static unsigned long long* dbg = nullptr;
            if (!dbg) AVL_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&dbg), 256 * 8 * 8 * sizeof(unsigned long long), 0));
            memset(dbg, 0, 256 * 8 * 8 * sizeof(unsigned long long));
            a.dbg = dbg;
            a.same_tile = AVL_EXP_INT("AVL_MX_SAMETILE", 0);
            AVL_PIPE_LAUNCH(IO, 8, 0, 3);
            AVL_HIP_CHECK(hipStreamSynchronize(s));
            double sum[2][8] = {};
            for (int b = 0; b < grid; ++b)
                for (int w = 0; w < 8; ++w)
                    for (int i = 0; i < 8; ++i) sum[w >= 4][i] += (double)dbg[(b * 8 + w) * 8 + i];
            for (int e = 0; e < 2; ++e) {
                const double n = sum[e][5] > 0 ? sum[e][5] : 1;
                fprintf(stderr, "[mx probe] M %d N %d K %d nmx %d %s waves: cycles per sub-step %.0f | EW wait+barrier %.0f | W burst %.0f | EV wait+barrier %.0f | A burst %.0f | epilogue %.0f (x %.1f per wave)\n",
                        a.g.M, a.g.N, a.g.K, a.nmx, e ? "early (4-7)" : "late (0-3)", sum[e][0] / n, sum[e][1] / n, sum[e][2] / n, sum[e][3] / n, sum[e][4] / n,
                        sum[e][7] > 0 ? sum[e][6] / sum[e][7] : 0., sum[e][7] / (4.0 * grid));
            }
        }
        else if (late_env == L1) AVL_PIPE_LAUNCH(IO, MI, L1);
        else if (MI == 8 && AVL_EXP_INT("AVL_MX_SPREAD", 0) == 1) AVL_PIPE_LAUNCH(IO, 8, 0, 0, 1);
        else if (MI == 8 && AVL_EXP_INT("AVL_MX_SPREAD", 0) == 2) AVL_PIPE_LAUNCH(IO, 8, 0, 0, 2);
        else if (MI == 8 && AVL_EXP_INT("AVL_MX_SPREAD", 0) == 3) AVL_PIPE_LAUNCH(IO, 8, 0, 0, 3);
        else if (MI == 8 && AVL_EXP_INT("AVL_MX_SPREAD", 0) == 4) AVL_PIPE_LAUNCH(IO, 8, 0, 0, 4);
        else if (MI == 8 && AVL_EXP_INT("AVL_MX_SPREAD", 0) == 5) AVL_PIPE_LAUNCH(IO, 8, 0, 0, 5);
        else
#endif
            AVL_PIPE_LAUNCH(IO, MI, L0);
#undef AVL_PIPE_LAUNCH
    } else {
        AVL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_ring_mx<IO, MI>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        hipLaunchKernelGGL((k_gemm_ring_mx<IO, MI>), dim3(grid), dim3(512), LDS, s, a, mtiles);
    }
    AVL_LAUNCH_CHECK();
    return AVL_OK;
}

int launch_ring_mx(const MxArgs& a, bool quantize_out, hipStream_t s) {
    // 128-row tiles when 256-row tiles would leave more than a quarter of the 256 CUs idle (or give a ragged second wave)
    const int force = AVL_EXP_INT("AVL_MX_TILE", 0);       // 128 / 256: experiments
    const long long t256 = (long long)((a.g.M + 255) / 256) * (a.g.N / 256);
    const bool small = force ? force == 128 : (t256 < 192 || (t256 > 256 && t256 < 384));
    if (small) return quantize_out ? launch_ring_mx_t<1, 4>(a, s) : launch_ring_mx_t<0, 4>(a, s);
    return quantize_out ? launch_ring_mx_t<1, 8>(a, s) : launch_ring_mx_t<0, 8>(a, s);
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
    if (op.out_f32) AVL_REQUIRE(!op.out_lo, "GEMM: an fp32 output has no low plane");
    if (op.out_f32 && op.out_mx)      // the fused arg-max (labels through out_mx): the small-N kernel, plain logits
        AVL_REQUIRE(N <= 32 && !op.in2 && !op.relu && op.w_split != 2 && t.bn == 64 && !ring_eligible(op),
                    "GEMM: an arg-max output (out_f32 with out_mx = uint8 labels[rows]) needs N <= 32, no residual, no ReLU (N %d)", N);
    if (op.w_split == 2) {
        AVL_REQUIRE(op.dtype == AVL_F16 && op.w_mx && op.in_mx, "MX GEMM needs AVL_F16 activations and the w_mx / in_mx bundles");
        AVL_REQUIRE(K % 256 == 0 && N % 256 == 0 && op.w_rows % 256 == 0 && op.w_rows >= N, "MX GEMM: K %d, N %d, w_rows %d must be multiples of 256", K, N, op.w_rows);
        AVL_REQUIRE(op.in_ld == K && op.in_rows % 256 == 0 && op.in_rows >= M, "MX GEMM input must be dense rows padded to 256 (ld %d, rows %d)", op.in_ld, op.in_rows);
        AVL_REQUIRE(!op.out_f32 && op.out_ld >= N && (op.out_ld * 2) % 16 == 0, "MX GEMM output");
        AVL_REQUIRE(!op.out_mx || op.out_ld == N, "MX GEMM can only quantise a dense output (out_ld %d, N %d)", op.out_ld, N);
        AVL_REQUIRE(!(op.mx_flags & AVL_MX_RES_LO) || (op.in2 && op.in2_mx && !op.in2_lo && op.in2_ld == N),
                    "AVL_MX_RES_LO needs in2, in2_mx (bundle of a dense [out_rows][N] residual) and no in2_lo");
        AVL_REQUIRE(!(op.mx_flags & AVL_MX_OUT_LO) || op.out_mx, "AVL_MX_OUT_LO without out_mx");
        AVL_REQUIRE(!(op.mx_flags & AVL_MX_IN_LO) || !op.in_lo, "AVL_MX_IN_LO together with in_lo");
        if (op.in3) {
            AVL_REQUIRE(op.in3_mx && op.in3_c > 0 && op.in3_c % 256 == 0 && op.in3_ld == op.in3_c,
                        "second GEMM input: in3_mx, in3_c %d (multiple of 256), dense rows (in3_ld %d)", op.in3_c, op.in3_ld);
            AVL_REQUIRE(!op.in_lo || (op.mx_flags & AVL_MX_IN_LO), "a second GEMM input needs both inputs' lo parts in their bundles (AVL_MX_IN_LO) or none");
            AVL_REQUIRE((reinterpret_cast<uintptr_t>(op.in3) | reinterpret_cast<uintptr_t>(op.in3_mx)) % 16 == 0, "second GEMM input must be 16-byte aligned");
        }
        AVL_REQUIRE((long long)op.in_rows * K * 2 < 0xffffff00LL && (long long)op.w_rows * K * 2 < 0xffffff00LL, "MX GEMM operand larger than 4 GB");
        AVL_REQUIRE((long long)op.out_rows * op.out_ld * 2 < (1LL << 31) && (!op.in2 || (long long)op.out_rows * op.in2_ld * 2 < (1LL << 31)) &&
                    (long long)(N / 256 + 1) * op.out_rows * 8 < (1LL << 31), "MX GEMM output / residual plane beyond 2 GB (32-bit epilogue addressing)");
        AVL_REQUIRE((reinterpret_cast<uintptr_t>(op.w_mx) | reinterpret_cast<uintptr_t>(op.in_mx) | reinterpret_cast<uintptr_t>(op.out_mx) |
                     reinterpret_cast<uintptr_t>(op.in_lo) | reinterpret_cast<uintptr_t>(op.in2_lo) | reinterpret_cast<uintptr_t>(op.out_lo)) % 16 == 0,
                    "MX GEMM bundles / low planes must be 16-byte aligned");
    } else if (op.w_split || op.in_lo || op.in2_lo || op.out_lo) {
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

#ifdef AVL_EXPERIMENTS
int launch_gemm_w4(const avl_seg_op& op, hipStream_t s);      // seg_gemm_w4.hip: one-wave-per-SIMD experiment (w_layout = 5)
#endif

int launch_gemm(const avl_seg_op& op, hipStream_t s) {
#ifdef AVL_EXPERIMENTS
    if (op.w_layout == 5) return launch_gemm_w4(op, s);
#else
    AVL_REQUIRE(op.w_layout != 5, "GEMM w_layout 5 (k_gemm_w4) exists in the experiments build only (make experiments)");
#endif
    GemmArgs a;
    a.A = op.in; a.W = op.weight; a.bias = op.bias; a.R = op.in2; a.C = op.out;
    a.lda = op.in_ld; a.ldr = op.in2_ld; a.ldc = op.out_ld;
    a.M = op.out_h * op.out_w; a.N = op.out_c; a.K = op.in_c;
    a.relu = op.relu; a.out_f32 = op.out_f32;
    if (op.w_split == 2) {
        MxArgs mx;
        memset(&mx, 0, sizeof(mx));
        a.nsub = 1;
        a.a_lo_delta = 0;
        a.R_lo = op.in2_lo; a.C_lo = op.out_lo;
        mx.g = a;
        // (with a second input appended along K: K1 = op.in_c comes from `in`, the rest from in3; weights span the whole K)
        const long long K1 = op.in_c, K2 = op.in3 ? op.in3_c : 0;
        auto abundle = [&](const void* base, long long rows, long long kk, const char** plane, const char** scales) {
            const char* b = static_cast<const char*>(base);
            const long long P = rows * (kk / 2), S = (kk / 256) * rows * 8;
            plane[0] = b; scales[0] = b + P; plane[1] = b + P + S; scales[1] = b + 2 * P + S;
        };
        abundle(op.in_mx, op.in_rows, K1, mx.Aq, mx.As);
        mx.nmb1 = (int)(K1 / 256);
        mx.lda2 = mx.g.lda; mx.ldaq2 = (int)(K1 / 2);
        if (K2) {
            a.K = (int)(K1 + K2);
            mx.g.K = a.K;
            mx.A2 = op.in3; mx.lda2 = op.in3_ld; mx.ldaq2 = (int)(K2 / 2);
            abundle(op.in3_mx, op.in_rows, K2, mx.Aq2, mx.As2);
            mx.a_srows2 = op.in_rows;
        }
        {
            const long long KT = K1 + K2;
            const char* b = static_cast<const char*>(op.w_mx);
            const long long P = (long long)op.w_rows * (KT / 2), S = (KT / 256) * op.w_rows * 8;
            mx.Wq[0] = b; mx.Ws[0] = b + P; mx.Wq[1] = b + P + S; mx.Ws[1] = b + 2 * P + S;
        }
        mx.ldaq = (int)(K1 / 2);
        mx.a_srows = op.in_rows;
        mx.w_srows = op.w_rows;
        mx.nmx = (op.in_lo || (op.mx_flags & AVL_MX_IN_LO)) ? 2 : 1;
        if (op.mx_flags & AVL_MX_RES_LO) {
            const char* b = static_cast<const char*>(op.in2_mx);
            const long long rows = op.out_rows, P = rows * (a.N / 2), S = (long long)(a.N / 256) * rows * 8;
            mx.Rq = b + P + S; mx.Rs = b + 2 * P + S;
            mx.ldrq = a.N / 2;
            mx.r_srows = rows;
        }
        if (op.out_mx) {
            char* b = static_cast<char*>(op.out_mx);
            const long long rows = op.out_rows, P = rows * (a.N / 2), S = (long long)(a.N / 256) * rows * 8;
            mx.Cq[0] = b; mx.Cs[0] = b + P;
            if (op.out_lo || (op.mx_flags & AVL_MX_OUT_LO)) { mx.Cq[1] = b + P + S; mx.Cs[1] = b + 2 * P + S; }
            mx.ldcq = a.N / 2;
            mx.c_srows = rows;
        }
        return launch_ring_mx(mx, op.out_mx != nullptr, s);
    }
    a.nsub = op.w_split ? (op.in_lo ? 3 : 2) : 1;
    a.a_lo_delta = op.in_lo ? static_cast<const char*>(op.in_lo) - static_cast<const char*>(op.in) : 0;
    a.R_lo = op.in2_lo;
    a.C_lo = op.out_f32 ? op.out_mx : op.out_lo;      // (out_f32 ops have neither a low plane nor an MX bundle: out_mx carries the label map)
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
            const int deep = AVL_EXP_INT("AVL_GEMM_DEEP", 1);      // A/B: 0 = 2 + 2 tiles, one sub-step ahead
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
