// k_gemm_w4 -- EXPERIMENT (round 3): the 1x1-conv GEMM with ONE wave per SIMD.
//
// VERDICT r2 item 3 / DESIGN r2 section 5 proposed it for the MX GEMM: 128 x 128 wave tiles, accumulators in AGPRs, four waves per
// workgroup.  Round 3's measurements say the 8-wave kernels are bound by the CU's L2 -> LDS staging path
// (~27 B/clk while a burst lasts) which idles ~30 % of the time because an LDS slot can only be refilled once it is released and two
// 64 KB slots fill the LDS.  This kernel tests the other way to use the 160 KB: FOUR stages of 32 K-values (64-byte LDS rows, 16 KB
// per operand and stage), the LDS-DMA of stage s + 3 spread evenly between the MFMAs of stage s (one instruction per eight MFMAs),
// one barrier per stage between four waves instead of eight.  Plain GEMM only (one 16-bit plane in, one out, optional single-plane
// residual): it is reachable through AVL_OP_GEMM with w_layout = 5 (tools/bench_gemm.py --variants 5) and is NOT used by the network.
//
// Layout: tile 256 x 256, waves 2 x 2, wave tile 128 rows (pixels) x 128 columns (channels) = acc[h 2][mi 8][nj 4] (256 registers).
// The product is computed transposed as in k_gemm_ring (weights on the MFMA's row operand, permuted so that a lane ends with 16
// consecutive channels of one pixel per 64-column half h).
// LDS rows are 64 bytes (4 chunks of 16); chunk c of row r is stored at position c ^ ((r >> 2) & 3): a 16-lane phase of a
// ds_read_b128 (rows r0 .. r0 + 15, one logical chunk) then touches all 64 banks once.  The swizzle is applied on the DMA source side
// (the DMA writes LDS lane-linearly: lane l of an instruction -> row (l >> 2), position l & 3).
#include "seg_types.h"

namespace avl {
namespace {

struct W4Args {
    const void* A;
    const void* W;
    const float* bias;
    const void* R;
    void* C;
    int lda, ldr, ldc;
    int M, N, K;
    int relu, ntiles;
};

constexpr int W4_ST = 4;                       // stages in the ring
constexpr int W4_TILE = 256 * 64;              // bytes of one operand tile of one stage

template <typename H, int W4_SCHED>
__global__ void __launch_bounds__(256) k_gemm_w4(W4Args p, int mtiles) {
    typedef typename Half16<H>::v8 v8;
    constexpr int W_REGION = W4_ST * W4_TILE;
    constexpr int NI = 4;                      // DMA instructions per wave, operand and stage (16 rows each)
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int nwg = gridDim.x;
    int vb;
    {
        const int bid = blockIdx.x, xcd = bid & 7, local = bid >> 3, q = nwg >> 3, r = nwg & 7;
        vb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    }
    const int total = mtiles * p.ntiles;
    const int nks = p.K / 32;                  // stages per tile
    const unsigned lds_base = lds_addr(lds);

    // ---- producer: instruction i of this wave covers rows (i * 4 + wave) * 16 + (lane >> 2); (row >> 2) & 3 = (lane >> 4) & 3
    const unsigned prow = (unsigned)lane >> 2, pchunk = ((unsigned)lane & 3u) ^ (((unsigned)lane >> 4) & 3u);
    const unsigned a_voff = prow * (unsigned)(p.lda * 2) + (pchunk << 4);
    const unsigned w_voff = prow * (unsigned)(p.K * 2) + (pchunk << 4);
    int pt = vb, pk = 0, issued = 0;           // tile / stage being issued next, stages issued so far
    const char* a_tile = nullptr;
    const char* w_tile = nullptr;
    auto set_tile = [&]() {
        const int nt = pt % p.ntiles, mt = pt / p.ntiles;
        a_tile = static_cast<const char*>(p.A) + ((long long)mt * 256 + wave * 16) * p.lda * 2;
        w_tile = static_cast<const char*>(p.W) + ((long long)nt * 256 + wave * 16) * p.K * 2;
    };
    // DMA instruction j (0 .. 7) of the stage being issued: 0 .. 3 activations, 4 .. 7 weights
    auto issue_one = [&](int j) {
        const unsigned slot = (unsigned)(issued & (W4_ST - 1)) * W4_TILE + (unsigned)wave * 1024u;
        if (j < NI) glds16_saddr(a_tile + (long long)j * 64 * p.lda * 2 + pk * 64, a_voff, lds_base + slot + j * 4096);
        else glds16_saddr(w_tile + (long long)(j - NI) * 64 * p.K * 2 + pk * 64, w_voff, lds_base + W_REGION + slot + (j - NI) * 4096);
    };
    auto stage_issued = [&]() {
        ++issued;
        if (++pk == nks) {
            pk = 0;
            pt += nwg;
            if (pt < total) set_tile();
        }
    };
    if (pt < total) set_tile();
    for (int s = 0; s < W4_ST - 1; ++s)
        if (pt < total) {
#pragma unroll
            for (int j = 0; j < 2 * NI; ++j) issue_one(j);
            stage_issued();
        }

    // ---- consumer addressing (byte offsets inside a stage's tile)
    const int fr = lane & 15, kq = lane >> 4;
    const unsigned a_v0 = (unsigned)((wm * 128 + fr) * 64 + ((kq ^ ((fr >> 2) & 3)) << 4));                  // + mi * 1024
    const unsigned w_r0 = (unsigned)(W_REGION + (wn * 128 + (fr >> 2) * 16 + (fr & 3)) * 64);               // + h * 4096 + nj * 256 + ((kq ^ nj) << 4)

    f32x4 acc[2][8][4];
    auto init_acc = [&](int t) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int nb = (t % p.ntiles) * 256 + wn * 128 + h * 64 + kq * 16;
#pragma unroll
            for (int nj = 0; nj < 4; ++nj) {
                const float4 b = *reinterpret_cast<const float4*>(p.bias + nb + 4 * nj);
#pragma unroll
                for (int mi = 0; mi < 8; ++mi) acc[h][mi][nj] = f32x4{b.x, b.y, b.z, b.w};
            }
        }
    };
    if (vb < total) init_acc(vb);

    int g = 0;                                 // stages consumed so far
    for (int t = vb; t < total; t += nwg) {
        const int nt = t % p.ntiles, mt = t / p.ntiles;
        for (int ks = 0; ks < nks; ++ks, ++g) {
            // stage g has landed (this wave's share); up to two younger stages may stay in flight.  Younger compiler-visible
            // operations (epilogue stores, bias loads) only make the counted wait over-wait (in-order retirement).
            const int ahead = issued - (g + 1);
            if (ahead >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const bool feed = pt < total;          // stage g + 3 -> the slot read in stage g - 1 (free since this barrier)
            const char* ab = lds + (g & (W4_ST - 1)) * W4_TILE;
            v8 wf[2][4], af[2];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int nj = 0; nj < 4; ++nj)
                    wf[h][nj] = *reinterpret_cast<const v8*>(ab + w_r0 + h * 4096 + nj * 256 + ((kq ^ nj) << 4));
            af[0] = *reinterpret_cast<const v8*>(ab + a_v0);
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
                if (mi + 1 < 8) af[(mi + 1) & 1] = *reinterpret_cast<const v8*>(ab + a_v0 + (mi + 1) * 1024);
                if (W4_SCHED == 0) { if (feed) issue_one(mi); }
                else if (W4_SCHED == 1) {        // half of the stage's DMA under the fragment-read latency at the head of the stage
                    if (feed && mi == 0) { issue_one(0); issue_one(1); issue_one(2); issue_one(3); }
                    if (feed && mi == 4) { issue_one(4); issue_one(5); issue_one(6); issue_one(7); }
                } else if (W4_SCHED == 2) {
                    if (feed && (mi & 1) == 0) { issue_one(mi); issue_one(mi + 1); }
                }                                // W4_SCHED == 3: no DMA after the prologue (timing probe: results are garbage)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int nj = 0; nj < 4; ++nj) acc[h][mi][nj] = Half16<H>::mfma(wf[h][nj], af[mi & 1], acc[h][mi][nj]);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (feed) stage_issued();
        }
        // ---- epilogue (the next tile's first stages are in flight)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int nbase = nt * 256 + wn * 128 + h * 64 + kq * 16;
            if (nbase + 16 > p.N) continue;
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
                const int m = mt * 256 + wm * 128 + mi * 16 + fr;
                if (m >= p.M) continue;
                float v[16];
#pragma unroll
                for (int nj = 0; nj < 4; ++nj)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[nj * 4 + r] = acc[h][mi][nj][r];
                if (p.R) {
                    const H* rp = static_cast<const H*>(p.R) + (long long)m * p.ldr + nbase;
                    const v8 r0 = *reinterpret_cast<const v8*>(rp), r1 = *reinterpret_cast<const v8*>(rp + 8);
#pragma unroll
                    for (int i = 0; i < 8; ++i) { v[i] += (float)r0[i]; v[8 + i] += (float)r1[i]; }
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
            }
        }
        if (t + nwg < total) init_acc(t + nwg);
    }
}

template <typename H>
int launch_w4_typed(const W4Args& a0, hipStream_t s) {
    constexpr int LDS = 2 * W4_ST * W4_TILE;
    static_assert(LDS <= 160 * 1024, "k_gemm_w4 LDS");
    W4Args a = a0;
    a.ntiles = a.N / 256;
    const int mtiles = (a.M + 255) / 256;
    const int total = mtiles * a.ntiles;
    const int sched = AVL_EXP_INT("AVL_W4_SCHED", 0);     // where in a stage the DMA instructions go
#define AVL_W4_LAUNCH(S)                                                                                                            \
    do {                                                                                                                            \
        AVL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_w4<H, S>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS)); \
        hipLaunchKernelGGL((k_gemm_w4<H, S>), dim3(total < 256 ? total : 256), dim3(256), LDS, s, a, mtiles);                       \
    } while (0)
    if (sched == 1) AVL_W4_LAUNCH(1);
    else if (sched == 2) AVL_W4_LAUNCH(2);
    else if (sched == 3) AVL_W4_LAUNCH(3);
    else AVL_W4_LAUNCH(0);
#undef AVL_W4_LAUNCH
    AVL_LAUNCH_CHECK();
    return AVL_OK;
}

}  // namespace

// experiment entry (w_layout = 5): one 16-bit plane in / out, N % 256 == 0, K % 32 == 0, rows padded to 256
int launch_gemm_w4(const avl_seg_op& op, hipStream_t s) {
    AVL_REQUIRE(is_half(op.dtype) && !op.w_split && !op.in_lo && !op.in2_lo && !op.out_lo && !op.out_f32 && !op.in3, "k_gemm_w4: plain 16-bit GEMM only");
    AVL_REQUIRE(op.out_c % 256 == 0 && op.in_c % 32 == 0 && op.w_rows >= op.out_c, "k_gemm_w4: N %% 256, K %% 32");
    const int M = op.out_h * op.out_w;
    AVL_REQUIRE(op.in_rows >= (M + 255) / 256 * 256 && op.in_ld >= op.in_c && (op.in_ld * 2) % 16 == 0, "k_gemm_w4: input rows padded to 256");
    AVL_REQUIRE((long long)op.in_rows * op.in_ld * 2 < (1LL << 32) && (long long)op.w_rows * op.in_c * 2 < (1LL << 32), "k_gemm_w4: 32-bit DMA offsets");
    W4Args a;
    a.A = op.in; a.W = op.weight; a.bias = op.bias; a.R = op.in2; a.C = op.out;
    a.lda = op.in_ld; a.ldr = op.in2_ld; a.ldc = op.out_ld;
    a.M = M; a.N = op.out_c; a.K = op.in_c;
    a.relu = op.relu; a.ntiles = 0;
    return op.dtype == AVL_F16 ? launch_w4_typed<f16>(a, s) : launch_w4_typed<bf16>(a, s);
}

}  // namespace avl
