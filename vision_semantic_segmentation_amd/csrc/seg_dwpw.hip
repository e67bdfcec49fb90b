// Fused depthwise-separable block (core/nn/modules/conv.py:103-141, the three dilated ASPP branches aspp.py:45-63):
//
//   out[m][n] = relu( b2[n] + sum_k W2[n][k] * a[m][k] ),   a[m][k] = T( relu( b1[k] + sum_t w1[t][k] * x[pix(m) + off_t][k] ) )
//
// Unfused, the depthwise result a (M x 2048, 132 MB at 1080p) is written to HBM and read back by the 1x1 GEMM, and
// both kernels are bound by exactly that traffic (75 + 44 us per branch).  Here a workgroup owns 128 pixels x all
// 256 output channels; for every 64-channel K-step its 512 lanes compute the 128 x 64 slice of `a` on the vector ALUs
// (9 taps, 16 outputs per lane, the same paired v_dot2c order as k_dwconv, so `a` is bit-identical to the unfused one),
// write it into LDS in the swizzled layout the MFMA fragments are read from, and multiply it with the weight slice
// that was loaded meanwhile.  `a` never exists in HBM.
//   * taps: buffer loads with a range-checked descriptor -- a tap outside the image gets an out-of-range
//     offset and the hardware returns zeros (no pointer select, no zero page); the K-step advances the scalar offset;
//   * the weight slice of the next step travels through registers as well (see load_w: one kind of load, in order);
//   * depthwise weights/bias of all K-steps sit in LDS for the whole tile (K/64 x 1.5 KB <= 48 KB);
//   * while the MFMAs of step s run, the taps of step s+2 are in flight and the `a` slice of step s+1 is computed.
#include "seg_types.h"

namespace avl {
namespace {

// DW_EXP (tools/ab_dwpw.sh; 0 in every shipped build): phase ablations of k_dwpw_x / k_dwpw_xs for timing only (results are wrong) --
// bit 0: no MFMAs, bit 1: no depthwise arithmetic (zero tiles), bit 2: no tap loads, bit 3: no fragment reads from LDS (with bit 0);
// k_dwpw_xs only: bit 7: no result stores
#ifndef DW_EXP
#define DW_EXP 0
#endif
// (bit 8: depthwise parameters are constants instead of LDS reads)
#define DW_PARAM(T, ptr) ((DW_EXP & 256) ? T{} : *reinterpret_cast<const T*>(ptr))
constexpr int TM = 128;              // pixels per workgroup
constexpr int TN = 256;              // output channels per workgroup
constexpr int A_STAGE = TM * 128;    // 16 KB: 128 rows x 64 k x 2 B
constexpr int W_STAGE = TN * 128;    // 32 KB
constexpr int LDS_W = 0, LDS_A = 2 * W_STAGE, LDS_P = LDS_A + 2 * A_STAGE;   // params after the two rings
constexpr int P_STEP = 8 * 6 * 8 * 4;                                         // bytes of depthwise parameters per K-step

struct DwPwArgs {
    const void* X;
    const void* X_lo;        // k_dwpw_xs: low plane of the input (same row stride), else NULL
    // k_dwpw_xs<CLS>: the network's last 1x1 conv (decoder.py:42-43: 256 -> num_classes, bias, no BN / ReLU) and torch.argmax (semantic_segmentation.py:56)
    // in this kernel's epilogue -- the block's own result never goes to memory
    const void* Wc;          // classifier weights, f16 [2 planes hi | lo][32 rows][N]
    const float* bc;         // its bias [32]
    float* LG;               // logits [M][ncls]
    unsigned char* labels;   // arg-max [M]
    int ncls;
    int tiles_x;             // k_dwpw_xs: > 0 = the 128 pixels of a tile are an 8 x 16 block (tile mt = (mt / tiles_x, mt % tiles_x)) instead of 128 consecutive pixels
    const void* W;
    const float* bias;
    const uint32_t* dwp;     // [K/64][chunk 8][6][8] dwords: 5 tap pairs (lo = tap 2p, hi = tap 2p+1, 16-bit type) + fp32 bias
    const int* order;        // [mtiles] pixel-tile visited by the i-th workgroup slot (see launch_dwpw), behind the parameters
    int mtiles, per_xcd;
    void* C;
    void* C_lo;              // "mixed" precision: low plane of the result (value - f16(value)), or NULL
    int H, Wd, OW, ldx, ldc, M, N, K, dil, pad, ntiles;      // input H x Wd, output rows are OW wide, M = OH * OW
    unsigned x_bytes;
};

typedef int v4i __attribute__((ext_vector_type(4)));

// WSUB = 2 ("mixed" precision): the 1x1 weights come as f16 pairs, rows [K/64][hi 64 | lo 64]; every K-step then has
// two MFMA passes over the same depthwise slice (hi weights, lo weights), and the depthwise work of the next slice is
// split between them (pixel q = 0 during the first, q = 1 during the second).
template <typename HT, int WSUB = 1>
__global__ void __launch_bounds__(512) k_dwpw(DwPwArgs p) {
    typedef typename Half16<HT>::v8 v8;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    // Workgroups b, b+8, ... share an XCD and run at the same time; slot (xcd, i) takes the (xcd * per_xcd + i)-th tile
    // of an order in which tiles whose rows are a multiple of the dilation apart are neighbours: such tiles read the
    // same input rows (as top / centre / bottom taps), and only tiles that are in flight on the SAME L2 at the same
    // K-step can share them -- otherwise every input row comes in from HBM three times.
    const int nt = blockIdx.x % p.ntiles;
    const int bm = blockIdx.x / p.ntiles, slot = (bm & 7) * p.per_xcd + (bm >> 3);
    if (slot >= p.mtiles) return;
    const int mt = p.order[slot];
    const int nk = p.K / 64;

    // ---- depthwise parameters of every K-step -> LDS
    {
        const uint4* src = reinterpret_cast<const uint4*>(p.dwp);
        uint4* dst = reinterpret_cast<uint4*>(lds + LDS_P);
        for (int i = tid; i < nk * (P_STEP / 16); i += 512) dst[i] = src[i];
    }

    // ---- producer geometry: lane -> 8-channel chunk of two pixels (rows r0, r0 + 64 of the tile)
    const int chunk = tid & 7, r0 = tid >> 3;
    unsigned voff[2][9];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int m = mt * TM + r0 + q * 64;
        const int y = m / p.OW, x = m - y * p.OW;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int iy = y - p.pad + (t / 3) * p.dil, ix = x - p.pad + (t % 3) * p.dil;
            const bool ok = m < p.M && iy >= 0 && iy < p.H && ix >= 0 && ix < p.Wd;
            voff[q][t] = ok ? (unsigned)(((long long)iy * p.Wd + ix) * p.ldx + chunk * 8) * 2u : 0x7fffff00u;   // >= x_bytes: reads as 0
        }
    }
    // ---- weight slice (256 rows x 128 B per K-step) through registers: lane -> 16-byte chunk (tid & 7) of rows
    // (tid >> 3) + 64 i; the swizzle goes on the ds_write address.  (LDS-DMA was used first; mixing it with register loads
    // on one vmcnt counter needs waits that also drain the tap loads: an `s_waitcnt vmcnt(N)` is only safe when N does not
    // exceed the number of younger operations OF THE SAME KIND.)
    const int wchunk = tid & 7, wrow0 = tid >> 3;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(static_cast<const char*>(p.W) + (long long)nt * TN * p.K * (2 * WSUB)), 0, TN * p.K * (2 * WSUB), 0x00020000);
    int w_voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w_voff[i] = (wrow0 + 64 * i) * p.K * (2 * WSUB) + wchunk * 16;
    v4i wl[4];
    auto load_w = [&](int s) {
#pragma unroll
        for (int i = 0; i < 4; ++i) wl[i] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, w_voff[i], s * 128, 0);
    };
    auto store_w = [&](int s) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = wrow0 + 64 * i;
            const int key = ((r >> 1) & 1) | (((r >> 4) & 3) << 1);
            *reinterpret_cast<v4i*>(lds + LDS_W + (s & 1) * W_STAGE + r * 128 + ((wchunk ^ key) << 4)) = wl[i];
        }
    };

    // ---- consumer geometry (64 x 64 per wave, product transposed: a lane ends with 16 consecutive channels of a pixel)
    const int fr = lane & 15, kq = lane >> 4;
    int a_off[4], w_off[4], a_key[4], w_key[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int row = wm * 64 + mi * 16 + fr;
        a_off[mi] = LDS_A + row * 128;
        a_key[mi] = row & 7;
    }
#pragma unroll
    for (int nj = 0; nj < 4; ++nj) {
        const int row = wn * 64 + (fr >> 2) * 16 + nj * 4 + (fr & 3);
        w_off[nj] = LDS_W + row * 128;
        w_key[nj] = ((row >> 1) & 1) | (((row >> 4) & 3) << 1);
    }
    f32x4 acc[4][4];
    {
        const int nb = nt * TN + wn * 64 + kq * 16;
#pragma unroll
        for (int nj = 0; nj < 4; ++nj) {
            const float4 b = *reinterpret_cast<const float4*>(p.bias + nb + 4 * nj);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) acc[mi][nj] = f32x4{b.x, b.y, b.z, b.w};
        }
    }

    v4i raw[2][9];
    // Every load of the loop is a compiler-visible buffer load: hipcc's own s_waitcnt insertion counts them (in-order
    // return), also across the loop back-edge.  Hand-placed waits on inline-asm loads were tried first and were WRONG a
    // few times in 10^3 launches: the register allocator is free to copy an asm output (it believes the value is there
    // as soon as the asm statement has executed), and it did copy the tap registers at the top of the loop body, before
    // the data had landed.
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.X), 0, (int)p.x_bytes, 0x00020000);
    auto load_taps = [&](int s, int q) {
#pragma unroll
        for (int t = 0; t < 9; ++t) raw[q][t] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)voff[q][t], s * 128, 0);
    };

    // depthwise 3x3 of K-step s for pixel q of this lane -> A ring slot s & 1 (tap pairs exactly as k_dwconv)
    auto produce_a = [&](int s, int q) {
        const uint32_t* pp = reinterpret_cast<const uint32_t*>(lds + LDS_P + s * P_STEP + chunk * (6 * 8 * 4));
        float o[8];
        {
            const float4 b0 = *reinterpret_cast<const float4*>(pp + 5 * 8), b1 = *reinterpret_cast<const float4*>(pp + 5 * 8 + 4);
            o[0] = b0.x; o[1] = b0.y; o[2] = b0.z; o[3] = b0.w;
            o[4] = b1.x; o[5] = b1.y; o[6] = b1.z; o[7] = b1.w;
        }
#pragma unroll
        for (int pr = 0; pr < 5; ++pr) {
            const uint4 w0 = *reinterpret_cast<const uint4*>(pp + pr * 8), w1 = *reinterpret_cast<const uint4*>(pp + pr * 8 + 4);
            const uint32_t wp[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
            const v4i ra = raw[q][2 * pr], rb = raw[q][pr < 4 ? 2 * pr + 1 : 8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t ua = (uint32_t)ra[j], ub = (uint32_t)rb[j];
                const uint32_t lo = pr < 4 ? __builtin_amdgcn_perm(ub, ua, 0x05040100u) : (ua & 0xffffu);
                const uint32_t hi = pr < 4 ? __builtin_amdgcn_perm(ub, ua, 0x07060302u) : (ua >> 16);
                o[2 * j] = Half16<HT>::dot2(lo, wp[2 * j], o[2 * j]);
                o[2 * j + 1] = Half16<HT>::dot2(hi, wp[2 * j + 1], o[2 * j + 1]);
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = fmaxf(o[i], 0.f);
        const int row = r0 + q * 64;
        HT* dst = reinterpret_cast<HT*>(lds + LDS_A + (s & 1) * A_STAGE + row * 128 + ((chunk ^ (row & 7)) << 4));
        Vec8<HT>::store(dst, o);
    };

    // ---- prologue: taps and weights of slice 0 -> A(0), W(0) in LDS; then the taps of slice 1
    load_taps(0, 0);
    load_taps(0, 1);
    load_w(0);
    __syncthreads();                 // parameters are in LDS
    produce_a(0, 0);
    produce_a(0, 1);
    store_w(0);
    if (nk > 1) { load_taps(1, 0); load_taps(1, 1); }

    // Step s: the MFMAs of slice s, the depthwise slice s+1 (its taps were requested a whole step ago), the weight slice
    // s+1 (requested at the top of the step, written to LDS at its end) and, as each pixel's tap registers free up, the
    // taps of slice s+2.  hipcc places the vmcnt waits (all loads are compiler-visible and return in issue order).
    // (one half of the depthwise work between the two MFMA groups measured 10 % faster than all of it after them)
    // (load_w / store_w count weight sub-slices h = s * WSUB + j; the ring slot of sub-slice h is h & 1)
    for (int h = 0; h < nk * WSUB; ++h) {
        const int s = h / WSUB, j = h % WSUB;
        const bool more = s + 1 < nk, more2 = s + 2 < nk, morew = h + 1 < nk * WSUB;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // own A(s) / W(h) writes are in LDS
        __builtin_amdgcn_s_barrier();
        if (morew) load_w(h + 1);
        const char* base = lds;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            v8 wa[4], af[4];
#pragma unroll
            for (int nj = 0; nj < 4; ++nj)
                wa[nj] = *reinterpret_cast<const v8*>(base + (h & 1) * W_STAGE + w_off[nj] + (((kk * 4 + kq) ^ w_key[nj]) << 4));
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
                af[mi] = *reinterpret_cast<const v8*>(base + (s & 1) * A_STAGE + a_off[mi] + (((kk * 4 + kq) ^ a_key[mi]) << 4));
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int nj = 0; nj < 4; ++nj) acc[mi][nj] = Half16<HT>::mfma(wa[nj], af[mi], acc[mi][nj]);
            if (more && (WSUB == 1 || kk == j)) {
                produce_a(s + 1, kk);
                if (more2) load_taps(s + 2, kk);
            }
        }
        if (morew) store_w(h + 1);
    }

    // ---- epilogue: ReLU, convert, store (bias was the accumulators' start value)
    const int nbase = nt * TN + wn * 64 + kq * 16;
    if (nbase + 16 <= p.N) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int m = mt * TM + wm * 64 + mi * 16 + fr;
            if (m < p.M) {
                float lo[8], hi[8];
#pragma unroll
                for (int nj = 0; nj < 2; ++nj)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        lo[nj * 4 + r] = fmaxf(acc[mi][nj][r], 0.f);
                        hi[nj * 4 + r] = fmaxf(acc[mi][2 + nj][r], 0.f);
                    }
                HT* cp = static_cast<HT*>(p.C) + (long long)m * p.ldc + nbase;
                Vec8<HT>::store(cp, lo);
                Vec8<HT>::store(cp + 8, hi);
                if constexpr (WSUB != 1) {
                    if (p.C_lo) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) { lo[i] -= (float)(HT)lo[i]; hi[i] -= (float)(HT)hi[i]; }
                        HT* cl = static_cast<HT*>(p.C_lo) + (long long)m * p.ldc + nbase;
                        Vec8<HT>::store(cl, lo);
                        Vec8<HT>::store(cl + 8, hi);
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_dwpw_x: the "mixed" precision variant with an EXACT depthwise stage (round 3).
// Measured on the worst weights draw at 1080p (tools/opt_sweep.py): the two roundings the fused kernel above adds to the split
// pipeline -- depthwise weights as ONE f16 each, the depthwise result as ONE f16 plane in LDS -- carried 47 % of the logits' error
// energy (rms 1.59e-4 -> 1.16e-4 with the branches run unfused through the split depthwise kernel; weights 2/3 of it, result 1/3).
// Here: FP32 depthwise weights (w_split = 3; float32 [K/64][chunk 8][row 10][8]: rows 0 .. 8 = the taps, row 9 = the bias), one v_fma_mix_f32
// per tap and channel (f16 operand by op_sel, fp32 weight and sum); the result split into an f16 hi tile and an f16 lo tile in LDS, and three
// MFMA passes per 64-wide K-step: Wh.th + Wh.tl (sub-step j = 0, weight slice hi) and Wl.th (j = 1, slice lo).
// (Round 3's depthwise stage -- weights as f16 pairs hi + lo, tap-pair permutes + two v_dot2c -- took 120 vector instructions per pixel chunk
//  instead of 72 and 6 % longer per launch: profiles/r05/dwpw_phase_ablation.log.)
//   LDS: weight ring 2 x 32 KB | hi tiles 2 x 16 KB | lo tile 16 KB (written in sub-step j = 1 of step s - 1, read in j = 0 of
//   step s) | a ring of four K-steps' depthwise parameters, refilled three steps ahead by the first 160 lanes (the parameters of all 32 K-steps
//   of an ASPP branch, 80 KB, do not fit).
//   The whole depthwise slice s + 1 is computed during sub-step j = 1 (32 MFMAs), j = 0 carries 64 MFMAs.
constexpr int X_LDS_W = 0, X_LDS_AH = 2 * W_STAGE, X_LDS_AL = X_LDS_AH + 2 * A_STAGE, X_LDS_P = X_LDS_AL + A_STAGE;
constexpr int XP_RING = 4;
constexpr int FP_STEP = 8 * 10 * 8 * 4;              // 2560 bytes of depthwise parameters per K-step

__global__ void __launch_bounds__(512) k_dwpw_x(DwPwArgs p) {
    constexpr int PSTEP = FP_STEP;          // bytes of depthwise parameters per K-step
    typedef f16 HT;
    typedef typename Half16<HT>::v8 v8;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int nt = blockIdx.x % p.ntiles;
    const int bm = blockIdx.x / p.ntiles, slot = (bm & 7) * p.per_xcd + (bm >> 3);
    if (slot >= p.mtiles) return;
    const int mt = p.order[slot];
    const int nk = p.K / 64;

    // ---- depthwise parameters: K-step s -> ring slot s & 3, 160 lanes x 16 bytes
    const uint4* psrc = reinterpret_cast<const uint4*>(p.dwp);
    uint4 pnext = make_uint4(0u, 0u, 0u, 0u);
    auto load_p = [&](int s) { if (tid < PSTEP / 16 && s < nk) pnext = psrc[s * (PSTEP / 16) + tid]; };
    auto store_p = [&](int s) { if (tid < PSTEP / 16 && s < nk) *reinterpret_cast<uint4*>(lds + X_LDS_P + (s & (XP_RING - 1)) * PSTEP + tid * 16) = pnext; };
    for (int s = 0; s < 3; ++s) { load_p(s); store_p(s); }

    // ---- producer geometry: lane -> 8-channel chunk of two pixels (rows r0, r0 + 64 of the tile)
    const int chunk = tid & 7, r0 = tid >> 3;
    unsigned voff[2][9];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int m = mt * TM + r0 + q * 64;
        const int y = m / p.OW, x = m - y * p.OW;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int iy = y - p.pad + (t / 3) * p.dil, ix = x - p.pad + (t % 3) * p.dil;
            const bool ok = m < p.M && iy >= 0 && iy < p.H && ix >= 0 && ix < p.Wd;
            voff[q][t] = ok ? (unsigned)(((long long)iy * p.Wd + ix) * p.ldx + chunk * 8) * 2u : 0x7fffff00u;   // >= x_bytes: reads as 0
        }
    }
    // ---- pointwise weight sub-slices (256 rows x 128 B) through registers, as in k_dwpw: rows [K/64][hi 64 | lo 64]
    const int wchunk = tid & 7, wrow0 = tid >> 3;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(static_cast<const char*>(p.W) + (long long)nt * TN * p.K * 4), 0, TN * p.K * 4, 0x00020000);
    int w_voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w_voff[i] = (wrow0 + 64 * i) * p.K * 4 + wchunk * 16;
    v4i wl[4];
    auto load_w = [&](int h) {
#pragma unroll
        for (int i = 0; i < 4; ++i) wl[i] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, w_voff[i], h * 128, 0);
    };
    auto store_w = [&](int h) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = wrow0 + 64 * i;
            const int key = ((r >> 1) & 1) | (((r >> 4) & 3) << 1);
            *reinterpret_cast<v4i*>(lds + X_LDS_W + (h & 1) * W_STAGE + r * 128 + ((wchunk ^ key) << 4)) = wl[i];
        }
    };

    // ---- consumer geometry (64 x 64 per wave, product transposed: a lane ends with 16 consecutive channels of a pixel)
    const int fr = lane & 15, kq = lane >> 4;
    int a_off[4], w_off[4], a_key[4], w_key[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int row = wm * 64 + mi * 16 + fr;
        a_off[mi] = row * 128;
        a_key[mi] = row & 7;
    }
#pragma unroll
    for (int nj = 0; nj < 4; ++nj) {
        const int row = wn * 64 + (fr >> 2) * 16 + nj * 4 + (fr & 3);
        w_off[nj] = X_LDS_W + row * 128;
        w_key[nj] = ((row >> 1) & 1) | (((row >> 4) & 3) << 1);
    }
    f32x4 acc[4][4];
    {
        const int nb = nt * TN + wn * 64 + kq * 16;
#pragma unroll
        for (int nj = 0; nj < 4; ++nj) {
            const float4 b = *reinterpret_cast<const float4*>(p.bias + nb + 4 * nj);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) acc[mi][nj] = f32x4{b.x, b.y, b.z, b.w};
        }
    }

    v4i raw[2][9];
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.X), 0, (int)p.x_bytes, 0x00020000);
    auto load_taps = [&](int s, int q) {
#pragma unroll
        for (int t = 0; t < 9; ++t) raw[q][t] = (DW_EXP & 4) ? v4i{0, 0, 0, 0} : __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)voff[q][t], s * 128, 0);
    };

    // depthwise 3x3 of K-step s for pixel q of this lane: fp32 sums of f16 x (f16 hi + f16 lo) products -> hi tile slot s & 1, lo tile
    auto produce_a = [&](int s, int q) {
        if constexpr ((DW_EXP & 2) != 0) {
#pragma unroll
            for (int t = 0; t < 9; ++t) asm volatile("" :: "v"(raw[q][t]));
            const int row = r0 + q * 64;
            const int sw = row * 128 + ((chunk ^ (row & 7)) << 4);
            *reinterpret_cast<v4i*>(lds + X_LDS_AH + (s & 1) * A_STAGE + sw) = v4i{0, 0, 0, 0};
            *reinterpret_cast<v4i*>(lds + X_LDS_AL + sw) = v4i{0, 0, 0, 0};
            return;
        }
        float o[8];
        const float* pf = reinterpret_cast<const float*>(lds + X_LDS_P + (s & (XP_RING - 1)) * PSTEP + chunk * (10 * 8 * 4));
        {
            const float4 b0 = DW_PARAM(float4, pf + 9 * 8), b1 = DW_PARAM(float4, pf + 9 * 8 + 4);
            o[0] = b0.x; o[1] = b0.y; o[2] = b0.z; o[3] = b0.w;
            o[4] = b1.x; o[5] = b1.y; o[6] = b1.z; o[7] = b1.w;
        }
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float4 w0 = DW_PARAM(float4, pf + t * 8), w1 = DW_PARAM(float4, pf + t * 8 + 4);
            const float wt[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int u = raw[q][t][j];        // (a copy first: __builtin_bit_cast of the vector-element expression itself read element 0 for every j)
                const h2 v = __builtin_bit_cast(h2, u);
                o[2 * j] = __builtin_fmaf((float)v[0], wt[2 * j], o[2 * j]);
                o[2 * j + 1] = __builtin_fmaf((float)v[1], wt[2 * j + 1], o[2 * j + 1]);
            }
        }
        float ol[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            o[i] = fmaxf(o[i], 0.f);
            ol[i] = o[i] - (float)(HT)o[i];
        }
        const int row = r0 + q * 64;
        const int sw = row * 128 + ((chunk ^ (row & 7)) << 4);
        Vec8<HT>::store(reinterpret_cast<HT*>(lds + X_LDS_AH + (s & 1) * A_STAGE + sw), o);
        Vec8<HT>::store(reinterpret_cast<HT*>(lds + X_LDS_AL + sw), ol);
    };

    // ---- prologue: taps and weights of slice 0 -> tiles(0), W hi(0) in LDS; then the taps of slice 1
    load_taps(0, 0);
    load_taps(0, 1);
    load_w(0);
    __syncthreads();                 // parameters of K-steps 0 .. 2 are in LDS
    produce_a(0, 0);
    produce_a(0, 1);
    store_w(0);
    if (nk > 1) { load_taps(1, 0); load_taps(1, 1); }

    for (int s = 0; s < nk; ++s) {
        const bool more = s + 1 < nk, more2 = s + 2 < nk;
        // ---- j = 0: weight slice hi (ring slot 0) x hi tile (slot s & 1) and x lo tile
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // own tile / weight / parameter writes are in LDS
        __builtin_amdgcn_s_barrier();
        load_w(2 * s + 1);
        load_p(s + 3);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            v8 wa[4], ah[4], al[4];
#pragma unroll
            for (int nj = 0; nj < 4; ++nj) wa[nj] = *reinterpret_cast<const v8*>(lds + w_off[nj] + (((kk * 4 + kq) ^ w_key[nj]) << 4));
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                ah[mi] = *reinterpret_cast<const v8*>(lds + X_LDS_AH + (s & 1) * A_STAGE + a_off[mi] + (((kk * 4 + kq) ^ a_key[mi]) << 4));
                al[mi] = *reinterpret_cast<const v8*>(lds + X_LDS_AL + a_off[mi] + (((kk * 4 + kq) ^ a_key[mi]) << 4));
            }
            // (the lo products first -- 2^-11 of the hi ones --, and the two products of one accumulator 16 MFMAs apart)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int nj = 0; nj < 4; ++nj) { if constexpr (!(DW_EXP & 1)) acc[mi][nj] = Half16<HT>::mfma(wa[nj], al[mi], acc[mi][nj]); else if constexpr (!(DW_EXP & 8)) asm volatile("" :: "v"(wa[nj]), "v"(al[mi])); }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int nj = 0; nj < 4; ++nj) { if constexpr (!(DW_EXP & 1)) acc[mi][nj] = Half16<HT>::mfma(wa[nj], ah[mi], acc[mi][nj]); else if constexpr (!(DW_EXP & 8)) asm volatile("" :: "v"(wa[nj]), "v"(ah[mi])); }
        }
        store_w(2 * s + 1);
        store_p(s + 3);
        // ---- j = 1: weight slice lo (ring slot 1) x hi tile; the depthwise slice s + 1 -> the other hi slot, the lo tile
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (more) load_w(2 * s + 2);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            v8 wa[4], ah[4];
#pragma unroll
            for (int nj = 0; nj < 4; ++nj) wa[nj] = *reinterpret_cast<const v8*>(lds + W_STAGE + w_off[nj] + (((kk * 4 + kq) ^ w_key[nj]) << 4));
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
                ah[mi] = *reinterpret_cast<const v8*>(lds + X_LDS_AH + (s & 1) * A_STAGE + a_off[mi] + (((kk * 4 + kq) ^ a_key[mi]) << 4));
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int nj = 0; nj < 4; ++nj) { if constexpr (!(DW_EXP & 1)) acc[mi][nj] = Half16<HT>::mfma(wa[nj], ah[mi], acc[mi][nj]); else if constexpr (!(DW_EXP & 8)) asm volatile("" :: "v"(wa[nj]), "v"(ah[mi])); }
            if (more) {
                produce_a(s + 1, kk);
                if (more2) load_taps(s + 2, kk);
            }
        }
        if (more) store_w(2 * s + 2);
    }

    // ---- epilogue: ReLU, split, store (bias was the accumulators' start value)
    const int nbase = nt * TN + wn * 64 + kq * 16;
    if (nbase + 16 <= p.N) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int m = mt * TM + wm * 64 + mi * 16 + fr;
            if (m < p.M) {
                float lo[8], hi[8];
#pragma unroll
                for (int nj = 0; nj < 2; ++nj)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        lo[nj * 4 + r] = fmaxf(acc[mi][nj][r], 0.f);
                        hi[nj * 4 + r] = fmaxf(acc[mi][2 + nj][r], 0.f);
                    }
                HT* cp = static_cast<HT*>(p.C) + (long long)m * p.ldc + nbase;
                Vec8<HT>::store(cp, lo);
                Vec8<HT>::store(cp + 8, hi);
                if (p.C_lo) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) { lo[i] -= (float)(HT)lo[i]; hi[i] -= (float)(HT)hi[i]; }
                    HT* cl = static_cast<HT*>(p.C_lo) + (long long)m * p.ldc + nbase;
                    Vec8<HT>::store(cl, lo);
                    Vec8<HT>::store(cl + 8, hi);
                }
            }
        }
    }
}

int launch_dwpw_x(const DwPwArgs& a, hipStream_t s) {
    constexpr int lds_bytes = X_LDS_P + XP_RING * FP_STEP;
    static_assert(lds_bytes <= 160 * 1024, "k_dwpw_x LDS");
    AVL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dwpw_x), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(k_dwpw_x, dim3(8 * a.per_xcd * a.ntiles), dim3(512), lds_bytes, s, a);
    AVL_LAUNCH_CHECK();
    return AVL_OK;
}

// ---------------------------------------------------------------------------------------------
// k_dwpw_xs: k_dwpw_x for an input that is itself split (hi + lo f16 planes) -- the "mixed" decoder's two refine blocks
// (decoder.py:33-43: depthwise 3x3 pad 0 + BN + ReLU, 1x1 + BN + ReLU on concat(upsampled ASPP, low-level) and on its result) and the
// ASPP branches of the split16 plan.  Unfused, the depthwise result went to HBM as two planes (+ FP4 copies) and came back into the
// 1x1 GEMM: 2 x 265 MB + 2 x 131 MB per frame at 1080p for 2.7 GMAC.
//   depthwise stage: (xh + xl) . w with fp32 weights: one v_fma_mix_f32 per tap, channel and plane, the lo plane's products first;
//   the tap registers hold ONE pixel's nine taps of both planes (72 VGPRs, as many as k_dwpw_x's two pixels of one plane): pixel 0 of
//   slice s + 1 is produced during sub-step j = 0 of step s and pixel 1 during j = 1, each pixel's taps requested as soon as the other
//   pixel's registers are free (48 MFMAs per wave ahead of their use); the lo tile has two slots like the hi tile for that;
//   weight sub-slices and depthwise parameters come by LDS-DMA (no staging registers: 246 VGPRs).
//   LDS: weight ring 2 x 32 KB | hi tiles 2 x 16 KB | lo tiles 2 x 16 KB | parameter ring 4 x 2.5 KB = 138 KB.
constexpr int S_LDS_W = 0, S_LDS_AH = 2 * W_STAGE, S_LDS_AL = S_LDS_AH + 2 * A_STAGE, S_LDS_P = S_LDS_AL + 2 * A_STAGE;
template <bool CLS>
__global__ void __launch_bounds__(512) k_dwpw_xs(DwPwArgs p) {
    constexpr int PSTEP = FP_STEP;          // bytes of depthwise parameters per K-step
    typedef f16 HT;
    typedef typename Half16<HT>::v8 v8;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int nt = blockIdx.x % p.ntiles;
    const int bm = blockIdx.x / p.ntiles, slot = (bm & 7) * p.per_xcd + (bm >> 3);
    if (slot >= p.mtiles) return;
    const int mt = p.order[slot];
    const int nk = p.K / 64;

    const unsigned lds0 = lds_addr(lds);
    // ---- depthwise parameters: K-step s -> ring slot s & 3; 2560 bytes = 160 lanes x 16 bytes, by LDS-DMA (waves 0 .. 2)
    auto dma_p = [&](int s) {
        if (tid < PSTEP / 16 && s < nk)
            glds16_saddr(reinterpret_cast<const char*>(p.dwp) + s * PSTEP, (unsigned)tid * 16u, lds0 + S_LDS_P + (s & (XP_RING - 1)) * PSTEP + wave * 1024);
    };
    for (int s = 0; s < 3; ++s) dma_p(s);
    // ---- producer geometry: lane -> 8-channel chunk of two pixels (rows r0, r0 + 64 of the tile).  A tap's byte offset is the pixel's
    // base (tap 0, possibly outside the image: signed) + a wave-uniform step; a 9-bit mask says which taps exist (the others get an
    // out-of-range offset and read as 0): 4 VGPRs instead of k_dwpw_x's 18, which this kernel does not have
    const int chunk = tid & 7, r0 = tid >> 3;
    int vbase[2];
    unsigned vmask[2];
    // 8 x 16 blocks (tiles_x > 0; stride-1 taps): a tile's taps fall on 10 x 18 input pixels instead of 3 x 130 -- the row-segment form
    // fetched its input 3x from HBM (PMC: 585 MB per launch against 198 MB of input, L2 hit 0.52)
    const int tty = p.tiles_x > 0 ? mt / p.tiles_x : 0, ttx = p.tiles_x > 0 ? mt - tty * p.tiles_x : 0;
    const int OHt = p.M / p.OW;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int ml = r0 + q * 64;
        int y, x;
        bool live;
        if (p.tiles_x > 0) {
            y = tty * 8 + (ml >> 4); x = ttx * 16 + (ml & 15);
            live = (y < OHt) & (x < p.OW);
        } else {
            const int m = mt * TM + ml;
            y = m / p.OW; x = m - y * p.OW;
            live = m < p.M;
        }
        vbase[q] = (int)((((long long)(y - p.pad) * p.Wd + (x - p.pad)) * p.ldx + chunk * 8) * 2);
        unsigned mk = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int iy = y - p.pad + (t / 3) * p.dil, ix = x - p.pad + (t % 3) * p.dil;
            mk |= (unsigned)(live & (iy >= 0) & (iy < p.H) & (ix >= 0) & (ix < p.Wd)) << t;
        }
        vmask[q] = mk;
    }
    const int tstep_x = p.dil * p.ldx * 2, tstep_y = p.dil * p.Wd * p.ldx * 2;
    // ---- pointwise weight sub-slices (256 rows x 128 B: rows [K/64][hi 64 | lo 64]) by LDS-DMA: instruction i of a wave brings rows
    // 64 i + 8 wave .. + 7 (1 KB, linear in LDS); the swizzle goes on the SOURCE chunk (its key does not depend on i)
    const char* wbase = static_cast<const char*>(p.W) + (long long)nt * TN * p.K * 4;
    unsigned w_voff;
    {
        const int r = wave * 8 + (lane >> 3);
        const int key = ((r >> 1) & 1) | (((r >> 4) & 3) << 1);
        w_voff = (unsigned)(r * p.K * 4 + (((lane & 7) ^ key) << 4));
    }
    const int w_rows64 = 64 * p.K * 4;
    auto dma_w = [&](int h) {
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16_saddr(wbase + h * 128 + i * w_rows64, w_voff, lds0 + S_LDS_W + (h & 1) * W_STAGE + i * 8192 + wave * 1024);
    };
    // (the swizzle keys do not depend on mi / nj: row & 7 = fr & 7 for the tiles, bits (fr >> 1) & 1 and (fr >> 2) & 3 for the weights --
    //  one address per K half each, the rest are immediate offsets)
    const int fr = lane & 15, kq = lane >> 4;
    int a_sw[2], w_sw[2];
    {
        const int a_key = fr & 7, w_key = ((fr >> 1) & 1) | (((fr >> 2) & 3) << 1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            a_sw[kk] = (wm * 64 + fr) * 128 + (((kk * 4 + kq) ^ a_key) << 4);
            w_sw[kk] = S_LDS_W + (wn * 64 + (fr >> 2) * 16 + (fr & 3)) * 128 + (((kk * 4 + kq) ^ w_key) << 4);
        }
    }
    f32x4 acc[4][4];
    {
        const int nb = nt * TN + wn * 64 + kq * 16;
#pragma unroll
        for (int nj = 0; nj < 4; ++nj) {
            const float4 b = *reinterpret_cast<const float4*>(p.bias + nb + 4 * nj);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) acc[mi][nj] = f32x4{b.x, b.y, b.z, b.w};
        }
    }

    // one pixel's taps of both planes (all loads compiler-visible: hipcc counts the vmcnt waits, see k_dwpw)
    v4i rh[9], rl[9];
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.X), 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t xlrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.X_lo), 0, (int)p.x_bytes, 0x00020000);
    auto load_taps = [&](int s, int q) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int off = (vmask[q] >> t) & 1u ? vbase[q] + (t / 3) * tstep_y + (t % 3) * tstep_x : 0x7fffff00;   // >= x_bytes: reads as 0
            rh[t] = (DW_EXP & 4) ? v4i{off, 0, 0, 0} : __builtin_amdgcn_raw_buffer_load_b128(xrsrc, off, s * 128, 0);
            rl[t] = (DW_EXP & 4) ? v4i{off, 0, 0, 0} : __builtin_amdgcn_raw_buffer_load_b128(xlrsrc, off, s * 128, 0);
        }
    };

    // depthwise 3x3 of K-step s for pixel q of this lane -> hi / lo tile slot s & 1
    auto produce_a = [&](int s, int q) {
        if constexpr ((DW_EXP & 2) != 0) {
#pragma unroll
            for (int t = 0; t < 9; ++t) asm volatile("" :: "v"(rh[t]), "v"(rl[t]));
            const int row = r0 + q * 64;
            const int sw = (s & 1) * A_STAGE + row * 128 + ((chunk ^ (row & 7)) << 4);
            *reinterpret_cast<v4i*>(lds + S_LDS_AH + sw) = v4i{0, 0, 0, 0};
            *reinterpret_cast<v4i*>(lds + S_LDS_AL + sw) = v4i{0, 0, 0, 0};
            return;
        }
        float o[8];
        const float* pf = reinterpret_cast<const float*>(lds + S_LDS_P + (s & (XP_RING - 1)) * PSTEP + chunk * (10 * 8 * 4));
        {
            const float4 b0 = DW_PARAM(float4, pf + 9 * 8), b1 = DW_PARAM(float4, pf + 9 * 8 + 4);
            o[0] = b0.x; o[1] = b0.y; o[2] = b0.z; o[3] = b0.w;
            o[4] = b1.x; o[5] = b1.y; o[6] = b1.z; o[7] = b1.w;
        }
        float wt[9][8];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float4 w0 = DW_PARAM(float4, pf + t * 8), w1 = DW_PARAM(float4, pf + t * 8 + 4);
            wt[t][0] = w0.x; wt[t][1] = w0.y; wt[t][2] = w0.z; wt[t][3] = w0.w;
            wt[t][4] = w1.x; wt[t][5] = w1.y; wt[t][6] = w1.z; wt[t][7] = w1.w;
        }
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        // (the lo plane's products first: they are 2^-11 of the others)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int u = rl[t][j];            // (a copy first: __builtin_bit_cast of the vector-element expression itself read element 0 for every j)
                const h2 v = __builtin_bit_cast(h2, u);
                o[2 * j] = __builtin_fmaf((float)v[0], wt[t][2 * j], o[2 * j]);
                o[2 * j + 1] = __builtin_fmaf((float)v[1], wt[t][2 * j + 1], o[2 * j + 1]);
            }
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int u = rh[t][j];
                const h2 v = __builtin_bit_cast(h2, u);
                o[2 * j] = __builtin_fmaf((float)v[0], wt[t][2 * j], o[2 * j]);
                o[2 * j + 1] = __builtin_fmaf((float)v[1], wt[t][2 * j + 1], o[2 * j + 1]);
            }
        float ol[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            o[i] = fmaxf(o[i], 0.f);
            ol[i] = o[i] - (float)(HT)o[i];
        }
        const int row = r0 + q * 64;
        const int sw = (s & 1) * A_STAGE + row * 128 + ((chunk ^ (row & 7)) << 4);
        Vec8<HT>::store(reinterpret_cast<HT*>(lds + S_LDS_AH + sw), o);
        Vec8<HT>::store(reinterpret_cast<HT*>(lds + S_LDS_AL + sw), ol);
    };

    // ---- prologue: parameters and W hi(0) by DMA, slice 0 of both pixels; then pixel 0's taps of slice 1
    dma_w(0);
    load_taps(0, 0);
    asm volatile("s_waitcnt vmcnt(18)" ::: "memory");          // (the 18 tap loads are younger than every DMA above)
    __syncthreads();                 // parameters of K-steps 0 .. 2 and W hi(0) are in LDS
    produce_a(0, 0);
    load_taps(0, 1);
    produce_a(0, 1);
    if (nk > 1) load_taps(1, 0);

    // vmcnt: hipcc counts its own (tap) loads; the DMAs are invisible to it, which only ever makes its waits longer.  The DMAs of a
    // sub-step are issued at its top -- the 18 tap loads that follow are the only younger operations -- and awaited at its end.
    // (Tried and measured equal or 4 % slower, profiles/r05/dwpw_phase_ablation.log: the two waves of a SIMD running the halves of a
    //  sub-step in opposite order -- depthwise work first / matrix work first --, in three wave groupings.  Written as a three-slot
    //  runtime loop that variant also met a hipcc hazard miss: the accumulators rotate through v_mov_b64 copies at the loop latch, and
    //  the first copy read the last MFMA's result three instructions after its issue, without an s_nop: stale acc[3][3][0:1].
    //  Also tried: the depthwise VALU work in the MFMAs' issue shadows -- one scheduling region per sub-step with
    //  sched_group_barrier(MFMA 1 : VALU 4 / 6 / 8): +1..+4 % time.)
    for (int s = 0; s < nk; ++s) {
        const bool more = s + 1 < nk, more2 = s + 2 < nk;
        const int ts = (s & 1) * A_STAGE;
        auto mfma_hi = [&](int kk) {            // weight slice hi (ring slot 0) x lo tile and x hi tile
            v8 wa[4], ah[4], al[4];
#pragma unroll
            for (int nj = 0; nj < 4; ++nj) wa[nj] = *reinterpret_cast<const v8*>(lds + (kk ? w_sw[1] : w_sw[0]) + nj * 512);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                ah[mi] = *reinterpret_cast<const v8*>(lds + S_LDS_AH + ts + (kk ? a_sw[1] : a_sw[0]) + mi * 2048);
                al[mi] = *reinterpret_cast<const v8*>(lds + S_LDS_AL + ts + (kk ? a_sw[1] : a_sw[0]) + mi * 2048);
            }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int nj = 0; nj < 4; ++nj) { if constexpr (!(DW_EXP & 1)) acc[mi][nj] = Half16<HT>::mfma(wa[nj], al[mi], acc[mi][nj]); else if constexpr (!(DW_EXP & 8)) asm volatile("" :: "v"(wa[nj]), "v"(al[mi])); }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int nj = 0; nj < 4; ++nj) { if constexpr (!(DW_EXP & 1)) acc[mi][nj] = Half16<HT>::mfma(wa[nj], ah[mi], acc[mi][nj]); else if constexpr (!(DW_EXP & 8)) asm volatile("" :: "v"(wa[nj]), "v"(ah[mi])); }
        };
        auto mfma_lo = [&](int kk) {            // weight slice lo (ring slot 1) x hi tile
            v8 wa[4], ah[4];
#pragma unroll
            for (int nj = 0; nj < 4; ++nj) wa[nj] = *reinterpret_cast<const v8*>(lds + W_STAGE + (kk ? w_sw[1] : w_sw[0]) + nj * 512);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) ah[mi] = *reinterpret_cast<const v8*>(lds + S_LDS_AH + ts + (kk ? a_sw[1] : a_sw[0]) + mi * 2048);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int nj = 0; nj < 4; ++nj) { if constexpr (!(DW_EXP & 1)) acc[mi][nj] = Half16<HT>::mfma(wa[nj], ah[mi], acc[mi][nj]); else if constexpr (!(DW_EXP & 8)) asm volatile("" :: "v"(wa[nj]), "v"(ah[mi])); }
        };
        // ---- j = 0: 64 MFMAs; pixel 0 of the depthwise slice s + 1, then the request for pixel 1's taps
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // own tile writes are in LDS
        __builtin_amdgcn_s_barrier();
        dma_p(s + 3);
        dma_w(2 * s + 1);
        mfma_hi(0);
        __builtin_amdgcn_sched_barrier(0);          // (keeps the next half's fragment reads below the depthwise work: registers)
        if (more) { produce_a(s + 1, 0); load_taps(s + 1, 1); }
        __builtin_amdgcn_sched_barrier(0);
        mfma_hi(1);
        if (more) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // ---- j = 1: 32 MFMAs; pixel 1 of the depthwise slice s + 1, then the request for pixel 0's taps of slice s + 2
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (more) dma_w(2 * s + 2);
        mfma_lo(0);
        __builtin_amdgcn_sched_barrier(0);
        if (more) { produce_a(s + 1, 1); if (more2) load_taps(s + 2, 0); }
        __builtin_amdgcn_sched_barrier(0);
        mfma_lo(1);
        if (more2) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

    // ---- epilogue: ReLU, split, store (bias was the accumulators' start value)
    // (lane geometry from an opaque copy: hipcc otherwise computes the 64-bit store offsets in front of the K loop and spills them)
    int elane = lane;
    asm volatile("" : "+v"(elane));
    const int efr = elane & 15;
    if constexpr (CLS) {
        // ---- the classifier on this tile's 128 x 256 result.  (1) ReLU, hi / lo split, into LDS as four 64-channel A tiles per plane (hi at 0
        // over the weight ring, lo at 64 KB over the depthwise tiles: both are done with), in the layout the fragment reads above use;
        // (2) wave w takes pixels 16 w .. 16 w + 15: logits[class][pixel] = Wc . y in three passes (Wh.yl, Wl.yh, Wh.yh), weights straight
        // from L2 (32 KB); (3) a lane ends with classes 4 kq .. + 3 and 16 + 4 kq .. + 3 of pixel fr: fp32 logits out, arg-max across the
        // four lanes of a pixel (first maximal index wins, a NaN counts as maximal, as torch.argmax)
        const int ekq = elane >> 4;
        __syncthreads();
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int row = wm * 64 + mi * 16 + efr;
            float v0[8], v1[8], l0[8], l1[8];
#pragma unroll
            for (int nj = 0; nj < 2; ++nj)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v0[nj * 4 + r] = fmaxf(acc[mi][nj][r], 0.f);
                    v1[nj * 4 + r] = fmaxf(acc[mi][2 + nj][r], 0.f);
                }
#pragma unroll
            for (int i = 0; i < 8; ++i) { l0[i] = v0[i] - (float)(HT)v0[i]; l1[i] = v1[i] - (float)(HT)v1[i]; }
            const int base = wn * A_STAGE + row * 128;
            const int o0 = base + (((2 * ekq) ^ (row & 7)) << 4), o1 = base + (((2 * ekq + 1) ^ (row & 7)) << 4);
            Vec8<HT>::store(reinterpret_cast<HT*>(lds + o0), v0);
            Vec8<HT>::store(reinterpret_cast<HT*>(lds + o1), v1);
            Vec8<HT>::store(reinterpret_cast<HT*>(lds + 65536 + o0), l0);
            Vec8<HT>::store(reinterpret_cast<HT*>(lds + 65536 + o1), l1);
        }
        __syncthreads();
        f32x4 a2[2];
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            const float4 b = *reinterpret_cast<const float4*>(p.bc + t2 * 16 + 4 * ekq);
            a2[t2] = f32x4{b.x, b.y, b.z, b.w};
        }
        const int crow = wave * 16 + efr;
        const char* wcl = static_cast<const char*>(p.Wc) + ((long long)efr * p.N + ekq * 8) * 2;          // + (t2 * 16 rows, K offset) ; lo plane 32 rows further
        const long long wplane = 32LL * p.N * 2;
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int off = st * A_STAGE + crow * 128 + (((kk * 4 + ekq) ^ (efr & 7)) << 4);
                const v8 yh = *reinterpret_cast<const v8*>(lds + off), yl = *reinterpret_cast<const v8*>(lds + 65536 + off);
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) {
                    const char* wp = wcl + ((long long)t2 * 16 * p.N + st * 64 + kk * 32) * 2;
                    const v8 wh = *reinterpret_cast<const v8*>(wp), wl = *reinterpret_cast<const v8*>(wp + wplane);
                    a2[t2] = Half16<HT>::mfma(wh, yl, a2[t2]);
                    a2[t2] = Half16<HT>::mfma(wl, yh, a2[t2]);
                    a2[t2] = Half16<HT>::mfma(wh, yh, a2[t2]);
                }
            }
        // pixel of this lane
        int m;
        {
            const int ml = crow;
            if (p.tiles_x > 0) {
                const int y = tty * 8 + (ml >> 4), x = ttx * 16 + (ml & 15);
                m = ((y < OHt) & (x < p.OW)) ? y * p.OW + x : p.M;
            } else {
                m = mt * TM + ml;
            }
        }
        // (a lane whose classes 4 kq .. and 16 + 4 kq .. all lie beyond ncls -- kq >= 2 at five classes -- takes no part: index kNone)
        constexpr int kNone = 1 << 20;
        float best = 0.f;
        int bi = kNone;
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int cls = t2 * 16 + 4 * ekq + r;
                const float v = a2[t2][r];
                if (cls < p.ncls) {
                    if (m < p.M) p.LG[(long long)m * p.ncls + cls] = v;
                    if (bi == kNone || v > best || (v != v && best == best)) { best = v; bi = cls; }
                }
            }
#pragma unroll
        for (int sh = 16; sh <= 32; sh <<= 1) {
            const float ov = __shfl_xor(best, sh, 64);
            const int oi = __shfl_xor(bi, sh, 64);
            const bool onan = ov != ov, bnan = best != best;
            const bool better = onan ? (!bnan || oi < bi) : (!bnan && (ov > best || (ov == best && oi < bi)));
            if (oi != kNone && (bi == kNone || better)) { best = ov; bi = oi; }
        }
        if (ekq == 0 && m < p.M) p.labels[m] = (unsigned char)bi;
        return;
    }
    const int nbase = nt * TN + wn * 64 + (elane >> 4) * 16;
    if (nbase + 16 <= p.N) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            int m = mt * TM + wm * 64 + mi * 16 + efr;
            if (p.tiles_x > 0) {           // row wm * 64 + mi * 16 + efr of the tile = block row wm * 4 + mi, column efr
                const int y = tty * 8 + wm * 4 + mi, x = ttx * 16 + efr;
                m = ((y < OHt) & (x < p.OW)) ? y * p.OW + x : p.M;
            }
            if (m < ((DW_EXP & 128) ? (acc[mi][0][0] == 12345.f ? 1 : 0) : p.M)) {
                float lo[8], hi[8];
#pragma unroll
                for (int nj = 0; nj < 2; ++nj)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        lo[nj * 4 + r] = fmaxf(acc[mi][nj][r], 0.f);
                        hi[nj * 4 + r] = fmaxf(acc[mi][2 + nj][r], 0.f);
                    }
                HT* cp = static_cast<HT*>(p.C) + (long long)m * p.ldc + nbase;
                Vec8<HT>::store(cp, lo);
                Vec8<HT>::store(cp + 8, hi);
                if (p.C_lo) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) { lo[i] -= (float)(HT)lo[i]; hi[i] -= (float)(HT)hi[i]; }
                    HT* cl = static_cast<HT*>(p.C_lo) + (long long)m * p.ldc + nbase;
                    Vec8<HT>::store(cl, lo);
                    Vec8<HT>::store(cl + 8, hi);
                }
            }
        }
    }
}

template <bool CLS>
int launch_dwpw_xs(const DwPwArgs& a, hipStream_t s) {
    constexpr int lds_bytes = S_LDS_P + XP_RING * FP_STEP;
    static_assert(lds_bytes <= 160 * 1024 && S_LDS_AH == 65536 && 4 * A_STAGE == 65536, "k_dwpw_xs LDS (the classifier epilogue keeps y hi at 0 and y lo at 64 KB, 64 KB each)");
    AVL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dwpw_xs<CLS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(k_dwpw_xs<CLS>, dim3(8 * a.per_xcd * a.ntiles), dim3(512), lds_bytes, s, a);
    AVL_LAUNCH_CHECK();
    return AVL_OK;
}

template <typename HT, int WSUB>
int launch_dwpw_typed(const DwPwArgs& a, int mtiles, hipStream_t s) {
    const int lds_bytes = LDS_P + (a.K / 64) * P_STEP;
    AVL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dwpw<HT, WSUB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL((k_dwpw<HT, WSUB>), dim3(8 * a.per_xcd * a.ntiles), dim3(512), lds_bytes, s, a);
    AVL_LAUNCH_CHECK();
    return AVL_OK;
}

}  // namespace

int validate_dwpw(const avl_seg_op& op) {
    AVL_REQUIRE(is_half(op.dtype), "fused depthwise+pointwise needs a 16-bit activation type");
    AVL_REQUIRE(op.in && op.out && op.weight && op.bias && op.in2, "dwpw has NULL buffers");
    AVL_REQUIRE(!op.w_split || op.dtype == AVL_F16, "split weights need AVL_F16 activations");
    AVL_REQUIRE((!op.in2_lo || op.out_f32) && (!op.out_lo || op.w_split), "dwpw: the output may be split only with w_split (in2_lo: the classifier's bias, with out_f32 only)");
    if (op.out_f32) {
        AVL_REQUIRE(op.w_split == 3 && op.in_lo && !op.out_lo && op.out_c == TN && op.in3 && op.in2_lo && op.out_mx && op.in3_c >= 1 && op.in3_c <= 32 && op.out_ld == op.in3_c,
                    "dwpw with out_f32 (classifier + arg-max in the epilogue): needs a split input, w_split 3, out_c 256, in3 = f16 [2][32][256] weights, in2_lo = "
                    "fp32 bias[32], out = fp32 logits [rows][in3_c <= 32] (out_ld = in3_c), out_mx = uint8 labels[rows], no out_lo");
        AVL_REQUIRE((reinterpret_cast<uintptr_t>(op.in3) | reinterpret_cast<uintptr_t>(op.in2_lo)) % 16 == 0 && reinterpret_cast<uintptr_t>(op.out) % 4 == 0, "dwpw classifier operands unaligned");
    }
    AVL_REQUIRE(op.w_split != 2, "dwpw: w_split 2 (f16 depthwise weight pairs) was replaced by w_split 3 (fp32 depthwise weights)");
    AVL_REQUIRE(!op.in_lo || op.w_split == 3, "dwpw: a split input (in_lo) needs the exact depthwise stage (w_split 3: k_dwpw_xs)");
    AVL_REQUIRE(op.w_layout == 0 || (op.w_layout == 1 && op.in_lo && op.w_split == 3), "dwpw: w_layout 1 (8 x 16-pixel tiles) exists for the split-input kernel only");
    AVL_REQUIRE((reinterpret_cast<uintptr_t>(op.out_lo) | reinterpret_cast<uintptr_t>(op.in_lo)) % 16 == 0, "dwpw low planes must be 16-byte aligned");
    const int M = op.out_h * op.out_w, K = op.in_c, N = op.out_c;
    AVL_REQUIRE(op.stride == 1 && op.ksize == 3 && op.dil >= 1 && op.pad >= 0 && op.out_h == op.in_h + 2 * op.pad - 2 * op.dil &&
                    op.out_w == op.in_w + 2 * op.pad - 2 * op.dil && M > 0,
                "dwpw geometry: 3x3, stride 1, output = input + 2 pad - 2 dilation");
    AVL_REQUIRE(K % 64 == 0 && K <= 2048, "dwpw K = %d (multiple of 64, <= 2048: the depthwise parameters live in LDS)", K);
    AVL_REQUIRE(N % 16 == 0 && op.w_rows >= (N + TN - 1) / TN * TN, "dwpw N = %d / weight rows %d", N, op.w_rows);
    AVL_REQUIRE(op.in_ld >= K && (op.in_ld * 2) % 16 == 0, "dwpw in_ld %d", op.in_ld);
    if (!op.out_f32) AVL_REQUIRE(op.out_ld >= N && (op.out_ld * 2) % 16 == 0, "dwpw out_ld %d", op.out_ld);
    AVL_REQUIRE(op.in_rows >= op.in_h * op.in_w && op.out_rows >= M, "dwpw rows");
    AVL_REQUIRE((long long)op.in_rows * op.in_ld * 2 < 0x7fffff00LL, "dwpw input larger than a buffer descriptor's range");
    AVL_REQUIRE((reinterpret_cast<uintptr_t>(op.in) | reinterpret_cast<uintptr_t>(op.weight) | reinterpret_cast<uintptr_t>(op.out) |
                 reinterpret_cast<uintptr_t>(op.bias) | reinterpret_cast<uintptr_t>(op.in2)) % 16 == 0, "dwpw buffers must be 16-byte aligned");
    return AVL_OK;
}

int launch_dwpw(const avl_seg_op& op, hipStream_t s) {
    DwPwArgs a;
    a.X = op.in; a.X_lo = op.in_lo; a.W = op.weight;
    a.Wc = nullptr; a.bc = nullptr; a.LG = nullptr; a.labels = nullptr; a.ncls = 0;
    a.tiles_x = (op.w_layout == 1) ? (op.out_w + 15) / 16 : 0; a.bias = op.bias; a.dwp = static_cast<const uint32_t*>(op.in2); a.C = op.out;
    a.H = op.in_h; a.Wd = op.in_w; a.OW = op.out_w; a.ldx = op.in_ld; a.ldc = op.out_ld; a.pad = op.pad;
    a.M = op.out_h * op.out_w; a.N = op.out_c; a.K = op.in_c; a.dil = op.dil;
    a.ntiles = (a.N + TN - 1) / TN;
    a.x_bytes = (unsigned)((long long)op.in_rows * op.in_ld * 2);
    const int mtiles = a.tiles_x > 0 ? a.tiles_x * ((op.out_h + 7) / 8) : (a.M + TM - 1) / TM;     // (the host's visiting order in2 has this many entries)
    a.mtiles = mtiles;
    a.per_xcd = (mtiles + 7) / 8;
    a.order = reinterpret_cast<const int*>(a.dwp + (a.K / 64) * ((op.w_split == 3 ? FP_STEP : P_STEP) / 4));
    a.C_lo = op.out_lo;
    // exact depthwise stage (fp32 depthwise weights: pack_dw_f32; split tile, three MFMA passes); with in_lo the input is two planes (the "mixed"
    // decoder, the split16 ASPP)
    if (op.w_split == 3 && op.out_f32) {         // + the classifier and the arg-max in the epilogue (validate_dwpw)
        a.Wc = op.in3; a.bc = static_cast<const float*>(op.in2_lo); a.LG = static_cast<float*>(op.out); a.labels = static_cast<unsigned char*>(op.out_mx);
        a.ncls = op.in3_c;
        return launch_dwpw_xs<true>(a, s);
    }
    if (op.w_split == 3) return op.in_lo ? launch_dwpw_xs<false>(a, s) : launch_dwpw_x(a, s);
    if (op.w_split) return launch_dwpw_typed<f16, 2>(a, mtiles, s);
    return op.dtype == AVL_F16 ? launch_dwpw_typed<f16, 1>(a, mtiles, s) : launch_dwpw_typed<bf16, 1>(a, mtiles, s);
}

}  // namespace avl
