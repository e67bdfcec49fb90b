// One torchvision Bottleneck (backbone/resnet.py:24-43 over torchvision's Bottleneck: conv1 1x1 -> BN -> ReLU -> grouped 3x3 ->
// BN -> ReLU -> conv3 1x1 -> BN -> (+ identity | + downsample 1x1) -> ReLU) as ONE persistent kernel, "mixed" precision.
//
// What the three-launch form pays for in layer1 (width 128, 129 600 pixels at 1080p) is bytes: conv1's output, the 3x3's output as two
// planes, the block input a second time for the residual -- 0.35 ms for 8 % of the frame's flops, every launch HBM-bound.  Here a
// workgroup owns an 8 x 16 output-pixel tile for ALL channels and the intermediates never leave the CU:
//   conv1   on the 10 x 18 halo (180 pixels, 12 MFMA row tiles): X tile [192 px][CIN] f16 in LDS (LDS-DMA, 64-channel slabs of 128-byte
//           rows, XOR-swizzled on the DMA source), weights as f16 pairs hi + lo in MFMA fragment order straight from L2 into
//           registers.  Wave w owns output channels 16 w .. 16 w + 15 for all 12 row tiles.
//   t1      = ReLU(conv1 + b1), zero outside the image (the 3x3 pads t1, not x), f16 hi (+ lo plane where LDS allows) in LDS as
//           [halo px][128 ch], 256-byte rows, 16-byte chunks XOR-swizzled with px & 15.
//   conv2   grouped 3x3 as dense block-diagonal 16-CHANNEL windows: K = 32 = TWO TAPS x 16 channels, so nine taps are 5 MFMAs per
//           window and row tile (the 32-channel windows of k_gconv_mfma need 18).  Window w is exactly the 16 channels wave w
//           produced in conv1, for all pixels: conv1 -> conv2 needs NO barrier, a wave reads back only what it wrote itself.
//   t2      = ReLU(conv2 + b2) as hi + lo f16 planes, written IN PLACE over the wave's own t1 entries (its results sit in
//           registers by then).
//   conv3   [256 out][128] + bias + residual (the block input re-read from L2 in accumulator layout) or + the downsample 1x1 as
//           two more K steps on the X tile still in LDS; wave w owns 32 output channels (two interleaved n-tiles: a lane holds 8
//           consecutive channels of one pixel = one 16-byte store per plane).
// Per tile: two barriers (t2 visible | t2 read + next X tile landed).  The next tile's X DMA is issued right behind the first
// barrier and has all of conv3 to land.  Every product runs Wh.xh + Wl.xh (+ Wh.xl where the operand has a lo plane) on
// v_mfma_f32_16x16x32_f16 with fp32 accumulation.
#include <cstring>

#include "seg_types.h"

#ifndef BN_EXP
#define BN_EXP 0          // hook for same-box A/B builds of tuning variants (tools/ab_bottleneck.sh: make EXPFLAGS=-DBN_EXP=n OUT=... BUILD=...); 0 = what ships
#endif

namespace avl {
namespace {

constexpr int BT_H = 8, BT_W = 16;                  // output tile
constexpr int BH_W = BT_W + 2, BH_H = BT_H + 2;     // halo tile
constexpr int BN_HALO = BH_W * BH_H;                // 180 pixels
constexpr int BN_M1 = 12;                           // conv1 row tiles (192 rows, the last 12 are padding)
constexpr int BN_WIDTH = 128, BN_COUT = 256;
constexpr int XSLAB = 192 * 128;                    // one 64-channel slab of the X tile
constexpr int PLANE1 = 192 * 256;                   // t1 plane (hi or lo)
constexpr int PLANE2 = 128 * 256;                   // t2 plane

struct BnArgs {
    const f16* x;        // block input [rows][in_ld], hi plane
    const f16* x_lo;     // its lo plane (enters the residual sum only) or NULL
    f16* out;
    f16* out_lo;         // NULL: single plane
    const f16* w1;       // [n 8][ks CIN/32][hi, lo][lane 64][8]
    const f16* w2;       // [window 8][ks 5][hi, lo][lane 64][8]
    const f16* w3;       // [wave 8][ks 4 (+ CIN/32 downsample steps)][nj 2][hi, lo][lane 64][8]
    const float* b1;     // [128]
    const float* b2;     // [128]
    const float* b3;     // [256] (downsample bias folded in)
    int H, W, in_ld, out_ld;
    int tiles_x, ntiles;
    int out_bytes;       // one output plane (buffer stores are range-checked against it)
    unsigned long long* dbg;   // experiments build, AVL_BN_PROBE=1: per-wave cycle sums of the tile loop's phases (host-visible memory), else NULL
};

template <int CIN, bool DS, bool T1LO>
struct BnLayout {
    static constexpr int NCH = CIN / 64;                          // 64-channel slabs of the X tile
    static constexpr int XTILE = NCH * XSLAB;
    static constexpr int XBYTES = (DS ? 2 : 1) * XTILE;           // DS: the X tile stays for conv3's downsample steps -> two of them
    static constexpr int T1LO_OFF = PLANE1;
    static constexpr int T2LO_OFF = T1LO ? PLANE1 : PLANE2;
    static constexpr int ABYTES = T1LO ? 2 * PLANE1 : 2 * PLANE2;
    static constexpr int LDS = XBYTES + ABYTES;
    static_assert(LDS <= 160 * 1024, "bottleneck tile does not fit the LDS");
};

typedef f16 h2v __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// ReLU on fp32 as ONE instruction (v_med3_f32; fmaxf costs a canonicalising v_max first)
__device__ __forceinline__ float relu_f32(float v) { return __builtin_amdgcn_fmed3f(v, 0.f, 3.0e38f); }

// four fp32 -> packed f16 with ReLU applied on the packed halves (v_cvt_pk_f16_f32 + v_pk_max_f16: rounding is monotone and keeps the sign)
__device__ __forceinline__ uint2 relu_pack4(f32x4 v) {
    const h2v z = {(f16)0.f, (f16)0.f};
    const h2v a = __builtin_elementwise_max(h2v{(f16)v[0], (f16)v[1]}, z), b = __builtin_elementwise_max(h2v{(f16)v[2], (f16)v[3]}, z);
    return make_uint2(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b));
}

// f16(v - hi_half) in ONE instruction (v_fma_mixlo / mixhi_f16: an fma whose first operand is an f16 half, rounded once to f16), written into
// the low / high half of `d`.  hipcc does not form these from (f16)(v - (float)h): it emits cvt, sub, cvt.  The inputs must come from ordinary
// vector instructions (here: v_med3 / v_cvt_pk), never straight from an MFMA: hipcc pads MFMA -> VALU hazards only for instructions it knows.
__device__ __forceinline__ unsigned lo_pair(unsigned hipk, float v0, float v1) {
    unsigned d;
    asm("v_fma_mixlo_f16 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hipk), "v"(v0));
    asm("v_fma_mixhi_f16 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(d) : "v"(hipk), "v"(v1));
    return d;
}

// four (already rectified) fp32 -> packed f16 hi parts and the packed f16 lo parts of what the rounding dropped
__device__ __forceinline__ void split4(const float (&v)[4], uint2& hi, uint2& lo) {
    const h2v a = {(f16)v[0], (f16)v[1]}, b = {(f16)v[2], (f16)v[3]};
    hi = make_uint2(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b));
    lo = make_uint2(lo_pair(hi.x, v[0], v[1]), lo_pair(hi.y, v[2], v[3]));
}

template <int CIN, bool DS, bool T1LO, bool XLO, bool OLO>
__global__ void __launch_bounds__(512) k_bottleneck(BnArgs p) {
    typedef BnLayout<CIN, DS, T1LO> L;
    constexpr int KS1 = CIN / 32;                 // conv1 K steps
    constexpr int KS3 = 4 + (DS ? CIN / 32 : 0);  // conv3 K steps (+ downsample)
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, q = lane >> 4;
    const unsigned lds0 = lds_addr(lds);
    char* const A = lds + L::XBYTES;

    int bid = blockIdx.x;
    {   // XCD-aware bijective remap: the workgroups of one XCD take a contiguous run of tiles (neighbours share halo rows in one L2)
        const int nwg = gridDim.x, xcd = bid & 7, local = bid >> 3, qq = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + local;
    }

    // ---- X tile by LDS-DMA: one wave-instruction = 8 halo pixels x 128 B of one slab; lane -> (pixel, physical chunk), the swizzle
    // goes on the source address.  Pixels outside the image come from a clamped address: conv1 of them is never used (t1 is zeroed there).
    auto stage_group = [&](int tile, unsigned xbase, int j) __attribute__((always_inline)) {
        const int ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
        int lo_ = lane;
        asm volatile("" : "+v"(lo_));            // (keeps the per-lane source offsets out of the loop preheader)
        const int gi = wave + 8 * j;
        const int pix = gi * 8 + (lo_ >> 3);
        const int hy = (pix * 3641) >> 16, hx = pix - hy * BH_W;          // pix / 18 for pix < 192
        const int iy = min(max(ty * BT_H - 1 + hy, 0), p.H - 1), ix = min(max(tx * BT_W - 1 + hx, 0), p.W - 1);
        const unsigned voff = ((unsigned)(iy * p.W + ix) * (unsigned)p.in_ld + (unsigned)(((lo_ & 7) ^ (pix & 7)) << 3)) * 2u;
#pragma unroll
        for (int kc = 0; kc < L::NCH; ++kc) glds16_saddr(p.x + kc * 64, voff, xbase + kc * XSLAB + gi * 1024);
    };
    auto stage = [&](int tile, unsigned xbase) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 3; ++j) stage_group(tile, xbase, j);
    };

    int tile = bid;
    int it = 0;                                   // tiles done by this workgroup (DS: X tile slot = it & 1)
    if (tile < p.ntiles) stage(tile, lds0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // lane constants
    const unsigned x_rd = (unsigned)(c * 128) + (unsigned)((q ^ (c & 7)) << 4);                        // + m * 2048 (+ 64 with bit flip for the second K step of a slab)
    const unsigned a_wr = (unsigned)(c * 256) + (unsigned)((((2 * wave) | (q >> 1)) ^ c) << 4) + (unsigned)((q & 1) * 8);   // t1 / t2 store: + m * 4096
    const unsigned t2_rd = (unsigned)(c * 256) + (unsigned)((q ^ c) << 4);                              // conv3 read: ^ (ks << 6), + m * 4096

    unsigned long long tsum[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // probe: conv1 | t1 store + conv2 + t2 store | t2 lo store + conv3 preloads | B2 | DMA issue | conv3 | landing + B3 | stores | - | tiles
    unsigned long long t0 = 0;
    auto stamp = [&](int slot) __attribute__((always_inline)) {
        if (kStamps && p.dbg) {
            unsigned long long t;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
            if (slot >= 0) tsum[slot] += t - t0;
            t0 = t;
        }
    };
    // weight fragments by BUFFER loads: wave-uniform resource (SGPRs) + 32-bit lane offset + scalar fragment offset -- per-request 64-bit
    // VGPR pointers were what hipcc hoisted out of the tile loop and spilled
    const __amdgpu_buffer_rsrc_t w1r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.w1)) + (size_t)wave * (KS1 * 2 * 1024), 0, KS1 * 2 * 1024, 0x00020000);
    const __amdgpu_buffer_rsrc_t w2r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.w2)) + (size_t)wave * (5 * 2 * 1024), 0, 5 * 2 * 1024, 0x00020000);
    const __amdgpu_buffer_rsrc_t w3r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.w3)) + (size_t)wave * (KS3 * 4 * 1024), 0, KS3 * 4 * 1024, 0x00020000);
    const int wlane = lane * 16;
    auto wfrag = [&](__amdgpu_buffer_rsrc_t r, int frag) __attribute__((always_inline)) {
        return __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(r, wlane, frag * 1024, 0));
    };
    const int ch0 = 32 * wave + 8 * q;                     // conv3: this lane's 8 output channels
    // conv1's weights stay in registers for the whole tile (row tiles outside, K steps inside); requested at the top of the tile.
    // (Requesting the NEXT tile's during conv3, so that their L2 round trip is not exposed at the top of the tile, was measured: -1 us on
    // the single-plane variant, +8 us with a split output -- the 64 registers are missed in conv3 -- and spills with a split residual.)
    f16x8 w1f[KS1][2];

    for (; tile < p.ntiles; tile += gridDim.x, ++it) {
        const int ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
        stamp(-1);
        const char* X = lds + (DS ? (it & 1) * L::XTILE : 0);
        // (opaque copies of the lane coordinates: hipcc would otherwise hoist the ~70 tile-invariant LDS offsets and pixel masks out of
        // the tile loop and spill them)
        int co = c, qo = q;
        asm volatile("" : "+v"(co), "+v"(qo));
        const bool more = tile + (int)gridDim.x < p.ntiles;
        f16x8 w2f[5][2];
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) { w1f[ks][0] = wfrag(w1r, 2 * ks); w1f[ks][1] = wfrag(w1r, 2 * ks + 1); }
#pragma unroll
        for (int k2 = 0; k2 < 5; ++k2) { w2f[k2][0] = wfrag(w2r, 2 * k2); w2f[k2][1] = wfrag(w2r, 2 * k2 + 1); }
        // DS: two X tiles, the other one is free from the top of the tile on; the weight requests above go first, so nothing waits
        // behind the DMA burst before conv3's preloads -- a whole conv1 + conv2 later
        if (DS && more) stage(tile + gridDim.x, lds0 + ((it & 1) ^ 1) * L::XTILE);

        // ================================================= conv1 + t1 -> LDS + conv2 + t2 -> LDS as ONE software-pipelined stream
        // conv1 walks the 12 halo row tiles (K steps inside, weights in registers): the X fragments of row tile m + 1 are read while the 16
        // MFMAs of tile m run.  Output row r of the 3x3 reads halo pixels up to 18 r + 53, i.e. t1 row tiles 0 .. (18 r + 53) / 16: it is
        // issued as soon as those are stored, so the 3x3's MFMAs and all the vector work (ReLU, f16 split, masks, LDS addresses) run
        // beside conv1's MFMAs instead of in a phase of their own.  t2 row r goes IN PLACE over t1 rows 16 r .. 16 r + 15 of this wave's
        // own channels, which no later output row reads (those start at 18 (r + 1)).
        auto store_t1 = [&](int m, f32x4 acc) __attribute__((always_inline)) {
            const int h = m * 16 + co;
            const int hy = (h * 3641) >> 16, hx = h - hy * BH_W;
            const int iy = ty * BT_H - 1 + hy, ix = tx * BT_W - 1 + hx;
            const bool valid = ((unsigned)(h < BN_HALO) & (unsigned)((unsigned)iy < (unsigned)p.H) & (unsigned)((unsigned)ix < (unsigned)p.W)) != 0u;   // (branch-free)
            const unsigned vm = valid ? 0xffffffffu : 0u;
            if constexpr (T1LO) {
                float v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = relu_f32(acc[k]);
                uint2 hi, lo;
                split4(v, hi, lo);
                *reinterpret_cast<uint2*>(A + a_wr + m * 4096) = make_uint2(hi.x & vm, hi.y & vm);
                *reinterpret_cast<uint2*>(A + L::T1LO_OFF + a_wr + m * 4096) = make_uint2(lo.x & vm, lo.y & vm);
            } else {
                const uint2 hi = relu_pack4(acc);
                *reinterpret_cast<uint2*>(A + a_wr + m * 4096) = make_uint2(hi.x & vm, hi.y & vm);
            }
        };
        uint2 t2lo[T1LO ? 1 : BT_H];                       // no t1 lo plane: the t2 lo plane shares LDS with t1 rows >= 128 -> stored after the last row
        const float4 bias1 = *reinterpret_cast<const float4*>(p.b1 + 16 * wave + 4 * q);
        const float4 bias2 = *reinterpret_cast<const float4*>(p.b2 + 16 * wave + 4 * q);
        const unsigned cw16 = (unsigned)(((2 * wave) | (qo & 1)) << 4);
        unsigned hk[5];
#pragma unroll
        for (int ks = 0; ks < 5; ++ks) {
            const int t = min(2 * ks + (qo >> 1), 8);      // this lane's tap in the K step (the tenth "tap" has zero weights)
            hk[ks] = (unsigned)((t / 3) * BH_W + (t % 3) + co);         // halo pixel of output row 0
        }
        auto conv2_row = [&](int r) __attribute__((always_inline)) {
            f32x4 acc = f32x4{bias2.x, bias2.y, bias2.z, bias2.w};
            f16x8 th[5], tl[T1LO ? 5 : 1];
#pragma unroll
            for (int ks = 0; ks < 5; ++ks) {
                const unsigned h = hk[ks] + 18u * r;
                const unsigned off = (h << 8) + (cw16 ^ ((h & 15u) << 4));
                th[ks] = *reinterpret_cast<const f16x8*>(A + off);
                if constexpr (T1LO) tl[ks] = *reinterpret_cast<const f16x8*>(A + L::T1LO_OFF + off);
            }
            __builtin_amdgcn_sched_barrier(0);      // all of the row's fragments are requested before its first MFMA (hipcc otherwise pairs them up: one LDS round trip per two MFMAs; same-box A/B: -1.5 .. -4 us per block)
#pragma unroll
            for (int ks = 0; ks < 5; ++ks) {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2f[ks][0], th[ks], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2f[ks][1], th[ks], acc, 0, 0, 0);
                if constexpr (T1LO) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2f[ks][0], tl[ks], acc, 0, 0, 0);
            }
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = relu_f32(acc[k]);
            uint2 hi, lo;
            split4(v, hi, lo);
            *reinterpret_cast<uint2*>(A + a_wr + r * 4096) = hi;
            if constexpr (T1LO) *reinterpret_cast<uint2*>(A + L::T2LO_OFF + a_wr + r * 4096) = lo;
            else t2lo[r] = lo;
        };
        // everything conv3 needs from global memory is requested BEFORE the X DMA of the next tile goes out (vmcnt counts in issue order: a load
        // behind the DMA burst would wait for the burst's HBM round trip at its first use) -- and as early as registers allow
        f16x8 w3f[KS3][4];
        uint4 resh[DS ? 1 : BT_H], resl[XLO ? BT_H : 1];
        static_assert(!(DS && XLO), "the downsample variant has no identity residual");
        auto preload_conv3 = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int ks = 0; ks < KS3; ++ks)
#pragma unroll
                for (int f = 0; f < 4; ++f) w3f[ks][f] = wfrag(w3r, ks * 4 + f);
            if constexpr (!DS) {
                // identity: the block input at the tile's own pixels, in accumulator layout (16 B per lane; L2 / MALL hits: the tile was staged from there)
#pragma unroll
                for (int rr = 0; rr < BT_H; ++rr) {
                    const int oy = min(ty * BT_H + rr, p.H - 1), ox = min(tx * BT_W + co, p.W - 1);
                    const unsigned roff = (unsigned)(oy * p.W + ox) * (unsigned)p.in_ld + (unsigned)ch0;
                    resh[rr] = *reinterpret_cast<const uint4*>(p.x + roff);
                    if constexpr (XLO) resl[rr] = *reinterpret_cast<const uint4*>(p.x_lo + roff);
                }
            }
        };
        {
            // X fragment addresses: slab pair (0,1 | 2,3) x K-step parity -> immediate offsets stay below 64 KB
            const char* xb[KS1 > 4 ? 4 : 2];
#pragma unroll
            for (int i = 0; i < (KS1 > 4 ? 4 : 2); ++i) xb[i] = X + (i >> 1) * 2 * XSLAB + (x_rd ^ ((i & 1) << 6));
            auto read_x = [&](int m, f16x8 (&xf)[KS1]) __attribute__((always_inline)) {
#pragma unroll
                for (int ks = 0; ks < KS1; ++ks)
                    xf[ks] = *reinterpret_cast<const f16x8*>(xb[(ks >> 2) * 2 + (ks & 1)] + ((ks >> 1) & 1) * XSLAB + m * 2048);
            };
            f16x8 xf[2][KS1];
            read_x(0, xf[0]);
            int next_row = 0;
#pragma unroll
            for (int m = 0; m < BN_M1; ++m) {
                if (m + 1 < BN_M1) read_x(m + 1, xf[(m + 1) & 1]);
                f32x4 acc = f32x4{bias1.x, bias1.y, bias1.z, bias1.w};
#pragma unroll
                for (int ks = 0; ks < KS1; ++ks) {
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1f[ks][0], xf[m & 1][ks], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1f[ks][1], xf[m & 1][ks], acc, 0, 0, 0);
                }
                // the 3x3's output rows whose t1 row tiles (0 .. m - 1 are stored at this point) are complete
#pragma unroll
                for (int r = 0; r < BT_H; ++r)
                    if (r >= next_row && (18 * r + 53) / 16 + 1 <= m) conv2_row(r);
#pragma unroll
                for (int r = 0; r < BT_H; ++r)
                    if (r >= next_row && (18 * r + 53) / 16 + 1 <= m) next_row = r + 1;
                store_t1(m, acc);
            }
            preload_conv3();                   // conv1's weight registers are free from here on: beside the last output rows of the 3x3
#pragma unroll
            for (int r = 0; r < BT_H; ++r)
                if (r >= next_row) conv2_row(r);
        }
        stamp(0);
        if constexpr (!T1LO) {
#pragma unroll
            for (int r = 0; r < BT_H; ++r) *reinterpret_cast<uint2*>(A + L::T2LO_OFF + a_wr + r * 4096) = t2lo[r];
        }
        stamp(1);
        stamp(2);
        __syncthreads();                                   // B2: t2 visible; every wave is done with the X tile it read in conv1
        stamp(3);
        stamp(4);

        // ================================================= conv3 (+ downsample) + residual, row by row; results packed in registers
        // The rows' results leave as they are produced, by BUFFER stores: a pixel outside the image gets an offset past the buffer's end
        // and the hardware drops it -- no branch, so the NUMBER of vector-memory instructions behind the last DMA instruction is fixed and
        // the landing of the next X tile can be awaited by count (vmcnt(N) passes once all but the N youngest have completed) without
        // waiting for these stores.
        const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t olrsrc = __builtin_amdgcn_make_buffer_rsrc(OLO ? p.out_lo : p.out, 0, p.out_bytes, 0x00020000);
        {
            const float4 b0 = *reinterpret_cast<const float4*>(p.b3 + ch0), b1 = *reinterpret_cast<const float4*>(p.b3 + ch0 + 4);
            auto read_t2 = [&](int r, f16x8 (&tf)[8]) __attribute__((always_inline)) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const unsigned base = t2_rd ^ (unsigned)(ks << 6);
                    tf[2 * ks] = *reinterpret_cast<const f16x8*>(A + base + r * 4096);
                    tf[2 * ks + 1] = *reinterpret_cast<const f16x8*>(A + L::T2LO_OFF + base + r * 4096);
                }
            };
            f16x8 tf[2][8];
            read_t2(0, tf[0]);
#pragma unroll
            for (int r = 0; r < BT_H; ++r) {
                if (!DS && r < 3 && more) stage_group(tile + gridDim.x, lds0, r);       // the next X tile, one third per row: B2 is behind us
                if (r + 1 < BT_H) read_t2(r + 1, tf[(r + 1) & 1]);                        // next row's fragments beside this row's MFMAs
                f32x4 a0 = f32x4{b0.x, b0.y, b0.z, b0.w}, a1 = f32x4{b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const f16x8 th = tf[r & 1][2 * ks], tl = tf[r & 1][2 * ks + 1];
                    a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3f[ks][0], th, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3f[ks][2], th, a1, 0, 0, 0);
                    a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3f[ks][1], th, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3f[ks][3], th, a1, 0, 0, 0);
                    a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3f[ks][0], tl, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3f[ks][2], tl, a1, 0, 0, 0);
                }
                if constexpr (DS) {
                    // downsample 1x1 (stride 1) on the tile's own pixels of the X tile: KS1 more K steps
                    const int h = (r + 1) * BH_W + 1 + co;
#pragma unroll
                    for (int ks = 0; ks < KS1; ++ks) {
                        const f16x8 xb = *reinterpret_cast<const f16x8*>(X + (ks >> 1) * XSLAB + h * 128 + ((((ks & 1) * 4 + qo) ^ (h & 7)) << 4));
                        a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3f[4 + ks][0], xb, a0, 0, 0, 0);
                        a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3f[4 + ks][2], xb, a1, 0, 0, 0);
                        a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3f[4 + ks][1], xb, a0, 0, 0, 0);
                        a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3f[4 + ks][3], xb, a1, 0, 0, 0);
                    }
                }
                const int oy = ty * BT_H + r, ox = tx * BT_W + co;
                const bool live = ((unsigned)(oy < p.H) & (unsigned)(ox < p.W)) != 0u;
                const unsigned ooff = live ? ((unsigned)(oy * p.W + ox) * (unsigned)p.out_ld + (unsigned)ch0) * 2u : 0x80000000u;
                float v0[4] = {a0[0], a0[1], a0[2], a0[3]}, v1[4] = {a1[0], a1[1], a1[2], a1[3]};
                if constexpr (!DS) {
                    // + identity.  Plain C (cvt + add): an inline-asm v_fma_mix_f32 reading the MFMA result directly is NOT safe -- hipcc
                    // inserts the MFMA -> VALU wait states only for instructions it knows (measured: garbage sums)
                    const f16x8 rh = __builtin_bit_cast(f16x8, resh[r]);
#pragma unroll
                    for (int k = 0; k < 4; ++k) { v0[k] += (float)rh[k]; v1[k] += (float)rh[4 + k]; }
                    if constexpr (XLO) {
                        const f16x8 rl = __builtin_bit_cast(f16x8, resl[r]);
#pragma unroll
                        for (int k = 0; k < 4; ++k) { v0[k] += (float)rl[k]; v1[k] += (float)rl[4 + k]; }
                    }
                }
                if constexpr (OLO) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) { v0[k] = relu_f32(v0[k]); v1[k] = relu_f32(v1[k]); }
                    uint2 h0, l0, h1, l1;
                    split4(v0, h0, l0);
                    split4(v1, h1, l1);
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{h0.x, h0.y, h1.x, h1.y}, orsrc, ooff, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{l0.x, l0.y, l1.x, l1.y}, olrsrc, ooff, 0, 0);
                } else {
                    const uint2 h0 = relu_pack4(f32x4{v0[0], v0[1], v0[2], v0[3]}), h1 = relu_pack4(f32x4{v1[0], v1[1], v1[2], v1[3]});
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{h0.x, h0.y, h1.x, h1.y}, orsrc, ooff, 0, 0);
                }
            }
        }
        stamp(5);
        // the next X tile has landed: behind its last DMA instruction this wave issued the stores of rows 2 .. 7 (all 8 rows where the DMA went
        // out at the top of the tile) -- at least; anything hipcc adds only makes the wait stricter
        {
            constexpr int NTAIL = (DS ? BT_H : BT_H - 2) * (OLO ? 2 : 1);
            if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NTAIL) : "memory");
        }
        __syncthreads();                                       // B3: ... for every wave; t2 (and, DS, this X tile) is read
        stamp(6);
        stamp(7);
        if (kStamps && p.dbg) tsum[9] += 1;
    }
    if (kStamps && p.dbg && lane == 0) {
#pragma unroll
        for (int i = 0; i < 10; ++i) p.dbg[((size_t)blockIdx.x * 8 + wave) * 10 + i] = tsum[i];
    }
}

int device_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

template <int CIN, bool DS, bool T1LO, bool XLO, bool OLO>
int launch_bn(const BnArgs& a, hipStream_t s) {
    typedef BnLayout<CIN, DS, T1LO> L;
    AVL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bottleneck<CIN, DS, T1LO, XLO, OLO>), hipFuncAttributeMaxDynamicSharedMemorySize, L::LDS));
    const int grid = a.ntiles < device_cus() ? a.ntiles : device_cus();
    BnArgs b = a;
    b.dbg = nullptr;
#ifdef AVL_EXPERIMENTS
    // timing experiment: where do a wave's cycles go (s_memtime stamps; synchronises the stream: never inside a graph capture)
    static unsigned long long* dbg = nullptr;
    if (AVL_EXP_INT("AVL_BN_PROBE", 0)) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        AVL_HIP_CHECK(hipStreamIsCapturing(s, &cap));
        AVL_REQUIRE(cap == hipStreamCaptureStatusNone, "AVL_BN_PROBE synchronises the stream: not while it is being captured");
        if (!dbg) AVL_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&dbg), 256 * 8 * 10 * sizeof(unsigned long long), 0));
        memset(dbg, 0, 256 * 8 * 10 * sizeof(unsigned long long));
        if (grid <= 256) b.dbg = dbg;
    }
#endif
    hipLaunchKernelGGL((k_bottleneck<CIN, DS, T1LO, XLO, OLO>), dim3(grid), dim3(512), L::LDS, s, b);
    AVL_LAUNCH_CHECK();
#ifdef AVL_EXPERIMENTS
    if (b.dbg) {
        AVL_HIP_CHECK(hipStreamSynchronize(s));
        double sum[10] = {};
        for (int i = 0; i < grid * 8; ++i)
            for (int k = 0; k < 10; ++k) sum[k] += (double)dbg[i * 10 + k];
        const double n = sum[9] > 0 ? sum[9] : 1;
        fprintf(stderr, "[bottleneck probe] cin %d ds %d t1lo %d xlo %d olo %d, %.1f tiles per wave; cycles per tile: conv1 + conv2 %.0f | t2 lo + preloads %.0f | - %.0f | B2 %.0f | - %.0f | conv3 + pack + stores %.0f | landing + B3 %.0f | - %.0f | - %.0f | sum %.0f\n",
                CIN, (int)DS, (int)T1LO, (int)XLO, (int)OLO, n / (grid * 8), sum[0] / n, sum[1] / n, sum[2] / n, sum[3] / n, sum[4] / n, sum[5] / n, sum[6] / n, sum[7] / n, sum[8] / n,
                (sum[0] + sum[1] + sum[2] + sum[3] + sum[4] + sum[5] + sum[6] + sum[7] + sum[8]) / n);
    }
#endif
    return AVL_OK;
}

}  // namespace

// AVL_OP_BOTTLENECK (include/avl_hip.h): in_c = 64 (with the downsample 1x1 folded into conv3: w_layout = 1) or 256 (identity
// residual: w_layout = 0); w_split = 1: t1 keeps a lo plane too (in_c = 64 only: the LDS has no room for it at 256).
int validate_bottleneck(const avl_seg_op& op) {
    AVL_REQUIRE(op.dtype == AVL_F16, "fused bottleneck: AVL_F16 activations only");
    AVL_REQUIRE(op.groups == 32 && op.ksize == 3 && op.stride == 1 && op.dil == 1 && op.pad == 1, "fused bottleneck: 3x3, 32 groups, stride 1, dilation 1");
    AVL_REQUIRE(op.out_c == BN_COUT && op.in3_c == BN_WIDTH, "fused bottleneck: width %d -> %d output channels only (got %d -> %d)", BN_WIDTH, BN_COUT, op.in3_c, op.out_c);
    AVL_REQUIRE((op.in_c == 64 && op.w_layout == 1) || (op.in_c == 256 && op.w_layout == 0 && op.w_split == 0),
                "fused bottleneck: 64 input channels with the downsample folded in, or 256 with the identity residual and no t1 lo plane");
    AVL_REQUIRE(op.in_h == op.out_h && op.in_w == op.out_w && op.in_h > 0 && op.in_w > 0, "fused bottleneck: same-size output");
    AVL_REQUIRE(op.in_ld >= op.in_c && op.in_ld % 8 == 0 && op.out_ld >= op.out_c && op.out_ld % 8 == 0, "fused bottleneck: row strides");
    AVL_REQUIRE((long long)op.in_rows >= (long long)op.in_h * op.in_w && (long long)op.out_rows >= (long long)op.out_h * op.out_w, "fused bottleneck: rows allocated");
    AVL_REQUIRE((long long)op.in_rows * op.in_ld * 2 < (1LL << 31) && (long long)op.out_rows * op.out_ld * 2 < (1LL << 31), "fused bottleneck: planes beyond 2 GB (32-bit offsets)");
    AVL_REQUIRE(op.in && op.out && op.weight && op.in2 && op.in3 && op.bias, "fused bottleneck: in, out, weight (conv1), in2 (conv2 weights), in3 (conv3 weights), bias");
    AVL_REQUIRE(((reinterpret_cast<uintptr_t>(op.in) | reinterpret_cast<uintptr_t>(op.in_lo) | reinterpret_cast<uintptr_t>(op.out) | reinterpret_cast<uintptr_t>(op.out_lo) |
                  reinterpret_cast<uintptr_t>(op.weight) | reinterpret_cast<uintptr_t>(op.in2) | reinterpret_cast<uintptr_t>(op.in3) | reinterpret_cast<uintptr_t>(op.bias)) & 15) == 0,
                "fused bottleneck: 16-byte aligned buffers");
    AVL_REQUIRE(!(op.in_c == 64 && op.in_lo), "fused bottleneck: the 64-channel variant takes a single-plane input");
    AVL_REQUIRE(op.in != op.out && op.in != op.out_lo, "fused bottleneck: not in place (tiles read their neighbours' pixels)");
    return AVL_OK;
}

int launch_bottleneck(const avl_seg_op& op, hipStream_t s) {
    BnArgs a;
    a.x = static_cast<const f16*>(op.in);
    a.x_lo = static_cast<const f16*>(op.in_lo);
    a.out = static_cast<f16*>(op.out);
    a.out_lo = static_cast<f16*>(op.out_lo);
    a.w1 = static_cast<const f16*>(op.weight);
    a.w2 = static_cast<const f16*>(op.in2);
    a.w3 = static_cast<const f16*>(op.in3);
    a.b1 = op.bias; a.b2 = op.bias + BN_WIDTH; a.b3 = op.bias + 2 * BN_WIDTH;
    a.H = op.in_h; a.W = op.in_w; a.in_ld = op.in_ld; a.out_ld = op.out_ld;
    a.tiles_x = (op.in_w + BT_W - 1) / BT_W;
    a.ntiles = a.tiles_x * ((op.in_h + BT_H - 1) / BT_H);
    a.dbg = nullptr;
    a.out_bytes = (int)((long long)op.out_rows * op.out_ld * 2);
    const int variant = (op.in_c == 64 ? 4 + 2 * (op.w_split ? 1 : 0) : 2 * (a.x_lo ? 1 : 0)) + (a.out_lo ? 1 : 0);
    switch (variant) {
        case 0: return launch_bn<256, false, false, false, false>(a, s);
        case 1: return launch_bn<256, false, false, false, true>(a, s);
        case 2: return launch_bn<256, false, false, true, false>(a, s);
        case 3: return launch_bn<256, false, false, true, true>(a, s);
        case 4: return launch_bn<64, true, false, false, false>(a, s);
        case 5: return launch_bn<64, true, false, false, true>(a, s);
        case 6: return launch_bn<64, true, true, false, false>(a, s);
        default: return launch_bn<64, true, true, false, true>(a, s);
    }
}

}  // namespace avl
