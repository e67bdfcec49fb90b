// Grouped 3x3 convolution (torchvision Bottleneck.conv2, groups = 32) on the matrix cores, bf16.
//
// The 32 groups have 4/8/16/32 channels each (layer1..4), far too narrow for an MFMA tile, so the
// host packs the weights of every 32-CHANNEL WINDOW as one dense block-diagonal [32 out][9 taps x 32 in]
// matrix (zeros between different groups).  MFMA time is then a few microseconds per layer even with
// the zeros; what bounds the kernel is bytes, so the work is organised around memory:
//   * a workgroup owns an 8 x 32 output-pixel tile x 64 channels (two windows); the input tile with
//     its dilation halo is staged ONCE into LDS (zero-filled outside the image), so each of the 9
//     taps is an LDS read instead of another trip to L2;
//   * LDS rows are one pixel = 128 B; 16-byte chunks are XOR-swizzled with the pixel index;
//   * each wave keeps its window's 18 weight fragments (2 n-tiles x 9 taps) in registers for the
//     whole tile and walks 8 sub-tiles of 16 pixels: 9 ds_read_b128 + 18 v_mfma_f32_16x16x32_bf16 each;
//   * product is computed transposed (channels on MFMA rows, permuted) so a lane owns 8 consecutive
//     output channels of one pixel: bias + ReLU + one 16-byte store per lane.
#include <cstring>

#include "seg_types.h"

namespace avl {
namespace {

constexpr int TW = 32;          // output tile width  (2 sub-tiles of 16 pixels)
constexpr int CC = 64;          // channels per workgroup (2 windows of 32)

template <typename HT>
struct GconvArgs {
    const HT* in;
    const HT* in_lo;      // XS (split input: the complete hi + lo pipeline): the input's lo plane, same shape and stride
    const HT* w;          // [window][nj 2][tap 9][i 16][ci 32]
    const float* bias;   // [C]
    HT* out;
    HT* out_lo;          // "mixed" precision: low plane of the result (value - f16(value)), or NULL
    char* oq[2];         // MX-FP4 copies of the output planes for the next MX GEMM (avl_hip.h, w_split = 2), or NULL ...
    char* os[2];         // ... and their E8M0 scales [C/256][rows][8]
    long long o_srows;
    int H, W, in_ld, OH, OW, out_ld, C;
    int stride, dil;
    int th;              // output tile height (8, or 4 for stride 2)
    int tiles_x, tiles_y, cchunks;
    int comb;            // 1: stride 1, dilation d > 1 -> tiles live on the d x d residue-class grids (see below)
    int nsp, nslots;     // pixel tiles (all residue classes), and how many of them are in flight (workgroups per channel chunk)
    int walk;            // 1: a workgroup takes a contiguous run of tiles, numbered DOWN the columns of the tile grid: its next tile is the one
                         // below, whose top halo rows are this tile's last input rows and still sit in L2 (4-row tiles read 6 rows: without this
                         // a third of every tile's input came from HBM a second time); 0: tiles slot, slot + nslots, ... numbered along the rows
    int tile_bytes;      // one LDS tile buffer
    int tw_magic;        // ceil(65536 / input tile width): pix / in_tw == (pix * tw_magic) >> 16 for every pixel of a tile
    // WS = 2 (MX variant, see below): FP4 copies of the weights and of the input
    const char* w4;      // FP4 fragments [window][nj 2][pair 2][g 3][lane 64][16 B]; pair 0 = Q4(W lo), 1 = Q4(W hi)
    const char* w4s;     // their scales [window][lane 64][12 B] (byte index nj * 6 + pair * 3 + g)
    const char* xq[2];   // FP4 planes of the input [rows][C/2]: hi part, lo part (NULL: no lo correction)
    const char* xs[2];   // scales [C/256][rows][8]
    long long x_srows;
    int dephase;               // 1: waves 4-7 stage the next tile in the middle of their MFMA phase (A/B switch AVL_GC_DEPHASE)
    unsigned long long* dbg;   // AVL_GC_PROBE=1: per-wave cycle sums of the tile loop's phases (host-visible memory), else NULL
};

// WS = 1 ("mixed" precision): the weights come as f16 pairs hi + lo ([window][nj 2][tap 18 = 9 hi, 9 lo][16][32]); both
// parts multiply the same input fragment into the same accumulator (fp32), i.e. the weights keep ~22 significant bits.
//
// WS = 2 (MX variant, MODEL.MIXED_GCONV_MX: blocks whose conv1 ran as an MX GEMM): the input is an f16 plane plus its MX
// bundle (FP4 copies of its hi part and of its lo part, include/avl_hip.h), the weights are f16 hi fragments plus FP4 copies:
//     out = Wh . xh                       f16 MFMA, one per tap and n-tile (K = the window's 32 channels)
//         + Q4(Wl) . Q4(xh) + Q4(Wh) . Q4(xl)     v_mfma_scale_f32_16x16x128_f8f6f4: K = 128 = FOUR TAPS x 32 channels,
//                                                 so three scaled MFMAs cover the nine taps (the last one is 3/4 zeros)
// An MX block is the 32 channels of one window at one pixel (activations) / of one output row and tap (weights): exactly
// the blocks the producers' epilogues quantise.  Lane (pixel fr, K block kq) of a scaled MFMA reads the 16 bytes of the
// pixel that tap 4 g + kq falls on, and that pixel's scale byte.  Both corrections come to 12 scaled MFMAs per sub-tile next
// to the 18 f16 ones (the f16-only form of split weights needs 36 and cannot correct for xl at all).  The LDS tile buffer
// then also holds the FP4 tiles (plane 2 x window 2, 16 B per pixel, 64 pixels per DMA instruction) and one scale dword per
// pixel and plane.
// XS (WS = 1 only): the input is split too (f16 planes hi + lo): a second f16 tile in LDS and a third pass Wh . xl per tap -- every
// operand of the 3x3 then carries ~22 bits (the complete-split plan of MODEL.MIXED_SELF_CHECK's ladder).
template <typename HT, int WS, int NJ, bool XS = false>      // NJ = tile height / 2: sub-tiles per wave and tile
__global__ void __launch_bounds__(512) k_gconv_mfma(GconvArgs<HT> p) {
    static_assert(!XS || WS == 1, "a split input goes with split weights");
    typedef typename Half16<HT>::v8 v8;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = blockIdx.x;
    {
        // XCD-aware, bijective remap (workgroups b, b+8, ... share an XCD and its L2): each XCD takes a contiguous run of
        // (slot, channel chunk) pairs, so neighbouring tiles share their halo in one L2
        const int nwg = gridDim.x, xcd = bid & 7, local = bid >> 3, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    }
    // PERSISTENT: the launch is one workgroup per CU; workgroup (slot, channel chunk) keeps its chunk's weight fragments in
    // registers and walks the pixel tiles slot, slot + nslots, ... with TWO LDS tile buffers: the LDS-DMA of the next tile is
    // in flight while this tile's sub-tiles go through the matrix cores (one barrier per tile).
    const int cchunk = bid % p.cchunks, slot = bid / p.cchunks;
    // Dilated layers (stride 1, pad = d): the outputs with (y mod d, x mod d) = (ry, rx) read only inputs of the same
    // residue class, so a tile is cut from that class's grid and is an ordinary d = 1 tile there: the halo is
    // (8+2) x (32+2) pixels instead of (8+2d) x (32+2d) (2.5x the tile at d = 4).
    const int s = p.stride;
    const int d = p.comb ? 1 : p.dil;            // tap distance inside the LDS tile
    const int step = p.comb ? p.dil : 1;         // image pixels per tile-grid step
    const int in_th = (p.th - 1) * s + 2 * d + 1, in_tw = (TW - 1) * s + 2 * d + 1;
    const int c0 = cchunk * CC;
    const int npix = in_th * in_tw;
    const int ngroups = (npix + 7) >> 3;
    const unsigned lds0 = lds_addr(lds);
    // MX: [f16 tile: ngroups KB][FP4 tiles (plane 2 x window 2): n64 KB each][scale dwords (plane 2): n64 x 256 B each]
    const int n64 = (npix + 63) >> 6;
    const int QBASE = ngroups * 1024, QT = n64 * 1024, SBASE = QBASE + 4 * QT, ST = n64 * 256;
    const bool has_lo = WS == 2 && p.xq[1] != nullptr;
    const int LOBASE = ngroups * 1024;            // XS: the lo plane's tile behind the hi plane's

    // ---- stage one input tile (+halo) by LDS-DMA: one wave-instruction = 8 pixels x 128 B, all of a wave's transfers in
    // flight at once.  The DMA writes LDS lane-linearly, so the chunk swizzle goes on the SOURCE address.  Pixels outside
    // the image are fetched from a clamped address and zeroed once the data has landed (bit mask returned).  The DMA is
    // issued from inline asm: the compiler does not see it, so it neither drains it in front of the ds_reads of the OTHER
    // buffer nor needs to -- completion is the hand-placed s_waitcnt vmcnt(0) at the top of the tile loop.
    const char* const xq0 = p.xq[0];
    const char* const xs0 = p.xs[0];
    const long long xq_delta = has_lo ? p.xq[1] - p.xq[0] : 0, xs_delta = has_lo ? p.xs[1] - p.xs[0] : 0;
    auto stage = [&](int sp, int buf) -> unsigned {
        int tx, ty, cmb;
        if (p.walk) { ty = sp % p.tiles_y; const int r1 = sp / p.tiles_y; tx = r1 % p.tiles_x; cmb = r1 / p.tiles_x; }
        else { tx = sp % p.tiles_x; const int r1 = sp / p.tiles_x; ty = r1 % p.tiles_y; cmb = r1 / p.tiles_y; }
        const int ry = p.comb ? cmb / p.dil : 0, rx = p.comb ? cmb % p.dil : 0;
        const int iy0 = ry + (ty * p.th * s - d) * step, ix0 = rx + (tx * TW * s - d) * step;
        const int prow = lane >> 3, cphys = lane & 7;
        unsigned oob = 0;
        int it = 0;
        // (branch-free: `&&` chains compile to a saveexec / branch ladder per DMA instruction; unsigned compares fold the >= 0 tests)
        for (int gi = wave; gi < ngroups; gi += 8, ++it) {
            const int pix = gi * 8 + prow;
            const int ly = (pix * p.tw_magic) >> 16, lx = pix - ly * in_tw;     // pix / in_tw (exact for the tile sizes: checked on the host)
            const int iy = iy0 + ly * step, ix = ix0 + lx * step;
            const unsigned outside = (unsigned)(pix >= npix) | (unsigned)((unsigned)iy >= (unsigned)p.H) | (unsigned)((unsigned)ix >= (unsigned)p.W);
            const int cy = min(max(iy, 0), p.H - 1), cx = min(max(ix, 0), p.W - 1);
            // uniform base (plane + channel chunk) + 32-bit lane offset
            const unsigned voff = ((unsigned)(cy * p.W + cx) * (unsigned)p.in_ld + ((cphys ^ (pix & 7)) << 3)) * (unsigned)sizeof(HT);
            glds16_saddr(p.in + c0, voff, lds0 + buf * p.tile_bytes + gi * 1024);
            if constexpr (XS) glds16_saddr(p.in_lo + c0, voff, lds0 + buf * p.tile_bytes + LOBASE + gi * 1024);
            oob |= outside << it;
        }
        if constexpr (WS == 2) {
            // FP4 tiles: item = (64-pixel group, plane, window) -- the (plane, window) pair in the low bits, so the split is a mask and
            // a shift (a run-time division by n64 is ~16 scalar instructions per item); lane = pixel; out-of-image pixels are zeroed
            // later (bits 16+)
            const int pwbits = has_lo ? 2 : 1;
            it = 16;
            for (int item = wave; item < (n64 << pwbits); item += 8, ++it) {
                const int pw = item & ((1 << pwbits) - 1), gq = item >> pwbits, w = pw & 1, pl = pw >> 1;
                const int pix = gq * 64 + lane;
                const int ly = (pix * p.tw_magic) >> 16, lx = pix - ly * in_tw;
                const int iy = iy0 + ly * step, ix = ix0 + lx * step;
                const unsigned outside = (unsigned)(pix >= npix) | (unsigned)((unsigned)iy >= (unsigned)p.H) | (unsigned)((unsigned)ix >= (unsigned)p.W);
                const int cy = min(max(iy, 0), p.H - 1), cx = min(max(ix, 0), p.W - 1);
                // (plane pointer = first plane + pl * distance, not p.xq[pl]: a dynamically indexed kernel argument -- or a select
                // between two of them -- becomes a scalar load + lgkmcnt(0) per item)
                glds16_saddr(xq0 + pl * xq_delta + c0 / 2 + w * 16, (unsigned)(cy * p.W + cx) * (unsigned)(p.C / 2),
                             lds0 + buf * p.tile_bytes + QBASE + (pl * 2 + w) * QT + gq * 1024);
                oob |= outside << it;
            }
            // one scale dword per pixel and plane (the four windows of this 128-channel half); a clamped pixel's scales are valid
            for (int item = wave; item < (n64 << (pwbits - 1)); item += 8) {
                const int pl = item & ((1 << (pwbits - 1)) - 1), gq = item >> (pwbits - 1);
                const int pix = gq * 64 + lane;
                const int ly = (pix * p.tw_magic) >> 16, lx = pix - ly * in_tw;
                const int cy = min(max(iy0 + ly * step, 0), p.H - 1), cx = min(max(ix0 + lx * step, 0), p.W - 1);
                glds4_saddr(xs0 + pl * xs_delta + ((long long)(c0 >> 8) * p.x_srows) * 8 + ((c0 >> 5) & 4), (unsigned)(cy * p.W + cx) * 8u,
                            lds0 + buf * p.tile_bytes + SBASE + pl * ST + gq * 256);
            }
        }
        return oob;
    };
    const int per = (p.nsp + p.nslots - 1) / p.nslots;
    const int sp_step = p.walk ? 1 : p.nslots, sp_end = p.walk ? min(p.nsp, (slot + 1) * per) : p.nsp;
    int sp = p.walk ? slot * per : slot, buf = 0;
    unsigned oob = sp < sp_end ? stage(sp, 0) : 0u;

    // ---- this wave's window and its weight fragments (registers for ALL of the workgroup's tiles)
    const int win = wave & 1;                  // window inside the 64-channel chunk
    const int part = wave >> 1;                // which quarter of each tile's sub-tiles
    const int fr = lane & 15, kq = lane >> 4;
    constexpr int NT = WS == 1 ? 18 : 9;
    v8 wf[2][NT];
    {
        const HT* wp = p.w + (long long)(cchunk * 2 + win) * (2 * NT * 16 * 32);
#pragma unroll
        for (int nj = 0; nj < 2; ++nj)
#pragma unroll
            for (int t = 0; t < NT; ++t)
                wf[nj][t] = *reinterpret_cast<const v8*>(wp + ((nj * NT + t) * 16 + fr) * 32 + kq * 8);
    }
    int4 w4[WS == 2 ? 2 : 1][2][3];
    unsigned w4sc[3] = {0u, 0u, 0u};
    if constexpr (WS == 2) {
        const long long widx = cchunk * 2 + win;
        const char* w4p = p.w4 + widx * (2 * 2 * 3 * 64 * 16) + lane * 16;
#pragma unroll
        for (int nj = 0; nj < 2; ++nj)
#pragma unroll
            for (int pr = 0; pr < 2; ++pr)
#pragma unroll
                for (int g = 0; g < 3; ++g) w4[nj][pr][g] = *reinterpret_cast<const int4*>(w4p + ((nj * 2 + pr) * 3 + g) * (64 * 16));
        const unsigned* sp = reinterpret_cast<const unsigned*>(p.w4s + widx * (64 * 12) + lane * 12);
        w4sc[0] = sp[0]; w4sc[1] = sp[1]; w4sc[2] = sp[2];
    }
    const int ssel = 8 * (((c0 >> 5) & 3) + win);           // MX: bit offset of this window's scale byte in a pixel's scale dword
    // lane's 8 output channels: window base + q*8 + nj*4 + r
    const int cbase = c0 + win * 32 + kq * 8;
    const int chunk_in = win * 4 + kq;         // this lane's 16-byte chunk of the pixel row (8 input channels)
    f32x4 bias0, bias1;
    {
        const float4 b0 = *reinterpret_cast<const float4*>(p.bias + cbase);
        const float4 b1 = *reinterpret_cast<const float4*>(p.bias + cbase + 4);
        bias0 = f32x4{b0.x, b0.y, b0.z, b0.w};
        bias1 = f32x4{b1.x, b1.y, b1.z, b1.w};
    }

    // the first tile has landed; zero what lies outside the image
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    auto zero_oob = [&](unsigned mask, int b) {
        if (mask) {
            int it = 0;
            for (int gi = wave; gi < ngroups; gi += 8, ++it)
                if ((mask >> it) & 1u) {
                    *reinterpret_cast<uint4*>(lds + b * p.tile_bytes + gi * 1024 + lane * 16) = make_uint4(0u, 0u, 0u, 0u);
                    if constexpr (XS) *reinterpret_cast<uint4*>(lds + b * p.tile_bytes + LOBASE + gi * 1024 + lane * 16) = make_uint4(0u, 0u, 0u, 0u);
                }
            if constexpr (WS == 2) {
                const int pwbits = has_lo ? 2 : 1;
                it = 16;
                for (int item = wave; item < (n64 << pwbits); item += 8, ++it)
                    if ((mask >> it) & 1u) {
                        const int pw = item & ((1 << pwbits) - 1), gq = item >> pwbits;
                        *reinterpret_cast<uint4*>(lds + b * p.tile_bytes + QBASE + pw * QT + gq * 1024 + lane * 16) = make_uint4(0u, 0u, 0u, 0u);
                    }
            }
        }
    };
    zero_oob(oob, 0);
    __syncthreads();

    // LDS byte offsets of the nine taps of each of this lane's sub-tiles: tile-invariant.  With one or two sub-tiles per wave
    // they are kept in registers (9 x NJ); with four there is no room (256 VGPRs) and they are recomputed per tile.
    constexpr bool HOIST = NJ <= 2 && WS != 2;      // (the MX variant has no registers to spare either)
    int toff[HOIST ? NJ : 1][9];
    if constexpr (HOIST) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int st = part + 4 * j;
            const int sy = st >> 1, sx = (st & 1) * 16 + fr;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int pix = (sy * s + (t / 3) * d) * in_tw + sx * s + (t % 3) * d;
                toff[j][t] = pix * 128 + ((chunk_in ^ (pix & 7)) << 4);
            }
        }
    }
    // output addressing in 32 bits off the (uniform) plane bases -- the epilogue is VALU-issue bound, 64-bit address chains cost
    // as much as the arithmetic it is there for
    const unsigned o_ld = (unsigned)p.out_ld, q_ld = (unsigned)(p.C / 2);
    const unsigned sc_lane = (unsigned)(cbase >> 8) * (unsigned)p.o_srows * 8u + (unsigned)((cbase >> 5) & 7);

    // Per tile: [DMA of the next tile -> other buffer] [MFMA phase: this wave's sub-tiles, results stay in registers]
    // [s_waitcnt vmcnt(0): the next tile has landed -- the only older stores are those of the PREVIOUS tile, long retired, so
    // the wait never sits on fresh stores] [epilogue phase: bias, ReLU, hi/lo split, FP4 copies, stores] [barrier].
    unsigned long long tsum[6] = {0, 0, 0, 0, 0, 0};     // probe: [0] stage [1] MFMA phase [2] landing wait + zeroing [3] epilogue [4] barrier [5] tiles
    auto stamp = [&]() __attribute__((always_inline)) {
        unsigned long long t;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        return t;
    };
    for (; sp < sp_end; sp += sp_step, buf ^= 1) {
        unsigned long long t0 = 0, t1 = 0;
        if (kStamps && p.dbg) t0 = stamp();
        // The two waves of a SIMD (w and w + 4) stage the next tile at DIFFERENT points: waves 0-3 in front of their MFMA phase,
        // waves 4-7 in the middle of theirs -- the staging is ~40 vector instructions per DMA instruction (pixel -> clamped source
        // address), so one wave of the SIMD computes addresses while the other one's MFMAs run.
        const bool late_stage = NJ >= 2 && wave >= 4 && p.dephase;
        oob = 0u;
        if (!late_stage && sp + sp_step < sp_end) oob = stage(sp + sp_step, buf ^ 1);
        if (kStamps && p.dbg) { t1 = stamp(); tsum[0] += t1 - t0; t0 = t1; }
        const char* tile = lds + buf * p.tile_bytes;
        f32x4 acc[NJ][2];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (NJ >= 2 && j == NJ / 2 && late_stage && sp + sp_step < sp_end) oob = stage(sp + sp_step, buf ^ 1);
            // all nine tap fragments are requested before the first MFMA: the LDS latency is paid once per sub-tile
            v8 a[9];
            if constexpr (HOIST) {
#pragma unroll
                for (int t = 0; t < 9; ++t) a[t] = *reinterpret_cast<const v8*>(tile + toff[j][t]);
            } else {
                const int st = part + 4 * j;
                int frv = fr;
                asm volatile("" : "+v"(frv));       // keeps hipcc from hoisting the 36 addresses (spills)
                const int sy = st >> 1, sx = (st & 1) * 16 + frv;        // output pixel (tile-local) of this lane
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int pix = (sy * s + (t / 3) * d) * in_tw + sx * s + (t % 3) * d;
                    a[t] = *reinterpret_cast<const v8*>(tile + pix * 128 + ((chunk_in ^ (pix & 7)) << 4));
                }
            }
            // the accumulators start from the bias (the epilogue is VALU-issue bound next to the MFMAs: every op saved there counts)
            f32x4 acc0 = bias0, acc1 = bias1;
            if constexpr (WS == 2) {
                // FP4 operands: this lane's K block of scaled MFMA g is tap 4 g + kq (taps past the ninth carry zero weights: any pixel)
                typedef int v8i __attribute__((ext_vector_type(8)));
                const int st2 = part + 4 * j;
                const int sy2 = st2 >> 1, sx2 = (st2 & 1) * 16 + fr;
                int4 bq[2][3];
                int bs[2][3];
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    const int t = min(4 * g + kq, 8);
                    const int pix = (sy2 * s + (t / 3) * d) * in_tw + sx2 * s + (t % 3) * d;
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) {
                        if (pl == 1 && !has_lo) { bq[pl][g] = make_int4(0, 0, 0, 0); bs[pl][g] = 127; continue; }
                        bq[pl][g] = *reinterpret_cast<const int4*>(tile + QBASE + (pl * 2 + win) * QT + pix * 16);
                        bs[pl][g] = (int)((*reinterpret_cast<const unsigned*>(tile + SBASE + pl * ST + pix * 4) >> ssel) & 0xffu);
                    }
                }
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    acc0 = Half16<HT>::mfma(wf[0][t], a[t], acc0);
                    acc1 = Half16<HT>::mfma(wf[1][t], a[t], acc1);
                }
#define AVL_GMX(NJ_, PR, G, ACC)                                                                                                \
    {                                                                                                                          \
        const v8i wa = {w4[NJ_][PR][G].x, w4[NJ_][PR][G].y, w4[NJ_][PR][G].z, w4[NJ_][PR][G].w, 0, 0, 0, 0};                    \
        const v8i xa = {bq[PR][G].x, bq[PR][G].y, bq[PR][G].z, bq[PR][G].w, 0, 0, 0, 0};                                        \
        ACC = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wa, xa, ACC, 4, 4, ((NJ_) * 6 + (PR) * 3 + (G)) & 3,              \
                                                               (int)w4sc[((NJ_) * 6 + (PR) * 3 + (G)) >> 2], 0, bs[PR][G]);     \
    }
                AVL_GMX(0, 0, 0, acc0) AVL_GMX(0, 0, 1, acc0) AVL_GMX(0, 0, 2, acc0)
                AVL_GMX(1, 0, 0, acc1) AVL_GMX(1, 0, 1, acc1) AVL_GMX(1, 0, 2, acc1)
                if (has_lo) {
                    AVL_GMX(0, 1, 0, acc0) AVL_GMX(0, 1, 1, acc0) AVL_GMX(0, 1, 2, acc0)
                    AVL_GMX(1, 1, 0, acc1) AVL_GMX(1, 1, 1, acc1) AVL_GMX(1, 1, 2, acc1)
                }
#undef AVL_GMX
            } else {
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    acc0 = Half16<HT>::mfma(wf[0][t], a[t], acc0);
                    acc1 = Half16<HT>::mfma(wf[1][t], a[t], acc1);
                    if constexpr (WS == 1) {
                        acc0 = Half16<HT>::mfma(wf[0][9 + t], a[t], acc0);
                        acc1 = Half16<HT>::mfma(wf[1][9 + t], a[t], acc1);
                    }
                }
                if constexpr (XS) {          // Wh . xl: the same nine addresses in the lo tile (a[] is reused)
                    if constexpr (HOIST) {
#pragma unroll
                        for (int t = 0; t < 9; ++t) a[t] = *reinterpret_cast<const v8*>(tile + LOBASE + toff[j][t]);
                    } else {
                        const int st = part + 4 * j;
                        int frv = fr;
                        asm volatile("" : "+v"(frv));
                        const int sy = st >> 1, sx = (st & 1) * 16 + frv;
#pragma unroll
                        for (int t = 0; t < 9; ++t) {
                            const int pix = (sy * s + (t / 3) * d) * in_tw + sx * s + (t % 3) * d;
                            a[t] = *reinterpret_cast<const v8*>(tile + LOBASE + pix * 128 + ((chunk_in ^ (pix & 7)) << 4));
                        }
                    }
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        acc0 = Half16<HT>::mfma(wf[0][t], a[t], acc0);
                        acc1 = Half16<HT>::mfma(wf[1][t], a[t], acc1);
                    }
                }
            }
            acc[j][0] = acc0; acc[j][1] = acc1;
            __builtin_amdgcn_sched_barrier(0);      // keep the next sub-tile's nine fragments out of this one's registers
        }
        if (kStamps && p.dbg) { t1 = stamp(); tsum[1] += t1 - t0; t0 = t1; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        zero_oob(oob, buf ^ 1);
        if (kStamps && p.dbg) { t1 = stamp(); tsum[2] += t1 - t0; t0 = t1; }

        int tx, ty, cmb;
        if (p.walk) { ty = sp % p.tiles_y; const int r1 = sp / p.tiles_y; tx = r1 % p.tiles_x; cmb = r1 / p.tiles_x; }
        else { tx = sp % p.tiles_x; const int r1 = sp / p.tiles_x; ty = r1 % p.tiles_y; cmb = r1 / p.tiles_y; }
        const int ry = p.comb ? cmb / p.dil : 0, rx = p.comb ? cmb % p.dil : 0;
        const int oy0 = ty * p.th, ox0 = tx * TW;    // tile origin on its grid
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int st = part + 4 * j;
            const int sy = st >> 1, sx = (st & 1) * 16 + fr;
            const int oy = ry + (oy0 + sy) * step, ox = rx + (ox0 + sx) * step;
            const bool live = oy < p.OH && ox < p.OW;
            const unsigned pix = live ? (unsigned)oy * (unsigned)p.OW + (unsigned)ox : 0u;
            float v[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = fmaxf(acc[j][0][r], 0.f);
                v[4 + r] = fmaxf(acc[j][1][r], 0.f);
            }
            if constexpr (WS == 0) {
                if (live) Vec8<HT>::store(p.out + (pix * o_ld + (unsigned)cbase), v);
            } else {
                // ONE conversion to the 16-bit type: the packed vector is what gets stored AND what hi is read back from
                typedef typename Half16<HT>::v8 h8;
                typedef HT h2 __attribute__((ext_vector_type(2)));
                typedef unsigned u4 __attribute__((ext_vector_type(4)));
                u4 hpk;
#pragma unroll
                for (int r = 0; r < 4; ++r) {          // pairs: one v_cvt_pk_f16_f32 each (element-wise conversion cost a cvt + a perm per pair)
                    const h2 t = {(HT)v[2 * r], (HT)v[2 * r + 1]};
                    hpk[r] = __builtin_bit_cast(unsigned, t);
                }
                const h8 hv = __builtin_bit_cast(h8, hpk);
                if (live) *reinterpret_cast<h8*>(p.out + (pix * o_ld + (unsigned)cbase)) = hv;
                float lo[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const float h = (float)hv[r];
                    lo[r] = v[r] - h;
                    v[r] = h;
                }
                if (p.out_lo) {        // a stored lo plane is f16: its FP4 copy is taken from what it holds
                    h8 lv;
#pragma unroll
                    for (int r = 0; r < 8; ++r) lv[r] = (HT)lo[r];
                    if (live) *reinterpret_cast<h8*>(p.out_lo + (pix * o_ld + (unsigned)cbase)) = lv;
#pragma unroll
                    for (int r = 0; r < 8; ++r) lo[r] = (float)lv[r];
                }
                // FP4 copies: the window's 32 channels (the four kq lanes of this pixel) are one MX block.  The four lanes sit
                // 16 apart: v_permlane16_swap / v_permlane32_swap exchange them at VALU speed (no LDS round trip).
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    if (p.oq[pl] == nullptr) continue;
                    const float* src = pl == 0 ? v : lo;
                    float amax = 0.f;
#pragma unroll
                    for (int r = 0; r < 8; ++r) amax = fmaxf(amax, fabsf(src[r]));
                    {
                        const unsigned ab = __float_as_uint(amax);
                        const auto r16 = __builtin_amdgcn_permlane16_swap(ab, ab, false, false);       // rows (0,0,2,2) / (1,1,3,3)
                        const unsigned m16 = max(r16[0], r16[1]);                                        // non-negative floats order as integers
                        const auto r32 = __builtin_amdgcn_permlane32_swap(m16, m16, false, false);
                        amax = __uint_as_float(max(r32[0], r32[1]));
                    }
                    const unsigned sbyte = mx_fp4_scale_byte(amax);
                    const float scale = __uint_as_float(sbyte << 23);
                    unsigned pk = 0u;
                    pk = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(pk, src[0], src[1], scale, 0);
                    pk = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(pk, src[2], src[3], scale, 1);
                    pk = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(pk, src[4], src[5], scale, 2);
                    pk = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(pk, src[6], src[7], scale, 3);
                    // every lane stores its own 4 bytes (the window's four lanes fill 16 contiguous bytes); the scale byte leaves
                    // through the kq = 0 lane
                    if (live) {
                        *reinterpret_cast<unsigned*>(p.oq[pl] + (pix * q_ld + (unsigned)(cbase / 2))) = pk;
                        if (kq == 0) p.os[pl][sc_lane + pix * 8u] = (char)sbyte;
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (kStamps && p.dbg) { t1 = stamp(); tsum[3] += t1 - t0; t0 = t1; }
        __syncthreads();
        if (kStamps && p.dbg) { t1 = stamp(); tsum[4] += t1 - t0; tsum[5] += 1; }
    }
    if (kStamps && p.dbg && lane == 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) p.dbg[((size_t)blockIdx.x * 8 + wave) * 6 + i] = tsum[i];
    }
}

}  // namespace

int gconv_mfma_lds_bytes(int stride, int dil, int& th) {
    th = stride == 1 ? 8 : 4;
    const int in_th = (th - 1) * stride + 2 * dil + 1, in_tw = (TW - 1) * stride + 2 * dil + 1;
    return ((in_th * in_tw + 7) / 8) * 1024;      // whole 8-pixel DMA groups
}

static int gconv_tile_bytes(int stride, int dil, int th, bool mx = false, bool xs = false) {
    const int in_th = (th - 1) * stride + 2 * dil + 1, in_tw = (TW - 1) * stride + 2 * dil + 1;
    const int npix = in_th * in_tw, n64 = (npix + 63) / 64;
    return ((npix + 7) / 8) * 1024 * (xs ? 2 : 1) + (mx ? 4 * n64 * 1024 + 2 * n64 * 256 : 0);      // + the lo plane's tile (XS) / + FP4 tiles and scale dwords (WS = 2)
}

static int device_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

// Tile height of the persistent kernel: two tile buffers must fit the CU's LDS, and among 8 and 4 (2 only when nothing else fits)
// the one with the shortest schedule -- rounds of the slowest workgroup x (rows a round computes + a per-tile overhead): layers with
// few tiles per workgroup lose less to the last, partly filled round and to the rows past the image with the lower tile
// (measured at 1080p: layer4 88 -> 83 us, layer2 32 -> 30 us with 4 rows; layer1 ties and keeps 8).
static int gconv_pick_th(const avl_seg_op& op, int comb, int cus, int* nsp_out, int* nslots_out) {
    const bool mx = op.w_split == 2, xs = op.in_lo != nullptr;
    const int d = comb ? 1 : op.dil;
    const int gh = comb ? (op.out_h + op.dil - 1) / op.dil : op.out_h, gw = comb ? (op.out_w + op.dil - 1) / op.dil : op.out_w;
    const int ncomb = comb ? op.dil * op.dil : 1, cchunks = op.in_c / CC;
    int best = 0;
    double best_cost = 0.;
    const int env_th = AVL_EXP_INT("AVL_GCONV_TH", 0);      // experiments build only
    for (int th = 8; th >= 2; th >>= 1) {
        if (2 * gconv_tile_bytes(op.stride, d, th, mx, xs) > 160 * 1024) continue;
        if ((mx || xs) && th == 8) continue;         // the MX variant with four sub-tiles per wave spills; the split-input one has no LDS for it
        const int nsp = ((gw + TW - 1) / TW) * ((gh + th - 1) / th) * ncomb;
        int nslots = cus / cchunks;
        if (nslots < 1) nslots = 1;
        if (nslots > nsp) nslots = nsp;
        const int rounds = (nsp + nslots - 1) / nslots;
        const double cost = (double)rounds * (th + 0.5);
        if (th == 2 && best != 0 && env_th != 2) break;
        if (best == 0 || cost < best_cost || th == env_th) {
            best = th; best_cost = th == env_th ? -1. : cost;
            *nsp_out = nsp; *nslots_out = nslots;
        }
    }
    return best;
}

template <typename HT, int WS, bool XS = false>
int launch_gconv_typed(const avl_seg_op& op, hipStream_t s) {
    GconvArgs<HT> a;
    a.in = static_cast<const HT*>(op.in);
    a.in_lo = static_cast<const HT*>(op.in_lo);
    a.w = static_cast<const HT*>(op.weight);
    a.bias = op.bias;
    a.out = static_cast<HT*>(op.out);
    a.out_lo = static_cast<HT*>(op.out_lo);
    a.oq[0] = a.oq[1] = a.os[0] = a.os[1] = nullptr;
    a.o_srows = op.out_rows;
    if (op.out_mx) {
        char* b = static_cast<char*>(op.out_mx);
        const long long rows = op.out_rows, P = rows * (op.out_c / 2), S = (long long)(op.out_c / 256) * rows * 8;
        a.oq[0] = b; a.os[0] = b + P;
        if (op.out_lo || (op.mx_flags & AVL_MX_OUT_LO)) { a.oq[1] = b + P + S; a.os[1] = b + 2 * P + S; }
    }
    a.H = op.in_h; a.W = op.in_w; a.in_ld = op.in_ld; a.OH = op.out_h; a.OW = op.out_w; a.out_ld = op.out_ld; a.C = op.in_c;
    a.stride = op.stride; a.dil = op.dil;
    a.comb = (op.stride == 1 && op.dil > 1 && op.pad == op.dil) ? 1 : 0;
    a.th = gconv_pick_th(op, a.comb, device_cus(), &a.nsp, &a.nslots);
    AVL_REQUIRE(a.th > 0, "grouped conv: no tile height fits two LDS buffers");
    a.tile_bytes = gconv_tile_bytes(op.stride, a.comb ? 1 : op.dil, a.th, WS == 2, XS);
    a.w4 = a.w4s = a.xq[0] = a.xq[1] = a.xs[0] = a.xs[1] = nullptr;
    a.x_srows = op.in_rows;
    if (WS == 2) {
        const long long nwin = op.in_c / 32;
        a.w4 = static_cast<const char*>(op.w_mx);
        a.w4s = a.w4 + nwin * (2 * 2 * 3 * 64 * 16);
        const char* b = static_cast<const char*>(op.in_mx);
        const long long rows = op.in_rows, P = rows * (op.in_c / 2), S = (long long)(op.in_c / 256) * rows * 8;
        a.xq[0] = b; a.xs[0] = b + P;
        if (op.mx_flags & AVL_MX_IN_LO) { a.xq[1] = b + P + S; a.xs[1] = b + 2 * P + S; }
    }
    {
        const int d = a.comb ? 1 : op.dil;
        const int in_th = (a.th - 1) * op.stride + 2 * d + 1, in_tw = (TW - 1) * op.stride + 2 * d + 1;
        a.tw_magic = (65536 + in_tw - 1) / in_tw;
        for (int pix = 0; pix < ((in_th * in_tw + 63) / 64) * 64; ++pix)
            AVL_REQUIRE(((pix * a.tw_magic) >> 16) == pix / in_tw, "grouped conv: tile %d x %d too large for the reciprocal division", in_th, in_tw);
    }
    const int gh = a.comb ? (op.out_h + op.dil - 1) / op.dil : op.out_h, gw = a.comb ? (op.out_w + op.dil - 1) / op.dil : op.out_w;
    a.tiles_x = (gw + TW - 1) / TW;
    a.tiles_y = (gh + a.th - 1) / a.th;
    a.cchunks = op.in_c / CC;
#define AVL_GCONV_LAUNCH(NJ)                                                                                                          \
    do {                                                                                                                              \
        AVL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gconv_mfma<HT, WS, NJ, XS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        hipLaunchKernelGGL((k_gconv_mfma<HT, WS, NJ, XS>), dim3(a.nslots * a.cchunks), dim3(512), 2 * a.tile_bytes, s, a);             \
    } while (0)
    a.dephase = AVL_EXP_INT("AVL_GC_DEPHASE", 1);
    a.walk = AVL_EXP_INT("AVL_GC_WALK", 1);
    a.dbg = nullptr;
#ifdef AVL_EXPERIMENTS
    // timing experiment: where do a wave's cycles go (s_memtime stamps; synchronises the stream: never inside a graph capture)
    static unsigned long long* dbg = nullptr;
    if (AVL_EXP_INT("AVL_GC_PROBE", 0)) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        AVL_HIP_CHECK(hipStreamIsCapturing(s, &cap));
        AVL_REQUIRE(cap == hipStreamCaptureStatusNone, "AVL_GC_PROBE synchronises the stream: not while it is being captured (MODEL.HIP_GRAPH = False)");
        if (!dbg) AVL_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&dbg), 256 * 8 * 6 * sizeof(unsigned long long), 0));
        memset(dbg, 0, 256 * 8 * 6 * sizeof(unsigned long long));
        if (a.nslots * a.cchunks <= 256) a.dbg = dbg;
    }
#endif
    // (the MX variant never runs four sub-tiles per wave -- gconv_pick_th: it would spill -- so that kernel is not even instantiated)
    if (a.th == 8) {
        if constexpr (WS != 2 && !XS) AVL_GCONV_LAUNCH(4);
        else return avl::set_error(AVL_E_ARG, "MX / split-input grouped conv: tile height 8 is not built");
    }
    else if (a.th == 4) AVL_GCONV_LAUNCH(2);
    else AVL_GCONV_LAUNCH(1);
#undef AVL_GCONV_LAUNCH
    AVL_LAUNCH_CHECK();
#ifdef AVL_EXPERIMENTS
    if (a.dbg) {
        AVL_HIP_CHECK(hipStreamSynchronize(s));
        double sum[6] = {};
        for (int i = 0; i < a.nslots * a.cchunks * 8; ++i)
            for (int k = 0; k < 6; ++k) sum[k] += (double)dbg[i * 6 + k];
        const double n = sum[5] > 0 ? sum[5] : 1;
        fprintf(stderr, "[gconv probe] C %d dil %d stride %d th %d WS %d tiles/wg %.1f: cycles per tile: stage %.0f | MFMA phase %.0f | landing wait + zeroing %.0f | epilogue %.0f | barrier %.0f\n",
                a.C, a.dil, a.stride, a.th, WS, n / (a.nslots * a.cchunks * 8), sum[0] / n, sum[1] / n, sum[2] / n, sum[3] / n, sum[4] / n);
    }
#endif
    return AVL_OK;
}

int launch_gconv_mfma(const avl_seg_op& op, hipStream_t s) {
    if (op.w_split == 2) return launch_gconv_typed<f16, 2>(op, s);
    if (op.w_split && op.in_lo) return launch_gconv_typed<f16, 1, true>(op, s);
    if (op.w_split) return launch_gconv_typed<f16, 1>(op, s);
    return op.dtype == AVL_F16 ? launch_gconv_typed<f16, 0>(op, s) : launch_gconv_typed<bf16, 0>(op, s);
}

int validate_gconv_mfma(const avl_seg_op& op) {
    AVL_REQUIRE(is_half(op.dtype), "MFMA grouped conv needs a 16-bit activation type");
    AVL_REQUIRE(op.in_c % CC == 0, "MFMA grouped conv needs channels %% 64 == 0 (got %d)", op.in_c);
    AVL_REQUIRE(!op.w_split || op.dtype == AVL_F16, "split weights need AVL_F16 activations");
    AVL_REQUIRE(!op.in2_lo && (!op.out_lo || op.w_split) && (!op.in_lo || op.w_split == 1), "grouped conv: a split output needs w_split, a split input w_split = 1");
    AVL_REQUIRE(!op.in_lo || reinterpret_cast<uintptr_t>(op.in_lo) % 16 == 0, "grouped conv: unaligned lo plane");
    if (op.w_split == 2) {
        AVL_REQUIRE(op.w_mx && op.in_mx && op.in_c % 256 == 0 && op.in_ld == op.in_c, "MX grouped conv: w_mx, in_mx, channels %% 256 == 0, dense input rows");
        AVL_REQUIRE((reinterpret_cast<uintptr_t>(op.w_mx) | reinterpret_cast<uintptr_t>(op.in_mx)) % 16 == 0, "MX grouped conv: unaligned bundles");
        AVL_REQUIRE(2 * gconv_tile_bytes(op.stride, (op.stride == 1 && op.pad == op.dil) ? 1 : op.dil, 2, true) <= 160 * 1024, "MX grouped conv: two tile buffers do not fit LDS");
        AVL_REQUIRE((long long)op.in_rows * (op.in_c / 2) < (1LL << 31), "MX grouped conv: FP4 plane beyond 2 GB (32-bit DMA offsets)");
    }
    AVL_REQUIRE(!op.out_mx || (op.w_split >= 1 && op.out_c % 256 == 0 && op.out_ld == op.out_c), "grouped conv: out_mx needs w_split, channels %% 256 == 0 and a dense output");
    AVL_REQUIRE(!(op.mx_flags & AVL_MX_OUT_LO) || (op.out_mx && !op.out_lo), "grouped conv: AVL_MX_OUT_LO needs out_mx and no out_lo");
    AVL_REQUIRE((long long)op.in_rows * op.in_ld * 2 < (1LL << 31), "grouped conv: input plane beyond 2 GB (32-bit DMA offsets)");
    AVL_REQUIRE((long long)op.out_rows * op.out_ld * 2 < (1LL << 31) && (long long)(op.out_c / 256 + 1) * op.out_rows * 8 < (1LL << 31),
                "grouped conv: output planes beyond 2 GB (32-bit epilogue addressing)");
    const int cg = op.in_c / op.groups;
    AVL_REQUIRE(cg <= 32 && 32 % cg == 0, "MFMA grouped conv needs <= 32 channels per group dividing 32 (got %d)", cg);
    AVL_REQUIRE(2 * gconv_tile_bytes(op.stride, (op.stride == 1 && op.pad == op.dil) ? 1 : op.dil, 2, false, op.in_lo != nullptr) <= 160 * 1024, "grouped conv: two tile buffers do not fit LDS (dilation %d)", op.dil);
    return AVL_OK;
}

}  // namespace avl
