// Stem on the matrix cores (bf16): uint8 RGB -> ToTensor/Normalize (semantic_segmentation.py:35-39) ->
// 7x7 stride-2 pad-3 conv 3->64 (torchvision ResNet.conv1) + folded bn1 + ReLU, NHWC bf16 out.
//
// Implicit GEMM with K laid out as [ky 7][kx*3+ci padded 21 -> 24]: inside one kernel row the 21 taps are 21
// CONTIGUOUS bytes of the image row, so a lane's 8-wide K fragment is 8 consecutive values of the normalised
// tile in LDS (the 3 pad positions per row carry zero weights).  K = 7*24 = 168 -> 6 MFMA steps of 32 (the last
// 24 are zero weights).  A workgroup converts an input tile (8x32 outputs -> 21x69 pixels) to normalised bf16 in
// LDS once (zeros outside the image: padding applies to the NORMALISED image), keeps all 24 weight fragments in
// registers and each wave walks 4 sub-tiles of 16 pixels: 21 LDS reads + 24 v_mfma_f32_16x16x32_bf16 per sub-tile.
//
// PRE = the node's pre-processing (vision_semantic_segmentation_node.py:83-98: BGR->RGB, cv2.undistort, INTER_AREA by an
// integer factor) happens in the loader: `img` is then the RAW camera frame and every pixel of the LDS tile is
// preprocessed_rgb() of seg_preprocess.h -- the function k_preprocess applies -- so the RGB network input is never written
// to memory or re-read.  The 5-pixel halo of a tile is recomputed (x1.41 pixels), which costs less than
// the round trip: the undistortion is ~60 double-precision flops per source pixel.
#include "seg_types.h"
#include "seg_preprocess.h"

namespace avl {
namespace {

constexpr int S_TH = 8, S_TW = 32;
constexpr int IN_TH = 2 * S_TH + 5, IN_TW = 2 * S_TW + 5;     // 21 x 69 input pixels
constexpr int ROW = 224;                                      // bf16 values per LDS row (>= 69*3 + slack for the pad taps)

template <typename HT>
struct StemArgs {
    const unsigned char* img;
    const HT* w;           // [nj 4][step 6][i 16][k 32]; SPLIT: the hi parts, then the same array of lo parts
    const float* bias;    // [64]
    HT* out;
    HT* out_lo;            // SPLIT: the result's lo plane
    int H, W, OH, OW, out_ld, tiles_x;
    const PreCamera* cam;      // PRE: device memory (one captured graph serves both cameras)
    int srcH, srcW, factor;    // PRE: the raw frame; H = srcH / factor, W = srcW / factor
};

// SPLIT (the complete hi + lo pipeline): the NORMALISED image is kept as two f16 tiles (value = hi + lo), the weights are
// f16 pairs, the product runs Wh.xh + Wl.xh + Wh.xl and the result leaves as hi + lo planes: no f16-class rounding anywhere.
template <typename HT, bool PRE, bool SPLIT = false>
__global__ void __launch_bounds__(256) k_stem_mfma(StemArgs<HT> p) {
    typedef typename Half16<HT>::v8 v8;
    __shared__ __attribute__((aligned(16))) HT tile[(IN_TH + 1) * ROW];
    __shared__ __attribute__((aligned(16))) HT tile_lo[SPLIT ? (IN_TH + 1) * ROW : 8];
    __shared__ HT lut[3 * 256];        // normalised value of every (channel, byte): exact divisions, done once
    __shared__ HT lut_lo[SPLIT ? 3 * 256 : 8];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tx = blockIdx.x % p.tiles_x, ty = blockIdx.x / p.tiles_x;
    const int oy0 = ty * S_TH, ox0 = tx * S_TW;
    const int iy0 = oy0 * 2 - 3, ix0 = ox0 * 2 - 3;
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    if constexpr (PRE) {
#pragma unroll
        for (int ci = 0; ci < 3; ++ci) {
            const float v = ((float)tid / 255.0f - mean[ci]) / stdv[ci];
            lut[ci * 256 + tid] = (HT)v;
            if constexpr (SPLIT) lut_lo[ci * 256 + tid] = (HT)(v - (float)(HT)v);
        }
        // the slack of every row and the extra row (read by the zero-weight pad taps)
        for (int e = tid; e < (IN_TH + 1) * ROW; e += 256) {
            const int ly = e / ROW, lc = e - ly * ROW;
            if (ly >= IN_TH || lc >= IN_TW * 3) { tile[e] = (HT)0.f; if constexpr (SPLIT) tile_lo[e] = (HT)0.f; }
        }
        const PreCamera cam = *p.cam;
        __syncthreads();
        for (int i = tid; i < IN_TH * IN_TW; i += 256) {
            const int ly = i / IN_TW, lx = i - ly * IN_TW;
            const int iy = iy0 + ly, ix = ix0 + lx;
            const bool ok = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            int rgb[3] = {0, 0, 0};
            if (ok) preprocessed_rgb(p.img, p.srcH, p.srcW, cam, p.factor, ix, iy, rgb);
            HT* t = tile + ly * ROW + lx * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) t[c] = ok ? lut[c * 256 + rgb[c]] : (HT)0.f;
            if constexpr (SPLIT) {
                HT* tl = tile_lo + ly * ROW + lx * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) tl[c] = ok ? lut_lo[c * 256 + rgb[c]] : (HT)0.f;
            }
        }
    } else {
        // all of a lane's byte loads are issued before the first conversion (the element-at-a-time loop was a chain of
        // ~20 dependent global-load latencies per workgroup and cost more than the MFMAs)
        constexpr int NE = ((IN_TH + 1) * ROW + 255) / 256;
        unsigned char px[NE];
        unsigned okmask = 0;
    #pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = tid + i * 256;
            const int ly = e / ROW, lc = e - ly * ROW;
            const int lx = lc / 3;
            const int iy = iy0 + ly, ix = ix0 + lx;
            const bool ok = ly < IN_TH && lx < IN_TW && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            const int cy = min(max(iy, 0), p.H - 1), cx = min(max(ix, 0), p.W - 1);      // unconditional load, masked below
            px[i] = p.img[((long long)cy * p.W + cx) * 3 + (lc - lx * 3)];
            okmask |= (ok ? 1u : 0u) << i;
        }
        // while the loads are in flight: the 768-entry table
    #pragma unroll
        for (int ci = 0; ci < 3; ++ci) {
            const float v = ((float)tid / 255.0f - mean[ci]) / stdv[ci];
            lut[ci * 256 + tid] = (HT)v;
            if constexpr (SPLIT) lut_lo[ci * 256 + tid] = (HT)(v - (float)(HT)v);
        }
        __syncthreads();
    #pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = tid + i * 256;
            if (e < (IN_TH + 1) * ROW) {
                const int ci = (e % ROW) % 3;
                tile[e] = ((okmask >> i) & 1u) ? lut[ci * 256 + px[i]] : (HT)0.f;
                if constexpr (SPLIT) tile_lo[e] = ((okmask >> i) & 1u) ? lut_lo[ci * 256 + px[i]] : (HT)0.f;
            }
        }
    }
    const int fr = lane & 15, kq = lane >> 4;
    v8 wf[4][6];
#pragma unroll
    for (int nj = 0; nj < 4; ++nj)
#pragma unroll
        for (int st = 0; st < 6; ++st)
            wf[nj][st] = *reinterpret_cast<const v8*>(p.w + ((nj * 6 + st) * 16 + fr) * 32 + kq * 8);
    float bias[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float4 b = *reinterpret_cast<const float4*>(p.bias + kq * 16 + 4 * j);
        bias[4 * j] = b.x; bias[4 * j + 1] = b.y; bias[4 * j + 2] = b.z; bias[4 * j + 3] = b.w;
    }
    __syncthreads();

    for (int sub = wave; sub < S_TH * (S_TW / 16); sub += 4) {
        const int sy = sub >> 1, sx = (sub & 1) * 16 + fr;
        f32x4 acc[4];
#pragma unroll
        for (int nj = 0; nj < 4; ++nj) acc[nj] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < 6; ++st) {
            // chunk index q = st*4 + kq in [0,24): kernel row ky = q/3, 8-wide piece (q%3) of its 24 taps
            const int q = st * 4 + kq;
            const int ky = q / 3, piece = q - ky * 3;
            v8 a;
            if (q < 21) {
                const uint32_t* src = reinterpret_cast<const uint32_t*>(tile + (sy * 2 + ky) * ROW + sx * 6 + piece * 8);
                uint32_t u[4] = {src[0], src[1], src[2], src[3]};
                __builtin_memcpy(&a, u, 16);
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] = (HT)0.f;
            }
#pragma unroll
            for (int nj = 0; nj < 4; ++nj) acc[nj] = Half16<HT>::mfma(wf[nj][st], a, acc[nj]);
            if constexpr (SPLIT) {
                v8 al;
                if (q < 21) {
                    const uint32_t* src = reinterpret_cast<const uint32_t*>(tile_lo + (sy * 2 + ky) * ROW + sx * 6 + piece * 8);
                    uint32_t u[4] = {src[0], src[1], src[2], src[3]};
                    __builtin_memcpy(&al, u, 16);
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) al[i] = (HT)0.f;
                }
#pragma unroll
                for (int nj = 0; nj < 4; ++nj) {
                    // the lo parts of the weights are not kept in registers (96 more): 24 L1-resident 16-byte loads per sub-tile
                    const v8 wl = *reinterpret_cast<const v8*>(p.w + 4 * 6 * 16 * 32 + ((nj * 6 + st) * 16 + fr) * 32 + kq * 8);
                    acc[nj] = Half16<HT>::mfma(wf[nj][st], al, acc[nj]);
                    acc[nj] = Half16<HT>::mfma(wl, a, acc[nj]);
                }
            }
        }
        const int oy = oy0 + sy, ox = ox0 + sx;
        if (oy < p.OH && ox < p.OW) {
            float lo[8], hi[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                lo[r] = fmaxf(acc[0][r] + bias[r], 0.f);
                lo[4 + r] = fmaxf(acc[1][r] + bias[4 + r], 0.f);
                hi[r] = fmaxf(acc[2][r] + bias[8 + r], 0.f);
                hi[4 + r] = fmaxf(acc[3][r] + bias[12 + r], 0.f);
            }
            HT* op = p.out + ((long long)oy * p.OW + ox) * p.out_ld + kq * 16;
            if constexpr (SPLIT) {
                float l0[8], l1[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const float h0 = (float)(HT)lo[r], h1 = (float)(HT)hi[r];
                    l0[r] = lo[r] - h0; l1[r] = hi[r] - h1;
                    lo[r] = h0; hi[r] = h1;
                }
                HT* ol = p.out_lo + ((long long)oy * p.OW + ox) * p.out_ld + kq * 16;
                Vec8<HT>::store(ol, l0);
                Vec8<HT>::store(ol + 8, l1);
            }
            Vec8<HT>::store(op, lo);
            Vec8<HT>::store(op + 8, hi);
        }
    }
}

}  // namespace

template <typename HT>
int launch_stem_typed(const avl_seg_op& op, hipStream_t s) {
    StemArgs<HT> a;
    a.img = static_cast<const unsigned char*>(op.in);
    a.w = static_cast<const HT*>(op.weight);
    a.bias = op.bias;
    a.out = static_cast<HT*>(op.out);
    a.out_lo = static_cast<HT*>(op.out_lo);
    a.H = op.in_h; a.W = op.in_w; a.OH = op.out_h; a.OW = op.out_w; a.out_ld = op.out_ld;
    a.tiles_x = (op.out_w + S_TW - 1) / S_TW;
    const int tiles_y = (op.out_h + S_TH - 1) / S_TH;
    a.cam = static_cast<const PreCamera*>(op.in2);
    a.srcW = op.in2_ld;
    a.srcH = op.in2_ld > 0 ? op.in_rows / op.in2_ld : 0;
    a.factor = op.in_w > 0 ? a.srcW / op.in_w : 1;
    if (op.w_split) {
        if (!op.out_lo) return set_error(AVL_E_ARG, "split stem (w_split = 1): out_lo is NULL");
        if (op.in2)
            hipLaunchKernelGGL((k_stem_mfma<HT, true, true>), dim3(a.tiles_x * tiles_y), dim3(256), 0, s, a);
        else
            hipLaunchKernelGGL((k_stem_mfma<HT, false, true>), dim3(a.tiles_x * tiles_y), dim3(256), 0, s, a);
    } else if (op.in2)
        hipLaunchKernelGGL((k_stem_mfma<HT, true>), dim3(a.tiles_x * tiles_y), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((k_stem_mfma<HT, false>), dim3(a.tiles_x * tiles_y), dim3(256), 0, s, a);
    AVL_LAUNCH_CHECK();
    return AVL_OK;
}

int launch_stem_mfma(const avl_seg_op& op, hipStream_t s) {
    return op.dtype == AVL_F16 ? launch_stem_typed<f16>(op, s) : launch_stem_typed<bf16>(op, s);
}

}  // namespace avl
