// The non-GEMM layers of the segmentation network (gfx950): stem, max-pool, grouped 3x3,
// depthwise 3x3, bilinear upsample, global average pool, the pooled branch's GEMVs, arg-max.
// All are NHWC, 16-byte vector accesses along channels, fp32 arithmetic whatever the storage type.
// Weights that every lane reads at the same address (stem, grouped conv) are left in global memory
// on purpose: the address is wave-uniform, so hipcc fetches them through the scalar cache and the
// inner loops are pure v_fmac with an SGPR operand.
#include "seg_types.h"
#include "seg_preprocess.h"

namespace avl {
namespace {

constexpr int kThreads = 256;

// ------------------------------------------------------------------------------------------ stem
// semantic_segmentation.py:35-39 (ToTensor: u8/255, Normalize: (x-mean)/std) + torchvision
// ResNet.conv1 (7x7, stride 2, pad 3, 3->64) + folded bn1 + ReLU.  Zero padding applies to the
// NORMALISED image, so normalisation cannot be folded into the weights at the border.
// weight: float [7][7][3][64]; one lane = one output pixel x 64 channels.
template <typename T>
__global__ void __launch_bounds__(kThreads) k_stem(const unsigned char* __restrict__ img, int H, int W,
                                                  const float* __restrict__ w, const float* __restrict__ bias,
                                                  T* __restrict__ out, int OH, int OW, int out_ld) {
    const int idx = blockIdx.x * kThreads + threadIdx.x;
    if (idx >= OH * OW) return;
    const int oy = idx / OW, ox = idx % OW;
    float acc[64];
#pragma unroll
    for (int c = 0; c < 64; ++c) acc[c] = bias[c];
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    for (int ky = 0; ky < 7; ++ky) {
        const int iy = oy * 2 - 3 + ky;
        for (int kx = 0; kx < 7; ++kx) {
            const int ix = ox * 2 - 3 + kx;
            const bool in = iy >= 0 && iy < H && ix >= 0 && ix < W;
            const unsigned char* px = img + 3ll * ((long long)(in ? iy : 0) * W + (in ? ix : 0));
#pragma unroll
            for (int ci = 0; ci < 3; ++ci) {
                const float x = in ? ((float)px[ci] / 255.0f - mean[ci]) / stdv[ci] : 0.0f;
                const float* wr = w + ((ky * 7 + kx) * 3 + ci) * 64;
#pragma unroll
                for (int c = 0; c < 64; ++c) acc[c] = fmaf(x, wr[c], acc[c]);
            }
        }
    }
    T* op = out + (long long)idx * out_ld;
#pragma unroll
    for (int c8 = 0; c8 < 8; ++c8) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = fmaxf(acc[c8 * 8 + i], 0.f);
        Vec8<T>::store(op + c8 * 8, v);
    }
}

// --------------------------------------------------------------------------------------- maxpool
// Split planes (in_lo / out_lo, f16): the maximum is taken over the VALUES hi + lo (exact in fp32: 22 bits), and the winner is split again --
// which reproduces that input's own (hi, lo) pair.
template <typename T>
__global__ void __launch_bounds__(kThreads) k_maxpool(const T* __restrict__ in, int H, int W, int C, int in_ld,
                                                     T* __restrict__ out, int OH, int OW, int out_ld,
                                                     const T* __restrict__ in_lo = nullptr, T* __restrict__ out_lo = nullptr) {
    const int c8n = C / 8;
    const long long idx = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (idx >= (long long)OH * OW * c8n) return;
    const int c8 = (int)(idx % c8n);
    const int pix = (int)(idx / c8n), oy = pix / OW, ox = pix % OW;
    float m[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) m[i] = -INFINITY;
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy * 2 - 1 + ky;
        if (iy < 0 || iy >= H) continue;
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = ox * 2 - 1 + kx;
            if (ix < 0 || ix >= W) continue;
            float v[8];
            Vec8<T>::load(in + ((long long)iy * W + ix) * in_ld + c8 * 8, v);
            if (in_lo) {
                float l[8];
                Vec8<T>::load(in_lo + ((long long)iy * W + ix) * in_ld + c8 * 8, l);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] += l[i];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) m[i] = fmaxf(m[i], v[i]);
        }
    }
    if (out_lo) {
        float h[8], l[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { h[i] = to_f32(from_f32<T>(m[i])); l[i] = m[i] - h[i]; }
        Vec8<T>::store(out + (long long)pix * out_ld + c8 * 8, h);
        Vec8<T>::store(out_lo + (long long)pix * out_ld + c8 * 8, l);
    } else {
        Vec8<T>::store(out + (long long)pix * out_ld + c8 * 8, m);
    }
}

// ------------------------------------------------------------------------------- grouped 3x3 conv
// torchvision Bottleneck.conv2 (groups = 32) + folded bn2 + ReLU.  weight: float
// [group][tap = ky*3+kx][ci][co]; one lane = one output pixel of one group (CG in, CG out).
template <typename T, int CG>
__global__ void __launch_bounds__(kThreads) k_gconv(const T* __restrict__ in, int H, int W, int in_ld,
                                                   const float* __restrict__ w, const float* __restrict__ bias,
                                                   T* __restrict__ out, int OH, int OW, int out_ld, int stride, int dil) {
    const int g = blockIdx.y;
    const int pix = blockIdx.x * kThreads + threadIdx.x;
    if (pix >= OH * OW) return;
    const int oy = pix / OW, ox = pix % OW;
    float acc[CG];
#pragma unroll
    for (int c = 0; c < CG; ++c) acc[c] = bias[g * CG + c];
    const float* wg = w + (long long)g * 9 * CG * CG;
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy * stride + (ky - 1) * dil;
        if (iy < 0 || iy >= H) continue;
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = ox * stride + (kx - 1) * dil;
            if (ix < 0 || ix >= W) continue;
            const T* ip = in + ((long long)iy * W + ix) * in_ld + g * CG;
            float x[CG];
            if constexpr (CG >= 8) {
#pragma unroll
                for (int c8 = 0; c8 < CG / 8; ++c8) {
                    float v[8];
                    Vec8<T>::load(ip + c8 * 8, v);
#pragma unroll
                    for (int i = 0; i < 8; ++i) x[c8 * 8 + i] = v[i];
                }
            } else {
#pragma unroll
                for (int c = 0; c < CG; ++c) x[c] = to_f32(ip[c]);
            }
            const float* wt = wg + (ky * 3 + kx) * CG * CG;
#pragma unroll
            for (int ci = 0; ci < CG; ++ci)
#pragma unroll
                for (int co = 0; co < CG; ++co) acc[co] = fmaf(x[ci], wt[ci * CG + co], acc[co]);
        }
    }
    T* op = out + (long long)pix * out_ld + g * CG;
    if constexpr (CG >= 8) {
#pragma unroll
        for (int c8 = 0; c8 < CG / 8; ++c8) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = fmaxf(acc[c8 * 8 + i], 0.f);
            Vec8<T>::store(op + c8 * 8, v);
        }
    } else {
#pragma unroll
        for (int c = 0; c < CG; ++c) op[c] = from_f32<T>(fmaxf(acc[c], 0.f));
    }
}

// ------------------------------------------------------------------------------ depthwise 3x3 conv
// core/nn/modules/conv.py:131-134 (groups = in_channels) + folded BN + ReLU.
// weight: float [9][C]; one lane = one output pixel x 8 channels.
// Work decomposition is by "comb": for dilation d the outputs with (y mod d, x mod d) = (ry, rx) only ever read
// inputs of that same residue class (pad == d in ASPP), so a workgroup that owns a comb shares no input with any
// other workgroup: no halo is re-fetched by another XCD (PMC showed 3-4.6x read amplification for the pixel-major
// order, whose vertical neighbours land on different XCDs / L2s).  256 lanes = up to 64 channel chunks of 8 (fastest:
// 1 KB of a pixel is one contiguous read; wider groups push the rows a workgroup re-reads out of L2) x column lanes.
// A lane keeps its 9 x 8 weights in registers, walks DOWN its grid column with a rolling 3-row window (3 loads per
// output, two rows prefetched ahead) and chains small combs until it has ~32 outputs per weight load.
// Dilation 1 (decoder) is a single comb, cut into bands of 16 rows x 2 columns per lane.
// Measured at 1080p (5 layers): 0.44 ms pixel-per-lane with 9 loads -> 0.335 ms.
struct DwGeom {
    int H, W, C, in_ld, OH, OW, out_ld, pad, dil, relu;
    int gh, gw;            // comb-grid size: outputs per residue class
    int band_h, band_w;    // grid rows / columns per workgroup
    int nby, nbx;          // bands per comb
    int nchunk, cl;        // channel-chunk lanes and pixel lanes per workgroup
    int cgroups;           // workgroups along channels (C / 8 / nchunk)
    int units, upb;        // (comb, band) units in total / per workgroup
};

// SPLIT ("mixed" precision, f16): input and output are two planes hi + lo (in_lo / out_lo, same offsets); the taps are
// summed to fp32 (exact) and go through the fp32 FMA chain with the fp32 weights, nothing is rounded away at the output.
// MX-FP4 copies of an output (include/avl_hip.h, "MX bundle"): planes [rows][C/2] bytes + E8M0 scales [C/256][rows][8]
struct MxOut {
    char* q[2];          // FP4 plane of the hi part / of the lo part (NULL = not written)
    char* s[2];
    long long srows;
    int ldq;
};

template <typename T, bool SPLIT = false>
__global__ void __launch_bounds__(kThreads) k_dwconv(const T* __restrict__ in, const float* __restrict__ w,
                                                    const float* __restrict__ bias, T* __restrict__ out, const T* __restrict__ zero,
                                                    DwGeom g, const T* __restrict__ in_lo = nullptr, T* __restrict__ out_lo = nullptr,
                                                    MxOut mx = MxOut()) {
    const int cgrp = blockIdx.x % g.cgroups, ub = blockIdx.x / g.cgroups;
    const int chunk = threadIdx.x % g.nchunk, cl = threadIdx.x / g.nchunk;
    if (cl >= g.cl) return;
    const int c8 = cgrp * g.nchunk + chunk;
    float wt[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const float4 w0 = *reinterpret_cast<const float4*>(w + t * g.C + c8 * 8);
        const float4 w1 = *reinterpret_cast<const float4*>(w + t * g.C + c8 * 8 + 4);
        wt[t][0] = w0.x; wt[t][1] = w0.y; wt[t][2] = w0.z; wt[t][3] = w0.w;
        wt[t][4] = w1.x; wt[t][5] = w1.y; wt[t][6] = w1.z; wt[t][7] = w1.w;
    }
    float bs[8];
    {
        const float4 b0 = *reinterpret_cast<const float4*>(bias + c8 * 8);
        const float4 b1 = *reinterpret_cast<const float4*>(bias + c8 * 8 + 4);
        bs[0] = b0.x; bs[1] = b0.y; bs[2] = b0.z; bs[3] = b0.w; bs[4] = b1.x; bs[5] = b1.y; bs[6] = b1.z; bs[7] = b1.w;
    }
    constexpr int NR = (sizeof(T) == 2 && !SPLIT) ? 1 : 2;        // 16-byte registers per tap: f32 = 8 floats; split = hi chunk, lo chunk
    // 16-bit types: taps are paired (0,1)(2,3)(4,5)(6,7)(8,-) and each channel's two taps go through one
    // v_dot2c_f32_{bf16,f16} (exact products, fp32 accumulate) -- 1 vector op per MAC instead of convert + FMA.
    // The folded weights are rounded to the activation type for it, like every GEMM weight of the network.
    uint32_t wp[5][8];
    if constexpr (sizeof(T) == 2 && !SPLIT) {
#pragma unroll
        for (int pr = 0; pr < 5; ++pr)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const T lo = (T)wt[2 * pr][i], hi = (T)(pr < 4 ? wt[2 * pr + 1][i] : 0.f);
                wp[pr][i] = (uint32_t)__builtin_bit_cast(unsigned short, lo) | ((uint32_t)__builtin_bit_cast(unsigned short, hi) << 16);
            }
    }
    const int u_end = min(g.units, (ub + 1) * g.upb);
    for (int u = ub * g.upb; u < u_end; ++u) {
        int b = u;
        const int bx = b % g.nbx; b /= g.nbx;
        const int by = b % g.nby; b /= g.nby;
        const int rx = b % g.dil, ry = b / g.dil;
        const int gy0 = by * g.band_h, gx0 = bx * g.band_w;
        const int bw = min(g.band_w, g.gw - gx0), bh = min(g.band_h, g.gh - gy0);
        // A lane walks DOWN one grid column with a rolling window: the three taps of the next input row are fetched
        // two outputs ahead of the one being computed, so an output costs 3 loads instead of 9 (the kernel is bound by
        // L1/L2 request bandwidth, not by VALU or HBM).  A tap outside the image reads a zero page (pointer select).
        for (int col = cl; col < bw; col += g.cl) {
            const int ox = rx + (gx0 + col) * g.dil;
            if (ox >= g.OW) break;
            bool cok[3];
            long long coff[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int ix = ox - g.pad + k * g.dil;
                cok[k] = ix >= 0 && ix < g.W;
                coff[k] = (long long)ix * g.in_ld + c8 * 8;
            }
            uint4 raw[9][NR], nxt[3][NR], nx2[3][NR];
            auto load_row = [&](int iy, uint4 (*dst)[NR]) {
                const bool rok = iy >= 0 && iy < g.H;
                const T* rowp = in + (long long)iy * g.W * g.in_ld;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const uint4* src = reinterpret_cast<const uint4*>((rok && cok[k]) ? rowp + coff[k] : zero);
                    dst[k][0] = src[0];
                    if constexpr (SPLIT) {
                        const uint4* srl = reinterpret_cast<const uint4*>((rok && cok[k]) ? in_lo + (long long)iy * g.W * g.in_ld + coff[k] : zero);
                        dst[k][1] = srl[0];
                    } else if constexpr (NR == 2) dst[k][1] = src[1];
                }
            };
            const int oy0 = ry + gy0 * g.dil;
            load_row(oy0 - g.pad, &raw[3]);
            load_row(oy0 - g.pad + g.dil, &raw[6]);
            load_row(oy0 - g.pad + 2 * g.dil, nxt);
            if (bh > 1) load_row(oy0 - g.pad + 3 * g.dil, nx2);
            for (int r = 0; r < bh; ++r) {
                const int oy = oy0 + r * g.dil;
                if (oy >= g.OH) break;
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int q = 0; q < NR; ++q) {
                        raw[k][q] = raw[3 + k][q];
                        raw[3 + k][q] = raw[6 + k][q];
                        raw[6 + k][q] = nxt[k][q];
                        nxt[k][q] = nx2[k][q];
                    }
                if (r + 2 < bh) load_row(oy - g.pad + 4 * g.dil, nx2);
            float acc[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = bs[i];
            if constexpr (SPLIT) {
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    float vh[8], vl[8];
                    Vec8<T>::load(reinterpret_cast<const T*>(&raw[t][0]), vh);
                    Vec8<T>::load(reinterpret_cast<const T*>(&raw[t][1]), vl);
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i] = fmaf(vh[i] + vl[i], wt[t][i], acc[i]);
                }
            } else if constexpr (sizeof(T) == 2) {
#pragma unroll
                for (int pr = 0; pr < 5; ++pr) {
                    const uint32_t* ra = reinterpret_cast<const uint32_t*>(&raw[2 * pr][0]);
                    const uint32_t* rb = reinterpret_cast<const uint32_t*>(&raw[pr < 4 ? 2 * pr + 1 : 8][0]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        // [tap a ch 2j | tap b ch 2j] and [tap a ch 2j+1 | tap b ch 2j+1]
                        const uint32_t lo = pr < 4 ? __builtin_amdgcn_perm(rb[j], ra[j], 0x05040100u) : (ra[j] & 0xffffu);
                        const uint32_t hi = pr < 4 ? __builtin_amdgcn_perm(rb[j], ra[j], 0x07060302u) : (ra[j] >> 16);
                        acc[2 * j] = Half16<T>::dot2(lo, wp[pr][2 * j], acc[2 * j]);
                        acc[2 * j + 1] = Half16<T>::dot2(hi, wp[pr][2 * j + 1], acc[2 * j + 1]);
                    }
                }
            } else {
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    float v[8];
                    Vec8<T>::load(reinterpret_cast<const T*>(&raw[t][0]), v);
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i] = fmaf(v[i], wt[t][i], acc[i]);
                }
            }
            if (g.relu) {
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = fmaxf(acc[i], 0.f);
            }
            Vec8<T>::store(out + ((long long)oy * g.OW + ox) * g.out_ld + c8 * 8, acc);
            if constexpr (SPLIT) {
                float hi8[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) { hi8[i] = to_f32(from_f32<T>(acc[i])); acc[i] -= hi8[i]; }
                if (out_lo) {
                    Vec8<T>::store(out_lo + ((long long)oy * g.OW + ox) * g.out_ld + c8 * 8, acc);
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i] = to_f32(from_f32<T>(acc[i]));        // the FP4 copy is taken from what the plane holds
                }
                // FP4 copies for the MX GEMM that follows: four consecutive lanes (channel chunks of one pixel) form a 32-block
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    if (mx.q[pl] == nullptr) continue;
                    const float* src = pl == 0 ? hi8 : acc;
                    float amax = 0.f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fabsf(src[i]));
                    amax = fmaxf(amax, __shfl_xor(amax, 1));
                    amax = fmaxf(amax, __shfl_xor(amax, 2));
                    const unsigned sbyte = mx_fp4_scale_byte(amax);
                    const float scale = __uint_as_float(sbyte << 23);
                    unsigned pk = 0u;
                    pk = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(pk, src[0], src[1], scale, 0);
                    pk = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(pk, src[2], src[3], scale, 1);
                    pk = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(pk, src[4], src[5], scale, 2);
                    pk = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(pk, src[6], src[7], scale, 3);
                    const long long pix = (long long)oy * g.OW + ox;
                    // the block's 16 bytes leave through its first lane as ONE store (4-byte stores from four lanes cost more)
                    const unsigned p1 = __shfl_down(pk, 1), p2 = __shfl_down(pk, 2), p3 = __shfl_down(pk, 3);
                    if ((c8 & 3) == 0) {
                        *reinterpret_cast<uint4*>(mx.q[pl] + pix * mx.ldq + c8 * 4) = make_uint4(pk, p1, p2, p3);
                        mx.s[pl][((long long)(c8 >> 5) * mx.srows + pix) * 8 + ((c8 >> 2) & 7)] = (char)sbyte;
                    }
                }
            }
            }
        }
    }
}

// ------------------------------------------------------------------------------------- bilinear
// F.interpolate(mode='bilinear', align_corners=True): src = dst * (in-1)/(out-1)
// in_lo / out_lo ("mixed" precision, f16): optional low planes, value = hi + lo (exact in fp32).
template <typename T>
__global__ void __launch_bounds__(kThreads) k_bilinear(const T* __restrict__ in, const T* __restrict__ in_lo, int H, int W, int C, int in_ld,
                                                      T* __restrict__ out, T* __restrict__ out_lo, int OH, int OW, int out_ld) {
    const int c8n = C / 8;
    const long long idx = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (idx >= (long long)OH * OW * c8n) return;
    const int c8 = (int)(idx % c8n);
    const int pix = (int)(idx / c8n), oy = pix / OW, ox = pix % OW;
    const float sy = OH > 1 ? (float)(H - 1) / (float)(OH - 1) : 0.f;
    const float sx = OW > 1 ? (float)(W - 1) / (float)(OW - 1) : 0.f;
    const float fy = sy * oy, fx = sx * ox;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float ly = fy - y0, lx = fx - x0, hy = 1.f - ly, hx = 1.f - lx;
    float a[8], b[8], c[8], d[8], r[8];
    const long long o00 = ((long long)y0 * W + x0) * in_ld + c8 * 8, o01 = ((long long)y0 * W + x1) * in_ld + c8 * 8;
    const long long o10 = ((long long)y1 * W + x0) * in_ld + c8 * 8, o11 = ((long long)y1 * W + x1) * in_ld + c8 * 8;
    Vec8<T>::load(in + o00, a);
    Vec8<T>::load(in + o01, b);
    Vec8<T>::load(in + o10, c);
    Vec8<T>::load(in + o11, d);
    if (in_lo) {
        float al[8], bl[8], cl[8], dl[8];
        Vec8<T>::load(in_lo + o00, al);
        Vec8<T>::load(in_lo + o01, bl);
        Vec8<T>::load(in_lo + o10, cl);
        Vec8<T>::load(in_lo + o11, dl);
#pragma unroll
        for (int i = 0; i < 8; ++i) { a[i] += al[i]; b[i] += bl[i]; c[i] += cl[i]; d[i] += dl[i]; }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = hy * (hx * a[i] + lx * b[i]) + ly * (hx * c[i] + lx * d[i]);
    Vec8<T>::store(out + (long long)pix * out_ld + c8 * 8, r);
    if (out_lo) {
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i] -= to_f32(from_f32<T>(r[i]));
        Vec8<T>::store(out_lo + (long long)pix * out_ld + c8 * 8, r);
    }
}

// Depthwise 3x3 on split (hi + lo) f16 planes, fp32 weights, fp32 FMA chain in tap order, split result: the "mixed"
// precision decoder (its activations keep ~22 significant bits).  One lane = one output pixel x 8 channels; consecutive
// lanes = consecutive channel chunks of a pixel.  A tap outside the image contributes zero.
template <typename T>
__global__ void __launch_bounds__(kThreads) k_dwconv_split(const T* __restrict__ in, const T* __restrict__ in_lo, const float* __restrict__ w,
                                                          const float* __restrict__ bias, T* __restrict__ out, T* __restrict__ out_lo,
                                                          int H, int W, int C, int in_ld, int OH, int OW, int out_ld, int pad, int dil, int relu) {
    const int c8n = C / 8;
    const long long idx = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (idx >= (long long)OH * OW * c8n) return;
    const int c8 = (int)(idx % c8n);
    const int pix = (int)(idx / c8n), oy = pix / OW, ox = pix % OW;
    float acc[8];
    {
        const float4 b0 = *reinterpret_cast<const float4*>(bias + c8 * 8), b1 = *reinterpret_cast<const float4*>(bias + c8 * 8 + 4);
        acc[0] = b0.x; acc[1] = b0.y; acc[2] = b0.z; acc[3] = b0.w; acc[4] = b1.x; acc[5] = b1.y; acc[6] = b1.z; acc[7] = b1.w;
    }
    float xv[9][8];
    bool ok[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int iy = oy - pad + (t / 3) * dil, ix = ox - pad + (t % 3) * dil;
        ok[t] = iy >= 0 && iy < H && ix >= 0 && ix < W;
        const long long o = ok[t] ? ((long long)iy * W + ix) * in_ld + c8 * 8 : (long long)c8 * 8;
        Vec8<T>::load(in + o, xv[t]);
        if (in_lo) {
            float l[8];
            Vec8<T>::load(in_lo + o, l);
#pragma unroll
            for (int i = 0; i < 8; ++i) xv[t][i] += l[i];
        }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const float4 w0 = *reinterpret_cast<const float4*>(w + t * C + c8 * 8), w1 = *reinterpret_cast<const float4*>(w + t * C + c8 * 8 + 4);
        const float wt[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = fmaf(ok[t] ? xv[t][i] : 0.f, wt[i], acc[i]);
    }
    if (relu) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = fmaxf(acc[i], 0.f);
    }
    Vec8<T>::store(out + (long long)pix * out_ld + c8 * 8, acc);
    if (out_lo) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] -= to_f32(from_f32<T>(acc[i]));
        Vec8<T>::store(out_lo + (long long)pix * out_ld + c8 * 8, acc);
    }
}

// ---------------------------------------------------------------------------- global average pool
// stage 1: block b sums rows b, b+G, b+2G, ... for its 8-channel chunks -> partial[b][C] (fp32)
template <typename T>
__global__ void __launch_bounds__(kThreads) k_gap_partial(const T* __restrict__ in, int M, int C, int in_ld,
                                                         float* __restrict__ partial) {
    const int c8n = C / 8;
    const int G = gridDim.x;
    for (int c8 = threadIdx.x; c8 < c8n; c8 += kThreads) {
        float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int r = blockIdx.x;
        // four rows in flight per lane (a dependent one-load-at-a-time loop left the kernel latency-bound)
        for (; r + 3 * G < M; r += 4 * G) {
            float v0[8], v1[8], v2[8], v3[8];
            Vec8<T>::load(in + (long long)r * in_ld + c8 * 8, v0);
            Vec8<T>::load(in + (long long)(r + G) * in_ld + c8 * 8, v1);
            Vec8<T>::load(in + (long long)(r + 2 * G) * in_ld + c8 * 8, v2);
            Vec8<T>::load(in + (long long)(r + 3 * G) * in_ld + c8 * 8, v3);
#pragma unroll
            for (int i = 0; i < 8; ++i) s[i] += (v0[i] + v1[i]) + (v2[i] + v3[i]);
        }
        for (; r < M; r += G) {
            float v[8];
            Vec8<T>::load(in + (long long)r * in_ld + c8 * 8, v);
#pragma unroll
            for (int i = 0; i < 8; ++i) s[i] += v[i];
        }
        Vec8<float>::store(partial + (long long)blockIdx.x * C + c8 * 8, s);
    }
}
// stage 2: fixed-order sum of the partials, divide by M.  One workgroup = 32 channels x 8 slices of G.
__global__ void __launch_bounds__(kThreads) k_gap_final(const float* __restrict__ partial, int G, int C, int M,
                                                       float* __restrict__ out) {
    __shared__ float red[8][32];
    const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    float s = 0.f;
    if (c < C)
        for (int g = sl; g < G; g += 8) s += partial[(long long)g * C + c];
    red[sl][cl] = s;
    __syncthreads();
    if (sl == 0 && c < C) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k][cl];
        out[c] = t / (float)M;
    }
}

// ------------------------------------------------------------------------------------------ gemv
// out[n] = act(sum_k w[n][k] * in[k] + bias[n]); fp32; one wave per output.  A lane takes 16-byte pieces of the row (K % 4 == 0:
// four independent partial sums, all of a row's loads in flight at once -- with one float per lane and iteration the K = 2048
// product of the ASPP image-pooling branch was a chain of 32 dependent load + FMA steps, 18 us for half a MFLOP).
__global__ void __launch_bounds__(kThreads) k_gemv(const float* __restrict__ in, const float* __restrict__ w,
                                                  const float* __restrict__ bias, float* __restrict__ out, int N, int K, int relu) {
    const int n = blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (n >= N) return;
    const float* wr = w + (long long)n * K;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if ((K & 3) == 0 && (reinterpret_cast<uintptr_t>(wr) & 15) == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0) {
#pragma unroll 8
        for (int k = lane * 4; k < K; k += 256) {
            const float4 a = *reinterpret_cast<const float4*>(wr + k), b = *reinterpret_cast<const float4*>(in + k);
            s0 = fmaf(a.x, b.x, s0); s1 = fmaf(a.y, b.y, s1); s2 = fmaf(a.z, b.z, s2); s3 = fmaf(a.w, b.w, s3);
        }
    } else {
        for (int k = lane; k < K; k += 64) s0 = fmaf(wr[k], in[k], s0);
    }
    float s = (s0 + s1) + (s2 + s3);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if (lane == 0) {
        s += bias ? bias[n] : 0.f;
        out[n] = relu ? fmaxf(s, 0.f) : s;
    }
}

// ---------------------------------------------------------------------------------------- argmax
// torch.argmax(dim=1): first maximal index wins on ties.
__global__ void __launch_bounds__(kThreads) k_argmax(const float* __restrict__ logits, int M, int C, int ld,
                                                    unsigned char* __restrict__ out) {
    const int m = blockIdx.x * kThreads + threadIdx.x;
    if (m >= M) return;
    const float* p = logits + (long long)m * ld;
    float best = p[0];
    int bi = 0;
    for (int c = 1; c < C; ++c) {
        const float v = p[c];
        if (v > best || (v != v && best == best)) { best = v; bi = c; }   // NaN counts as maximal, like torch
    }
    out[m] = (unsigned char)bi;
}

// ------------------------------------------------------------------------------------- subsample
template <typename T>
__global__ void __launch_bounds__(kThreads) k_subsample(const T* __restrict__ in, int W, int C, int in_ld,
                                                       T* __restrict__ out, int OH, int OW, int out_ld, int stride) {
    const int c8n = C / 8;
    const long long idx = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (idx >= (long long)OH * OW * c8n) return;
    const int c8 = (int)(idx % c8n);
    const int pix = (int)(idx / c8n), oy = pix / OW, ox = pix % OW;
    float v[8];
    Vec8<T>::load(in + ((long long)(oy * stride) * W + ox * stride) * in_ld + c8 * 8, v);
    Vec8<T>::store(out + (long long)pix * out_ld + c8 * 8, v);
}

// ---------------------------------------------------------------------------------- pre-processing
// vision_semantic_segmentation_node.py:83-98 as a stand-alone kernel (seg_preprocess.h); one lane = one output pixel.
// The network path does not use it: the stem's loader applies the same function while it fills its LDS tile.
__global__ void __launch_bounds__(kThreads) k_preprocess(const unsigned char* __restrict__ bgr, int H, int W, PreParams q,
                                                        unsigned char* __restrict__ out, int OH, int OW) {
    const int idx = blockIdx.x * kThreads + threadIdx.x;
    if (idx >= OH * OW) return;
    int v[3];
    preprocessed_rgb(bgr, H, W, q.cam, q.factor, idx % OW, idx / OW, v);
#pragma unroll
    for (int c = 0; c < 3; ++c) out[3ll * idx + c] = (unsigned char)v[c];
}

// cv2.resize(INTER_AREA) to ANY smaller size (vision_semantic_segmentation_node.py:92-98 accepts every IMAGE_SCALE in (0, 1)): OpenCV's
// ResizeArea for a non-integer ratio -- per axis the destination cell [d s, (d + 1) s) covers a partial first source pixel, whole ones and a
// partial last one (computeResizeAreaTab: weights below 1e-3 dropped, cell width min(s, size - d s)); the rows are reduced in float, row by
// row, then across rows, and rounded half to even.  One lane = one output pixel; the undistorted source pixels are computed on the fly.
// OpenCV is absent: parity unpinned (oracle/preprocess_oracle.py restates the same steps in NumPy float32, mul then add, no FMA).
struct AreaAxis { int s1, s2; float w_first, w_mid, w_last; };       // source span [s1 - (w_first > 0), s2 + (w_last > 0))
__device__ __forceinline__ AreaAxis area_axis(int d, double scale, int ssize) {
    const double f1 = d * scale, f2 = f1 + scale;
    const double cell = fmin(scale, (double)ssize - f1);
    int s1 = (int)ceil(f1), s2 = (int)floor(f2);
    s2 = min(s2, ssize - 1);
    s1 = min(s1, s2);
    AreaAxis a;
    a.s1 = s1; a.s2 = s2;
    a.w_first = (s1 - f1 > 1e-3) ? (float)((s1 - f1) / cell) : 0.f;
    a.w_mid = (float)(1.0 / cell);
    a.w_last = (f2 - s2 > 1e-3) ? (float)(fmin(fmin(f2 - s2, 1.0), cell) / cell) : 0.f;
    return a;
}
__global__ void __launch_bounds__(kThreads) k_preprocess_area(const unsigned char* __restrict__ bgr, int H, int W, PreCamera cam,
                                                             unsigned char* __restrict__ out, int OH, int OW) {
    const int idx = blockIdx.x * kThreads + threadIdx.x;
    if (idx >= OH * OW) return;
    const int ox = idx % OW, oy = idx / OW;
    const AreaAxis ax = area_axis(ox, (double)W / OW, W), ay = area_axis(oy, (double)H / OH, H);
    float sum[3] = {0.f, 0.f, 0.f};
    for (int sy = ay.s1 - (ay.w_first > 0.f ? 1 : 0); sy < ay.s2 + (ay.w_last > 0.f ? 1 : 0); ++sy) {
        const float beta = sy < ay.s1 ? ay.w_first : (sy < ay.s2 ? ay.w_mid : ay.w_last);
        float buf[3] = {0.f, 0.f, 0.f};
        for (int sx = ax.s1 - (ax.w_first > 0.f ? 1 : 0); sx < ax.s2 + (ax.w_last > 0.f ? 1 : 0); ++sx) {
            const float alpha = sx < ax.s1 ? ax.w_first : (sx < ax.s2 ? ax.w_mid : ax.w_last);
            int rgb[3];
            undistorted_rgb(bgr, H, W, cam, sx, sy, rgb);
#pragma unroll
            for (int c = 0; c < 3; ++c) buf[c] = __fadd_rn(buf[c], __fmul_rn((float)rgb[c], alpha));
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) sum[c] = __fadd_rn(sum[c], __fmul_rn(buf[c], beta));
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) out[3ll * idx + c] = (unsigned char)min(max(__float2int_rn(sum[c]), 0), 255);
}

inline unsigned blocks_for(long long n) { return (unsigned)((n + kThreads - 1) / kThreads); }

template <typename T>
int launch_typed(const avl_seg_op& op, hipStream_t s) {
    const T* in = static_cast<const T*>(op.in);
    T* out = static_cast<T*>(op.out);
    const float* w = static_cast<const float*>(op.weight);
    switch (op.kind) {
        case AVL_OP_STEM:
            hipLaunchKernelGGL(k_stem<T>, dim3(blocks_for((long long)op.out_h * op.out_w)), dim3(kThreads), 0, s,
                               static_cast<const unsigned char*>(op.in), op.in_h, op.in_w, w, op.bias, out, op.out_h, op.out_w,
                               op.out_ld);
            break;
        case AVL_OP_MAXPOOL:
            hipLaunchKernelGGL(k_maxpool<T>, dim3(blocks_for((long long)op.out_h * op.out_w * (op.in_c / 8))), dim3(kThreads), 0, s,
                               in, op.in_h, op.in_w, op.in_c, op.in_ld, out, op.out_h, op.out_w, op.out_ld, static_cast<const T*>(op.in_lo),
                               static_cast<T*>(op.out_lo));
            break;
        case AVL_OP_GCONV: {
            const int cg = op.in_c / op.groups;
            const dim3 grid(blocks_for((long long)op.out_h * op.out_w), op.groups);
#define AVL_GCONV(CG)                                                                                                   \
    hipLaunchKernelGGL((k_gconv<T, CG>), grid, dim3(kThreads), 0, s, in, op.in_h, op.in_w, op.in_ld, w, op.bias, out, \
                       op.out_h, op.out_w, op.out_ld, op.stride, op.dil)
            if (cg == 4) AVL_GCONV(4);
            else if (cg == 8) AVL_GCONV(8);
            else if (cg == 16) AVL_GCONV(16);
            else if (cg == 32) AVL_GCONV(32);
            else if (cg == 2) AVL_GCONV(2);
            else return set_error(AVL_E_UNSUPPORTED, "grouped conv with %d channels per group", cg);
#undef AVL_GCONV
            break;
        }
        case AVL_OP_DWCONV: {
            if ((op.in_lo == nullptr) != (op.out_lo == nullptr) && !op.out_mx) {      // one side split only: the simple kernel
                hipLaunchKernelGGL(k_dwconv_split<T>, dim3(blocks_for((long long)op.out_h * op.out_w * (op.in_c / 8))), dim3(kThreads), 0, s,
                                   in, static_cast<const T*>(op.in_lo), w, op.bias, out, static_cast<T*>(op.out_lo), op.in_h, op.in_w,
                                   op.in_c, op.in_ld, op.out_h, op.out_w, op.out_ld, op.pad, op.dil, op.relu);
                break;
            }
            DwGeom g;
            g.H = op.in_h; g.W = op.in_w; g.C = op.in_c; g.in_ld = op.in_ld; g.OH = op.out_h; g.OW = op.out_w; g.out_ld = op.out_ld;
            g.pad = op.pad; g.dil = op.dil; g.relu = op.relu;
            g.gh = (op.out_h + op.dil - 1) / op.dil;
            g.gw = (op.out_w + op.dil - 1) / op.dil;
            const int chunks = op.in_c / 8;
            // channel lanes: at most 64 chunks (1 KB of a pixel) per workgroup, so that the rows a workgroup re-reads
            // two grid rows later (2 x gw x 512 B) are still in its XCD's L2 with ~1000 workgroups in flight
            const int max_chunk = AVL_EXP_INT("AVL_DW_NCHUNK", 64);
            g.nchunk = chunks < max_chunk ? chunks : max_chunk;
            while (chunks % g.nchunk) --g.nchunk;      // lanes beyond nchunk x cl idle (C = 304: 19 x 13 of 256)
            g.cl = kThreads / g.nchunk;
            g.cgroups = chunks / g.nchunk;
            const long long combs = (long long)op.dil * op.dil;
            g.band_w = g.gw; g.band_h = g.gh;
            if (op.dil == 1) {
                // a single comb: bands of 2 columns per lane x 16 rows (2 halo rows re-read per band)
                g.band_w = 2 * g.cl < g.gw ? 2 * g.cl : g.gw;
                g.band_h = g.gh < 16 ? g.gh : 16;
                const int nbx = (g.gw + g.band_w - 1) / g.band_w;
                while (g.band_h > 4 && (long long)g.cgroups * nbx * ((g.gh + g.band_h - 1) / g.band_h) < 512) g.band_h = (g.band_h + 1) / 2;
            }
            g.nbx = (g.gw + g.band_w - 1) / g.band_w;
            g.nby = (g.gh + g.band_h - 1) / g.band_h;
            g.units = (int)(combs * g.nby * g.nbx);
            // ~32 outputs per lane amortise the 18 weight loads: small combs are chained in one workgroup
            const int target = 32 * g.cl, unit_px = g.band_h * g.band_w;
            g.upb = (op.dil == 1 || unit_px >= target) ? 1 : (target + unit_px - 1) / unit_px;
            const unsigned nblk = (unsigned)((g.units + g.upb - 1) / g.upb) * g.cgroups;
            if constexpr (sizeof(T) == 2) {
                if (op.in_lo && (op.out_lo || op.out_mx)) {
                    MxOut mx;
                    memset(&mx, 0, sizeof(mx));
                    if (op.out_mx) {
                        char* b = static_cast<char*>(op.out_mx);
                        const long long rows = op.out_rows, P = rows * (op.out_c / 2), S = (long long)(op.out_c / 256) * rows * 8;
                        mx.q[0] = b; mx.s[0] = b + P;
                        if (op.out_lo || (op.mx_flags & AVL_MX_OUT_LO)) { mx.q[1] = b + P + S; mx.s[1] = b + 2 * P + S; }
                        mx.srows = rows; mx.ldq = op.out_c / 2;
                    }
                    hipLaunchKernelGGL((k_dwconv<T, true>), dim3(nblk), dim3(kThreads), 0, s, in, w, op.bias, out,
                                       static_cast<const T*>(op.in2), g, static_cast<const T*>(op.in_lo), static_cast<T*>(op.out_lo), mx);
                    break;
                }
            }
            hipLaunchKernelGGL((k_dwconv<T, false>), dim3(nblk), dim3(kThreads), 0, s, in, w, op.bias, out,
                               static_cast<const T*>(op.in2), g, static_cast<const T*>(nullptr), static_cast<T*>(nullptr), MxOut());
            break;
        }
        case AVL_OP_BILINEAR:
            hipLaunchKernelGGL(k_bilinear<T>, dim3(blocks_for((long long)op.out_h * op.out_w * (op.in_c / 8))), dim3(kThreads), 0, s,
                               in, static_cast<const T*>(op.in_lo), op.in_h, op.in_w, op.in_c, op.in_ld, out, static_cast<T*>(op.out_lo),
                               op.out_h, op.out_w, op.out_ld);
            break;
        case AVL_OP_GAP: {
            const int G = 256;
            float* partial = static_cast<float*>(const_cast<void*>(op.in2));
            hipLaunchKernelGGL(k_gap_partial<T>, dim3(G), dim3(kThreads), 0, s, in, op.in_h * op.in_w, op.in_c, op.in_ld, partial);
            AVL_LAUNCH_CHECK();
            hipLaunchKernelGGL(k_gap_final, dim3((op.in_c + 31) / 32), dim3(kThreads), 0, s, partial, G, op.in_c,
                               op.in_h * op.in_w, static_cast<float*>(op.out));
            break;
        }
        case AVL_OP_SUBSAMPLE:
            hipLaunchKernelGGL(k_subsample<T>, dim3(blocks_for((long long)op.out_h * op.out_w * (op.in_c / 8))), dim3(kThreads), 0, s,
                               in, op.in_w, op.in_c, op.in_ld, out, op.out_h, op.out_w, op.out_ld, op.stride);
            break;
        default:
            return set_error(AVL_E_ARG, "op kind %d is not a typed conv op", op.kind);
    }
    AVL_LAUNCH_CHECK();
    return AVL_OK;
}

}  // namespace

int validate_conv_op(const avl_seg_op& op) {
    const int es = elem_size(op.dtype);
    AVL_REQUIRE(op.in && op.out, "op %d has NULL in/out", op.kind);
    AVL_REQUIRE(op.in_h > 0 && op.in_w > 0 && op.out_h > 0 && op.out_w > 0, "op %d has empty spatial dims", op.kind);
    const long long in_pix = (long long)op.in_h * op.in_w, out_pix = (long long)op.out_h * op.out_w;
    switch (op.kind) {
        case AVL_OP_STEM:
            AVL_REQUIRE(op.weight && op.bias && op.out_c == 64 && op.in_c == 3, "stem expects 3 -> 64 channels");
            AVL_REQUIRE(op.out_h == (op.in_h + 6 - 7) / 2 + 1 && op.out_w == (op.in_w + 6 - 7) / 2 + 1, "stem output size");
            AVL_REQUIRE(op.out_rows >= out_pix && op.out_ld >= 64 && (op.out_ld * es) % 16 == 0, "stem output buffer");
            AVL_REQUIRE(op.w_layout == 0 || (op.w_layout == 1 && is_half(op.dtype)), "stem weight layout %d", op.w_layout);
            if (op.in2) {      // pre-processing in the loader: `in` is the raw BGR frame [in_rows / in2_ld][in2_ld][3], in2 the camera block
                AVL_REQUIRE(op.w_layout == 1, "the pre-processing stem is the MFMA kernel (w_layout 1)");
                AVL_REQUIRE(reinterpret_cast<uintptr_t>(op.in2) % 4 == 0, "stem camera block alignment");
                AVL_REQUIRE(op.in2_ld > 0 && op.in_rows > 0 && op.in_rows % op.in2_ld == 0, "stem raw frame: in_rows = src_h * src_w, in2_ld = src_w");
                const int src_w = op.in2_ld, src_h = op.in_rows / op.in2_ld, f = src_w / op.in_w;
                AVL_REQUIRE(f >= 1 && op.in_w == src_w / f && op.in_h == src_h / f, "stem raw frame %dx%d does not scale to %dx%d by an integer factor",
                            src_h, src_w, op.in_h, op.in_w);
            } else {
                AVL_REQUIRE(op.in_rows >= in_pix, "stem input buffer");
            }
            return AVL_OK;
        case AVL_OP_GEMV:
            AVL_REQUIRE(op.weight && op.in_c > 0 && op.out_c > 0, "gemv shapes");
            return AVL_OK;
        case AVL_OP_ARGMAX:
            AVL_REQUIRE(op.in_c > 0 && op.in_c <= 256 && op.in_ld >= op.in_c && op.in_rows >= in_pix && op.out_rows >= in_pix, "argmax shapes");
            return AVL_OK;
        case AVL_OP_GAP:
            AVL_REQUIRE(op.in2 && op.in_c % 8 == 0 && op.in_rows >= in_pix && op.in_ld >= op.in_c, "gap shapes / scratch");
            AVL_REQUIRE((op.in_ld * es) % 16 == 0, "gap in_ld");
            return AVL_OK;
        default:
            break;
    }
    AVL_REQUIRE(is_half(op.dtype) || op.dtype == AVL_F32, "op %d dtype %d", op.kind, op.dtype);
    AVL_REQUIRE(op.in_c % 8 == 0 || (op.kind == AVL_OP_GCONV), "op %d: channels %d not a multiple of 8", op.kind, op.in_c);
    AVL_REQUIRE(op.in_rows >= in_pix && op.out_rows >= out_pix, "op %d: allocated rows too small", op.kind);
    AVL_REQUIRE(op.in_ld >= op.in_c && op.out_ld >= op.out_c, "op %d: leading dims", op.kind);
    AVL_REQUIRE((op.in_ld * es) % 16 == 0 && (op.out_ld * es) % 16 == 0, "op %d: row strides must be 16-byte multiples", op.kind);
    AVL_REQUIRE((reinterpret_cast<uintptr_t>(op.in) | reinterpret_cast<uintptr_t>(op.out)) % 16 == 0, "op %d: unaligned buffers", op.kind);
    if (op.out_mx && op.kind == AVL_OP_DWCONV) {
        AVL_REQUIRE(op.dtype == AVL_F16 && op.in_lo && op.out_c % 256 == 0 && op.out_ld == op.out_c && (op.in_c / 8) % 4 == 0,
                    "dwconv: out_mx needs the split f16 form, channels %% 256 == 0 and a dense output");
        AVL_REQUIRE(!(op.mx_flags & AVL_MX_OUT_LO) || !op.out_lo, "dwconv: AVL_MX_OUT_LO together with out_lo");
    }
    if (op.in_lo || op.out_lo || op.in2_lo) {
        // split (hi + lo) planes: same shape and stride as the high plane; f16 only
        AVL_REQUIRE(op.dtype == AVL_F16 && !op.in2_lo, "op %d: split planes need AVL_F16 (and no in2_lo)", op.kind);
        AVL_REQUIRE(op.kind == AVL_OP_BILINEAR || op.kind == AVL_OP_DWCONV || (op.kind == AVL_OP_GCONV && op.w_layout == 1) || op.kind == AVL_OP_MAXPOOL ||
                        (op.kind == AVL_OP_STEM && op.w_layout == 1 && op.w_split == 1 && !op.in_lo),
                    "op %d does not take split planes", op.kind);
        AVL_REQUIRE(op.kind != AVL_OP_MAXPOOL || (op.in_lo != nullptr) == (op.out_lo != nullptr), "maxpool: both sides split or none");
        AVL_REQUIRE((reinterpret_cast<uintptr_t>(op.in_lo) | reinterpret_cast<uintptr_t>(op.out_lo)) % 16 == 0, "op %d: unaligned low planes", op.kind);
    }
    switch (op.kind) {
        case AVL_OP_MAXPOOL:
            AVL_REQUIRE(op.out_c == op.in_c && op.out_h == (op.in_h + 2 - 3) / 2 + 1 && op.out_w == (op.in_w + 2 - 3) / 2 + 1, "maxpool shapes");
            break;
        case AVL_OP_GCONV: {
            AVL_REQUIRE(op.weight && op.bias && op.groups > 0 && op.in_c % op.groups == 0 && op.out_c == op.in_c, "gconv channels");
            AVL_REQUIRE(op.ksize == 3 && (op.stride == 1 || op.stride == 2) && op.dil >= 1 && op.pad == op.dil, "gconv geometry");
            AVL_REQUIRE(op.out_h == (op.in_h + 2 * op.pad - 2 * op.dil - 1) / op.stride + 1 &&
                        op.out_w == (op.in_w + 2 * op.pad - 2 * op.dil - 1) / op.stride + 1, "gconv output size");
            if (op.w_layout == 1) return validate_gconv_mfma(op);
            AVL_REQUIRE(op.w_layout == 0, "gconv weight layout %d", op.w_layout);
            break;
        }
        case AVL_OP_DWCONV:
            AVL_REQUIRE(op.weight && op.bias && op.out_c == op.in_c && op.ksize == 3 && op.stride == 1 && op.dil >= 1 && op.pad >= 0, "dwconv geometry");
            AVL_REQUIRE(op.in2 && reinterpret_cast<uintptr_t>(op.in2) % 16 == 0, "dwconv needs in2 = a 32-byte zero page (taps outside the image read it)");
            AVL_REQUIRE(op.out_h == op.in_h + 2 * op.pad - 2 * op.dil && op.out_w == op.in_w + 2 * op.pad - 2 * op.dil, "dwconv output size");
            AVL_REQUIRE(op.in_c % (es == 2 ? 64 : 32) == 0, "dwconv channels %d not a multiple of one 128-byte line", op.in_c);
            break;
        case AVL_OP_BILINEAR:
            AVL_REQUIRE(op.out_c == op.in_c, "bilinear channels");
            break;
        case AVL_OP_SUBSAMPLE:
            AVL_REQUIRE(op.out_c == op.in_c && op.stride >= 1 && (op.out_h - 1) * op.stride < op.in_h && (op.out_w - 1) * op.stride < op.in_w, "subsample geometry");
            break;
        default:
            return set_error(AVL_E_ARG, "unknown op kind %d", op.kind);
    }
    return AVL_OK;
}

int launch_preprocess(const unsigned char* bgr, int H, int W, const double* K, const double* dist, int factor, unsigned char* out,
                      hipStream_t s) {
    PreParams q;
    q.cam = make_pre_camera(K, dist);
    q.factor = factor;
    const int OH = H / factor, OW = W / factor;
    hipLaunchKernelGGL(k_preprocess, dim3(blocks_for((long long)OH * OW)), dim3(kThreads), 0, s, bgr, H, W, q, out, OH, OW);
    AVL_LAUNCH_CHECK();
    return AVL_OK;
}

int launch_preprocess_area(const unsigned char* bgr, int H, int W, const double* K, const double* dist, int OH, int OW, unsigned char* out,
                           hipStream_t s) {
    hipLaunchKernelGGL(k_preprocess_area, dim3(blocks_for((long long)OH * OW)), dim3(kThreads), 0, s, bgr, H, W, make_pre_camera(K, dist), out, OH, OW);
    AVL_LAUNCH_CHECK();
    return AVL_OK;
}

__global__ void k_set_camera(PreCamera q, PreCamera* out) { *out = q; }

// writes the camera block a pre-processing stem reads (stream-ordered, so a captured plan can switch cameras between frames)
int launch_set_camera(void* cam_dev, const double* K, const double* dist, hipStream_t s) {
    static_assert(sizeof(PreCamera) <= 64, "AVL_STEM_CAMERA_BYTES");
    hipLaunchKernelGGL(k_set_camera, dim3(1), dim3(1), 0, s, make_pre_camera(K, dist), static_cast<PreCamera*>(cam_dev));
    AVL_LAUNCH_CHECK();
    return AVL_OK;
}

int launch_conv_op(const avl_seg_op& op, hipStream_t s) {
    if (op.kind == AVL_OP_GCONV && op.w_layout == 1) return launch_gconv_mfma(op, s);
    if (op.kind == AVL_OP_STEM && op.w_layout == 1) return launch_stem_mfma(op, s);
    if (op.kind == AVL_OP_GEMV) {
        hipLaunchKernelGGL(k_gemv, dim3((op.out_c + 3) / 4), dim3(kThreads), 0, s, static_cast<const float*>(op.in),
                           static_cast<const float*>(op.weight), op.bias, static_cast<float*>(op.out), op.out_c, op.in_c, op.relu);
        AVL_LAUNCH_CHECK();
        return AVL_OK;
    }
    if (op.kind == AVL_OP_ARGMAX) {
        const int M = op.in_h * op.in_w;
        hipLaunchKernelGGL(k_argmax, dim3(blocks_for(M)), dim3(kThreads), 0, s, static_cast<const float*>(op.in), M, op.in_c,
                           op.in_ld, static_cast<unsigned char*>(op.out));
        AVL_LAUNCH_CHECK();
        return AVL_OK;
    }
    if (op.dtype == AVL_BF16) return launch_typed<bf16>(op, s);
    if (op.dtype == AVL_F16) return launch_typed<f16>(op, s);
    return launch_typed<float>(op, s);
}

}  // namespace avl
