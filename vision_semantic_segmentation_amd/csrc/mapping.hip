// LiDAR -> image -> BEV grid kernels (gfx950).  Compiled with -ffp-contract=off: the float64
// arithmetic below has to round exactly where the reference's NumPy does, so every fused
// multiply-add is written out.
//
// Reference arithmetic (src/mapping.py):
//   :371  Xv  = T @ [x y z 1]^T            (np.matmul -> OpenBLAS dgemm, K = 4: p0*x0 then an FMA chain)
//   :375  IXY = ((P @ Xv)[0:2] / (P @ Xv)[2]).astype(int32)
//   :378  0 < Xv.x < RANGE_MAX             :381-383 0 <= IX < W, 0 <= IY < H
//   :405-411 cell = (((x,y) + offset) - (b00,b10)) / res -> astype(int32), on-grid test
//   :419-437 per class: R,G colour match, buffered += of CM[:, i] (once per cell), lane bonus
//
// Kernels are HBM/latency bound integer + f64 work: one point per lane, coalesced 16 B (f32 AoS)
// or 8 B (f64 SoA) loads, a label gather served by L2, and an idempotent atomicOr per point.
#include "avl_common.h"

#include <climits>
#include <cstdlib>

namespace {

constexpr int kBlock = 256;

struct ProjParams {
    double P[12];
    double T[16];
    int has_T;
    double range_max;
    int img_w, img_h;
};

struct PtsView {
    const char* base;
    int n;
    int dtype;
    long long point_stride, comp_stride;
    int aos_f32;  // 1: float4-per-point fast path (16-byte aligned, comp_stride 4, point_stride 16)
};

struct GridParams {
    double off_x, off_y, b00, b10, resolution;
    int Hm, Wm, C;
    unsigned bonus_classes;
    unsigned char colors[AVL_MAX_MAP_CLASSES * 3];
};

struct CmParams {
    double cm[AVL_MAX_MAP_CLASSES * AVL_MAX_MAP_CLASSES];  // row-major [C][C]
};

struct LutParams {
    unsigned lut[256];
};

__device__ __forceinline__ void load_point(const PtsView& v, int k, double& x, double& y, double& z, double& it) {
    if (v.aos_f32) {
        const float4 p = *reinterpret_cast<const float4*>(v.base + (long long)k * 16);
        x = (double)p.x; y = (double)p.y; z = (double)p.z; it = (double)p.w;
        return;
    }
    const char* p = v.base + (long long)k * v.point_stride;
    if (v.dtype == AVL_F64) {
        x = *reinterpret_cast<const double*>(p);
        y = *reinterpret_cast<const double*>(p + v.comp_stride);
        z = *reinterpret_cast<const double*>(p + 2 * v.comp_stride);
        it = *reinterpret_cast<const double*>(p + 3 * v.comp_stride);
    } else {
        x = (double)*reinterpret_cast<const float*>(p);
        y = (double)*reinterpret_cast<const float*>(p + v.comp_stride);
        z = (double)*reinterpret_cast<const float*>(p + 2 * v.comp_stride);
        it = (double)*reinterpret_cast<const float*>(p + 3 * v.comp_stride);
    }
}

// float64 -> int32 as NumPy on x86-64 does it (cvttsd2si): truncate toward zero; NaN, +-inf and
// anything that does not fit yield INT_MIN.  v_cvt_i32_f64 would saturate / give 0.
__device__ __forceinline__ int cvt_i32_numpy(double v) {
    if (!(v > -2147483649.0 && v < 2147483648.0)) return INT_MIN;
    return (int)v;
}

// int32(num / den) exactly as NumPy computes it -- the correctly rounded float64 quotient, truncated (cvt_i32_numpy) -- without
// the ~35-instruction IEEE division sequence in the common case (round 4; four divisions per point).  `y` approximates 1 / den to
// a few ulp (refine_rcp).  q = num * y corrected once by its own residual is within ~2 ulp of the true quotient; the truncation of
// that equals the truncation of the correctly rounded quotient unless an integer lies within those ulps -- so lanes whose q is
// within 2^-44 (relative) of an integer, or is not an ordinary number (NaN, infinity, zero, |q| >= 2^40: zero / non-finite
// denominators, overflow), take the exact division instead.  Both paths give the same int32 for every input; the exact one runs
// for ~1e-10 of ordinary points (tests/test_gpu_mapping.py::test_truncating_divisions_on_integer_boundaries forces it).
__device__ __forceinline__ double refine_rcp(double den) {
    double y = __builtin_amdgcn_rcp(den);                 // v_rcp_f64: ~26 bits
    double e = __builtin_fma(-den, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-den, y, 1.0);
    return __builtin_fma(y, e, y);
}
__device__ __forceinline__ int div_i32_numpy(double num, double den, double y) {
    double q = num * y;
    const double r = __builtin_fma(-q, den, num);
    q = __builtin_fma(r, y, q);
    const double aq = __builtin_fabs(q);
    const bool safe = __builtin_fabs(q - __builtin_rint(q)) > aq * 0x1p-44 && aq > 0x1p-500 && aq < 0x1p40;
    if (!safe) q = num / den;
    return cvt_i32_numpy(q);
}

// dot of a 4-vector row with (a,b,c,d) in OpenBLAS's order: r0*a, then fma chain.
__device__ __forceinline__ double dot4(const double* r, double a, double b, double c, double d) {
    double s = r[0] * a;
    s = __builtin_fma(r[1], b, s);
    s = __builtin_fma(r[2], c, s);
    s = __builtin_fma(r[3], d, s);
    return s;
}

__device__ __forceinline__ bool project(const ProjParams& pp, double x, double y, double z, int& ix, int& iy) {
    double v0 = x, v1 = y, v2 = z, v3 = 1.0;
    if (pp.has_T) {
        v0 = dot4(pp.T + 0, x, y, z, 1.0);
        v1 = dot4(pp.T + 4, x, y, z, 1.0);
        v2 = dot4(pp.T + 8, x, y, z, 1.0);
        v3 = dot4(pp.T + 12, x, y, z, 1.0);
    }
    const double p0 = dot4(pp.P + 0, v0, v1, v2, v3);
    const double p1 = dot4(pp.P + 4, v0, v1, v2, v3);
    const double p2 = dot4(pp.P + 8, v0, v1, v2, v3);
    const double rp2 = refine_rcp(p2);
    ix = div_i32_numpy(p0, p2, rp2);
    iy = div_i32_numpy(p1, p2, rp2);
    const bool positive = (0.0 < v0) && (v0 < pp.range_max);
    return positive && ix >= 0 && ix < pp.img_w && iy >= 0 && iy < pp.img_h;
}

// grid cell of a point in its ORIGINAL frame (:404-411); -1 if off-grid / non-finite
__device__ __forceinline__ int grid_cell(const GridParams& g, double x, double y, double z) {
    const double xl = x + g.off_x, yl = y + g.off_y, zl = z + 0.0;
    // :406 subtracts 0*z-like terms built from every coordinate: one non-finite coordinate makes both NaN
    if (!(__builtin_isfinite(xl) && __builtin_isfinite(yl) && __builtin_isfinite(zl))) return -1;
    const double rres = refine_rcp(g.resolution);
    const int cx = div_i32_numpy(xl - g.b00, g.resolution, rres);
    const int cy = div_i32_numpy(yl - g.b10, g.resolution, rres);
    if (cx < 0 || cx >= g.Hm || cy < 0 || cy >= g.Wm) return -1;
    return cx * g.Wm + cy;
}

__device__ __forceinline__ unsigned vote_from_rg(const GridParams& g, unsigned r, unsigned gch) {
    unsigned vote = 0;
    for (int i = 0; i < g.C; ++i)
        if (g.colors[3 * i] == r && g.colors[3 * i + 1] == gch) vote |= 1u << i;  // blue ignored (Q2)
    return vote;
}

__device__ __forceinline__ unsigned add_bonus(unsigned vote, unsigned bonus_classes, double intensity) {
    if ((vote & bonus_classes) && (intensity < 2.0 || intensity > 14.0)) vote |= (vote & bonus_classes) << 16;
    return vote;
}

// OR the vote into the cell; the lane that turns the mask non-zero records the cell.  The append
// counter is bumped once per wave (ballot + popcount) instead of once per lane.
__device__ __forceinline__ void cast_vote(unsigned* cell_mask, int* touched, int* counter, int cell, unsigned vote) {
    bool first = false;
    if (cell >= 0 && vote != 0) first = (atomicOr(&cell_mask[cell], vote) == 0u);
    const unsigned long long m = __ballot(first);
    if (m == 0) return;
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == __builtin_ctzll(m)) base = atomicAdd(counter, __builtin_popcountll(m));
    base = __shfl(base, __builtin_ctzll(m));
    if (first) touched[base + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = cell;
}

// Dense clouds: no touched list -- a fire-and-forget OR (no-return atomic, nothing depends on it); the apply
// pass then sweeps the whole mask instead of a list (k_grid_apply_scan).
__device__ __forceinline__ void cast_vote_nolist(unsigned* cell_mask, int cell, unsigned vote) {
    if (cell >= 0 && vote != 0) atomicOr(&cell_mask[cell], vote);
}

// Byte-wide vote mask (dense-cloud path, when C class bits + the bonus bits fit 8): bit i = class i seen, bit C + r = lane
// bonus of the r-th class of bonus_classes.  A quarter of the bytes of the 32-bit mask for the sweep to read; the OR goes
// to the containing dword (idempotent, so any interleaving of lanes gives the same mask).
__device__ __forceinline__ unsigned encode_vote_byte(unsigned vote, int C, unsigned bonus_classes) {
    unsigned enc = vote & ((1u << C) - 1u);
    unsigned rest = bonus_classes;
    int r = 0;
    while (rest) {
        const int j = __builtin_ctz(rest);
        rest &= rest - 1u;
        if ((vote >> (16 + j)) & 1u) enc |= 1u << (C + r);
        ++r;
    }
    return enc;
}
__device__ __forceinline__ void cast_vote_byte(unsigned* cell_mask, int cell, unsigned vote, int C, unsigned bonus_classes) {
    if (cell >= 0 && vote != 0) atomicOr(&cell_mask[cell >> 2], encode_vote_byte(vote, C, bonus_classes) << ((cell & 3) * 8));
}

// ---------------------------------------------------------------- projection only (:367-383)
__global__ void __launch_bounds__(kBlock) k_project_points(PtsView pv, ProjParams pp, int* __restrict__ out_ixy,
                                                           unsigned char* __restrict__ out_mask) {
    const int k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= pv.n) return;
    double x, y, z, it;
    load_point(pv, k, x, y, z, it);
    int ix, iy;
    const bool ok = project(pp, x, y, z, ix, iy);
    if (out_ixy) {
        out_ixy[k] = ix;
        out_ixy[pv.n + k] = iy;
    }
    if (out_mask) out_mask[k] = ok ? 1 : 0;
}

// ---------------------------------------------------------------- project_pcd with compaction
// pass 1: mask + pixel per point, survivors per block
__global__ void __launch_bounds__(kBlock) k_pcd_mask(PtsView pv, ProjParams pp, int* __restrict__ pix,
                                                     int* __restrict__ block_count) {
    __shared__ int wave_cnt[kBlock / 64];
    const int k = blockIdx.x * kBlock + threadIdx.x;
    bool ok = false;
    if (k < pv.n) {
        double x, y, z, it;
        load_point(pv, k, x, y, z, it);
        int ix, iy;
        ok = project(pp, x, y, z, ix, iy);
        pix[k] = ok ? iy * pp.img_w + ix : -1;
    }
    const unsigned long long m = __ballot(ok);
    if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = __builtin_popcountll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        int s = 0;
        for (int w = 0; w < kBlock / 64; ++w) s += wave_cnt[w];
        block_count[blockIdx.x] = s;
    }
}

// pass 2: exclusive scan of the block counts (one workgroup), total -> out_count
__global__ void __launch_bounds__(1024) k_scan_blocks(int* __restrict__ block_count, int nb, int* __restrict__ out_count) {
    __shared__ int part[1024];
    const int t = threadIdx.x;
    const int per = (nb + 1023) / 1024;
    const int lo = t * per, hi = min(lo + per, nb);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += block_count[i];
    part[t] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan
        int v = (t >= off) ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - s;  // exclusive prefix of this thread's chunk
    for (int i = lo; i < hi; ++i) {
        const int c = block_count[i];
        block_count[i] = run;
        run += c;
    }
    if (t == 1023) *out_count = part[1023];
}

// pass 3: order-preserving scatter of the survivors + RGB gather (:385-387)
__global__ void __launch_bounds__(kBlock) k_pcd_scatter(PtsView pv, const int* __restrict__ pix,
                                                        const int* __restrict__ block_off,
                                                        const unsigned char* __restrict__ image,
                                                        double* __restrict__ out_pcd, unsigned char* __restrict__ out_label,
                                                        long long out_ld) {
    __shared__ int wave_cnt[kBlock / 64];
    const int k = blockIdx.x * kBlock + threadIdx.x;
    const int p = (k < pv.n) ? pix[k] : -1;
    const bool ok = p >= 0;
    const unsigned long long m = __ballot(ok);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wave_cnt[wave] = __builtin_popcountll(m);
    __syncthreads();
    if (!ok) return;
    int dst = block_off[blockIdx.x] + __builtin_popcountll(m & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; ++w) dst += wave_cnt[w];
    double x, y, z, it;
    load_point(pv, k, x, y, z, it);
    out_pcd[dst] = x;
    out_pcd[out_ld + dst] = y;
    out_pcd[2 * out_ld + dst] = z;
    out_pcd[3 * out_ld + dst] = it;
    const unsigned char* px = image + 3ll * p;
    out_label[dst] = px[0];
    out_label[out_ld + dst] = px[1];
    out_label[2 * out_ld + dst] = px[2];
}

// ---------------------------------------------------------------- update_map front end (:403-437)
__global__ void __launch_bounds__(kBlock) k_vote_labelled(const double* __restrict__ pcd, const unsigned char* __restrict__ label,
                                                          long long ld, int m_host, const int* __restrict__ m_dev,
                                                          GridParams g, unsigned* __restrict__ cell_mask,
                                                          int* __restrict__ touched, int* __restrict__ counter) {
    const int m = m_dev ? min(*m_dev, m_host) : m_host;
    const int k = blockIdx.x * kBlock + threadIdx.x;
    int cell = -1;
    unsigned vote = 0;
    if (k < m) {
        cell = grid_cell(g, pcd[k], pcd[ld + k], pcd[2 * ld + k]);
        if (cell >= 0) {
            vote = vote_from_rg(g, label[k], label[ld + k]);
            vote = add_bonus(vote, g.bonus_classes, pcd[3 * ld + k]);
        }
    }
    cast_vote(cell_mask, touched, counter, cell, vote);
}

// ---------------------------------------------------------------- fused project + vote
// Partitioned touched lists (MODE 3).  The single-counter list of cast_vote serialises on one L2 address (185 us for the 1 M
// points of config E); here a workgroup compacts its first-touch cells in LDS, reserves their range with ONE global atomic on
// counter kListBase + (workgroup % kLists) and appends to that list: 64 addresses, ~60 atomics each.  List k holds at most
// (workgroups of k) x 256 entries, so `cap` = ceil(nwg / kLists) * 256 never overflows.
constexpr int kLists = 64;
constexpr int kListBase = 4;                 // counter[4 .. 4 + kLists): cursors; counter[4 + kLists .. 4 + 2 kLists): apply tickets
__device__ __forceinline__ void cast_vote_byte_lists(unsigned* cell_mask, int* touched, int* counter, int cap, int cell, unsigned vote,
                                                     int C, unsigned bonus_classes) {
    __shared__ int wg_count, wg_base;
    if (threadIdx.x == 0) wg_count = 0;
    __syncthreads();
    bool first = false;
    if (cell >= 0 && vote != 0) {
        const unsigned enc = encode_vote_byte(vote, C, bonus_classes);
        const unsigned sh = 8u * ((unsigned)cell & 3u);
        const unsigned old = atomicOr(&cell_mask[cell >> 2], enc << sh);
        first = ((old >> sh) & 0xffu) == 0u;          // this lane turned the cell's byte non-zero: it lists the cell
    }
    const unsigned long long m = __ballot(first);
    const int lane = threadIdx.x & 63;
    int wave_off = 0;
    if (m != 0ull && lane == 0) wave_off = atomicAdd(&wg_count, __builtin_popcountll(m));
    wave_off = __shfl(wave_off, 0);
    __syncthreads();
    const int list = blockIdx.x % kLists;
    if (threadIdx.x == 0 && wg_count > 0) wg_base = atomicAdd(&counter[kListBase + list], wg_count);
    __syncthreads();
    // cap = (vote workgroups of this list) x 256 cannot overflow while the cursors start from zero; they would not after an apply
    // launch that failed (the host then resets them, avl_fused_frame), so the append is bounded all the same: an entry
    // past cap is dropped, never written outside the list
    const int slot = wg_base + wave_off + __builtin_popcountll(m & ((1ull << lane) - 1ull));
    if (first && slot < cap) touched[(long long)list * cap + slot] = cell;
}

// MODE: 0 = 32-bit mask + touched list, 1 = 32-bit mask only (sweep), 2 = byte mask only (sweep, see encode_vote_byte),
//       3 = byte mask + partitioned touched lists (k_grid_apply_lists)
template <int SRC, int MODE>
__global__ void __launch_bounds__(kBlock) k_fused_vote(PtsView pv, ProjParams pp, GridParams g,
                                                       const unsigned char* __restrict__ src, int src_w, int src_h,
                                                       LutParams lut, unsigned* __restrict__ cell_mask,
                                                       int* __restrict__ touched, int* __restrict__ counter, int list_cap) {
    const int k = blockIdx.x * kBlock + threadIdx.x;
    int cell = -1;
    unsigned vote = 0;
    if (k < pv.n) {
        double x, y, z, it;
        load_point(pv, k, x, y, z, it);
        int ix, iy;
        if (project(pp, x, y, z, ix, iy)) {
            cell = grid_cell(g, x, y, z);
            if (cell >= 0) {
                if (SRC == AVL_SRC_RGB) {
                    const unsigned char* px = src + 3ll * ((long long)iy * src_w + ix);
                    vote = vote_from_rg(g, px[0], px[1]);
                } else {
                    // cv2 INTER_NEAREST source index: min(floor(d * (src/dst)), src - 1), double arithmetic
                    int sx = ix, sy = iy;
                    if (src_w != pp.img_w) sx = min((int)__builtin_floor((double)ix * ((double)src_w / (double)pp.img_w)), src_w - 1);
                    if (src_h != pp.img_h) sy = min((int)__builtin_floor((double)iy * ((double)src_h / (double)pp.img_h)), src_h - 1);
                    vote = lut.lut[src[(long long)sy * src_w + sx]];
                }
                vote = add_bonus(vote, g.bonus_classes, it);
            }
        }
    }
    if (MODE == 0) cast_vote(cell_mask, touched, counter, cell, vote);
    else if (MODE == 1) cast_vote_nolist(cell_mask, cell, vote);
    else if (MODE == 2) cast_vote_byte(cell_mask, cell, vote, g.C, g.bonus_classes);
    else cast_vote_byte_lists(cell_mask, touched, counter, list_cap, cell, vote, g.C, g.bonus_classes);
}

// ---------------------------------------------------------------- touched cells -> grid (:424,437)
// One lane per touched cell: adds CM[:, i] for every class bit in reference order, the bonus right
// after its class, clears the mask.  Each cell appears once in `touched`, so no atomics on the map.
template <typename MapT>
__global__ void __launch_bounds__(kBlock) k_grid_apply(MapT* __restrict__ map, MapT* __restrict__ rows, int C, CmParams cm,
                                                       unsigned* __restrict__ cell_mask, const int* __restrict__ touched,
                                                       const int* __restrict__ counter) {
    const int n = *counter;
    for (int k = blockIdx.x * kBlock + threadIdx.x; k < n; k += gridDim.x * kBlock) {
        const int cell = touched[k];
        const unsigned m = cell_mask[cell];
        cell_mask[cell] = 0u;
        MapT* row = rows ? rows + (long long)k * C : map + (long long)cell * C;
        double v[AVL_MAX_MAP_CLASSES];
#pragma unroll
        for (int c = 0; c < AVL_MAX_MAP_CLASSES; ++c)
            if (c < C) v[c] = (double)row[c];
        for (int i = 0; i < C; ++i) {
            if (m & (1u << i)) {
#pragma unroll
                for (int c = 0; c < AVL_MAX_MAP_CLASSES; ++c)
                    if (c < C) v[c] = (double)(MapT)(v[c] + cm.cm[c * C + i]);
            }
            if (m & (1u << (16 + i))) {
#pragma unroll
                for (int c = 0; c < AVL_MAX_MAP_CLASSES; ++c)
                    if (c == i) v[c] = (double)(MapT)(v[c] + 2.0);
            }
        }
#pragma unroll
        for (int c = 0; c < AVL_MAX_MAP_CLASSES; ++c)
            if (c < C) row[c] = (MapT)v[c];
    }
}

// Sweep variant of the apply pass for dense clouds: every lane inspects 4 consecutive cells' masks (one
// 16-byte load); non-zero ones are applied exactly like k_grid_apply and cleared.  Cost ~ Hm*Wm*4 bytes,
// independent of the number of points.
template <typename MapT>
__global__ void __launch_bounds__(kBlock) k_grid_apply_scan(MapT* __restrict__ map, int C, CmParams cm,
                                                            unsigned* __restrict__ cell_mask, long long ncell4) {
    for (long long q = (long long)blockIdx.x * kBlock + threadIdx.x; q < ncell4; q += (long long)gridDim.x * kBlock) {
        const uint4 m4 = reinterpret_cast<const uint4*>(cell_mask)[q];
        if ((m4.x | m4.y | m4.z | m4.w) == 0u) continue;
        const unsigned mm[4] = {m4.x, m4.y, m4.z, m4.w};
        for (int j = 0; j < 4; ++j) {
            const unsigned m = mm[j];
            if (m == 0u) continue;
            MapT* row = map + (q * 4 + j) * C;
            for (int i = 0; i < C; ++i) {
                if (m & (1u << i))
                    for (int c = 0; c < C; ++c) row[c] = (MapT)((double)row[c] + cm.cm[c * C + i]);
                if (m & (1u << (16 + i))) row[i] = (MapT)((double)row[i] + 2.0);
            }
        }
        reinterpret_cast<uint4*>(cell_mask)[q] = make_uint4(0u, 0u, 0u, 0u);
    }
}

// Partitioned lists -> grid (MODE 3 of k_fused_vote).  Workgroup (list k, slice j) applies entries j*256 + lane, + W*256, ... of
// list k: reads the cell's vote byte, clears it, updates the row in the reference's class order.  The last workgroup of a list
// to finish (a ticket per list: 64 addresses again) zeroes the list's cursor and ticket, so the counters are all zero on return.
template <typename MapT>
__global__ void __launch_bounds__(kBlock) k_grid_apply_lists(MapT* __restrict__ map, int C, unsigned bonus_classes, CmParams cm,
                                                             unsigned char* __restrict__ mask, const int* __restrict__ touched,
                                                             int* __restrict__ counter, int cap, int wgs_per_list) {
    const int k = blockIdx.x % kLists, j = blockIdx.x / kLists;
    // the append drops entries past cap but its cursor still advances (k_vote_append): clamp here as well, or a stale / overrun
    // cursor would walk into the next list
    const int n = min(counter[kListBase + k], cap);
    const int* list = touched + (long long)k * cap;
    for (int e = j * kBlock + threadIdx.x; e < n; e += wgs_per_list * kBlock) {
        const int cell = list[e];
        const unsigned m = mask[cell];
        mask[cell] = 0;
        MapT* row = map + (long long)cell * C;
        double vals[AVL_MAX_MAP_CLASSES];
#pragma unroll
        for (int c = 0; c < AVL_MAX_MAP_CLASSES; ++c)
            if (c < C) vals[c] = (double)row[c];
        int r = 0;
        for (int i = 0; i < C; ++i) {
            if (m & (1u << i)) {
#pragma unroll
                for (int c = 0; c < AVL_MAX_MAP_CLASSES; ++c)
                    if (c < C) vals[c] = (double)(MapT)(vals[c] + cm.cm[c * C + i]);
            }
            if ((bonus_classes >> i) & 1u) {
                if (m & (1u << (C + r))) {
#pragma unroll
                    for (int c = 0; c < AVL_MAX_MAP_CLASSES; ++c)
                        if (c == i) vals[c] = (double)(MapT)(vals[c] + 2.0);
                }
                ++r;
            }
        }
#pragma unroll
        for (int c = 0; c < AVL_MAX_MAP_CLASSES; ++c)
            if (c < C) row[c] = (MapT)vals[c];
    }
    __syncthreads();                  // every lane of this workgroup has read n
    if (threadIdx.x == 0) {
        const int t = atomicAdd(&counter[kListBase + kLists + k], 1);
        if (t == wgs_per_list - 1) {
            counter[kListBase + k] = 0;
            counter[kListBase + kLists + k] = 0;
        }
    }
}

// Sweep of the BYTE mask.  The 32-bit sweep above is latency-bound (one dependent 16-byte load per lane and iteration, then
// a serial read-modify-write chain in the few lanes that found something: 57 us for the 64 MB mask of config E).  Here a
// workgroup takes 16 K cells at a time: every lane has four independent 16-byte mask loads in flight, the non-zero bytes
// are compacted into an LDS list (local cell index << 8 | bits) with one LDS atomic each, the mask bytes are cleared,
// and then ALL lanes apply one listed cell each: the grid rows are read, updated in the reference's class order
// (class i, then its lane bonus) and written back with every lane busy and every load independent.
// Cells per workgroup round: a camera frustum puts most touched cells into a few grid rows, and the sweep ends when its BUSIEST
// workgroup does -- 4096-cell rounds (one 16-byte mask vector per lane) instead of 16384: config E 38.4 -> 34.6 us per frame
// (profiles/r04/mapping_sweep_chunk_ab.log; AVL_SWEEP_VEC = 4 / 2 / 1 in the experiments build).
constexpr int kSweepVecDefault = 1;
template <typename MapT, int kSweepVec>                       // 16-byte mask vectors per lane and round: 4 / 2 / 1 -> 16384 / 8192 / 4096 cells
__global__ void __launch_bounds__(kBlock) k_grid_sweep_bytes(MapT* __restrict__ map, int C, unsigned bonus_classes, CmParams cm,
                                                             unsigned char* __restrict__ mask, long long ncell, int exp) {
    constexpr int kSweepCells = kBlock * kSweepVec * 16;      // cells per workgroup round
    __shared__ unsigned list[kSweepCells];
    __shared__ int count;
    const long long rounds = (ncell + kSweepCells - 1) / kSweepCells;
    for (long long rd = blockIdx.x; rd < rounds; rd += gridDim.x) {
        if (threadIdx.x == 0) count = 0;
        __syncthreads();
        const long long base = rd * kSweepCells;
        uint4 v[kSweepVec];
#pragma unroll
        for (int u = 0; u < kSweepVec; ++u) {
            const long long c0 = base + ((long long)u * kBlock + threadIdx.x) * 16;
            v[u] = c0 < ncell ? *reinterpret_cast<const uint4*>(mask + c0) : make_uint4(0u, 0u, 0u, 0u);
        }
        // list slots: every lane counts its non-zero bytes, one ballot-free wave scan (shuffles) gives its offset inside the
        // wave, ONE LDS atomic per wave reserves the wave's range -- instead of one returning LDS atomic per listed cell
        int mine = 0;
#pragma unroll
        for (int u = 0; u < kSweepVec; ++u) {
            const unsigned w4[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned x = w4[q];
                // one bit per non-zero byte: (x | x >> 1 | ... | x >> 7) & 0x01010101
                unsigned t = x | (x >> 4);
                t |= t >> 2;
                t |= t >> 1;
                mine += __builtin_popcount(t & 0x01010101u);
            }
        }
#ifdef AVL_EXPERIMENTS
        if (exp & 2) mine = 0;
#endif
        const int lane = threadIdx.x & 63;
        int incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(incl, off);
            if (lane >= off) incl += up;
        }
        int wave_base = 0;
        if (lane == 63 && incl > 0) wave_base = atomicAdd(&count, incl);
        wave_base = __shfl(wave_base, 63);
        int slot = wave_base + incl - mine;
        if (mine > 0) {
#pragma unroll
            for (int u = 0; u < kSweepVec; ++u) {
                if ((v[u].x | v[u].y | v[u].z | v[u].w) == 0u) continue;
                const int local0 = (u * kBlock + threadIdx.x) * 16;
                auto take = [&](unsigned x, int first) {          // the non-zero bytes of one dword -> list (registers only: no indexed arrays)
                    while (x) {
                        const int b = __builtin_ctz(x) >> 3;
                        list[slot++] = ((unsigned)(local0 + first + b) << 8) | ((x >> (8 * b)) & 0xffu);
                        x &= ~(0xffu << (8 * b));
                    }
                };
                take(v[u].x, 0);
                take(v[u].y, 4);
                take(v[u].z, 8);
                take(v[u].w, 12);
            }
        }
#pragma unroll
        for (int u = 0; u < kSweepVec; ++u)
            if ((v[u].x | v[u].y | v[u].z | v[u].w) != 0u)
                *reinterpret_cast<uint4*>(mask + base + (u * kBlock + threadIdx.x) * 16) = make_uint4(0u, 0u, 0u, 0u);
        __syncthreads();
#ifdef AVL_EXPERIMENTS
        const int n = (exp & 1) ? 0 : count;
#else
        const int n = count;
#endif
        for (int k = threadIdx.x; k < n; k += kBlock) {
            const unsigned e = list[k];
            const unsigned m = e & 0xffu;
            MapT* row = map + (base + (e >> 8)) * C;
            double vals[AVL_MAX_MAP_CLASSES];
#pragma unroll
            for (int c = 0; c < AVL_MAX_MAP_CLASSES; ++c)
                if (c < C) vals[c] = (double)row[c];
            int r = 0;
            for (int i = 0; i < C; ++i) {
                if (m & (1u << i)) {
#pragma unroll
                    for (int c = 0; c < AVL_MAX_MAP_CLASSES; ++c)
                        if (c < C) vals[c] = (double)(MapT)(vals[c] + cm.cm[c * C + i]);
                }
                if ((bonus_classes >> i) & 1u) {
                    if (m & (1u << (C + r))) {
#pragma unroll
                        for (int c = 0; c < AVL_MAX_MAP_CLASSES; ++c)
                            if (c == i) vals[c] = (double)(MapT)(vals[c] + 2.0);
                    }
                    ++r;
                }
            }
#pragma unroll
            for (int c = 0; c < AVL_MAX_MAP_CLASSES; ++c)
                if (c < C) row[c] = (MapT)vals[c];
        }
        __syncthreads();
    }
}

// ================================================================ end-of-run rendering (src/renderer.py), SURVEY 8f row 3
struct RenderParams {
    unsigned char colors[AVL_MAX_MAP_CLASSES * 3];
    int priority[AVL_MAX_MAP_CLASSES];
    double thresholds[AVL_MAX_MAP_CLASSES];
};

// render_bev_map (renderer.py:32-59): colour of np.argmax (first maximum wins), black where the channel sum is 0
template <typename MapT>
__global__ void __launch_bounds__(kBlock) k_render_bev(const MapT* __restrict__ map, long long ncell, int C, RenderParams rp,
                                                       unsigned char* __restrict__ out) {
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= ncell) return;
    const MapT* row = map + i * C;
    double best = (double)row[0], sum = 0.0;
    int bi = 0;
    for (int c = 0; c < C; ++c) {
        const double v = (double)row[c];
        sum = sum + v;                                            // np.sum over 5 contiguous values: sequential from 0
        if (c > 0 && (v > best || (v != v && best == best))) { best = v; bi = c; }
    }
    unsigned char r = rp.colors[3 * bi], g = rp.colors[3 * bi + 1], b = rp.colors[3 * bi + 2];
    if (sum == 0.0) r = g = b = 0;
    out[3 * i] = r; out[3 * i + 1] = g; out[3 * i + 2] = b;
}

// render_bev_map_with_thresholds (renderer.py:131-172): later priorities overwrite earlier ones
template <typename MapT>
__global__ void __launch_bounds__(kBlock) k_render_thresholds(const MapT* __restrict__ map, long long ncell, int C, RenderParams rp,
                                                              unsigned char* __restrict__ out) {
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= ncell) return;
    const MapT* row = map + i * C;
    MapT s = (MapT)0;
    for (int c = 0; c < C; ++c) s = s + row[c];
    unsigned char r = 0, g = 0, b = 0;
    if (s != (MapT)0) {
        for (int k = 0; k < C; ++k) {
            const int ch = rp.priority[k];
            const MapT pn = row[ch] / s;                          // np.divide(map, channel_sum), same element type as the map
            if ((double)pn >= rp.thresholds[k]) { r = rp.colors[3 * ch]; g = rp.colors[3 * ch + 1]; b = rp.colors[3 * ch + 2]; }
        }
    }
    out[3 * i] = r; out[3 * i + 1] = g; out[3 * i + 2] = b;
}

// apply_filter (renderer.py:175-189): cv2.filter2D with ones(3,3,float32)/9, BORDER_REFLECT_101, double accumulation
template <typename MapT>
__global__ void __launch_bounds__(kBlock) k_box_filter3(const MapT* __restrict__ src, MapT* __restrict__ dst, int Hm, int Wm, int C) {
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    const long long total = (long long)Hm * Wm * C;
    if (i >= total) return;
    const int c = (int)(i % C);
    const long long cell = i / C;
    const int x = (int)(cell % Wm), y = (int)(cell / Wm);
    const double k = (double)(1.0f / 9.0f);
    double acc = 0.0;
    for (int dy = -1; dy <= 1; ++dy) {
        int yy = y + dy;
        yy = yy < 0 ? -yy : (yy >= Hm ? 2 * Hm - 2 - yy : yy);
        for (int dx = -1; dx <= 1; ++dx) {
            int xx = x + dx;
            xx = xx < 0 ? -xx : (xx >= Wm ? 2 * Wm - 2 - xx : xx);
            acc = acc + k * (double)src[((long long)yy * Wm + xx) * C + c];
        }
    }
    dst[i] = (MapT)acc;
}

__global__ void __launch_bounds__(kBlock) k_colorize(const unsigned char* __restrict__ labels, int lw, int lh,
                                                     const unsigned* __restrict__ pal_packed_dummy, LutParams pal,
                                                     unsigned char* __restrict__ out, int out_w, int out_h) {
    (void)pal_packed_dummy;
    const long long idx = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (idx >= (long long)out_w * out_h) return;
    const int dx = (int)(idx % out_w), dy = (int)(idx / out_w);
    int sx = dx, sy = dy;
    if (lw != out_w) sx = min((int)__builtin_floor((double)dx * ((double)lw / (double)out_w)), lw - 1);
    if (lh != out_h) sy = min((int)__builtin_floor((double)dy * ((double)lh / (double)out_h)), lh - 1);
    const unsigned c = pal.lut[labels[(long long)sy * lw + sx]];
    out[3 * idx + 0] = (unsigned char)(c & 255u);
    out[3 * idx + 1] = (unsigned char)((c >> 8) & 255u);
    out[3 * idx + 2] = (unsigned char)((c >> 16) & 255u);
}

// ---------------------------------------------------------------- host helpers
int fill_proj(ProjParams& pp, const double* P, const double* T, double range_max, int img_w, int img_h) {
    if (!P) return avl::set_error(AVL_E_ARG, "P_host is NULL");
    if (img_w <= 0 || img_h <= 0) return avl::set_error(AVL_E_ARG, "image size %dx%d", img_w, img_h);
    memcpy(pp.P, P, sizeof(pp.P));
    pp.has_T = T ? 1 : 0;
    if (T) memcpy(pp.T, T, sizeof(pp.T));
    else memset(pp.T, 0, sizeof(pp.T));
    pp.range_max = range_max;
    pp.img_w = img_w;
    pp.img_h = img_h;
    return AVL_OK;
}

int fill_pts(PtsView& pv, const void* pts, int n, int dtype, int64_t point_stride, int64_t comp_stride) {
    if (n < 0) return avl::set_error(AVL_E_ARG, "n = %d", n);
    if (n > 0 && !pts) return avl::set_error(AVL_E_ARG, "pts is NULL");
    if (dtype != AVL_F32 && dtype != AVL_F64) return avl::set_error(AVL_E_ARG, "point dtype %d", dtype);
    const int64_t es = dtype == AVL_F64 ? 8 : 4;
    if (point_stride % es || comp_stride % es || point_stride <= 0 || comp_stride <= 0)
        return avl::set_error(AVL_E_ARG, "point strides %lld/%lld not multiples of %lld", (long long)point_stride,
                              (long long)comp_stride, (long long)es);
    pv.base = static_cast<const char*>(pts);
    pv.n = n;
    pv.dtype = dtype;
    pv.point_stride = point_stride;
    pv.comp_stride = comp_stride;
    pv.aos_f32 = (dtype == AVL_F32 && point_stride == 16 && comp_stride == 4 && (reinterpret_cast<uintptr_t>(pts) & 15) == 0);
    return AVL_OK;
}

int fill_grid(GridParams& gp, const avl_grid* g, const uint8_t* colors, uint32_t bonus) {
    if (!g || !g->map || !g->cell_mask || !g->touched || !g->counter) return avl::set_error(AVL_E_ARG, "avl_grid has NULL members");
    if (g->Hm <= 0 || g->Wm <= 0 || g->C <= 0 || g->C > AVL_MAX_MAP_CLASSES)
        return avl::set_error(AVL_E_ARG, "grid %dx%dx%d (C must be 1..%d)", g->Hm, g->Wm, g->C, AVL_MAX_MAP_CLASSES);
    if ((long long)g->Hm * g->Wm > 0x7fffffffLL) return avl::set_error(AVL_E_ARG, "grid has more than 2^31 cells");
    if (g->map_dtype != AVL_F32 && g->map_dtype != AVL_F64) return avl::set_error(AVL_E_ARG, "map dtype %d", g->map_dtype);
    if (!(g->resolution > 0.0)) return avl::set_error(AVL_E_ARG, "resolution %g", g->resolution);
    if (bonus >> g->C) return avl::set_error(AVL_E_ARG, "bonus_classes has bits beyond C");
    gp.off_x = g->off_x; gp.off_y = g->off_y; gp.b00 = g->b00; gp.b10 = g->b10; gp.resolution = g->resolution;
    gp.Hm = g->Hm; gp.Wm = g->Wm; gp.C = g->C;
    gp.bonus_classes = bonus;
    memset(gp.colors, 0, sizeof(gp.colors));
    if (colors) memcpy(gp.colors, colors, 3 * g->C);
    return AVL_OK;
}

int launch_apply(const avl_grid* g, const double* cm_host, void* rows, int rows_dtype, hipStream_t s) {
    if (!cm_host) return avl::set_error(AVL_E_ARG, "cm_host is NULL");
    const int dt = rows ? rows_dtype : g->map_dtype;
    if (dt != AVL_F32 && dt != AVL_F64) return avl::set_error(AVL_E_ARG, "rows dtype %d", dt);
    CmParams cm;
    memset(&cm, 0, sizeof(cm));
    memcpy(cm.cm, cm_host, sizeof(double) * g->C * g->C);
    const int cap = g->touched_cap;
    int blocks = (cap + kBlock - 1) / kBlock;
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    if (dt == AVL_F64)
        hipLaunchKernelGGL(k_grid_apply<double>, dim3(blocks), dim3(kBlock), 0, s, static_cast<double*>(g->map),
                           static_cast<double*>(rows), g->C, cm, g->cell_mask, g->touched, g->counter);
    else
        hipLaunchKernelGGL(k_grid_apply<float>, dim3(blocks), dim3(kBlock), 0, s, static_cast<float*>(g->map),
                           static_cast<float*>(rows), g->C, cm, g->cell_mask, g->touched, g->counter);
    AVL_LAUNCH_CHECK();
    return AVL_OK;
}

int launch_apply_scan(const avl_grid* g, const double* cm_host, hipStream_t s) {
    if (!cm_host) return avl::set_error(AVL_E_ARG, "cm_host is NULL");
    CmParams cm;
    memset(&cm, 0, sizeof(cm));
    memcpy(cm.cm, cm_host, sizeof(double) * g->C * g->C);
    const long long ncell4 = (long long)g->Hm * g->Wm / 4;
    if (g->map_dtype == AVL_F64)
        hipLaunchKernelGGL(k_grid_apply_scan<double>, dim3(2048), dim3(kBlock), 0, s, static_cast<double*>(g->map), g->C, cm, g->cell_mask, ncell4);
    else
        hipLaunchKernelGGL(k_grid_apply_scan<float>, dim3(2048), dim3(kBlock), 0, s, static_cast<float*>(g->map), g->C, cm, g->cell_mask, ncell4);
    AVL_LAUNCH_CHECK();
    return AVL_OK;
}

int launch_sweep_bytes(const avl_grid* g, const double* cm_host, unsigned bonus, hipStream_t s) {
    if (!cm_host) return avl::set_error(AVL_E_ARG, "cm_host is NULL");
    CmParams cm;
    memset(&cm, 0, sizeof(cm));
    memcpy(cm.cm, cm_host, sizeof(double) * g->C * g->C);
    const long long ncell = (long long)g->Hm * g->Wm;
    unsigned char* mask = reinterpret_cast<unsigned char*>(g->cell_mask);
    const int exp = AVL_EXP_INT("AVL_SWEEP_EXP", 0);      // timing experiments only (experiments build; 0 in the release library)
#ifdef AVL_EXPERIMENTS
    const int vec = AVL_EXP_INT("AVL_SWEEP_VEC", kSweepVecDefault);
#else
    constexpr int vec = kSweepVecDefault;
#endif
#define AVL_SWEEP_LAUNCH(T, V)                                                                                                        \
    do {                                                                                                                              \
        const long long rounds = (ncell + kBlock * V * 16 - 1) / (kBlock * V * 16);                                                   \
        const unsigned blocks = (unsigned)(rounds < 8192 ? rounds : 8192);                                                            \
        hipLaunchKernelGGL((k_grid_sweep_bytes<T, V>), dim3(blocks), dim3(kBlock), 0, s, static_cast<T*>(g->map), g->C, bonus, cm, mask, ncell, exp); \
    } while (0)
    if (g->map_dtype == AVL_F64) {
#ifdef AVL_EXPERIMENTS
        if (vec == 4) AVL_SWEEP_LAUNCH(double, 4); else if (vec == 2) AVL_SWEEP_LAUNCH(double, 2); else
#endif
            AVL_SWEEP_LAUNCH(double, kSweepVecDefault);
    } else {
#ifdef AVL_EXPERIMENTS
        if (vec == 4) AVL_SWEEP_LAUNCH(float, 4); else if (vec == 2) AVL_SWEEP_LAUNCH(float, 2); else
#endif
            AVL_SWEEP_LAUNCH(float, kSweepVecDefault);
    }
#undef AVL_SWEEP_LAUNCH
    (void)vec;
    AVL_LAUNCH_CHECK();
    return AVL_OK;
}

// byte mask usable: class bits + bonus bits fit a byte, whole 16-byte vectors, aligned scratch
bool byte_mask_ok(const avl_grid* g, unsigned bonus) {
    const char* mode = AVL_EXP_STR("AVL_MASK_MODE");          // "32": keep the 32-bit mask (experiments build only)
    if (mode && mode[0] == '3') return false;
    const long long cells = (long long)g->Hm * g->Wm;
    return g->C + __builtin_popcount(bonus) <= 8 && cells % 16 == 0 && (reinterpret_cast<uintptr_t>(g->cell_mask) & 15) == 0;
}

// partitioned lists: geometry shared by the vote and the apply launch
struct ListGeom { int cap, wgs_per_list, apply_wgs; };
ListGeom list_geom(int n) {
    ListGeom lg;
    const int nwg = (n + kBlock - 1) / kBlock;
    const int per = (nwg + kLists - 1) / kLists;                 // vote workgroups per list
    lg.cap = per * kBlock;
    lg.wgs_per_list = per < 16 ? per : 16;                       // apply workgroups per list (grid-stride over the list)
    lg.apply_wgs = lg.wgs_per_list * kLists;
    return lg;
}
// MODE 3 needs the big counter block, room for kLists x cap entries and a byte mask.  Its cost grows with the cloud (returning
// atomics in the vote, rows visited in point order -- no locality -- in the apply), the sweep's is ~15 us whatever the cloud:
// measured 17 vs 22 us per frame at 120 k points on 4 M cells (config C), 54 vs 38 us at 1 M points on 16 M cells (config E).
// AVL_APPLY_MODE=plist forces it (experiments).
bool use_lists(const avl_grid* g, int n, unsigned bonus) {
    const char* mode = AVL_EXP_STR("AVL_APPLY_MODE");
    if (g->counter_len < kListBase + 2 * kLists || !byte_mask_ok(g, bonus)) return false;
    if ((long long)list_geom(n).cap * kLists > g->touched_cap) return false;
    if (mode && mode[0] == 'p') return true;
    if (mode) return false;
    return n <= 250000 && (long long)n * 2 <= (long long)g->Hm * g->Wm;
}
int launch_apply_lists(const avl_grid* g, const double* cm_host, unsigned bonus, int n, hipStream_t s) {
    CmParams cm;
    memset(&cm, 0, sizeof(cm));
    for (int i = 0; i < g->C * g->C; ++i) cm.cm[i] = cm_host[i];
    const ListGeom lg = list_geom(n);
    unsigned char* mask = reinterpret_cast<unsigned char*>(g->cell_mask);
    if (g->map_dtype == AVL_F64)
        hipLaunchKernelGGL(k_grid_apply_lists<double>, dim3(lg.apply_wgs), dim3(kBlock), 0, s, static_cast<double*>(g->map), g->C, bonus, cm,
                           mask, g->touched, g->counter, lg.cap, lg.wgs_per_list);
    else
        hipLaunchKernelGGL(k_grid_apply_lists<float>, dim3(lg.apply_wgs), dim3(kBlock), 0, s, static_cast<float*>(g->map), g->C, bonus, cm,
                           mask, g->touched, g->counter, lg.cap, lg.wgs_per_list);
    AVL_LAUNCH_CHECK();
    return AVL_OK;
}

// list (sparse) vs sweep (dense) apply: the sweep reads Hm*Wm*4 bytes whatever the cloud; the list costs a
// returning atomic + an append per first touch.  AVL_APPLY_MODE=list|scan overrides (experiments).
bool use_scan(const avl_grid* g, int n, unsigned bonus) {
    const char* mode = AVL_EXP_STR("AVL_APPLY_MODE");
    const long long cells = (long long)g->Hm * g->Wm;
    if (cells % 4 != 0 || (reinterpret_cast<uintptr_t>(g->cell_mask) & 15)) return false;
    if (mode && mode[0] == 'l') return false;
    if (mode && mode[0] == 's') return true;
    // measured (32-bit sweep): sweep wins at 120 k points on 4 M cells (25 vs 36 us) and 1 M on 16 M (73 vs 213 us); the byte
    // sweep reads a quarter of that, so it stays ahead down to much sparser clouds
    return (long long)n * (byte_mask_ok(g, bonus) ? 1024 : 128) >= cells;
}

}  // namespace

// ============================================================================ C ABI
extern "C" int avl_project_points(const void* pts, int n, int dtype, int64_t point_stride, int64_t comp_stride,
                                  const double* P_host, const double* T_host, double range_max, int img_w, int img_h,
                                  int32_t* out_ixy, uint8_t* out_mask, void* stream) {
    PtsView pv;
    ProjParams pp;
    int rc;
    if ((rc = fill_pts(pv, pts, n, dtype, point_stride, comp_stride))) return rc;
    if ((rc = fill_proj(pp, P_host, T_host, range_max, img_w, img_h))) return rc;
    if (n == 0) return AVL_OK;
    hipLaunchKernelGGL(k_project_points, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, avl::as_stream(stream), pv, pp,
                       out_ixy, out_mask);
    AVL_LAUNCH_CHECK();
    return AVL_OK;
}

extern "C" int64_t avl_project_pcd_scratch_bytes(int n) {
    if (n < 0) return 0;
    const int64_t nb = (n + kBlock - 1) / kBlock;
    return (int64_t)sizeof(int) * ((int64_t)n + nb + 16);
}

extern "C" int avl_project_pcd(const void* pts, int n, int dtype, int64_t point_stride, int64_t comp_stride,
                               const double* P_host, const double* T_host, double range_max, const uint8_t* image,
                               int img_w, int img_h, double* out_pcd, uint8_t* out_label, int64_t out_ld,
                               int32_t* out_count, void* scratch, void* stream) {
    PtsView pv;
    ProjParams pp;
    int rc;
    if ((rc = fill_pts(pv, pts, n, dtype, point_stride, comp_stride))) return rc;
    if ((rc = fill_proj(pp, P_host, T_host, range_max, img_w, img_h))) return rc;
    AVL_REQUIRE(out_count, "out_count is NULL");
    hipStream_t s = avl::as_stream(stream);
    if (n == 0) {
        AVL_HIP_CHECK(hipMemsetAsync(out_count, 0, sizeof(int), s));
        return AVL_OK;
    }
    AVL_REQUIRE(image && out_pcd && out_label && scratch, "NULL buffer");
    AVL_REQUIRE(out_ld >= n, "out_ld %lld < n %d", (long long)out_ld, n);
    AVL_REQUIRE((long long)img_w * img_h < 0x7fffffffLL, "image too large");
    const int nb = (n + kBlock - 1) / kBlock;
    int* pix = static_cast<int*>(scratch);
    int* block_count = pix + n;
    hipLaunchKernelGGL(k_pcd_mask, dim3(nb), dim3(kBlock), 0, s, pv, pp, pix, block_count);
    AVL_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, s, block_count, nb, out_count);
    AVL_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_pcd_scatter, dim3(nb), dim3(kBlock), 0, s, pv, pix, block_count, image, out_pcd, out_label,
                       (long long)out_ld);
    AVL_LAUNCH_CHECK();
    return AVL_OK;
}

extern "C" int avl_vote_points(const avl_grid* g, const double* pcd, const uint8_t* label, int64_t ld, int m_host,
                               const int32_t* m_dev, const uint8_t* label_colors_host, uint32_t bonus_classes,
                               void* stream) {
    GridParams gp;
    int rc;
    AVL_REQUIRE(label_colors_host, "label_colors_host is NULL");
    if ((rc = fill_grid(gp, g, label_colors_host, bonus_classes))) return rc;
    AVL_REQUIRE(m_host >= 0, "m = %d", m_host);
    hipStream_t s = avl::as_stream(stream);
    AVL_HIP_CHECK(hipMemsetAsync(g->counter, 0, 16, s));
    if (m_host == 0) return AVL_OK;
    AVL_REQUIRE(pcd && label, "NULL buffer");
    AVL_REQUIRE(ld >= m_host, "ld %lld < m %d", (long long)ld, m_host);
    AVL_REQUIRE(g->touched_cap >= (m_host < g->Hm * g->Wm ? m_host : g->Hm * g->Wm), "touched_cap %d too small", g->touched_cap);
    hipLaunchKernelGGL(k_vote_labelled, dim3((m_host + kBlock - 1) / kBlock), dim3(kBlock), 0, s, pcd, label, (long long)ld,
                       m_host, m_dev, gp, g->cell_mask, g->touched, g->counter);
    AVL_LAUNCH_CHECK();
    return AVL_OK;
}

extern "C" int avl_grid_apply(const avl_grid* g, const double* cm_host, void* rows, int rows_dtype, void* stream) {
    AVL_REQUIRE(g && g->cell_mask && g->touched && g->counter && (rows || g->map), "avl_grid has NULL members");
    AVL_REQUIRE(g->C > 0 && g->C <= AVL_MAX_MAP_CLASSES, "C = %d", g->C);
    return launch_apply(g, cm_host, rows, rows_dtype, avl::as_stream(stream));
}

extern "C" int avl_update_map(const avl_grid* g, const double* pcd, const uint8_t* label, int64_t ld, int m_host,
                              const int32_t* m_dev, const uint8_t* label_colors_host, const double* cm_host,
                              uint32_t bonus_classes, void* stream) {
    AVL_REQUIRE(cm_host, "cm_host is NULL");
    int rc = avl_vote_points(g, pcd, label, ld, m_host, m_dev, label_colors_host, bonus_classes, stream);
    if (rc) return rc;
    return launch_apply(g, cm_host, nullptr, 0, avl::as_stream(stream));
}

extern "C" int avl_fused_frame(const avl_grid* g, const void* pts, int n, int dtype, int64_t point_stride,
                               int64_t comp_stride, const double* P_host, const double* T_host, double range_max,
                               int src_kind, const uint8_t* src, int src_w, int src_h, int img_w, int img_h,
                               const uint32_t* lut_host, const uint8_t* label_colors_host, const double* cm_host,
                               uint32_t bonus_classes, void* stream) {
    PtsView pv;
    ProjParams pp;
    GridParams gp;
    LutParams lut;
    int rc;
    if ((rc = fill_pts(pv, pts, n, dtype, point_stride, comp_stride))) return rc;
    if ((rc = fill_proj(pp, P_host, T_host, range_max, img_w, img_h))) return rc;
    if ((rc = fill_grid(gp, g, label_colors_host, bonus_classes))) return rc;
    AVL_REQUIRE(src_kind == AVL_SRC_RGB || src_kind == AVL_SRC_CLASSMAP, "src_kind %d", src_kind);
    AVL_REQUIRE(src && src_w > 0 && src_h > 0, "bad semantic source");
    memset(&lut, 0, sizeof(lut));
    if (src_kind == AVL_SRC_RGB) {
        AVL_REQUIRE(label_colors_host, "label_colors_host is NULL");
        AVL_REQUIRE(src_w == img_w && src_h == img_h, "RGB source must be %dx%d", img_w, img_h);
    } else {
        AVL_REQUIRE(lut_host, "lut_host is NULL");
        memcpy(lut.lut, lut_host, sizeof(lut.lut));
    }
    if (n == 0) return AVL_OK;
    AVL_REQUIRE(g->touched_cap >= (n < g->Hm * g->Wm ? n : g->Hm * g->Wm), "touched_cap %d too small", g->touched_cap);
    hipStream_t s = avl::as_stream(stream);
    const dim3 grid((n + kBlock - 1) / kBlock), block(kBlock);
    const bool scan = use_scan(g, n, bonus_classes);
    const int mode = use_lists(g, n, bonus_classes) ? 3 : !scan ? 0 : (byte_mask_ok(g, bonus_classes) ? 2 : 1);
    if (mode == 0) AVL_HIP_CHECK(hipMemsetAsync(g->counter, 0, 16, s));      // only the single touched-list path counts there
    const int list_cap = list_geom(n).cap;
#define AVL_FV(SRC, MODE) hipLaunchKernelGGL((k_fused_vote<SRC, MODE>), grid, block, 0, s, pv, pp, gp, src, src_w, src_h, lut, g->cell_mask, g->touched, g->counter, list_cap)
    if (src_kind == AVL_SRC_RGB) { if (mode == 3) AVL_FV(AVL_SRC_RGB, 3); else if (mode == 2) AVL_FV(AVL_SRC_RGB, 2); else if (mode == 1) AVL_FV(AVL_SRC_RGB, 1); else AVL_FV(AVL_SRC_RGB, 0); }
    else { if (mode == 3) AVL_FV(AVL_SRC_CLASSMAP, 3); else if (mode == 2) AVL_FV(AVL_SRC_CLASSMAP, 2); else if (mode == 1) AVL_FV(AVL_SRC_CLASSMAP, 1); else AVL_FV(AVL_SRC_CLASSMAP, 0); }
#undef AVL_FV
    AVL_LAUNCH_CHECK();
    if (mode == 3) {
        rc = launch_apply_lists(g, cm_host, bonus_classes, n, s);
        // only the apply kernel returns the list cursors to zero: if it did not launch, reset them here so that the next frame's
        // vote appends from the start of each list again
        if (rc != AVL_OK) (void)hipMemsetAsync(g->counter + kListBase, 0, 2 * kLists * sizeof(int), s);
        return rc;
    }
    if (mode == 2) return launch_sweep_bytes(g, cm_host, bonus_classes, s);
    return mode == 1 ? launch_apply_scan(g, cm_host, s) : launch_apply(g, cm_host, nullptr, 0, s);
}

extern "C" int avl_colorize_labels(const uint8_t* labels, int lw, int lh, const uint8_t* palette_host, uint8_t* out,
                                   int out_w, int out_h, void* stream) {
    AVL_REQUIRE(labels && palette_host && out, "NULL buffer");
    AVL_REQUIRE(lw > 0 && lh > 0 && out_w > 0 && out_h > 0, "bad sizes");
    LutParams pal;
    for (int k = 0; k < 256; ++k)
        pal.lut[k] = (unsigned)palette_host[3 * k] | ((unsigned)palette_host[3 * k + 1] << 8) | ((unsigned)palette_host[3 * k + 2] << 16);
    const long long total = (long long)out_w * out_h;
    hipLaunchKernelGGL(k_colorize, dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0, avl::as_stream(stream),
                       labels, lw, lh, nullptr, pal, out, out_w, out_h);
    AVL_LAUNCH_CHECK();
    return AVL_OK;
}

// ---------------------------------------------------------------------------------------- semantic point cloud
namespace {
// create_point_cloud (src/utils/utils_ros.py:31-59): one PointCloud2 record per point, fields x,y,z FLOAT32 at
// offsets 0/4/8 and rgba UINT32 at 12 = struct.pack('BBBB', r, g, b, 255) read back as a little-endian 'I'.
__global__ void __launch_bounds__(kBlock) k_pack_cloud(const double* __restrict__ pcd, const unsigned char* __restrict__ label,
                                                       long long ld, int m_host, const int* __restrict__ m_dev, uint4* __restrict__ out) {
    const int m = m_dev ? min(*m_dev, m_host) : m_host;
    const int k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= m) return;
    const float x = (float)pcd[k], y = (float)pcd[ld + k], z = (float)pcd[2 * ld + k];
    const unsigned rgba = (unsigned)label[k] | ((unsigned)label[ld + k] << 8) | ((unsigned)label[2 * ld + k] << 16) | (255u << 24);
    out[k] = make_uint4(__float_as_uint(x), __float_as_uint(y), __float_as_uint(z), rgba);
}
}  // namespace

extern "C" int avl_pack_semantic_cloud(const double* pcd, const uint8_t* label, int64_t ld, int m_host, const int32_t* m_dev,
                                       void* out_records, void* stream) {
    AVL_REQUIRE(m_host >= 0, "m = %d", m_host);
    if (m_host == 0) return AVL_OK;
    AVL_REQUIRE(pcd && label && out_records && ld >= m_host, "bad buffers");
    AVL_REQUIRE(reinterpret_cast<uintptr_t>(out_records) % 16 == 0, "out_records must be 16-byte aligned");
    hipLaunchKernelGGL(k_pack_cloud, dim3((m_host + kBlock - 1) / kBlock), dim3(kBlock), 0, avl::as_stream(stream), pcd, label,
                       (long long)ld, m_host, m_dev, static_cast<uint4*>(out_records));
    AVL_LAUNCH_CHECK();
    return AVL_OK;
}

namespace {
struct PlanarParams {
    double hi[9];
    unsigned char colors[AVL_MAX_MAP_CLASSES * 3];
    int C, sep, match, img_h, img_w;
};
// One lane per grid cell: inverse-map the cell into the image (float64, the oracle's expression order: this file is
// compiled with -ffp-contract=off), bilinear R and G with a zero border, class test, +1, clamp.
template <typename MapT>
__global__ void __launch_bounds__(kBlock) k_planar_update(MapT* __restrict__ map, int Hm, int Wm, const unsigned char* __restrict__ img,
                                                          PlanarParams pp) {
    const long long cell = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (cell >= (long long)Hm * Wm) return;
    const int y = (int)(cell / Wm), x = (int)(cell % Wm);
    MapT* row = map + cell * pp.C;
    if (pp.match && x >= pp.sep) {
        const double xs = (double)x, ys = (double)y;
        const double den = pp.hi[6] * xs + pp.hi[7] * ys + pp.hi[8];
        double sx = (pp.hi[0] * xs + pp.hi[1] * ys + pp.hi[2]) / den;
        double sy = (pp.hi[3] * xs + pp.hi[4] * ys + pp.hi[5]) / den;
        if (!(__builtin_isfinite(sx) && __builtin_isfinite(sy) && __builtin_fabs(sx) < 1e9 && __builtin_fabs(sy) < 1e9)) sx = sy = -10.0;
        const double fx = __builtin_floor(sx), fy = __builtin_floor(sy);
        const double ax = sx - fx, ay = sy - fy;
        const long long x0 = (long long)fx, y0 = (long long)fy;
        double acc[2] = {0.0, 0.0};
        for (int dy = 0; dy < 2; ++dy)
            for (int dx = 0; dx < 2; ++dx) {
                const long long xx = x0 + dx, yy = y0 + dy;
                const bool inside = xx >= 0 && xx < pp.img_w && yy >= 0 && yy < pp.img_h;
                const double wgt = (dx ? ax : 1.0 - ax) * (dy ? ay : 1.0 - ay);
                const long long cx = xx < 0 ? 0 : (xx >= pp.img_w ? pp.img_w - 1 : xx), cy = yy < 0 ? 0 : (yy >= pp.img_h ? pp.img_h - 1 : yy);
                const unsigned char* px = img + 3 * (cy * pp.img_w + cx);
                const double wv = inside ? wgt : 0.0;
                acc[0] = acc[0] + wv * (double)px[0];
                acc[1] = acc[1] + wv * (double)px[1];
            }
        const double r = __builtin_rint(acc[0]), g = __builtin_rint(acc[1]);
        const int ri = r < 0.0 ? 0 : (r > 255.0 ? 255 : (int)r), gi = g < 0.0 ? 0 : (g > 255.0 ? 255 : (int)g);
        for (int i = 0; i < pp.C; ++i)
            if (pp.colors[3 * i] == ri && pp.colors[3 * i + 1] == gi) row[i] = (MapT)((double)row[i] + 1.0);
    }
    for (int i = 0; i < pp.C; ++i)
        if (row[i] < (MapT)0) row[i] = (MapT)0;                       // map_local[map_local < 0] = 0 (:481)
}
}  // namespace

extern "C" int avl_planar_update(void* map, int map_dtype, int Hm, int Wm, int C, const uint8_t* image, int img_h, int img_w,
                                 const double* Hinv_host, int sep, const uint8_t* label_colors_host, int match_colour, void* stream) {
    AVL_REQUIRE(map && Hm > 0 && Wm > 0 && C > 0 && C <= AVL_MAX_MAP_CLASSES, "bad grid %dx%dx%d", Hm, Wm, C);
    AVL_REQUIRE(map_dtype == AVL_F64 || map_dtype == AVL_F32, "map dtype %d", map_dtype);
    AVL_REQUIRE(!match_colour || (image && Hinv_host && label_colors_host && img_h > 0 && img_w > 0), "colour matching needs image, Hinv_host and label_colors_host");
    PlanarParams pp;
    memset(&pp, 0, sizeof(pp));
    pp.C = C; pp.sep = sep; pp.match = match_colour ? 1 : 0; pp.img_h = img_h; pp.img_w = img_w;
    if (match_colour) {
        memcpy(pp.hi, Hinv_host, sizeof(pp.hi));
        memcpy(pp.colors, label_colors_host, 3 * C);
    }
    const long long n = (long long)Hm * Wm;
    const dim3 grid((unsigned)((n + kBlock - 1) / kBlock)), block(kBlock);
    if (map_dtype == AVL_F64) hipLaunchKernelGGL(k_planar_update<double>, grid, block, 0, avl::as_stream(stream), static_cast<double*>(map), Hm, Wm, image, pp);
    else hipLaunchKernelGGL(k_planar_update<float>, grid, block, 0, avl::as_stream(stream), static_cast<float*>(map), Hm, Wm, image, pp);
    AVL_LAUNCH_CHECK();
    return AVL_OK;
}

namespace {
// One lane per point; fields are 4-byte aligned, so each is one dword load (a point's record is usually 16-32 bytes:
// the wave reads a contiguous 1-2 KB).  The count goes through one ballot + one atomic per wave.
__global__ void __launch_bounds__(kBlock) k_unpack_cloud(const unsigned char* __restrict__ data, long long n, int step, int ox, int oy,
                                                         int oz, int oi, float4* __restrict__ out, int* __restrict__ n_valid) {
    const long long k = (long long)blockIdx.x * kBlock + threadIdx.x;
    bool ok = false;
    if (k < n) {
        const unsigned char* rec = data + k * step;
        float x = *reinterpret_cast<const float*>(rec + ox);
        const float y = *reinterpret_cast<const float*>(rec + oy), z = *reinterpret_cast<const float*>(rec + oz);
        const float i = *reinterpret_cast<const float*>(rec + oi);
        ok = !(x != x || y != y || z != z || i != i);
        if (!ok) x = __uint_as_float(0x7fc00000u);
        out[k] = make_float4(x, y, z, i);
    }
    if (n_valid) {
        const unsigned long long b = __ballot(ok);
        if ((threadIdx.x & 63) == 0 && b) atomicAdd(n_valid, __popcll(b));
    }
}
}  // namespace

extern "C" int avl_unpack_pointcloud2(const uint8_t* data, int64_t n_points, int point_step, int off_x, int off_y, int off_z, int off_i,
                                      float* out_xyzi, int32_t* n_valid, void* stream) {
    AVL_REQUIRE(n_points >= 0, "n_points = %lld", (long long)n_points);
    if (n_valid) AVL_HIP_CHECK(hipMemsetAsync(n_valid, 0, sizeof(int32_t), avl::as_stream(stream)));
    if (n_points == 0) return AVL_OK;
    AVL_REQUIRE(data && out_xyzi, "bad buffers");
    AVL_REQUIRE(point_step >= 16 && point_step % 4 == 0, "point_step = %d (FLOAT32 fields need 4-byte aligned records)", point_step);
    const int offs[4] = {off_x, off_y, off_z, off_i};
    for (int o : offs) AVL_REQUIRE(o >= 0 && o % 4 == 0 && o + 4 <= point_step, "field offset %d outside a %d-byte record / unaligned", o, point_step);
    AVL_REQUIRE(reinterpret_cast<uintptr_t>(data) % 4 == 0 && reinterpret_cast<uintptr_t>(out_xyzi) % 16 == 0, "unaligned buffers");
    hipLaunchKernelGGL(k_unpack_cloud, dim3((unsigned)((n_points + kBlock - 1) / kBlock)), dim3(kBlock), 0, avl::as_stream(stream), data,
                       (long long)n_points, point_step, off_x, off_y, off_z, off_i, reinterpret_cast<float4*>(out_xyzi), n_valid);
    AVL_LAUNCH_CHECK();
    return AVL_OK;
}

// ---------------------------------------------------------------------------------------- rendering
namespace {
int fill_render(RenderParams& rp, int C, const uint8_t* colors, const int32_t* priority, const double* thresholds) {
    if (C <= 0 || C > AVL_MAX_MAP_CLASSES) return avl::set_error(AVL_E_ARG, "C = %d", C);
    if (!colors) return avl::set_error(AVL_E_ARG, "colors_host is NULL");
    memset(&rp, 0, sizeof(rp));
    memcpy(rp.colors, colors, 3 * C);
    for (int k = 0; k < C; ++k) {
        rp.priority[k] = priority ? priority[k] : k;
        if (rp.priority[k] < 0 || rp.priority[k] >= C) return avl::set_error(AVL_E_ARG, "priority[%d] = %d", k, rp.priority[k]);
        rp.thresholds[k] = thresholds ? thresholds[k] : 0.01;
    }
    return AVL_OK;
}
}  // namespace

extern "C" int avl_render_bev_map(const void* map, int map_dtype, int Hm, int Wm, int C, const uint8_t* colors_host, uint8_t* out,
                                  void* stream) {
    RenderParams rp;
    int rc;
    if ((rc = fill_render(rp, C, colors_host, nullptr, nullptr))) return rc;
    AVL_REQUIRE(map && out && Hm > 0 && Wm > 0, "bad map / out");
    AVL_REQUIRE(map_dtype == AVL_F64 || map_dtype == AVL_F32, "map dtype %d", map_dtype);
    const long long n = (long long)Hm * Wm;
    const dim3 grid((unsigned)((n + kBlock - 1) / kBlock)), block(kBlock);
    if (map_dtype == AVL_F64) hipLaunchKernelGGL(k_render_bev<double>, grid, block, 0, avl::as_stream(stream), static_cast<const double*>(map), n, C, rp, out);
    else hipLaunchKernelGGL(k_render_bev<float>, grid, block, 0, avl::as_stream(stream), static_cast<const float*>(map), n, C, rp, out);
    AVL_LAUNCH_CHECK();
    return AVL_OK;
}

extern "C" int avl_render_bev_map_thresholds(const void* map, int map_dtype, int Hm, int Wm, int C, const uint8_t* colors_host,
                                             const int32_t* priority_host, const double* thresholds_host, uint8_t* out, void* stream) {
    RenderParams rp;
    int rc;
    if ((rc = fill_render(rp, C, colors_host, priority_host, thresholds_host))) return rc;
    AVL_REQUIRE(map && out && Hm > 0 && Wm > 0, "bad map / out");
    AVL_REQUIRE(map_dtype == AVL_F64 || map_dtype == AVL_F32, "map dtype %d", map_dtype);
    const long long n = (long long)Hm * Wm;
    const dim3 grid((unsigned)((n + kBlock - 1) / kBlock)), block(kBlock);
    if (map_dtype == AVL_F64) hipLaunchKernelGGL(k_render_thresholds<double>, grid, block, 0, avl::as_stream(stream), static_cast<const double*>(map), n, C, rp, out);
    else hipLaunchKernelGGL(k_render_thresholds<float>, grid, block, 0, avl::as_stream(stream), static_cast<const float*>(map), n, C, rp, out);
    AVL_LAUNCH_CHECK();
    return AVL_OK;
}

// ---- end-of-run evaluation (test/test_semantic_mapping.py: convert_labels :6-19, Test.iou :127-161) -----------------
// One pass over the rendered colour map: label = 1..5 for the five map colours (exact three-channel match, 0 otherwise
// or where the validity mask is 0), and -- if a ground-truth map is given -- the joint histogram counts[gt][label]
// (8 x 8 bins; gt bin 7 = "positive, but none of 1..6") from which IoU, accuracy and missing rate are integer sums.
namespace {
__global__ void __launch_bounds__(kBlock) k_eval_map(const uint8_t* __restrict__ color, int H, int W, const uint8_t* __restrict__ mask,
                                                   int mask_ld, const uint8_t* __restrict__ gt, int gt_ld, uint8_t* __restrict__ labels,
                                                   unsigned long long* __restrict__ counts) {
    __shared__ unsigned int hist[64];
    if (threadIdx.x < 64) hist[threadIdx.x] = 0;
    __syncthreads();
    const long long n = (long long)H * W;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) {
        const int y = (int)(i / W), x = (int)(i - (long long)y * W);
        const uint8_t r = color[i * 3], g = color[i * 3 + 1], b = color[i * 3 + 2];
        int lab = 0;
        if (r == 128 && g == 64 && b == 128) lab = 1;            // road
        else if (r == 140 && g == 140 && b == 200) lab = 2;      // crosswalk
        else if (r == 255 && g == 255 && b == 255) lab = 3;      // lane
        else if (r == 244 && g == 35 && b == 232) lab = 4;       // sidewalk
        else if (r == 107 && g == 142 && b == 35) lab = 5;       // vegetation
        if (mask && mask[(long long)y * mask_ld + x] == 0) lab = 0;
        if (labels) labels[i] = (uint8_t)lab;
        if (gt) {
            int gv = gt[(long long)y * gt_ld + x];
            gv = gv > 7 ? 7 : gv;
            atomicAdd(&hist[gv * 8 + lab], 1u);
        }
    }
    __syncthreads();
    if (gt && threadIdx.x < 64 && hist[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)hist[threadIdx.x]);
}
}  // namespace

extern "C" int avl_eval_map(const uint8_t* color_map, int H, int W, const uint8_t* mask, int mask_ld, const uint8_t* gt, int gt_ld,
                            uint8_t* labels_out, unsigned long long* counts, void* stream) {
    AVL_REQUIRE(color_map && H > 0 && W > 0, "bad colour map");
    AVL_REQUIRE(labels_out || gt, "avl_eval_map: nothing to compute (labels_out and gt are both NULL)");
    AVL_REQUIRE(!mask || mask_ld >= W, "mask row stride %d < W %d", mask_ld, W);
    AVL_REQUIRE(!gt || (gt_ld >= W && counts), "ground truth needs gt_ld >= W and a counts[64] buffer");
    hipStream_t s = avl::as_stream(stream);
    if (gt) AVL_HIP_CHECK(hipMemsetAsync(counts, 0, 64 * sizeof(unsigned long long), s));
    const long long n = (long long)H * W;
    long long blocks = (n + kBlock - 1) / kBlock;
    if (blocks > 2048) blocks = 2048;                      // grid-stride: one LDS histogram flush per workgroup
    hipLaunchKernelGGL(k_eval_map, dim3((unsigned)blocks), dim3(kBlock), 0, s, color_map, H, W, mask, mask_ld, gt, gt_ld, labels_out, counts);
    AVL_LAUNCH_CHECK();
    return AVL_OK;
}

extern "C" int avl_grid_box_filter(const void* src, void* dst, int map_dtype, int Hm, int Wm, int C, void* stream) {
    AVL_REQUIRE(src && dst && src != dst, "box filter needs distinct src and dst");
    AVL_REQUIRE(Hm > 1 && Wm > 1 && C > 0, "grid %dx%dx%d", Hm, Wm, C);
    AVL_REQUIRE(map_dtype == AVL_F64 || map_dtype == AVL_F32, "map dtype %d", map_dtype);
    const long long n = (long long)Hm * Wm * C;
    const dim3 grid((unsigned)((n + kBlock - 1) / kBlock)), block(kBlock);
    if (map_dtype == AVL_F64) hipLaunchKernelGGL(k_box_filter3<double>, grid, block, 0, avl::as_stream(stream), static_cast<const double*>(src), static_cast<double*>(dst), Hm, Wm, C);
    else hipLaunchKernelGGL(k_box_filter3<float>, grid, block, 0, avl::as_stream(stream), static_cast<const float*>(src), static_cast<float*>(dst), Hm, Wm, C);
    AVL_LAUNCH_CHECK();
    return AVL_OK;
}
