// Element-type helpers shared by the segmentation kernels (gfx950).
#pragma once
#include <hip/hip_runtime.h>

#include "avl_common.h"

namespace avl {

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

// 8 consecutive activation elements <-> 8 floats
template <typename T>
struct Vec8;

template <>
struct Vec8<bf16> {
    static __device__ __forceinline__ void load(const bf16* p, float (&v)[8]) {
        const bf16x8 x = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (float)x[i];
    }
    static __device__ __forceinline__ void store(bf16* p, const float (&v)[8]) {
        bf16x8 x;
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = (bf16)v[i];   // v_cvt_pk_bf16_f32: round to nearest even
        *reinterpret_cast<bf16x8*>(p) = x;
    }
};

template <>
struct Vec8<f16> {
    static __device__ __forceinline__ void load(const f16* p, float (&v)[8]) {
        const f16x8 x = *reinterpret_cast<const f16x8*>(p);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (float)x[i];
    }
    static __device__ __forceinline__ void store(f16* p, const float (&v)[8]) {
        f16x8 x;
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = (f16)v[i];     // v_cvt_pk_f16_f32 / v_cvt_f16_f32: round to nearest even
        *reinterpret_cast<f16x8*>(p) = x;
    }
};

// the two 16-bit activation types: fragment vector type + the matching 16x16x32 MFMA (same rate for both)
template <typename H>
struct Half16;
template <>
struct Half16<bf16> {
    typedef bf16x8 v8;
    static __device__ __forceinline__ f32x4 mfma(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
    // c + a.lo*b.lo + a.hi*b.hi on packed pairs (v_dot2c_f32_bf16)
    static __device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
        typedef __bf16 v2 __attribute__((ext_vector_type(2)));
        return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2, a), __builtin_bit_cast(v2, b), c, false);
    }
};
template <>
struct Half16<f16> {
    typedef f16x8 v8;
    static __device__ __forceinline__ f32x4 mfma(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
        typedef _Float16 v2 __attribute__((ext_vector_type(2)));
        return __builtin_amdgcn_fdot2(__builtin_bit_cast(v2, a), __builtin_bit_cast(v2, b), c, false);
    }
};

template <>
struct Vec8<float> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[8]) {
        const float4 a = *reinterpret_cast<const float4*>(p);
        const float4 b = *reinterpret_cast<const float4*>(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    static __device__ __forceinline__ void store(float* p, const float (&v)[8]) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
};

template <typename T>
__device__ __forceinline__ float to_f32(T v) { return (float)v; }
template <typename T>
__device__ __forceinline__ T from_f32(float v) { return (T)v; }

// The same DMA issued from inline asm: hipcc then does not know LDS is being written behind its back, so it
// neither drains the queue (vmcnt(0)) in front of every ds_read nor before the next DMA; completion is
// counted by hand (s_waitcnt vmcnt(N) + s_barrier in the caller).  M0 carries the wave-uniform LDS base.
__device__ __forceinline__ void glds16_asm(const void* g, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(g), "s"(lds_byte_addr)
        : "memory");
}

// The same with an SGPR base and a 32-bit VGPR byte offset (the tile / plane base is uniform, the lane keeps only its offset).
__device__ __forceinline__ void glds16_saddr(const void* sbase, unsigned voff, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_byte_addr)
        : "memory");
}

// 4 bytes per lane (64 x 4 B = 256 B per wave-instruction)
__device__ __forceinline__ void glds4_saddr(const void* sbase, unsigned voff, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dword %1, %2\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_byte_addr)
        : "memory");
}

// E8M0 scale byte of an MX-FP4 block whose largest magnitude is `amax` (>= 0): 2^(floor(log2 amax) - 2) puts the block maximum
// into [4, 8), where e2m1 has the two values 4 and 6 and everything above 6 saturates; when the maximum would land above 6.5 the
// scale is doubled instead (the maximum then rounds to 3.5 -> 4 rather than saturating to 6, at the price of a coarser grid for
// the small elements of that block): -8 % rms logits error in tools/precision_study.py ("r3s", both weight seeds tried; 6.0 and
// 7.0 as thresholds: about the same).  network.fp4_quant_blocks is the host twin, bit for bit.
__device__ __forceinline__ unsigned mx_fp4_scale_byte(float amax) {
    const unsigned ab = __float_as_uint(amax), e = ab >> 23;
    return (e >= 3u ? e - 2u : 1u) + ((ab & 0x7fffffu) > 0x500000u ? 1u : 0u);          // mantissa > 1.625 <=> 4 m > 6.5
}

__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) void*)p;
}

inline int elem_size(int dtype) { return dtype == AVL_F32 ? 4 : 2; }
inline bool is_half(int dtype) { return dtype == AVL_BF16 || dtype == AVL_F16; }

// launchers implemented in seg_gemm.hip / seg_conv.hip; each validates its op and returns AVL_*
int launch_gemm(const avl_seg_op& op, hipStream_t s);
int validate_gemm(const avl_seg_op& op);
int launch_conv_op(const avl_seg_op& op, hipStream_t s);
int validate_conv_op(const avl_seg_op& op);
int launch_gconv_mfma(const avl_seg_op& op, hipStream_t s);
int validate_gconv_mfma(const avl_seg_op& op);
int launch_stem_mfma(const avl_seg_op& op, hipStream_t s);
int launch_dwpw(const avl_seg_op& op, hipStream_t s);
int validate_dwpw(const avl_seg_op& op);
int launch_bottleneck(const avl_seg_op& op, hipStream_t s);
int validate_bottleneck(const avl_seg_op& op);

}  // namespace avl
