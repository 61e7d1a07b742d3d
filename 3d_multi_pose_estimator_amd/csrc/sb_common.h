// Shared pieces of the split-bf16 arithmetic (gemm_sb16.hip: the batch kernels; lat.hip: the small-batch latency kernels): the split
// of eight fp32 values into their three bf16 planes and the six products of one 32-deep K stage in the canonical order.  One
// definition, so that a row has the same bits whichever kernel computed it.
#pragma once
#include <hip/hip_runtime.h>

namespace mpe {
namespace sb {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

__device__ __forceinline__ unsigned pack2_bf16(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
    bf2 v = {(__bf16)lo, (__bf16)hi};                    // v_cvt_pk_bf16_f32: round to nearest even
    return __builtin_bit_cast(unsigned, v);
}

// x - y as ONE v_sub_f32: left to itself the compiler pairs the residual subtractions of split8 into v_pk_add_f32, and a packed
// fp32 instruction in the shadow of MFMAs costs 13-22 cycles more than the two plain ones it replaces (MI355X_MICROARCH.md, cycle
// constants: "an anti-lever beside MFMAs").  Not volatile: the scheduler still places it.
__device__ __forceinline__ float sub1(float x, float y) {
#ifdef SB_NO_SUB1
    return x - y;
#else
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
#endif
}

// eight fp32 values -> their three bf16 planes (exact: x = p0 + p1 + p2)
__device__ __forceinline__ void split8(const f32x4 &x0, const f32x4 &x1, bf16x8 &p0, bf16x8 &p1, bf16x8 &p2) {
    u32x4 q0, q1, q2;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a = j < 2 ? x0[2 * j] : x1[2 * j - 4], b = j < 2 ? x0[2 * j + 1] : x1[2 * j - 3];
        const unsigned u0 = pack2_bf16(a, b);
        const float ra = sub1(a, __uint_as_float(u0 << 16)), rb = sub1(b, __uint_as_float(u0 & 0xFFFF0000u));
        const unsigned u1 = pack2_bf16(ra, rb);
        const float sa = sub1(ra, __uint_as_float(u1 << 16)), sb = sub1(rb, __uint_as_float(u1 & 0xFFFF0000u));
        q0[j] = u0;
        q1[j] = u1;
        q2[j] = pack2_bf16(sa, sb);
    }
    p0 = __builtin_bit_cast(bf16x8, q0);
    p1 = __builtin_bit_cast(bf16x8, q1);
    p2 = __builtin_bit_cast(bf16x8, q2);
}

// the six products of one stage, canonical order; a[p], w[p] = plane p (0 = most significant)
#define SB_STAGE(ACC, A, W)                                                            \
    do {                                                                               \
        ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((W)[2], (A)[0], ACC, 0, 0, 0);   \
        ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((W)[1], (A)[1], ACC, 0, 0, 0);   \
        ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((W)[1], (A)[0], ACC, 0, 0, 0);   \
        ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((W)[0], (A)[2], ACC, 0, 0, 0);   \
        ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((W)[0], (A)[1], ACC, 0, 0, 0);   \
        ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((W)[0], (A)[0], ACC, 0, 0, 0);   \
    } while (0)

// the same with the chain started from zero (C operand = the inline constant 0: no accumulator to clear after a flush)
#define SB_STAGE0(ACC, A, W)                                                                                       \
    do {                                                                                                           \
        ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((W)[2], (A)[0], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);     \
        ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((W)[1], (A)[1], ACC, 0, 0, 0);                              \
        ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((W)[1], (A)[0], ACC, 0, 0, 0);                              \
        ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((W)[0], (A)[2], ACC, 0, 0, 0);                              \
        ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((W)[0], (A)[1], ACC, 0, 0, 0);                              \
        ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((W)[0], (A)[0], ACC, 0, 0, 0);                              \
    } while (0)

// v_cvt_pk_bf16_f32 by name.  pack2_bf16 (sb_common.h) leaves the choice to the compiler, which in these kernels converts the two
// values one by one and joins them with a v_perm_b32: 77 instead of 44 vector instructions per split, in kernels whose only
// arithmetic cost is the split.  Same instruction, same rounding (nearest even), same bits.
__device__ __forceinline__ unsigned pack2_lat(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

// split8 of sb_common.h with that conversion
__device__ __forceinline__ void split8_lat(const f32x4 &x0, const f32x4 &x1, bf16x8 &p0, bf16x8 &p1, bf16x8 &p2) {
    u32x4 q0, q1, q2;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a = j < 2 ? x0[2 * j] : x1[2 * j - 4], b = j < 2 ? x0[2 * j + 1] : x1[2 * j - 3];
        const unsigned u0 = pack2_lat(a, b);
        const float ra = sub1(a, __uint_as_float(u0 << 16)), rb = sub1(b, __uint_as_float(u0 & 0xFFFF0000u));
        const unsigned u1 = pack2_lat(ra, rb);
        const float sa = sub1(ra, __uint_as_float(u1 << 16)), sb_ = sub1(rb, __uint_as_float(u1 & 0xFFFF0000u));
        q0[j] = u0;
        q1[j] = u1;
        q2[j] = pack2_lat(sa, sb_);
    }
    p0 = __builtin_bit_cast(bf16x8, q0);
    p1 = __builtin_bit_cast(bf16x8, q1);
    p2 = __builtin_bit_cast(bf16x8, q2);
}

// the three bf16 planes of four values, as split8 makes them -> three 8-byte stores
__device__ __forceinline__ void store_planes4(unsigned short *dst, size_t plane, const f32x4 &v) {
    const unsigned a0 = pack2_lat(v[0], v[1]), a1 = pack2_lat(v[2], v[3]);
    const float r0 = sub1(v[0], __uint_as_float(a0 << 16)), r1 = sub1(v[1], __uint_as_float(a0 & 0xFFFF0000u));
    const float r2 = sub1(v[2], __uint_as_float(a1 << 16)), r3 = sub1(v[3], __uint_as_float(a1 & 0xFFFF0000u));
    const unsigned b0 = pack2_lat(r0, r1), b1 = pack2_lat(r2, r3);
    const float s0 = sub1(r0, __uint_as_float(b0 << 16)), s1 = sub1(r1, __uint_as_float(b0 & 0xFFFF0000u));
    const float s2 = sub1(r2, __uint_as_float(b1 << 16)), s3 = sub1(r3, __uint_as_float(b1 & 0xFFFF0000u));
    *reinterpret_cast<uint2 *>(dst) = make_uint2(a0, a1);
    *reinterpret_cast<uint2 *>(dst + plane) = make_uint2(b0, b1);
    *reinterpret_cast<uint2 *>(dst + 2 * plane) = make_uint2(pack2_lat(s0, s1), pack2_lat(s2, s3));
}

// the three planes of ONE value (plane stride in elements)
__device__ __forceinline__ void store_planes1(unsigned short *dst, size_t plane, float v) {
    const unsigned a0 = pack2_lat(v, 0.f);
    const float r0 = sub1(v, __uint_as_float(a0 << 16));
    const unsigned b0 = pack2_lat(r0, 0.f);
    const float s0 = sub1(r0, __uint_as_float(b0 << 16));
    dst[0] = (unsigned short)(a0 & 0xFFFFu);
    dst[plane] = (unsigned short)(b0 & 0xFFFFu);
    dst[2 * plane] = (unsigned short)(pack2_lat(s0, 0.f) & 0xFFFFu);
}

}  // namespace sb
}  // namespace mpe
