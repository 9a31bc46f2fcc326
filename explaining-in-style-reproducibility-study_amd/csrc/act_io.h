// act_io.h — typed activation loads/stores: activations are fp32 (parity mode, STYLEX_F32 / STYLEX_BF16)
// or bf16 (STYLEX_BF16_ACT: halves the HBM traffic of every bandwidth-bound kernel).
#pragma once
#include <hip/hip_runtime.h>

typedef __bf16 act_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float act_f32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned act_pack2(float lo, float hi) {
    act_f32x2_t v = {lo, hi};
    act_bf16x2_t r = __builtin_convertvector(v, act_bf16x2_t);  // v_cvt_pk_bf16_f32 (RNE)
    return *reinterpret_cast<unsigned*>(&r);
}
__device__ __forceinline__ float act_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float act_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ float4 act_unpack4(uint2 h) { return make_float4(act_lo(h.x), act_hi(h.x), act_lo(h.y), act_hi(h.y)); }
__device__ __forceinline__ uint2 act_pack4(float4 v) { return make_uint2(act_pack2(v.x, v.y), act_pack2(v.z, v.w)); }

// 4 consecutive channels at element offset `off` of an activation tensor
template <bool BF>
__device__ __forceinline__ float4 act_ld4(const void* base, long off) {
    if (BF) return act_unpack4(*reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + off));
    return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + off);
}
template <bool BF>
__device__ __forceinline__ void act_st4(void* base, long off, float4 v) {
    if (BF) *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(base) + off) = act_pack4(v);
    else *reinterpret_cast<float4*>(reinterpret_cast<float*>(base) + off) = v;
}
template <bool BF>
__device__ __forceinline__ float act_ld1(const void* base, long off) {
    if (BF) return __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(base)[off] << 16);
    return reinterpret_cast<const float*>(base)[off];
}
template <bool BF>
__device__ __forceinline__ void act_st1(void* base, long off, float v) {
    if (BF) reinterpret_cast<unsigned short*>(base)[off] = (unsigned short)(act_pack2(v, 0.f) & 0xffffu);
    else reinterpret_cast<float*>(base)[off] = v;
}
// runtime-flag forms
__device__ __forceinline__ float4 act_ld4(const void* base, long off, int bf) { return bf ? act_ld4<true>(base, off) : act_ld4<false>(base, off); }
__device__ __forceinline__ void act_st4(void* base, long off, float4 v, int bf) { if (bf) act_st4<true>(base, off, v); else act_st4<false>(base, off, v); }
__device__ __forceinline__ float act_ld1(const void* base, long off, int bf) { return bf ? act_ld1<true>(base, off) : act_ld1<false>(base, off); }
__device__ __forceinline__ void act_st1(void* base, long off, float v, int bf) { if (bf) act_st1<true>(base, off, v); else act_st1<false>(base, off, v); }
