// Shared device helpers for the EDITOR hot-path kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define EDITOR_WAVE 64

typedef uint16_t bf16_t;   // raw bfloat16 bits; arithmetic is always done in fp32

__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
    return __uint_as_float(((uint32_t)v) << 16);
}
// round-to-nearest-even, NaN stays NaN (matches torch's float->bfloat16 cast): gfx950's v_cvt_pk_bf16_f32, one
// instruction per PAIR of values (the shift/add software form costs ~9 VALU instructions per value and made the GEMM
// epilogues VALU-bound)
typedef float float2_t_ __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const float2_t_ v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t_));
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return (bf16_t)(pack_bf16x2(f, 0.f) & 0xffffu); }

// IEEE half activations (cfg.MODEL.COMPUTE_DTYPE = 'f16'): the reference's own GPU arithmetic (torch.cuda.amp autocast,
// engine/processor.py:79) - same MFMA rate as bf16, 3 more mantissa bits.  A distinct C++ type so templates dispatch on it;
// the wire format is the raw 16 bits.  Dtype CODES of the C ABI's `*_bf16` / dtype arguments: 0 = fp32, 1 = bf16, 2 = f16.
struct f16_t { uint16_t v; };
typedef _Float16 half2_t_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_f16x2(float lo, float hi) {
    const float2_t_ v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, half2_t_));      // round-to-nearest-even
}
__device__ __forceinline__ float f16_to_f32(uint16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
__device__ __forceinline__ uint16_t f32_to_f16(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }

// 16-bit operand format of a kernel instantiation: F16 = false -> bfloat16, true -> IEEE half
template <bool F16> struct H16 {
    static __device__ __forceinline__ uint32_t pack2(float lo, float hi) { return F16 ? pack_f16x2(lo, hi) : pack_bf16x2(lo, hi); }
    static __device__ __forceinline__ float2_t_ unpack2(uint32_t w) {
        if (F16) { const half2_t_ h = __builtin_bit_cast(half2_t_, w); return __builtin_convertvector(h, float2_t_); }
        return float2_t_{__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)};
    }
    static __device__ __forceinline__ float to_f32(uint16_t v) { return F16 ? f16_to_f32(v) : bf16_to_f32(v); }
    static __device__ __forceinline__ uint16_t from_f32(float f) { return F16 ? f32_to_f16(f) : f32_to_bf16(f); }
};

template <typename T> struct Elem;
template <> struct Elem<float> {
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
    static __device__ __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
    static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};
template <> struct Elem<f16_t> {
    static __device__ __forceinline__ float ld(const f16_t* p) { return f16_to_f32(p->v); }
    static __device__ __forceinline__ void st(f16_t* p, float v) { p->v = f32_to_f16(v); }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// block-wide sum for blocks of up to 1024 threads; `red` = >= 16 floats of LDS. All threads get it.
__device__ __forceinline__ float block_sum(float v, float* red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}

#define EDITOR_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)
