// Shared device helpers of the fused attention kernels (attention_bf16.hip, attention_split.hip): LDS images of HD-wide head
// slices, MFMA fragment loads in the k-major and transposed (ds_read_b64_tr_b16) forms, accumulator packing.  gfx950 only.
#pragma once
#include "common.h"

typedef __attribute__((ext_vector_type(4))) short short4_t;
typedef __attribute__((ext_vector_type(8))) short short8_t;
typedef __attribute__((ext_vector_type(4))) float float4_t;

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));

namespace {

// F16 = false: bfloat16 q/k/v/P/dS operands; true: IEEE half (the reference's autocast dtype) - same instruction rate
template <bool F16>
__device__ __forceinline__ float4_t mfma16(short8_t a, short8_t b, float4_t c)
{
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, a), __builtin_bit_cast(half8_t, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// Head width.  The kernels are written for 64-wide heads (ViT-B/L, DeiT-B: every shipped configuration); the same source is
// compiled again with -DATTN_HD=32 and -DATTN_HD=96 for the factory's other widths (DeiT-small's HMA heads, ViT-small's backbone:
// vit_pytorch.py:704-727) - KS k-steps of 32 and ND output tiles of 16 columns across the head instead of 2 and 4, image rows of
// CPR 16-byte chunks.  With ATTN_HD == 64 every expression below reduces to the original one.
#ifndef ATTN_HD
#define ATTN_HD 64
#endif
static_assert(ATTN_HD == 32 || ATTN_HD == 64 || ATTN_HD == 96, "head widths built: 32, 64, 96");
constexpr int HD = ATTN_HD;            // head dim
constexpr int KS = HD / 32;            // 32-deep MFMA k-steps across the head dim
constexpr int ND = HD / 16;            // 16-column output tiles across the head dim
constexpr int CPR = HD / 8;            // 16-byte chunks per LDS image row
constexpr int ROWB = HD * 2;           // bytes per LDS image row
constexpr int SWZ = (CPR % 8 == 0) ? 7 : 3;   // XOR swizzle mask of the chunk index (stays inside an aligned group of SWZ + 1 chunks)
constexpr float kLog2e = 1.4426950408889634f;

// LDS image of (rows x HD) bf16: 16-byte chunk c of row r lives at r*ROWB + ((c ^ (r & SWZ)) << 4)  (HD = 64: r*128, r & 7)
__device__ __forceinline__ int img_off(int row, int chunk) { return row * ROWB + ((chunk ^ (row & SWZ)) << 4); }

// cooperative load of rows [0,T) of one head slice (64 columns starting at `base` of a row-major matrix with leading
// dimension ld) into an LDS image of Tp rows (Tp a multiple of 8), as LDS-DMA (global_load_lds_dwordx4): no staging
// registers and every piece in flight at once - the register-staged loop it replaces (load, wait, ds_write per 16 bytes,
// seven times per thread and image) serialised HBM latencies in front of every workgroup's compute.  The DMA writes
// lane l's 16 bytes at piece_base + 16 l, so the XOR swizzle is applied to the SOURCE chunk index (as in the GEMMs).
// Rows >= T cannot be zero-filled by a DMA: they are loaded from row T-1 (finite data) - every consumer masks keys /
// queries beyond the sequence end (scores -> -inf, lse -> +inf), so those rows only ever meet exact zeros.
__device__ __forceinline__ void load_image(char* img, const bf16_t* __restrict__ base, long ld, int T, int Tp)
{
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    if constexpr (HD == 64) {
        const int r8 = lane >> 3, p = lane & 7;
        for (int piece = w; piece < (Tp >> 3); piece += nw) {          // 8 rows x 128 B = 1 KiB per piece
            const int row = piece * 8 + r8;
            const int c = p ^ (row & 7);
            const bf16_t* src = base + (long)min(row, T - 1) * ld + c * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(img + piece * 1024), 16, 0, 0);
        }
    } else {
        // other head widths: a 1 KiB piece is 64 consecutive chunk SLOTS of the flat image (16 rows of 4, or 5 1/3 rows of 12);
        // slot p of row r receives source chunk p ^ (r & SWZ).  Tp is a multiple of 16, so Tp * CPR / 64 is whole.
        for (int piece = w; piece < Tp * CPR / 64; piece += nw) {
            const int ci = piece * 64 + lane;
            const int row = ci / CPR, p = ci - row * CPR;
            const int c = p ^ (row & SWZ);
            const bf16_t* src = base + (long)min(row, T - 1) * ld + c * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(img + piece * 1024), 16, 0, 0);
        }
    }
}
// the DMA counts on vmcnt: drain it before the barrier that publishes the images
__device__ __forceinline__ void images_ready()
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// k-major fragment: lane (i = l&15, g = l>>4) <- image[row0 + i][s*32 + g*8 .. +8]
__device__ __forceinline__ short8_t frag_k(const char* img, int row0, int s, int lane)
{
    return *reinterpret_cast<const short8_t*>(img + img_off(row0 + (lane & 15), s * 4 + (lane >> 4)));
}

// transposed fragment for the reduction over image ROWS: lane (i, g) <- image[row(kappa)][dt*16 + i] with the
// permuted reduction index  kappa = 8g+e  <->  row = 32*s2 + 16*(e>>2) + 4g + (e&3)   (matches the packed
// accumulator operand).  Two ds_read_b64_tr_b16: each 16-lane group presents a [4 rows][16 cols] block.
__device__ __forceinline__ short8_t frag_t(const char* img, int s2, int dt, int lane)
{
    const int i = lane & 15, g = lane >> 4;
    const int r0 = 32 * s2 + 4 * g + (i >> 2), r1 = r0 + 16;
    const int d0 = dt * 16 + (i & 3) * 4;
    const int c = d0 >> 3, sub = (d0 & 4) << 1;
    typedef __attribute__((address_space(3))) short4_t* lds_p;
    const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(img + img_off(r0, c) + sub));
    const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(img + img_off(r1, c) + sub));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// own-side fragment straight from global: lane (i, g) <- M[row0 + i][s*32 + g*8 .. +8] (zeros beyond T)
__device__ __forceinline__ short8_t frag_own(const bf16_t* __restrict__ base, long ld, int row0, int T, int s, int lane)
{
    const int row = row0 + (lane & 15);
    if (row >= T) return short8_t{0, 0, 0, 0, 0, 0, 0, 0};
    return *reinterpret_cast<const short8_t*>(base + (long)row * ld + s * 32 + (lane >> 4) * 8);
}

template <bool F16>
__device__ __forceinline__ uint2 pack4(float a, float b, float c, float d)
{
    uint2 u; u.x = H16<F16>::pack2(a, b); u.y = H16<F16>::pack2(c, d); return u;
}
__device__ __forceinline__ short8_t join(uint2 lo, uint2 hi)
{
    union { uint32_t u[4]; short8_t s; } x;
    x.u[0] = lo.x; x.u[1] = lo.y; x.u[2] = hi.x; x.u[3] = hi.y;
    return x.s;
}

__device__ __forceinline__ float group_max(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); return fmaxf(v, __shfl_xor(v, 32, 64)); }
__device__ __forceinline__ float group_sum(float v) { v += __shfl_xor(v, 16, 64); return v + __shfl_xor(v, 32, 64); }

}  // namespace
