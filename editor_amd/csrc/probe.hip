// Hardware-semantics probes (test/bring-up only; never on the product path).
//   editor_probe_tr16: what ds_read_b64_tr_b16 returns for a known LDS image.
//   editor_probe_mfma16: lane->element maps of mfma_f32_16x16x32_bf16 for A, B and C/D.
#include "common.h"
#include "../../include/editor_debug.h"

typedef __attribute__((ext_vector_type(4))) short short4_t;
typedef __attribute__((ext_vector_type(8))) short short8_t;
typedef __attribute__((ext_vector_type(4))) float float4_t;

namespace {
// lds[i] = i (u16). Lane l supplies byte address addr[l]; out[l*4+j] = element j it received.
__global__ void probe_tr16_kernel(const int* __restrict__ addr, uint16_t* __restrict__ out)
{
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int a = addr[threadIdx.x];
    short4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) short4_t*)((__attribute__((address_space(3))) char*)lds + a));
#pragma unroll
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (uint16_t)v[j];
}

// D = A * B with A[i][k] (16x32) and B[k][j] (32x16) given as fp32; the kernel packs fragments with the
// ASSUMED map (A: row=l&15, k=(l>>4)*8+e ; B: col=l&15, k=(l>>4)*8+e ; D: col=l&15, row=(l>>4)*4+r).
__global__ void probe_mfma16_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ D)
{
    const int l = threadIdx.x;
    short8_t a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        a[e] = (short)f32_to_bf16(A[(l & 15) * 32 + (l >> 4) * 8 + e]);
        b[e] = (short)f32_to_bf16(B[((l >> 4) * 8 + e) * 16 + (l & 15)]);
    }
    float4_t c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}
// raw 16-bit operand lanes in, fp32 D out: v_mfma_f32_16x16x32_f16 (f16 != 0) or _bf16 - used to pin how the matrix core
// treats SUBNORMAL half inputs (the split-precision forward keeps low-order parts there)
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
__global__ void probe_mfma16_raw_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B, float* __restrict__ D, int f16)
{
    const int l = threadIdx.x;
    short8_t a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        a[e] = (short)A[(l & 15) * 32 + (l >> 4) * 8 + e];
        b[e] = (short)B[((l >> 4) * 8 + e) * 16 + (l & 15)];
    }
    float4_t c = {0.f, 0.f, 0.f, 0.f};
    if (f16) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, a), __builtin_bit_cast(half8_t, b), c, 0, 0, 0);
    else c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}
}  // namespace

extern "C" int editor_probe_tr16(const int* addr, uint16_t* out, hipStream_t stream)
{
    hipLaunchKernelGGL(probe_tr16_kernel, dim3(1), dim3(64), 0, stream, addr, out);
    EDITOR_LAUNCH_CHECK();
    return 0;
}
extern "C" int editor_probe_mfma16(const float* A, const float* B, float* D, hipStream_t stream)
{
    hipLaunchKernelGGL(probe_mfma16_kernel, dim3(1), dim3(64), 0, stream, A, B, D);
    EDITOR_LAUNCH_CHECK();
    return 0;
}
extern "C" int editor_probe_mfma16_raw(const uint16_t* A, const uint16_t* B, float* D, int f16, hipStream_t stream)
{
    hipLaunchKernelGGL(probe_mfma16_raw_kernel, dim3(1), dim3(64), 0, stream, A, B, D, f16);
    EDITOR_LAUNCH_CHECK();
    return 0;
}
