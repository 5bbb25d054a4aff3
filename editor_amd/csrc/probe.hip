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
// (s_memtime, s_memrealtime) pairs sampled by one lane every ~sleep * 64 * 127 shader cycles: the first counts the shader
// clock, the second a constant 100 MHz reference - their ratio is the frequency the CUs actually run at while another
// stream keeps the matrix cores busy (tools/clock_probe.py)
__global__ void clock_trace_kernel(unsigned long long* out, int n, int sleep)
{
    if (threadIdx.x != 0) return;
    for (int i = 0; i < n; ++i) {
        out[2 * i] = __builtin_amdgcn_s_memtime();
        out[2 * i + 1] = __builtin_amdgcn_s_memrealtime();
        for (int j = 0; j < sleep; ++j) __builtin_amdgcn_s_sleep(127);
    }
}

// nothing but matrix-core work: every wave issues `iters` x 8 independent v_mfma_f32_16x16x32_bf16 (register operands, no
// memory traffic) - the rate the chip SUSTAINS on dense bf16 MFMA under its power management, to hold against the 2.5 PFLOP/s
// that MI355X_MICROARCH.md quotes at the 2.4 GHz boost clock.  zero != 0: all-zero operands (the data-dependent part of the
// matrix core's power draw switched off).
typedef __attribute__((ext_vector_type(8))) short pshort8_t;
typedef __attribute__((ext_vector_type(4))) float pfloat4_t;
__global__ __launch_bounds__(256) void mfma_peak_kernel(float* out, int iters, int zero)
{
    const int l = threadIdx.x;
    pshort8_t a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        a[e] = zero ? (short)0 : (short)(0x3f80 + ((l * 7 + e * 13) & 0x3f));          // bf16 values in [1, 1.5)
        b[e] = zero ? (short)0 : (short)(0xbf80 + ((l * 5 + e * 11) & 0x3f));          // and in (-1.5, -1]
    }
    pfloat4_t c[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) c[i] = pfloat4_t{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)                 // (in place, as written: the builtin form gets its accumulators rotated through AGPR moves)
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c[i]) : "v"(a), "v"(b));
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    if (s == 12345.678f) out[0] = s;                               // (keeps the loop alive)
}

}  // namespace

// grid x 4 wavefronts; FLOPs = grid * 4 * iters * 8 * 16384
// ---- pricing experiment ("price the bf16-branch residual with a measurement"): x_out = x + rowscale * branch
// (branch: the bf16 output of a projection / fc2 product with the PLAIN 16-bit epilogue), y = LayerNorm(x_out) in bf16 - the
// residual add moved out of the GEMM's fp32 epilogue into the LayerNorm that follows.  D = 768, wave per row, same access
// pattern as layernorm_fwd_kernel.  Measurement only (tools/residual_pricing.py): the product path keeps the fp32 epilogue.
namespace {
__global__ __launch_bounds__(256) void resid_add_ln_kernel(const float* __restrict__ x, const uint16_t* __restrict__ branch,
    const float* __restrict__ rowscale, const float* __restrict__ gamma, const float* __restrict__ beta, float eps, long M,
    float* __restrict__ x_out, uint16_t* __restrict__ y, float* __restrict__ mean_out, float* __restrict__ rstd_out)
{
    constexpr int D = 768;
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float rs = rowscale ? rowscale[row] : 1.f;
    float v[12];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = i * 256 + lane * 4;
        const float4 a = *reinterpret_cast<const float4*>(x + row * D + c);
        const uint2 b = *reinterpret_cast<const uint2*>(branch + row * D + c);
        v[4 * i + 0] = a.x + rs * bf16_to_f32((bf16_t)(b.x & 0xffffu));
        v[4 * i + 1] = a.y + rs * bf16_to_f32((bf16_t)(b.x >> 16));
        v[4 * i + 2] = a.z + rs * bf16_to_f32((bf16_t)(b.y & 0xffffu));
        v[4 * i + 3] = a.w + rs * bf16_to_f32((bf16_t)(b.y >> 16));
        *reinterpret_cast<float4*>(x_out + row * D + c) = make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
        s += (v[4 * i] + v[4 * i + 1]) + (v[4 * i + 2] + v[4 * i + 3]);
    }
    const float mean = wave_sum(s) * (1.f / D);
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 12; ++e) { const float d = v[e] - mean; q += d * d; }
    const float rstd = rsqrtf(wave_sum(q) * (1.f / D) + eps);
    if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = i * 256 + lane * 4;
        const float4 g = *reinterpret_cast<const float4*>(gamma + c), bb = *reinterpret_cast<const float4*>(beta + c);
        uint2 o;
        o.x = pack_bf16x2((v[4 * i] - mean) * rstd * g.x + bb.x, (v[4 * i + 1] - mean) * rstd * g.y + bb.y);
        o.y = pack_bf16x2((v[4 * i + 2] - mean) * rstd * g.z + bb.z, (v[4 * i + 3] - mean) * rstd * g.w + bb.w);
        *reinterpret_cast<uint2*>(y + row * D + c) = o;
    }
}
}  // namespace

extern "C" int editor_probe_resid_add_layernorm(const float* x, const uint16_t* branch, const float* rowscale, const float* gamma,
    const float* beta, float eps, long M, int D, float* x_out, uint16_t* y, float* mean, float* rstd, hipStream_t stream)
{
    if (D != 768 || M < 1) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(resid_add_ln_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, stream, x, branch, rowscale, gamma, beta, eps,
                       M, x_out, y, mean, rstd);
    return (int)hipGetLastError();
}

extern "C" int editor_probe_mfma_peak(float* out, int grid, int iters, int zero, hipStream_t stream)
{
    hipLaunchKernelGGL(mfma_peak_kernel, dim3(grid), dim3(256), 0, stream, out, iters, zero);
    EDITOR_LAUNCH_CHECK();
    return 0;
}

extern "C" int editor_probe_clock_trace(unsigned long long* out, int n, int sleep, hipStream_t stream)
{
    hipLaunchKernelGGL(clock_trace_kernel, dim3(1), dim3(64), 0, stream, out, n, sleep);
    EDITOR_LAUNCH_CHECK();
    return 0;
}

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
