// Parity-mode (fp32) multi-head attention for the EDITOR hot path, gfx950.
// Restates Attention.forward (vit_pytorch.py:184-198) and AttentionMask.forward (:240-258) on packed
// qkv rows: S = scale * q k^T (exact-f32 MFMA batched GEMM) -> masked softmax -> P v, and the matching
// backward.  The probabilities ARE an output here (the backbone returns them, vit_pytorch.py:638-644).
// The performance path is attention_bf16.hip; this file exists so that end-to-end index parity can be
// checked with fp32 arithmetic (SURVEY.md 7 "Bit-identical indices vs bf16").
#include "common.h"
#include "../../include/editor_hip.h"

namespace {

// One wave per (b, h, q) row of S [B,h,T,T].  mask (B,T) uint8 or NULL:
//   s = (mask[b,q] && mask[b,key]) ? s : -65504 ; p = softmax(s) * mask[b,q]      (vit_pytorch.py:250-253)
template <int NI>            // 64*NI >= T keys per row, NI values per lane in registers
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ S, long rows, int T, int heads,
                                                           const uint8_t* __restrict__ mask)
{
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int q = (int)(row % T);
    const long b = row / ((long)T * heads);
    float* s = S + row * T;
    const uint8_t* mk = mask ? mask + b * T : nullptr;
    const bool qkeep = !mk || mk[q];
    float v[NI];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int k = i * 64 + lane;
        if (k < T) {
            float x = s[k];
            if (mk && !(qkeep && mk[k])) x = -65504.f;
            v[i] = x;
            mx = fmaxf(mx, x);
        }
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int k = i * 64 + lane;
        if (k < T) { v[i] = expf(v[i] - mx); sum += v[i]; }
    }
    sum = wave_sum(sum);
    const float inv = qkeep ? 1.f / sum : 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int k = i * 64 + lane;
        if (k < T) s[k] = v[i] * inv;
    }
}

// dS = P * (dP - rowsum(dP * P)), in place on dP
template <int NI>
__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const float* __restrict__ P, float* __restrict__ dP,
                                                               long rows, int T)
{
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* p = P + row * T;
    float* d = dP + row * T;
    float pv[NI], dv[NI];
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int k = i * 64 + lane;
        if (k < T) { pv[i] = p[k]; dv[i] = d[k]; dot += pv[i] * dv[i]; }
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int k = i * 64 + lane;
        if (k < T) d[k] = pv[i] * (dv[i] - dot);
    }
}

}  // namespace

extern "C" int editor_attention_fwd_f32(const float* qkv, int B, int T, int heads, int hd, float scale,
                                        const uint8_t* mask, float* out, float* probs, hipStream_t stream)
{
    if (!probs || T > 4096) return (int)hipErrorInvalidValue;
    const int D = heads * hd;
    const long TT = (long)T * T;
    // S[b,h] = scale * Q K^T       Q rows at qkv + h*hd, K rows at qkv + D + h*hd, row stride 3D
    int rc = editor_gemm_f32(qkv, qkv + D, probs, T, T, hd, 3L * D, 3L * D, T, 0, 0,
                             B, (long)T * 3 * D, (long)T * 3 * D, heads * TT, heads, hd, hd, TT,
                             scale, 0.f, nullptr, nullptr, 1, EDITOR_EPI_NONE, nullptr, 0, stream);
    if (rc) return rc;
    const long rows = (long)B * heads * T;
    const dim3 sgrid((unsigned)((rows + 3) / 4));
    if (T <= 1024) hipLaunchKernelGGL(softmax_rows_kernel<16>, sgrid, dim3(256), 0, stream, probs, rows, T, heads, mask);
    else if (T <= 2304) hipLaunchKernelGGL(softmax_rows_kernel<36>, sgrid, dim3(256), 0, stream, probs, rows, T, heads, mask);
    else hipLaunchKernelGGL(softmax_rows_kernel<64>, sgrid, dim3(256), 0, stream, probs, rows, T, heads, mask);
    EDITOR_LAUNCH_CHECK();
    // O[b,:,h] = P V               V stored [key][hd] -> transB = 1
    return editor_gemm_f32(probs, qkv + 2 * D, out, T, hd, T, T, 3L * D, D, 0, 1,
                           B, heads * TT, (long)T * 3 * D, (long)T * D, heads, TT, hd, hd,
                           1.f, 0.f, nullptr, nullptr, 1, EDITOR_EPI_NONE, nullptr, 0, stream);
}

extern "C" int editor_attention_bwd_f32(const float* qkv, const float* dout, const float* probs, int B, int T, int heads,
                                        int hd, float scale, float* dqkv, float* workspace, hipStream_t stream)
{
    const int D = heads * hd;
    const long TT = (long)T * T;
    const long sq = (long)T * 3 * D;
    // dV = P^T dO
    int rc = editor_gemm_f32(probs, dout, dqkv + 2 * D, T, hd, T, T, D, 3L * D, 1, 1,
                             B, heads * TT, (long)T * D, sq, heads, TT, hd, hd, 1.f, 0.f, nullptr, nullptr, 1, EDITOR_EPI_NONE, nullptr, 0, stream);
    if (rc) return rc;
    // dP = dO V^T   (V stored [key][hd] = [N][K] -> transB = 0)
    rc = editor_gemm_f32(dout, qkv + 2 * D, workspace, T, T, hd, D, 3L * D, T, 0, 0,
                         B, (long)T * D, sq, heads * TT, heads, hd, hd, TT, 1.f, 0.f, nullptr, nullptr, 1, EDITOR_EPI_NONE, nullptr, 0, stream);
    if (rc) return rc;
    const long rows = (long)B * heads * T;
    if (T > 4096) return (int)hipErrorInvalidValue;
    const dim3 sgrid((unsigned)((rows + 3) / 4));
    if (T <= 1024) hipLaunchKernelGGL(softmax_bwd_rows_kernel<16>, sgrid, dim3(256), 0, stream, probs, workspace, rows, T);
    else if (T <= 2304) hipLaunchKernelGGL(softmax_bwd_rows_kernel<36>, sgrid, dim3(256), 0, stream, probs, workspace, rows, T);
    else hipLaunchKernelGGL(softmax_bwd_rows_kernel<64>, sgrid, dim3(256), 0, stream, probs, workspace, rows, T);
    EDITOR_LAUNCH_CHECK();
    // dQ = scale * dS K     (K stored [key][hd] = [K][N] -> transB = 1)
    rc = editor_gemm_f32(workspace, qkv + D, dqkv, T, hd, T, T, 3L * D, 3L * D, 0, 1,
                         B, heads * TT, sq, sq, heads, TT, hd, hd, scale, 0.f, nullptr, nullptr, 1, EDITOR_EPI_NONE, nullptr, 0, stream);
    if (rc) return rc;
    // dK = scale * dS^T Q
    return editor_gemm_f32(workspace, qkv, dqkv + D, T, hd, T, T, 3L * D, 3L * D, 1, 1,
                           B, heads * TT, sq, sq, heads, TT, hd, hd, scale, 0.f, nullptr, nullptr, 1, EDITOR_EPI_NONE, nullptr, 0, stream);
}
