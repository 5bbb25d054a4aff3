// Small head kernels of the EDITOR hot path, gfx950.  All are tiny (B x 768 / B x 2304)
// and latency bound; they exist so that no stage of the path runs outside libeditor_hip.so.
//   BatchNorm1d fwd/bwd          FUSE_BN / BACKBONE_BN / AL_BN            (make_model.py:115,120,140)
//   OCFR centre update + loss    OCFR.forward/update/compute_intra_loss   (OCFR.py:22-84)
#include "common.h"
#include "../../include/editor_hip.h"

namespace {

// one thread per feature column, loop over the batch (coalesced across threads)
__global__ void bn1d_fwd_kernel(const float* __restrict__ x, long ldx, int B, int C, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ rmean, float* __restrict__ rvar, float momentum, float eps,
    int training, float* __restrict__ y, float* __restrict__ save_mean, float* __restrict__ save_invstd)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float mean, invstd;
    if (training) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += x[b * ldx + c];
        mean = s / (float)B;
        float q = 0.f;
        for (int b = 0; b < B; ++b) { const float d = x[b * ldx + c] - mean; q += d * d; }
        const float var = q / (float)B;                               // biased: normalisation
        invstd = rsqrtf(var + eps);
        rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;     // running stats use the unbiased variance
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * (B > 1 ? q / (float)(B - 1) : var);
        save_mean[c] = mean;
        save_invstd[c] = invstd;
    } else {
        mean = rmean[c];
        invstd = rsqrtf(rvar[c] + eps);
    }
    const float g = gamma[c] * invstd, bt = beta[c];
    for (int b = 0; b < B; ++b) y[(long)b * C + c] = (x[b * ldx + c] - mean) * g + bt;
}

__global__ void bn1d_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, long ldx, int B, int C,
    const float* __restrict__ gamma, const float* __restrict__ save_mean, const float* __restrict__ save_invstd,
    float* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float mean = save_mean[c], invstd = save_invstd[c];
    float sdy = 0.f, sdyx = 0.f;
    for (int b = 0; b < B; ++b) {
        const float d = dy[(long)b * C + c];
        sdy += d;
        sdyx += d * (x[b * ldx + c] - mean) * invstd;
    }
    dgamma[c] = sdyx;
    dbeta[c] = sdy;
    const float k = gamma[c] * invstd / (float)B;
    for (int b = 0; b < B; ++b) {
        const float xh = (x[b * ldx + c] - mean) * invstd;
        dx[(long)b * C + c] = k * ((float)B * dy[(long)b * C + c] - sdy - xh * sdyx);
    }
}

// The same two operators with the column held in REGISTERS: block = 64 columns x 4 row groups, thread (col, g) owns samples
// b = g, g + 4, ... (<= R of them, all loads issued before the first use), statistics folded through LDS in row-group order.
// The one-thread-per-column form above walks the batch three times with one dependent load per trip: 60 us per call for
// 0.3 MB, four calls per step.  Same formulas (two-pass variance); only the summation order of the batch sums differs.
template <int R>
__global__ __launch_bounds__(256) void bn1d_fwd_reg_kernel(const float* __restrict__ x, long ldx, int B, int C, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ rmean, float* __restrict__ rvar, float momentum, float eps,
    float* __restrict__ y, float* __restrict__ save_mean, float* __restrict__ save_invstd)
{
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const bool live = c < C;
    float v[R];
#pragma unroll
    for (int j = 0; j < R; ++j) { const int b = g + 4 * j; v[j] = (live && b < B) ? x[b * ldx + c] : 0.f; }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < R; ++j) s += v[j];
    red[g][lane] = s;
    __syncthreads();
    const float mean = (((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane]) / (float)B;
    __syncthreads();
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < R; ++j) { const float d = v[j] - mean; q += (g + 4 * j < B) ? d * d : 0.f; }
    red[g][lane] = q;
    __syncthreads();
    const float qs = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
    const float var = qs / (float)B, invstd = rsqrtf(var + eps);
    if (!live) return;
    if (g == 0) {
        rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * (B > 1 ? qs / (float)(B - 1) : var);
        save_mean[c] = mean;
        save_invstd[c] = invstd;
    }
    const float gm = gamma[c] * invstd, bt = beta[c];
#pragma unroll
    for (int j = 0; j < R; ++j) { const int b = g + 4 * j; if (b < B) y[(long)b * C + c] = (v[j] - mean) * gm + bt; }
}

template <int R>
__global__ __launch_bounds__(256) void bn1d_bwd_reg_kernel(const float* __restrict__ dy, const float* __restrict__ x, long ldx, int B, int C,
    const float* __restrict__ gamma, const float* __restrict__ save_mean, const float* __restrict__ save_invstd,
    float* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta)
{
    __shared__ float red[2][4][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const bool live = c < C;
    const float mean = live ? save_mean[c] : 0.f, invstd = live ? save_invstd[c] : 0.f;
    float d[R], xh[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int b = g + 4 * j;
        const bool ok = live && b < B;
        d[j] = ok ? dy[(long)b * C + c] : 0.f;
        xh[j] = ok ? (x[b * ldx + c] - mean) * invstd : 0.f;
    }
    float sdy = 0.f, sdyx = 0.f;
#pragma unroll
    for (int j = 0; j < R; ++j) { sdy += d[j]; sdyx += d[j] * xh[j]; }
    red[0][g][lane] = sdy; red[1][g][lane] = sdyx;
    __syncthreads();
    sdy = ((red[0][0][lane] + red[0][1][lane]) + red[0][2][lane]) + red[0][3][lane];
    sdyx = ((red[1][0][lane] + red[1][1][lane]) + red[1][2][lane]) + red[1][3][lane];
    if (!live) return;
    if (g == 0) { dgamma[c] = sdyx; dbeta[c] = sdy; }
    const float k = gamma[c] * invstd / (float)B;
#pragma unroll
    for (int j = 0; j < R; ++j) { const int b = g + 4 * j; if (b < B) dx[(long)b * C + c] = k * ((float)B * d[j] - sdy - xh[j] * sdyx); }
}

// ---- OCFR ------------------------------------------------------------------------------------------------------
// F.normalize(dim=1, eps=1e-12) (OCFR.py:46-49): block per sample
__global__ __launch_bounds__(256) void ocfr_normalize_kernel(const float* __restrict__ f, long ldf, int D,
                                                             float* __restrict__ fn, float* __restrict__ inv_norm)
{
    __shared__ float red[16];
    const long b = blockIdx.x;
    float s = 0.f;
    for (int c = threadIdx.x; c < D; c += 256) { const float v = f[b * ldf + c]; s += v * v; }
    s = block_sum(s, red);
    const float inv = 1.f / fmaxf(sqrtf(s), 1e-12f);
    for (int c = threadIdx.x; c < D; c += 256) fn[b * D + c] = f[b * ldf + c] * inv;
    if (threadIdx.x == 0) inv_norm[b] = inv;
}
// per class c present in the batch: centers[c] = m * mean_{b: label==c} fn[b] + (1-m) * centers[c]  (OCFR.py:22-29,70-84)
__global__ void ocfr_update_kernel(const float* __restrict__ fn, const long* __restrict__ label, int B, int D,
                                   float momentum, float* __restrict__ centers)
{
    const int cls = blockIdx.x, col = blockIdx.y * blockDim.x + threadIdx.x;
    if (col >= D) return;
    float s = 0.f;
    int cnt = 0;
    for (int b = 0; b < B; ++b)
        if (label[b] == cls) { s += fn[(long)b * D + col]; ++cnt; }
    if (cnt) centers[(long)cls * D + col] = momentum * (s / (float)cnt) + (1.f - momentum) * centers[(long)cls * D + col];
}
// partial[b] = sum_d (centers[label_b][d] - fn[b][d])^2   (nn.MSELoss numerator, OCFR.py:31-42)
__global__ __launch_bounds__(256) void ocfr_loss_kernel(const float* __restrict__ fn, const long* __restrict__ label,
    const float* __restrict__ centers, int D, float* __restrict__ partial)
{
    __shared__ float red[16];
    const long b = blockIdx.x;
    const float* c = centers + label[b] * D;
    float s = 0.f;
    for (int d = threadIdx.x; d < D; d += 256) { const float v = c[d] - fn[b * D + d]; s += v * v; }
    s = block_sum(s, red);
    if (threadIdx.x == 0) partial[b] = s;
}
// df = inv_norm * (g - fn * <fn, g>),  g = dloss * 2/(B*D) * (fn - center)
__global__ __launch_bounds__(256) void ocfr_bwd_kernel(const float* __restrict__ fn, const float* __restrict__ inv_norm,
    const float* __restrict__ centers, const long* __restrict__ label, const float* __restrict__ dloss, float gnorm, int D,
    float* __restrict__ df)
{
    __shared__ float red[16];
    const long b = blockIdx.x;
    const float* c = centers + label[b] * D;
    const float gs = dloss[0] * gnorm;
    float dot = 0.f;
    for (int d = threadIdx.x; d < D; d += 256) { const float v = fn[b * D + d]; dot += v * gs * (v - c[d]); }
    dot = block_sum(dot, red);
    const float inv = inv_norm[b];
    for (int d = threadIdx.x; d < D; d += 256) {
        const float v = fn[b * D + d];
        df[b * D + d] = inv * (gs * (v - c[d]) - v * dot);
    }
}

__global__ void sum_scaled_kernel(const float* __restrict__ partial, int n, float scale, float* __restrict__ out, int accumulate)
{
    __shared__ float red[16];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += partial[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.f) + s * scale;
}

}  // namespace

extern "C" int editor_bn1d_fwd(const float* x, long ldx, int B, int C, const float* gamma, const float* beta,
    float* running_mean, float* running_var, float momentum, float eps, int training, float* y, float* save_mean,
    float* save_invstd, hipStream_t stream)
{
    if (B < 1 || C < 1) return (int)hipErrorInvalidValue;
    if (training && B <= 128)
        hipLaunchKernelGGL(bn1d_fwd_reg_kernel<32>, dim3((C + 63) / 64), dim3(256), 0, stream, x, ldx, B, C, gamma, beta, running_mean,
                           running_var, momentum, eps, y, save_mean, save_invstd);
    else if (training && B <= 256)
        hipLaunchKernelGGL(bn1d_fwd_reg_kernel<64>, dim3((C + 63) / 64), dim3(256), 0, stream, x, ldx, B, C, gamma, beta, running_mean,
                           running_var, momentum, eps, y, save_mean, save_invstd);
    else
    hipLaunchKernelGGL(bn1d_fwd_kernel, dim3((C + 63) / 64), dim3(64), 0, stream, x, ldx, B, C, gamma, beta, running_mean,
                       running_var, momentum, eps, training, y, save_mean, save_invstd);
    EDITOR_LAUNCH_CHECK();
    return 0;
}
extern "C" int editor_bn1d_bwd(const float* dy, const float* x, long ldx, int B, int C, const float* gamma,
    const float* save_mean, const float* save_invstd, float* dx, float* dgamma, float* dbeta, hipStream_t stream)
{
    if (B <= 128)
        hipLaunchKernelGGL(bn1d_bwd_reg_kernel<32>, dim3((C + 63) / 64), dim3(256), 0, stream, dy, x, ldx, B, C, gamma, save_mean,
                           save_invstd, dx, dgamma, dbeta);
    else if (B <= 256)
        hipLaunchKernelGGL(bn1d_bwd_reg_kernel<64>, dim3((C + 63) / 64), dim3(256), 0, stream, dy, x, ldx, B, C, gamma, save_mean,
                           save_invstd, dx, dgamma, dbeta);
    else
    hipLaunchKernelGGL(bn1d_bwd_kernel, dim3((C + 63) / 64), dim3(64), 0, stream, dy, x, ldx, B, C, gamma, save_mean,
                       save_invstd, dx, dgamma, dbeta);
    EDITOR_LAUNCH_CHECK();
    return 0;
}

extern "C" int editor_ocfr_fwd(const float* feat, long ldf, const long* label, int B, int D, int C, float* centers,
    float momentum, float* fnorm, float* inv_norm, float* workspace, float* loss, int accumulate, hipStream_t stream)
{
    if (B < 1 || D < 1 || C < 1) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(ocfr_normalize_kernel, dim3(B), dim3(256), 0, stream, feat, ldf, D, fnorm, inv_norm);
    EDITOR_LAUNCH_CHECK();
    hipLaunchKernelGGL(ocfr_update_kernel, dim3(C, (D + 255) / 256), dim3(256), 0, stream, fnorm, label, B, D, momentum, centers);
    EDITOR_LAUNCH_CHECK();
    hipLaunchKernelGGL(ocfr_loss_kernel, dim3(B), dim3(256), 0, stream, fnorm, label, centers, D, workspace);
    EDITOR_LAUNCH_CHECK();
    hipLaunchKernelGGL(sum_scaled_kernel, dim3(1), dim3(256), 0, stream, workspace, B, 1.f / ((float)B * D), loss, accumulate);
    EDITOR_LAUNCH_CHECK();
    return 0;
}
extern "C" int editor_ocfr_bwd(const float* fnorm, const float* inv_norm, const float* centers, const long* label,
    const float* dloss, int B, int D, float* dfeat, hipStream_t stream)
{
    hipLaunchKernelGGL(ocfr_bwd_kernel, dim3(B), dim3(256), 0, stream, fnorm, inv_norm, centers, label, dloss,
                       2.f / ((float)B * D), D, dfeat);
    EDITOR_LAUNCH_CHECK();
    return 0;
}
