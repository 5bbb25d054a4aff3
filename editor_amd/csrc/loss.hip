// Loss head that consumes the hot path's outputs, gfx950.  Tiny, latency-bound kernels
// (B x num_classes, B x 2304) whose job is to keep the whole training step inside libeditor_hip.so.
//   CrossEntropyLabelSmooth(eps)        layers/softmax_loss.py:4-34
//   TripletLoss() soft-margin form      layers/triplet_loss.py:16-33 (euclidean_dist), 51-84 (hard_example_mining),
//                                       121-136 (SoftMarginLoss(dist_an - dist_ap, 1))
// Every reduction runs in a fixed order (no atomics): the loss and its gradients are run-to-run deterministic.
#include "common.h"
#include "../../include/editor_hip.h"

namespace {

__device__ __forceinline__ float block_max(float v, float* red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_max(v);
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = red[0];
    for (int i = 1; i < nw; ++i) t = fmaxf(t, red[i]);
    return t;
}

// one block per sample: row_loss[b] = -sum_c soft_c * log_softmax_c, soft = (1-eps) onehot + eps/C
__global__ void ce_smooth_fwd_kernel(const float* __restrict__ logits, const long* __restrict__ target, int C, float eps,
    float* __restrict__ row_loss)
{
    __shared__ float red[16];
    const float* x = logits + (long)blockIdx.x * C;
    float m = -INFINITY;
    for (int c = threadIdx.x; c < C; c += blockDim.x) m = fmaxf(m, x[c]);
    m = block_max(m, red);
    float s = 0.f, sx = 0.f;
    for (int c = threadIdx.x; c < C; c += blockDim.x) { s += expf(x[c] - m); sx += x[c]; }
    s = block_sum(s, red);
    sx = block_sum(sx, red);
    if (threadIdx.x == 0) {
        const float lse = m + logf(s);
        const float lp_t = x[target[blockIdx.x]] - lse;          // log p of the labelled class
        const float lp_sum = sx - (float)C * lse;                // sum_c log p_c
        row_loss[blockIdx.x] = -((1.f - eps) * lp_t + eps / (float)C * lp_sum);
    }
}

// dlogits[b,c] = g/B * (softmax_c - soft_c)
__global__ void ce_smooth_bwd_kernel(const float* __restrict__ logits, const long* __restrict__ target, int B, int C,
    float eps, const float* __restrict__ g, float* __restrict__ dlogits)
{
    __shared__ float red[16];
    const float* x = logits + (long)blockIdx.x * C;
    float m = -INFINITY;
    for (int c = threadIdx.x; c < C; c += blockDim.x) m = fmaxf(m, x[c]);
    m = block_max(m, red);
    float s = 0.f;
    for (int c = threadIdx.x; c < C; c += blockDim.x) s += expf(x[c] - m);
    s = block_sum(s, red);
    const float inv = 1.f / s, k = g[0] / (float)B, u = eps / (float)C;
    const int t = (int)target[blockIdx.x];
    float* d = dlogits + (long)blockIdx.x * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x)
        d[c] = k * (expf(x[c] - m) * inv - u - (c == t ? 1.f - eps : 0.f));
}

// out[0] (+)= scale * sum_i v[i], fixed order
__global__ void sum_scalar_kernel(const float* __restrict__ v, int n, float scale, float* __restrict__ out, int accumulate)
{
    __shared__ float red[16];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += v[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.f) + scale * s;
}

__global__ void add_const_kernel(float* v, float c) { v[0] += c; }

__global__ void row_sqnorm_kernel(const float* __restrict__ f, long ldf, int D, float* __restrict__ sq)
{
    __shared__ float red[16];
    const float* x = f + (long)blockIdx.x * ldf;
    float s = 0.f;
    for (int c = threadIdx.x; c < D; c += blockDim.x) s += x[c] * x[c];
    s = block_sum(s, red);
    if (threadIdx.x == 0) sq[blockIdx.x] = s;
}

// one block per anchor: hardest positive (max distance among equal labels, self included as in the reference) and
// hardest negative (min among different labels).  Ties resolve to the lowest index.
// coef[i] = sigmoid(d_ap - d_an); coef[B+i] = 1/d_ap (0 where clamped); coef[2B+i] = 1/d_an.
__global__ void triplet_mine_kernel(const float* __restrict__ gram, const float* __restrict__ sq,
    const long* __restrict__ label, int B, int* __restrict__ idx, float* __restrict__ coef, float* __restrict__ row_loss)
{
    __shared__ float sv[2][256];
    __shared__ int si[2][256];
    const int i = blockIdx.x, tid = threadIdx.x;
    const long li = label[i];
    const float sqi = sq[i];
    float bp = -INFINITY, bn = INFINITY;
    int ip = -1, in_ = -1;
    for (int j = tid; j < B; j += blockDim.x) {
        const float q = sqi + sq[j] - 2.f * gram[(long)i * B + j];
        const float d = sqrtf(fmaxf(q, 1e-12f));
        if (label[j] == li) { if (d > bp) { bp = d; ip = j; } }
        else                { if (d < bn) { bn = d; in_ = j; } }
    }
    sv[0][tid] = bp; si[0][tid] = ip; sv[1][tid] = bn; si[1][tid] = in_;
    __syncthreads();
    for (int o = blockDim.x >> 1; o > 0; o >>= 1) {
        if (tid < o) {
            const float a = sv[0][tid], b = sv[0][tid + o];
            const int ia = si[0][tid], ib = si[0][tid + o];
            if (ib >= 0 && (ia < 0 || b > a || (b == a && ib < ia))) { sv[0][tid] = b; si[0][tid] = ib; }
            const float c = sv[1][tid], e = sv[1][tid + o];
            const int ic = si[1][tid], ie = si[1][tid + o];
            if (ie >= 0 && (ic < 0 || e < c || (e == c && ie < ic))) { sv[1][tid] = e; si[1][tid] = ie; }
        }
        __syncthreads();
    }
    if (tid == 0) {
        const float ap = sv[0][0], an = sv[1][0];
        const int p = si[0][0], n = si[1][0];
        idx[i] = p; idx[B + i] = n;
        const float z = ap - an;                                         // SoftMarginLoss(an - ap, 1) = log(1 + e^z)
        row_loss[i] = z > 0.f ? z + log1pf(expf(-z)) : log1pf(expf(z));
        coef[i] = 1.f / (1.f + expf(-z));
        const float qp = sqi + sq[p] - 2.f * gram[(long)i * B + p];
        const float qn = n >= 0 ? sqi + sq[n] - 2.f * gram[(long)i * B + n] : 0.f;
        coef[B + i] = qp > 1e-12f ? 1.f / ap : 0.f;                      // clamp(min) passes no gradient below the floor
        coef[2 * B + i] = (n >= 0 && qn > 1e-12f) ? 1.f / an : 0.f;
    }
}

// one block per feature row r: gather every anchor's contribution to row r (as anchor, as its positive, as its
// negative) in anchor order.  The few anchors that touch row r are listed in LDS first - by ballots over 64-anchor chunks
// (anchor order preserved) - together with their partner rows and weights, so that the column loop issues the three feature
// loads of FOUR list entries before it uses any (it was one dependent idx -> coef -> feature chain per entry: 75 us).
__global__ __launch_bounds__(256) void triplet_bwd_kernel(const float* __restrict__ f, long ldf, int B, int D, const int* __restrict__ idx,
    const float* __restrict__ coef, const float* __restrict__ g, float* __restrict__ df)
{
    __shared__ int l_i[1024], l_p[1024], l_n[1024];
    __shared__ float l_wp[1024], l_wn[1024];
    __shared__ int nlist;
    const int r = blockIdx.x;
    if (threadIdx.x < 64) {                                            // wave 0 builds the list
        int n = 0;
        for (int i0 = 0; i0 < B; i0 += 64) {
            const int i = i0 + threadIdx.x;
            const bool hit = i < B && (i == r || idx[i] == r || idx[B + i] == r);
            const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
            if (hit) {
                const int pos = n + __builtin_popcountll(m & ((1ull << threadIdx.x) - 1ull));
                const float s = coef[i];
                l_i[pos] = i; l_p[pos] = idx[i]; l_n[pos] = idx[B + i];
                l_wp[pos] = s * coef[B + i]; l_wn[pos] = s * coef[2 * B + i];
            }
            n += __builtin_popcountll(m);
        }
        if (threadIdx.x == 0) nlist = n;
    }
    __syncthreads();
    const float k = g[0] / (float)B;
    const int n_l = nlist;
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
        float acc = 0.f;
        for (int q0 = 0; q0 < n_l; q0 += 4) {
            float fi[4], fp[4], fn[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int q = min(q0 + j, n_l - 1);
                const int n = l_n[q];
                fi[j] = f[(long)l_i[q] * ldf + c];
                fp[j] = f[(long)l_p[q] * ldf + c];
                fn[j] = f[(long)(n >= 0 ? n : l_i[q]) * ldf + c];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int q = q0 + j;
                if (q >= n_l) break;
                const int i = l_i[q], p = l_p[q], n = l_n[q];
                const float ep = (fi[j] - fp[j]) * l_wp[q];
                const float en = n >= 0 ? (fi[j] - fn[j]) * l_wn[q] : 0.f;
                if (r == i) acc += ep - en;
                if (r == p) acc -= ep;
                if (r == n) acc += en;
            }
        }
        df[(long)r * D + c] = k * acc;
    }
}

}  // namespace

// CenterLoss.forward (layers/center_loss.py:30-51): distmat[i,k] = |x_i|^2 + |c_k|^2 - 2 x_i . c_k (the expanded form the reference
// computes with addmm_), masked to k = label_i, EVERY entry clamped to [1e-12, 1e12] - the B (C - 1) masked-out zeros each contribute
// 1e-12 - summed and divided by B.  One workgroup per sample: row[i] = clamp(d_i) with d_i kept for the backward's clamp gate.
__global__ __launch_bounds__(256) void center_loss_fwd_kernel(const float* __restrict__ x, const float* __restrict__ centers,
    const long* __restrict__ label, int D, float* __restrict__ dist, float* __restrict__ row)
{
    __shared__ float red[16];
    const int i = blockIdx.x;
    const float* xi = x + (long)i * D;
    const float* ck = centers + label[i] * D;
    float xx = 0.f, cc = 0.f, xc = 0.f;
    for (int c = threadIdx.x; c < D; c += blockDim.x) { const float a = xi[c], b = ck[c]; xx += a * a; cc += b * b; xc += a * b; }
    xx = block_sum(xx, red); __syncthreads();
    cc = block_sum(cc, red); __syncthreads();
    xc = block_sum(xc, red);
    if (threadIdx.x == 0) {
        const float d = (xx + cc) - 2.f * xc;
        dist[i] = d;
        row[i] = fminf(fmaxf(d, 1e-12f), 1e12f);
    }
}
// dx_i = dloss * 2 (x_i - c_{y_i}) / B where d_i lies inside the clamp range, else 0
__global__ __launch_bounds__(256) void center_loss_bwd_x_kernel(const float* __restrict__ x, const float* __restrict__ centers,
    const long* __restrict__ label, const float* __restrict__ dist, const float* __restrict__ dloss, int B, int D, float* __restrict__ dx)
{
    const int i = blockIdx.x;
    const float gate = (dist[i] >= 1e-12f && dist[i] <= 1e12f) ? 2.f * dloss[0] / (float)B : 0.f;
    const float* xi = x + (long)i * D;
    const float* ck = centers + label[i] * D;
    for (int c = threadIdx.x; c < D; c += blockDim.x) dx[(long)i * D + c] = gate * (xi[c] - ck[c]);
}
// dc_k = -dloss * 2 / B * sum_{i: y_i = k} (x_i - c_k), samples in index order (deterministic); classes without a sample get zeros
__global__ __launch_bounds__(256) void center_loss_bwd_c_kernel(const float* __restrict__ x, const float* __restrict__ centers,
    const long* __restrict__ label, const float* __restrict__ dist, const float* __restrict__ dloss, int B, int D, float* __restrict__ dc)
{
    const long k = blockIdx.x;
    const float sc = -2.f * dloss[0] / (float)B;
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
        float s = 0.f;
        const float ckc = centers[k * D + c];
        for (int i = 0; i < B; ++i)
            if (label[i] == k && dist[i] >= 1e-12f && dist[i] <= 1e12f) s += x[(long)i * D + c] - ckc;
        dc[k * D + c] = sc * s;
    }
}

extern "C" {

int editor_center_loss_fwd(const float* x, const float* centers, const long* label, int B, int C, int D, float* dist, float* row,
                           float* loss, editor_stream_t stream)
{
    if (B <= 0 || C <= 0 || D <= 0 || !x || !centers || !label || !dist || !row || !loss) return 1;
    hipStream_t st = (hipStream_t)stream;
    center_loss_fwd_kernel<<<B, 256, 0, st>>>(x, centers, label, D, dist, row);
    sum_scalar_kernel<<<1, 256, 0, st>>>(row, B, 1.f / (float)B, loss, 0);
    // the masked-out entries of the reference's (B, C) matrix: B (C - 1) zeros clamped to 1e-12, / B
    add_const_kernel<<<1, 1, 0, st>>>(loss, (float)((double)(C - 1) * 1e-12));
    EDITOR_LAUNCH_CHECK();
    return 0;
}

int editor_center_loss_bwd(const float* x, const float* centers, const long* label, const float* dist, const float* dloss, int B, int C,
                           int D, float* dx, float* dcenters, editor_stream_t stream)
{
    if (B <= 0 || C <= 0 || D <= 0 || !x || !centers || !label || !dist || !dloss) return 1;
    hipStream_t st = (hipStream_t)stream;
    if (dx) center_loss_bwd_x_kernel<<<B, 256, 0, st>>>(x, centers, label, dist, dloss, B, D, dx);
    if (dcenters) center_loss_bwd_c_kernel<<<C, 256, 0, st>>>(x, centers, label, dist, dloss, B, D, dcenters);
    EDITOR_LAUNCH_CHECK();
    return 0;
}

int editor_ce_smooth_fwd(const float* logits, const long* target, int B, int C, float eps, float* row_loss, float* loss,
                         int accumulate, editor_stream_t stream)
{
    if (B <= 0 || C <= 0) return 1;
    hipStream_t st = (hipStream_t)stream;
    ce_smooth_fwd_kernel<<<B, 256, 0, st>>>(logits, target, C, eps, row_loss);
    sum_scalar_kernel<<<1, 256, 0, st>>>(row_loss, B, 1.f / (float)B, loss, accumulate);
    EDITOR_LAUNCH_CHECK();
    return 0;
}

int editor_ce_smooth_bwd(const float* logits, const long* target, int B, int C, float eps, const float* dloss,
                         float* dlogits, editor_stream_t stream)
{
    if (B <= 0 || C <= 0) return 1;
    ce_smooth_bwd_kernel<<<B, 256, 0, (hipStream_t)stream>>>(logits, target, B, C, eps, dloss, dlogits);
    EDITOR_LAUNCH_CHECK();
    return 0;
}

int editor_triplet_fwd(const float* feat, long ldf, const long* label, int B, int D, float* gram, float* sq, int* idx,
                       float* coef, float* row_loss, float* loss, int accumulate, editor_stream_t stream)
{
    if (B <= 1 || D <= 0) return 1;
    hipStream_t st = (hipStream_t)stream;
    row_sqnorm_kernel<<<B, 256, 0, st>>>(feat, ldf, D, sq);
    // gram = feat feat^T on the exact-fp32 matrix cores (B operand in the (N,K) nn.Linear layout = feat itself).  B x B
    // is only (B/64)^2 output tiles: the reduction over D runs as a batch of 8 chunks into slabs behind `gram`, folded
    // in a fixed order (100 -> ~15 us at B = 128, D = 2304; deterministic).
    const int S = (D % 128 == 0 && D >= 1024) ? 8 : 1;
    int rc;
    if (S > 1) {
        float* slabs = gram + (long)B * B;
        const int kc = D / S;
        rc = editor_gemm_f32(feat, feat, slabs, B, B, kc, ldf, ldf, B, 0, 0, S, kc, kc, (long)B * B, 1, 0, 0, 0, 1.f, 0.f, nullptr,
                             nullptr, 1, EDITOR_EPI_NONE, nullptr, 0, stream);
        if (!rc) rc = editor_reduce_rows(slabs, S, (long)B * B, gram, 0, 1.f, stream);
    } else {
        rc = editor_gemm_f32(feat, feat, gram, B, B, D, ldf, ldf, B, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1.f, 0.f, nullptr, nullptr, 1,
                             EDITOR_EPI_NONE, nullptr, 0, stream);
    }
    if (rc) return rc;
    triplet_mine_kernel<<<B, 256, 0, st>>>(gram, sq, label, B, idx, coef, row_loss);
    sum_scalar_kernel<<<1, 256, 0, st>>>(row_loss, B, 1.f / (float)B, loss, accumulate);
    EDITOR_LAUNCH_CHECK();
    return 0;
}

int editor_triplet_bwd(const float* feat, long ldf, int B, int D, const int* idx, const float* coef, const float* dloss,
                       float* dfeat, editor_stream_t stream)
{
    if (B <= 1 || B > 1024 || D <= 0) return 1;
    triplet_bwd_kernel<<<B, 256, 0, (hipStream_t)stream>>>(feat, ldf, B, D, idx, coef, dloss, dfeat);
    EDITOR_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
