// Retrieval evaluation of the hot path's eval-mode features, gfx950.
//   F.normalize(feats)                         utils/metrics.py:259-261
//   squared-Euclidean distance matrix          utils/metrics.py:12-18  (qq + gg^T, then addmm_(beta=1, alpha=-2))
//   per-query ascending ranking                utils/metrics.py:143    (np.argsort(distmat, axis=1))
//   CMC / AP with (pid, cam|scene) removal     utils/metrics.py:151-183 (eval_func), :66-123 (eval_func_msrv)
// HBM-bound index work: the ranking is a bitonic network over 64-bit (orderable distance << 32 | gallery index) keys,
// so exactly tied distances rank by gallery index (a stable argsort) and the result is bit-reproducible.
#include "common.h"
#include "../../include/editor_hip.h"

namespace {

typedef unsigned long long u64;
constexpr int CH = 4096;            // keys per LDS-resident chunk (32 KB)
constexpr int SORT_THREADS = 512;

__global__ void l2norm_rows_kernel(const float* __restrict__ x, long ldx, int D, float eps, float* __restrict__ y)
{
    __shared__ float red[16];
    const float* r = x + (long)blockIdx.x * ldx;
    float s = 0.f;
    for (int c = threadIdx.x; c < D; c += blockDim.x) s += r[c] * r[c];
    s = block_sum(s, red);
    const float den = fmaxf(sqrtf(s), eps);
    for (int c = threadIdx.x; c < D; c += blockDim.x) y[(long)blockIdx.x * D + c] = r[c] / den;
}

__global__ void sqnorm_rows_kernel(const float* __restrict__ x, long ldx, int D, float* __restrict__ sq)
{
    __shared__ float red[16];
    const float* r = x + (long)blockIdx.x * ldx;
    float s = 0.f;
    for (int c = threadIdx.x; c < D; c += blockDim.x) s += r[c] * r[c];
    s = block_sum(s, red);
    if (threadIdx.x == 0) sq[blockIdx.x] = s;
}

__global__ void outer_sum_kernel(const float* __restrict__ qq, const float* __restrict__ gg, int Q, int G,
                                 float* __restrict__ dist)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)Q * G) return;
    dist[i] = qq[i / G] + gg[i % G];
}

__device__ __forceinline__ unsigned orderable(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ void build_keys_kernel(const float* __restrict__ dist, int G, int P, u64* __restrict__ keys)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const long q = blockIdx.y;
    keys[q * P + i] = i < G ? ((u64)orderable(dist[q * G + i]) << 32) | (unsigned)i : ~0ull;
}

__device__ __forceinline__ void cmpx(u64& a, u64& b, bool asc) {
    if ((a > b) == asc) { const u64 t = a; a = b; b = t; }
}

// One chunk of n = min(CH, P) keys in LDS.  FULL: every (k, j) stage with k <= n.  Otherwise: the j < n tail of the
// single stage k (k > n), after the strided steps were done in global memory.
template <bool FULL>
__global__ void __launch_bounds__(SORT_THREADS) bitonic_local_kernel(u64* __restrict__ keys, int P, int k_merge)
{
    __shared__ u64 s[CH];
    const int n = P < CH ? P : CH;
    const long base = (long)blockIdx.y * P + (long)blockIdx.x * n;
    const int goff = blockIdx.x * n;
    for (int e = threadIdx.x; e < n; e += SORT_THREADS) s[e] = keys[base + e];
    __syncthreads();
    for (int k = FULL ? 2 : k_merge; k <= (FULL ? n : k_merge); k <<= 1) {
        for (int j = (k >> 1) < n ? (k >> 1) : (n >> 1); j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < (n >> 1); t += SORT_THREADS) {
                const int i = ((t / j) * 2 * j) + (t % j);
                const bool asc = ((goff + i) & k) == 0;
                u64 a = s[i], b = s[i + j];
                if ((a > b) == asc) { s[i] = b; s[i + j] = a; }
            }
            __syncthreads();
        }
    }
    for (int e = threadIdx.x; e < n; e += SORT_THREADS) keys[base + e] = s[e];
}

__global__ void bitonic_global_kernel(u64* __restrict__ keys, int P, int k, int j)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (P >> 1)) return;
    u64* row = keys + (long)blockIdx.y * P;
    const int i = ((t / j) * 2 * j) + (t % j);
    u64 a = row[i], b = row[i + j];
    const bool asc = (i & k) == 0;
    if ((a > b) == asc) { row[i] = b; row[i + j] = a; }
}

__global__ void extract_order_kernel(const u64* __restrict__ keys, int G, int P, int* __restrict__ order)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= G) return;
    order[(long)blockIdx.y * G + i] = (int)(unsigned)(keys[(long)blockIdx.y * P + i] & 0xffffffffull);
}

// one wave per query walks its ranking 64 ranks at a time
__global__ void rank_metrics_kernel(const int* __restrict__ order, const long* __restrict__ q_pids,
    const long* __restrict__ g_pids, const long* __restrict__ q_aux, const long* __restrict__ g_aux, int G,
    double* __restrict__ ap, int* __restrict__ first_pos)
{
    const int q = blockIdx.x, lane = threadIdx.x;
    const long qp = q_pids[q], qa = q_aux[q];
    const u64 lt = lane ? (~0ull >> (64 - lane)) : 0ull;
    int kept_before = 0, match_before = 0, first = -1;
    double acc = 0.0;
    for (int base = 0; base < G; base += 64) {
        const int r = base + lane;
        const bool valid = r < G;
        const int g = valid ? order[(long)q * G + r] : 0;
        const bool same = valid && g_pids[g] == qp;
        const bool keep = valid && !(same && g_aux[g] == qa);
        const bool match = keep && same;
        const u64 km = __ballot(keep), mm = __ballot(match);
        const int pos = kept_before + __popcll(km & lt);                  // 0-based rank among the kept entries
        const int cum = match_before + __popcll(mm & (lt | (1ull << lane)));
        if (match) acc += (double)cum / (double)(pos + 1);
        if (first < 0 && mm) {
            const int fl = __ffsll((long long)mm) - 1;
            first = kept_before + __popcll(km & (fl ? (~0ull >> (64 - fl)) : 0ull));
        }
        kept_before += __popcll(km);
        match_before += __popcll(mm);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) {
        ap[q] = match_before ? acc / (double)match_before : 0.0;
        first_pos[q] = match_before ? first : -1;
    }
}

// totals[0] = sum of AP over valid queries, totals[1] = #valid; cmc_counts[r] = #valid queries matched within rank r
__global__ void rank_reduce_kernel(const double* __restrict__ ap, const int* __restrict__ first_pos, int Q, int max_rank,
    double* __restrict__ totals, int* __restrict__ cmc_counts)
{
    const int r = threadIdx.x;
    if (r < max_rank) {
        int c = 0;
        for (int q = 0; q < Q; ++q) { const int f = first_pos[q]; c += (f >= 0 && f <= r); }
        cmc_counts[r] = c;
    }
    if (r == 0) {
        double s = 0.0; int v = 0;
        for (int q = 0; q < Q; ++q) if (first_pos[q] >= 0) { s += ap[q]; ++v; }
        totals[0] = s; totals[1] = (double)v;
    }
}

}  // namespace

extern "C" {

int editor_l2norm_rows(const float* x, long ldx, long M, int D, float eps, float* y, editor_stream_t stream)
{
    if (M <= 0 || D <= 0) return 1;
    l2norm_rows_kernel<<<(unsigned)M, 256, 0, (hipStream_t)stream>>>(x, ldx, D, eps, y);
    EDITOR_LAUNCH_CHECK();
    return 0;
}

int editor_distmat_f32(const float* qf, long ldq, const float* gf, long ldg, int Q, int G, int D, float* qq, float* gg,
                       float* dist, editor_stream_t stream)
{
    if (Q <= 0 || G <= 0 || D <= 0) return 1;
    hipStream_t st = (hipStream_t)stream;
    sqnorm_rows_kernel<<<Q, 256, 0, st>>>(qf, ldq, D, qq);
    sqnorm_rows_kernel<<<G, 256, 0, st>>>(gf, ldg, D, gg);
    const long n = (long)Q * G;
    outer_sum_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(qq, gg, Q, G, dist);
    EDITOR_LAUNCH_CHECK();
    return editor_gemm_f32(qf, gf, dist, Q, G, D, ldq, ldg, G, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, -2.f, 1.f, nullptr, nullptr, 1,
                           EDITOR_EPI_NONE, nullptr, 0, stream);
}

int editor_rank_sort(const float* dist, int Q, int G, int P, unsigned long long* keys, int* order, editor_stream_t stream)
{
    if (Q <= 0 || G <= 0 || P < G || (P & (P - 1)) || P < 2) return 1;
    hipStream_t st = (hipStream_t)stream;
    build_keys_kernel<<<dim3((P + 255) / 256, Q), 256, 0, st>>>(dist, G, P, keys);
    const int n = P < CH ? P : CH;
    bitonic_local_kernel<true><<<dim3(P / n, Q), SORT_THREADS, 0, st>>>(keys, P, 0);
    for (int k = 2 * n; k <= P; k <<= 1) {
        for (int j = k >> 1; j >= n; j >>= 1)
            bitonic_global_kernel<<<dim3(((P >> 1) + 255) / 256, Q), 256, 0, st>>>(keys, P, k, j);
        bitonic_local_kernel<false><<<dim3(P / n, Q), SORT_THREADS, 0, st>>>(keys, P, k);
    }
    extract_order_kernel<<<dim3((G + 255) / 256, Q), 256, 0, st>>>(keys, G, P, order);
    EDITOR_LAUNCH_CHECK();
    return 0;
}

int editor_rank_metrics(const int* order, const long* q_pids, const long* g_pids, const long* q_aux, const long* g_aux,
                        int Q, int G, int max_rank, double* ap, int* first_pos, double* totals, int* cmc_counts,
                        editor_stream_t stream)
{
    if (Q <= 0 || G <= 0 || max_rank <= 0 || max_rank > 1024) return 1;
    hipStream_t st = (hipStream_t)stream;
    rank_metrics_kernel<<<Q, 64, 0, st>>>(order, q_pids, g_pids, q_aux, g_aux, G, ap, first_pos);
    rank_reduce_kernel<<<1, 1024, 0, st>>>(ap, first_pos, Q, max_rank, totals, cmc_counts);
    EDITOR_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
