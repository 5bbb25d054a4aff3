// k-reciprocal re-ranking of the retrieval evaluation (utils/metrics.py:275-278 ->
// utils/reranking.py:30-101, Zhong et al. CVPR 2017), gfx950.  Dense N x N work over ALL images (N = queries + gallery):
//   editor_rerank_normalise   reranking.py:37-47   od[i,j] = dist[j,i] / max_k dist[k,i]           (fp32, HBM-bound)
//   (editor_rank_sort)        reranking.py:49      initial_rank = argsort(od) per row
//   editor_rerank_weights     reranking.py:51-72   k-reciprocal sets + 2/3-overlap expansion, V[i,.] = exp(-od) / sum  -> fp16
//   editor_rerank_expand      reranking.py:74-79   V[i,.] <- mean of the k2 nearest rows (fp32 accumulation in rank order, fp16 result)
//   editor_rerank_final       reranking.py:81-101  Jaccard distance over the common non-zeros IN FP16, ascending column order,
//                                                  final = fp16(jaccard * (1 - lambda)) + lambda * od   -> (Q, N - Q) fp32
// The float16 storage of V and the float16 arithmetic of the Jaccard step are part of the reference as shipped (np.float16 arrays);
// numpy evaluates a half operation as float(a) op float(b) rounded to half - reproduced literally here (NOT v_add_f16 / v_div:
// one rounding instead of two would differ on ties), as are numpy's pairwise summation of the weights and its reduction order.
#include "common.h"
#include "../../include/editor_hip.h"

namespace {

typedef unsigned long long u64;

__device__ __forceinline__ float h2f(uint16_t v) { return f16_to_f32(v); }
__device__ __forceinline__ uint16_t f2h(float f) { return f32_to_f16(f); }

// colmax[c] = max_k dist[k, c]: a thread per column walks the rows (coalesced across the wave)
__global__ void colmax_kernel(const float* __restrict__ dist, int N, float* __restrict__ colmax)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N) return;
    float m = dist[c];
    for (int k = 1; k < N; ++k) m = fmaxf(m, dist[(long)k * N + c]);
    colmax[c] = m;
}

// od[i, j] = dist[j, i] / colmax[i]: 32 x 32 tiles through LDS so that both sides move whole lines
__global__ void transpose_div_kernel(const float* __restrict__ dist, const float* __restrict__ colmax, int N, float* __restrict__ od)
{
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;           // 32 x 8
    const int j0 = blockIdx.y * 32, i0 = blockIdx.x * 32;             // source rows j, source columns i
    for (int r = ty; r < 32; r += 8) {
        const int j = j0 + r, i = i0 + tx;
        tile[r][tx] = (j < N && i < N) ? dist[(long)j * N + i] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int i = i0 + r, j = j0 + tx;
        if (i < N && j < N) od[(long)i * N + j] = tile[tx][r] / colmax[i];
    }
}

__global__ void transpose_u16_kernel(const uint16_t* __restrict__ in, int N, uint16_t* __restrict__ out)
{
    __shared__ uint16_t tile[64][66];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;           // 64 x 4
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    for (int r = ty; r < 64; r += 4)
        tile[r][tx] = (r0 + r < N && c0 + tx < N) ? in[(long)(r0 + r) * N + c0 + tx] : (uint16_t)0;
    __syncthreads();
    for (int r = ty; r < 64; r += 4)
        if (c0 + r < N && r0 + tx < N) out[(long)(c0 + r) * N + r0 + tx] = tile[tx][r];
}

// numpy's float32 add.reduce over a contiguous vector: 0 + pairwise_sum(a, n) (numpy/core/src/umath/loops_utils.h.src:
// < 8 elements a plain loop, <= 128 eight running sums combined as ((0+1)+(2+3))+((4+5)+(6+7)) then the tail, else two halves, the
// first a multiple of 8).  One lane, <= 2112 elements.
__device__ float np_pairwise_sum(const float* a, int n)
{
    if (n < 8) {
        float res = 0.f;
        for (int i = 0; i < n; ++i) res += a[i];
        return res;
    }
    if (n <= 128) {
        float r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int i = 8;
        for (; i < n - (n % 8); i += 8)
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
}

// R(c, k) (reranking.py:53-57 / :61-66): lane r < k holds near[r] = rank[c, r]; kept when c is in rank[near[r], :k].  Returns the
// ballot of kept lanes (their order = the order of `near`).
__device__ __forceinline__ u64 reciprocal_lanes(const int* __restrict__ rank, int N, int c, int k, int lane, int& near)
{
    bool keep = false;
    near = 0;
    if (lane < k) {
        near = rank[(long)c * N + lane];
        const int* back = rank + (long)near * N;
        for (int q = 0; q < k; ++q) keep |= (back[q] == c);
    }
    return __ballot(keep);
}

constexpr int kMaxK1 = 64, kMaxHalf = 32, kMaxSet = kMaxK1 + kMaxK1 * kMaxHalf;      // 2112 members at most

// one wave per image i
__global__ void __launch_bounds__(64) rerank_weights_kernel(const float* __restrict__ od, const int* __restrict__ rank, int N,
                                                            int K1, int K1H, uint16_t* __restrict__ V)
{
    extern __shared__ unsigned smem_w[];
    const int words = (N + 31) >> 5;
    unsigned* bitmap = smem_w;                                  // [words]
    int* base = reinterpret_cast<int*>(bitmap + words);         // [kMaxK1]   R(i, k1)
    int* members = base + kMaxK1;                               // [kMaxSet]  sorted unique members
    float* w = reinterpret_cast<float*>(members + kMaxSet);     // [kMaxSet]
    const int i = blockIdx.x, lane = threadIdx.x;
    const u64 below = lane ? (~0ull >> (64 - lane)) : 0ull;
    for (int q = lane; q < words; q += 64) bitmap[q] = 0u;
    int near;
    const u64 km = reciprocal_lanes(rank, N, i, K1, lane, near);
    const int nbase = __popcll(km);
    if ((km >> lane) & 1ull) base[__popcll(km & below)] = near;
    __syncthreads();
    if (lane < nbase) atomicOr(&bitmap[base[lane] >> 5], 1u << (base[lane] & 31));
    for (int j = 0; j < nbase; ++j) {
        const int c = base[j];
        int cn;
        const u64 cm = reciprocal_lanes(rank, N, c, K1H, lane, cn);
        const int ncand = __popcll(cm);
        bool common = false;
        if ((cm >> lane) & 1ull)
            for (int q = 0; q < nbase; ++q) common |= (base[q] == cn);
        const int ncommon = __popcll(__ballot(common));
        if ((double)ncommon > 2.0 / 3.0 * (double)ncand) {      // (wave-uniform)  len(intersect1d) > 2 / 3 * len(candidate set)
            if ((cm >> lane) & 1ull) atomicOr(&bitmap[cn >> 5], 1u << (cn & 31));
        }
    }
    __syncthreads();
    // np.unique: the set bits in ascending order
    int total = 0;
    for (int q0 = 0; q0 < words; q0 += 64) {
        const int q = q0 + lane;
        const unsigned bits = q < words ? bitmap[q] : 0u;
        const int cnt = __popc(bits);
        int incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
        int pos = total + incl - cnt;
        unsigned b = bits;
        while (b) { const int bit = __ffs((int)b) - 1; b &= b - 1; members[pos++] = (q << 5) + bit; }
        total += __shfl(incl, 63, 64);
    }
    __syncthreads();
    for (int l = lane; l < total; l += 64) w[l] = expf(-od[(long)i * N + members[l]]);
    __syncthreads();
    float s = 0.f;
    if (lane == 0) s = np_pairwise_sum(w, total);
    s = __shfl(s, 0, 64);
    for (int l = lane; l < total; l += 64) V[(long)i * N + members[l]] = f2h(w[l] / s);
}

// V_qe[i, c] = half( (sum over r < k2, in rank order, of float(V[rank[i, r], c])) / k2 )
__global__ void rerank_expand_kernel(const uint16_t* __restrict__ V, const int* __restrict__ rank, int N, int K2,
                                     uint16_t* __restrict__ out)
{
    const int i = blockIdx.y;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N) return;
    float acc = 0.f;
    for (int r = 0; r < K2; ++r) acc += h2f(V[(long)rank[(long)i * N + r] * N + c]);
    out[(long)i * N + c] = f2h(acc / (float)K2);
}

constexpr int JCH = 8192;           // columns of V per LDS-resident chunk of the query's non-zero list

// block (query i, 1024 gallery columns): t[g] = sum over the common non-zero columns j (ascending) of min(V[i,j], V[g,j]), every
// partial sum rounded to half; jaccard = 1 - t / (2 - t) in half; out = float(half(jaccard * w16)) + od[i, g] * lam
__global__ void __launch_bounds__(256) rerank_final_kernel(const uint16_t* __restrict__ V, const uint16_t* __restrict__ Vt,
    const float* __restrict__ od, int N, int Q, uint16_t w16, float lam, float* __restrict__ out)
{
    __shared__ int nz_j[JCH];
    __shared__ uint16_t nz_v[JCH];
    __shared__ int nz_n;
    const int i = blockIdx.y, G = N - Q;
    const int g0 = Q + blockIdx.x * 1024;
    const int lane = threadIdx.x & 63;
    const u64 below = lane ? (~0ull >> (64 - lane)) : 0ull;
    uint16_t t[4] = {0, 0, 0, 0};
    for (int j0 = 0; j0 < N; j0 += JCH) {
        __syncthreads();
        if (threadIdx.x < 64) {                                  // wave 0: ordered compaction of the chunk's non-zeros
            int n = 0;
            const int jend = min(N, j0 + JCH);
            for (int jb = j0; jb < jend; jb += 64) {
                const int j = jb + lane;
                const uint16_t v = j < jend ? V[(long)i * N + j] : (uint16_t)0;
                const bool nzv = (v & 0x7fffu) != 0;             // (+0 / -0 are zero; NaN cannot occur)
                const u64 m = __ballot(nzv);
                if (nzv) { const int p = n + __popcll(m & below); nz_j[p] = j; nz_v[p] = v; }
                n += __popcll(m);
            }
            if (lane == 0) nz_n = n;
        }
        __syncthreads();
        const int n = nz_n;
        for (int e = 0; e < n; ++e) {
            const int j = nz_j[e];
            const float vi = h2f(nz_v[e]);
            const uint16_t* col = Vt + (long)j * N;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int g = g0 + u * 256 + threadIdx.x;
                if (g < N) {
                    const uint16_t vg = col[g];
                    if ((vg & 0x7fffu) != 0) t[u] = f2h(h2f(t[u]) + fminf(vi, h2f(vg)));
                }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int g = g0 + u * 256 + threadIdx.x;
        if (g >= N) continue;
        const float tf = h2f(t[u]);
        const float den = h2f(f2h(2.f - tf));
        const float frac = h2f(f2h(tf / den));
        const float jac = h2f(f2h(1.f - frac));
        const float jw = h2f(f2h(jac * h2f(w16)));
        float scaled = od[(long)i * N + g] * lam;
        asm volatile("" : "+v"(scaled));                         // two roundings, as numpy: keeps hipcc's default contraction from fusing
        out[(long)i * G + (g - Q)] = jw + scaled;                // the product into the sum (an fma differs in the last bit on 18 % of entries)
    }
}

}  // namespace

// every stage indexes the N = Q + G images through grid.y (rows) somewhere in the pipeline (rank sort, expand, final): one limit for
// all entry points, checked before anything is launched
constexpr int kMaxImages = 65535;

extern "C" {

int editor_rerank_normalise(const float* dist, int N, float* colmax, float* od, editor_stream_t stream)
{
    if (N <= 0 || N > kMaxImages || !dist || !colmax || !od) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    colmax_kernel<<<(N + 255) / 256, 256, 0, st>>>(dist, N, colmax);
    transpose_div_kernel<<<dim3((N + 31) / 32, (N + 31) / 32), 256, 0, st>>>(dist, colmax, N, od);
    EDITOR_LAUNCH_CHECK();
    return 0;
}

int editor_rerank_weights(const float* od, const int* rank, int N, int k1, int k1_half, uint16_t* V, editor_stream_t stream)
{
    // (k + 1 entries of a ranking are looked at; every image has at least that many neighbours)
    if (N <= 0 || k1 < 1 || k1 + 1 > kMaxK1 || k1_half < 0 || k1_half + 1 > kMaxHalf || k1 + 1 > N || N > kMaxImages || !od || !rank || !V)
        return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(V, 0, (size_t)N * N * sizeof(uint16_t), st);
    if (e != hipSuccess) return (int)e;
    const size_t lds = (size_t)((N + 31) / 32) * 4 + kMaxK1 * 4 + (size_t)kMaxSet * 8;
    rerank_weights_kernel<<<N, 64, lds, st>>>(od, rank, N, k1 + 1, k1_half + 1, V);
    EDITOR_LAUNCH_CHECK();
    return 0;
}

int editor_rerank_expand(const uint16_t* V, const int* rank, int N, int k2, uint16_t* Vq, editor_stream_t stream)
{
    if (N <= 0 || N > kMaxImages || k2 < 1 || k2 > N || !V || !rank || !Vq) return (int)hipErrorInvalidValue;
    rerank_expand_kernel<<<dim3((N + 255) / 256, N), 256, 0, (hipStream_t)stream>>>(V, rank, N, k2, Vq);
    EDITOR_LAUNCH_CHECK();
    return 0;
}

int editor_rerank_final(const uint16_t* V, uint16_t* Vt, const float* od, int N, int Q, int one_minus_lambda_f16_bits,
                        float lambda, float* final_dist, editor_stream_t stream)
{
    if (N <= 0 || N > kMaxImages || Q <= 0 || Q >= N || !V || !Vt || !od || !final_dist) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    transpose_u16_kernel<<<dim3((N + 63) / 64, (N + 63) / 64), 256, 0, st>>>(V, N, Vt);
    rerank_final_kernel<<<dim3((N - Q + 1023) / 1024, Q), 256, 0, st>>>(V, Vt, od, N, Q, (uint16_t)one_minus_lambda_f16_bits, lambda, final_dist);
    EDITOR_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
