// Token-selection kernels of the EDITOR hot path, gfx950.
//   K8/K9 frequency branch : tile-local 4-level Haar (one wavefront per 16x16 patch, butterflies via
//                            cross-lane shuffles, no LDS) -> modality mean -> inverse -> positive count
//   K7/K9 top-k            : libstdc++ partial_sort / nth_element tie order, one lane per row in LDS
//   K6    attention rollout: row-vector form r <- r * A_l, one workgroup per (sample, head)
//   K10   SFTS mask apply + background-consistency loss (+ backward)
// Reference behaviour restated: modeling/fusion_part/Frequency.py:42-84, SFTS.py:145-164,181-230.
#include "common.h"
#include "../../include/editor_hip.h"

#pragma clang fp contract(off)   // keep s*a + s*b as two roundings + add, like the reference's fp32 ops

// ------------------------------------------------------------------------------------------------
// K8/K9a  per-patch positive-pixel counts
// ------------------------------------------------------------------------------------------------
namespace {

constexpr float kS = 0.70710678118654752440f;   // haar tap 1/sqrt(2) rounded to fp32 (lowlevel.py:970-974)

struct Quad { float ll, lh, hl, hh; };

// one analysis level on a 2x2 block (x00 x01 / x10 x11): rows first, then columns (AFB2D lowlevel.py:341-343)
__device__ __forceinline__ Quad haar_fwd(float x00, float x01, float x10, float x11) {
    // torch's CPU conv2d evaluates each 2-tap analysis dot product as round(s*even) then ONE fma with the
    // odd sample (measured bit-exact against the reference, see oracle/editor_ref.py:_analysis_1d).
    const float p0 = __fmul_rn(kS, x00), p1 = __fmul_rn(kS, x10);
    const float lo0 = __fmaf_rn(kS, x01, p0), hi0 = __fmaf_rn(-kS, x01, p0);
    const float lo1 = __fmaf_rn(kS, x11, p1), hi1 = __fmaf_rn(-kS, x11, p1);
    const float pl = __fmul_rn(kS, lo0), ph = __fmul_rn(kS, hi0);
    Quad q;
    q.ll = __fmaf_rn(kS, lo1, pl);    // row-lo, col-lo
    q.lh = __fmaf_rn(-kS, lo1, pl);   // row-lo, col-hi   (band 0 of the reference's yh)
    q.hl = __fmaf_rn(kS, hi1, ph);    // row-hi, col-lo   (band 1)
    q.hh = __fmaf_rn(-kS, hi1, ph);   // band 2
    return q;
}
// one synthesis level, returning the element at (row parity ry, col parity rx): columns then rows
// (SFB2D lowlevel.py:676-679)
__device__ __forceinline__ float haar_inv(const Quad& q, int ry, int rx) {
    const float lo = ry ? (kS * q.ll - kS * q.lh) : (kS * q.ll + kS * q.lh);
    const float hi = ry ? (kS * q.hl - kS * q.hh) : (kS * q.hl + kS * q.hh);
    return rx ? (kS * lo - kS * hi) : (kS * lo + kS * hi);
}

// gather the 2x2 neighbourhood of `v` across the lane bits (bx = column bit, by = row bit)
__device__ __forceinline__ Quad gather_level(float v, int lane, int bx, int by) {
    const float vx = __shfl_xor(v, bx, 64), vy = __shfl_xor(v, by, 64), vxy = __shfl_xor(v, bx | by, 64);
    const bool cx = lane & bx, cy = lane & by;
    const float x00 = cy ? (cx ? vxy : vy) : (cx ? vx : v);
    const float x01 = cy ? (cx ? vy : vxy) : (cx ? v : vx);
    const float x10 = cy ? (cx ? vx : v) : (cx ? vxy : vy);
    const float x11 = cy ? (cx ? v : vx) : (cx ? vy : vxy);
    return haar_fwd(x00, x01, x10, x11);
}

// NMOD / NC > 0: compile-time modality / channel counts (the path's 3 x 3 and the 4-modal extension): all NMOD * NC * 2
// row loads of a lane (8 bytes each: its 2 x 2 pixels) are REQUESTED UP FRONT - the butterflies of one (channel, modality)
// plane are ~80 dependent shuffles / FMAs, and with the loads inside the runtime loops at most two were in flight per lane
// while it waited (3.3 TB/s on 151 MB -> 3.9).  Measured and dropped: staging 16-row x 64-pixel strips of every plane through
// LDS with one 16-byte load per thread and plane (full cache lines): 54 us against 39 - 39 KiB of LDS per block leaves a
// quarter of the waves resident, and the kernel is bound by its shuffle chains, not by request granularity.
// NMOD == 0: the generic runtime-count form.
template <int NMOD, int NC>
__global__ __launch_bounds__(256) void freq_counts_kernel(
    const float* __restrict__ m0, const float* __restrict__ m1, const float* __restrict__ m2, const float* __restrict__ m3,
    int nmod_rt, int B, int C_rt, int H, int W, int32_t* __restrict__ counts)
{
    const int nmod = NMOD > 0 ? NMOD : nmod_rt, C = NMOD > 0 ? NC : C_rt;
    const int lane = threadIdx.x & 63;
    const int tiles_x = W >> 4, tiles_y = H >> 4, ntile = tiles_x * tiles_y;
    const long tile = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= (long)B * ntile) return;                 // whole wave exits together
    const int b = (int)(tile / ntile), p = (int)(tile % ntile);
    const int ty = p / tiles_x, tx = p % tiles_x;
    const int lx = lane & 7, ly = lane >> 3;             // lane grid 8x8, 2x2 pixels per lane
    const int y0 = ty * 16 + ly * 2, x0 = tx * 16 + lx * 2;
    const float* mods[4] = {m0, m1, m2, m3};
    const float fnm = (float)nmod;

    constexpr int NPRE = NMOD > 0 ? NMOD * NC : 1;
    float2 pre0[NPRE], pre1[NPRE];
    if constexpr (NMOD > 0) {
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int m = 0; m < NMOD; ++m) {
                const float* base = mods[m] + (((long)b * NC + c) * H + y0) * W + x0;
                pre0[c * NMOD + m] = *reinterpret_cast<const float2*>(base);
                pre1[c * NMOD + m] = *reinterpret_cast<const float2*>(base + W);
            }
    }

    float sum[4] = {0.f, 0.f, 0.f, 0.f};                 // channel sums of the reconstruction
    auto channel = [&](int c) {
        Quad acc[4];                                      // modality-summed coefficients per level
#pragma unroll
        for (int l = 0; l < 4; ++l) acc[l] = Quad{0.f, 0.f, 0.f, 0.f};
        float ll4 = 0.f;
#pragma unroll
        for (int m = 0; m < (NMOD > 0 ? NMOD : 4); ++m) {
            if (m >= nmod) break;
            float2 r0, r1;
            if constexpr (NMOD > 0) { r0 = pre0[c * NMOD + m]; r1 = pre1[c * NMOD + m]; }
            else {
                const float* base = mods[m] + (((long)b * C + c) * H + y0) * W + x0;
                r0 = *reinterpret_cast<const float2*>(base);
                r1 = *reinterpret_cast<const float2*>(base + W);
            }
            Quad q1 = haar_fwd(r0.x, r0.y, r1.x, r1.y);
            Quad q2 = gather_level(q1.ll, lane, 1, 8);
            Quad q3 = gather_level(q2.ll, lane, 2, 16);
            Quad q4 = gather_level(q3.ll, lane, 4, 32);
            // (Ylx + Yly + Ylz) accumulated in modality order (Frequency.py:71-74)
            if (m == 0) { acc[0] = q1; acc[1] = q2; acc[2] = q3; acc[3] = q4; ll4 = q4.ll; }
            else {
                acc[0].lh += q1.lh; acc[0].hl += q1.hl; acc[0].hh += q1.hh;
                acc[1].lh += q2.lh; acc[1].hl += q2.hl; acc[1].hh += q2.hh;
                acc[2].lh += q3.lh; acc[2].hl += q3.hl; acc[2].hh += q3.hh;
                acc[3].lh += q4.lh; acc[3].hl += q4.hl; acc[3].hh += q4.hh;
                ll4 += q4.ll;
            }
        }
#pragma unroll
        for (int l = 0; l < 4; ++l) { acc[l].lh /= fnm; acc[l].hl /= fnm; acc[l].hh /= fnm; }
        acc[3].ll = ll4 / fnm;
        // inverse: every lane rebuilds its own LL chain (no communication needed)
        acc[2].ll = haar_inv(acc[3], (lane >> 5) & 1, (lane >> 2) & 1);
        acc[1].ll = haar_inv(acc[2], (lane >> 4) & 1, (lane >> 1) & 1);
        acc[0].ll = haar_inv(acc[1], (lane >> 3) & 1, lane & 1);
        sum[0] += haar_inv(acc[0], 0, 0);
        sum[1] += haar_inv(acc[0], 0, 1);
        sum[2] += haar_inv(acc[0], 1, 0);
        sum[3] += haar_inv(acc[0], 1, 1);
    };
    if constexpr (NMOD > 0) {
#pragma unroll
        for (int c = 0; c < NC; ++c) channel(c);
    } else {
        for (int c = 0; c < C; ++c) channel(c);
    }
    // sign(mean over channels) == sign(sum); torch.mean then .gt(0) (Frequency.py:44,54)
    const float cdiv = (float)C;
    int cnt = ((sum[0] / cdiv) > 0.f) + ((sum[1] / cdiv) > 0.f) + ((sum[2] / cdiv) > 0.f) + ((sum[3] / cdiv) > 0.f);
    cnt = wave_sum_i(cnt);
    if (lane == 0) counts[tile] = cnt;
}

static int freq_counts_launch(const float* m0, const float* m1, const float* m2, const float* m3, int nmod, int B, int C, int H, int W,
                              int32_t* counts, hipStream_t stream)
{
    const long tiles = (long)B * (H >> 4) * (W >> 4);
    const dim3 grid((unsigned)((tiles + 3) / 4)), block(256);
    if (nmod == 3 && C == 3)
        hipLaunchKernelGGL((freq_counts_kernel<3, 3>), grid, block, 0, stream, m0, m1, m2, m3, nmod, B, C, H, W, counts);
    else if (nmod == 4 && C == 3)
        hipLaunchKernelGGL((freq_counts_kernel<4, 3>), grid, block, 0, stream, m0, m1, m2, m3, nmod, B, C, H, W, counts);
    else
        hipLaunchKernelGGL((freq_counts_kernel<0, 0>), grid, block, 0, stream, m0, m1, m2, m3, nmod, B, C, H, W, counts);
    EDITOR_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// K7/K9b  top-k with torch.topk's CPU tie order (libstdc++ partial_sort / nth_element on (value,index)
//         pairs).  One lane owns one row; rows live lane-interleaved in LDS.
// ------------------------------------------------------------------------------------------------
template <typename V> struct Row {
    V* val; uint16_t* idx; int stride;      // element j of this lane's row at [j*stride]
    __device__ __forceinline__ V& v(int j) { return val[j * stride]; }
    __device__ __forceinline__ uint16_t& i(int j) { return idx[j * stride]; }
};
template <typename V> struct Pair { V v; uint16_t i; };

__device__ __forceinline__ bool before(float x, float y) { return (isnan(x) && !isnan(y)) || (x > y); }
__device__ __forceinline__ bool before(int x, int y) { return x > y; }

template <typename V> __device__ __forceinline__ Pair<V> get(Row<V>& r, int j) { return Pair<V>{r.v(j), r.i(j)}; }
template <typename V> __device__ __forceinline__ void put(Row<V>& r, int j, Pair<V> p) { r.v(j) = p.v; r.i(j) = p.i; }
template <typename V> __device__ __forceinline__ void swp(Row<V>& r, int a, int b) {
    Pair<V> t = get(r, a); put(r, a, get(r, b)); put(r, b, t);
}

// heap over r[base .. base+len)
template <typename V> __device__ void push_heap_(Row<V>& r, int base, int hole, int top, Pair<V> val) {
    int parent = (hole - 1) / 2;
    while (hole > top && before(r.v(base + parent), val.v)) {
        put(r, base + hole, get(r, base + parent));
        hole = parent;
        parent = (hole - 1) / 2;
    }
    put(r, base + hole, val);
}
template <typename V> __device__ void adjust_heap_(Row<V>& r, int base, int hole, int len, Pair<V> val) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (before(r.v(base + child), r.v(base + child - 1))) child--;
        put(r, base + hole, get(r, base + child));
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        put(r, base + hole, get(r, base + child - 1));
        hole = child - 1;
    }
    push_heap_(r, base, hole, top, val);
}
template <typename V> __device__ void heap_select_(Row<V>& r, int base, int middle, int last) {
    if (middle >= 2) {
        int parent = (middle - 2) / 2;
        for (;;) {
            adjust_heap_(r, base, parent, middle, get(r, base + parent));
            if (parent == 0) break;
            parent--;
        }
    }
    for (int i = middle; i < last; ++i)
        if (before(r.v(base + i), r.v(base))) {
            Pair<V> val = get(r, base + i);
            put(r, base + i, get(r, base));
            adjust_heap_(r, base, 0, middle, val);
        }
}
template <typename V> __device__ void insertion_sort_(Row<V>& r, int first, int last) {
    if (first == last) return;
    for (int i = first + 1; i != last; ++i) {
        Pair<V> val = get(r, i);
        if (before(val.v, r.v(first))) {
            for (int j = i; j > first; --j) put(r, j, get(r, j - 1));
            put(r, first, val);
        } else {
            int cur = i, next = i - 1;
            while (before(val.v, r.v(next))) { put(r, cur, get(r, next)); cur = next; --next; }
            put(r, cur, val);
        }
    }
}
template <typename V> __device__ void introselect_(Row<V>& r, int first, int nth, int last, int depth) {
    while (last - first > 3) {
        if (depth == 0) {
            heap_select_(r, first, nth + 1 - first, last - first);
            swp(r, first, nth);
            return;
        }
        --depth;
        // __unguarded_partition_pivot: median of (first+1, mid, last-1) moved to first
        const int mid = first + (last - first) / 2;
        const int a = first + 1, b = mid, c = last - 1;
        const V va = r.v(a), vb = r.v(b), vc = r.v(c);
        int med;
        if (before(va, vb)) med = before(vb, vc) ? b : (before(va, vc) ? c : a);
        else                med = before(va, vc) ? a : (before(vb, vc) ? c : b);
        swp(r, first, med);
        int lo = first + 1, hi = last;
        const V pv = r.v(first);
        for (;;) {
            while (before(r.v(lo), pv)) ++lo;
            --hi;
            while (before(pv, r.v(hi))) --hi;
            if (!(lo < hi)) break;
            swp(r, lo, hi);
            ++lo;
        }
        if (lo <= nth) first = lo; else last = lo;
    }
    insertion_sort_(r, first, last);
}

template <typename V>
__global__ __launch_bounds__(64) void topk_mask_kernel(const V* __restrict__ vals, int rows, int n, int k,
                                                       int group, uint8_t* __restrict__ mask, int rows_per_block)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    V* sval = reinterpret_cast<V*>(smem);
    uint16_t* sidx = reinterpret_cast<uint16_t*>(smem + (size_t)rows_per_block * n * sizeof(V));
    const int lane = threadIdx.x;
    const long row0 = (long)blockIdx.x * rows_per_block;
    // coalesced fill: consecutive threads read consecutive elements of the block's rows
    const long total = (long)rows_per_block * n;
    for (long e = lane; e < total; e += 64) {
        const int rl = (int)(e / n), j = (int)(e % n);
        if (row0 + rl < rows) {
            sval[j * rows_per_block + rl] = vals[(row0 + rl) * n + j];
            sidx[j * rows_per_block + rl] = (uint16_t)j;
        }
    }
    __syncthreads();
    const long row = row0 + lane;
    if (lane >= rows_per_block || row >= rows) return;
    Row<V> r{sval + lane, sidx + lane, rows_per_block};
    if ((long)k * 64 <= n) heap_select_(r, 0, k, n);             // std::partial_sort's selection half
    else {
        int lg = 0; for (int t = n; t > 1; t >>= 1) ++lg;
        introselect_(r, 0, k - 1, n, 2 * lg);                    // std::nth_element
    }
    uint8_t* out = mask + (row / group) * (long)n;
    for (int j = 0; j < k; ++j) out[r.i(j)] = 1;                 // benign same-value races across heads
}

// ------------------------------------------------------------------------------------------------
// K6  attention rollout, row-vector form:  r = e0^T A_{L-1};  r <- r A_l  (l = L-2 .. 0)
//     probs layout [L][Bp][heads][T][T] fp32 (softmax outputs of the backbone, vit_pytorch.py:190)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void rollout_kernel(const float* __restrict__ probs, int L, long layer_stride,
                                                      int T, int ldp, float* __restrict__ scores)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* r = reinterpret_cast<float*>(smem);          // [T]
    float* part = r + ((T + 3) & ~3);                    // [G][T]
    const long bh = blockIdx.x;
    const int G = max(1, (int)blockDim.x / T);           // row groups (T > blockDim: one group, columns strided)
    const float* A = probs + (long)(L - 1) * layer_stride + bh * (long)T * ldp;
    for (int t = threadIdx.x; t < T; t += blockDim.x) r[t] = A[t];           // CLS row of the last layer
    __syncthreads();
    for (int l = L - 2; l >= 0; --l) {
        A = probs + (long)l * layer_stride + bh * (long)T * ldp;
        for (int e = threadIdx.x; e < G * T; e += blockDim.x) {
            const int g = e / T, j = e % T;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            int i = g;
            for (; i + 3 * G < T; i += 4 * G) {
                a0 += r[i] * A[(long)i * ldp + j];
                a1 += r[i + G] * A[(long)(i + G) * ldp + j];
                a2 += r[i + 2 * G] * A[(long)(i + 2 * G) * ldp + j];
                a3 += r[i + 3 * G] * A[(long)(i + 3 * G) * ldp + j];
            }
            for (; i < T; i += G) a0 += r[i] * A[(long)i * ldp + j];
            part[g * T + j] = (a0 + a1) + (a2 + a3);
        }
        __syncthreads();
        for (int t = threadIdx.x; t < T; t += blockDim.x) {
            float s = 0.f;
            for (int q = 0; q < G; ++q) s += part[q * T + t];
            r[t] = s;
        }
        __syncthreads();
    }
    for (int t = threadIdx.x + 1; t < T; t += blockDim.x) scores[bh * (long)(T - 1) + t - 1] = r[t];
}

// ------------------------------------------------------------------------------------------------
// mask utilities
// ------------------------------------------------------------------------------------------------
__global__ void mask_or4_kernel(const uint8_t* a, const uint8_t* b, const uint8_t* c, const uint8_t* d,
                                uint8_t* out, long n)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (a[i] | (b ? b[i] : 0) | (c ? c[i] : 0) | (d ? d[i] : 0)) ? 1 : 0;
}

}  // namespace

extern "C" int editor_freq_counts_f32(const float* rgb, const float* nir, const float* tir, int B, int C, int H,
                                      int W, int32_t* counts, hipStream_t stream)
{
    if ((H & 15) || (W & 15) || B <= 0 || C <= 0) return (int)hipErrorInvalidValue;
    const int nmod = tir ? 3 : 2;
    return freq_counts_launch(rgb, nir, tir, nullptr, nmod, B, C, H, W, counts, stream);
}

extern "C" int editor_freq_counts_nmod_f32(const float* m0, const float* m1, const float* m2, const float* m3, int nmod,
                                           int B, int C, int H, int W, int32_t* counts, hipStream_t stream)
{
    if ((H & 15) || (W & 15) || B <= 0 || C <= 0 || nmod < 2 || nmod > 4 || !m0 || !m1 || (nmod > 2 && !m2) || (nmod > 3 && !m3))
        return (int)hipErrorInvalidValue;
    return freq_counts_launch(m0, m1, m2, m3, nmod, B, C, H, W, counts, stream);
}

template <typename V>
static int topk_mask_launch(const V* vals, int rows, int n, int k, int group, uint8_t* mask, hipStream_t stream)
{
    if (k <= 0 || k > n || n > 4096 || group <= 0 || rows % group) return (int)hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(mask, 0, (size_t)(rows / group) * n, stream);
    if (e != hipSuccess) return (int)e;
    int rpb = 64;
    while ((size_t)rpb * n * (sizeof(V) + 2) > 96 * 1024 && rpb > 1) rpb >>= 1;
    const size_t lds = (size_t)rpb * n * (sizeof(V) + 2);
    auto kern = topk_mask_kernel<V>;
    if (lds > 48 * 1024) {
        e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, dim3((rows + rpb - 1) / rpb), dim3(64), lds, stream, vals, rows, n, k, group, mask, rpb);
    EDITOR_LAUNCH_CHECK();
    return 0;
}

extern "C" int editor_topk_mask_i32(const int32_t* vals, int rows, int n, int k, int group, uint8_t* mask,
                                    hipStream_t stream)
{ return topk_mask_launch<int>(vals, rows, n, k, group, mask, stream); }

extern "C" int editor_topk_mask_f32(const float* vals, int rows, int n, int k, int group, uint8_t* mask,
                                    hipStream_t stream)
{ return topk_mask_launch<float>(vals, rows, n, k, group, mask, stream); }

extern "C" int editor_attn_rollout_f32(const float* probs, int L, int BH, int T, int ldp, long layer_stride, float* scores,
                                       hipStream_t stream)
{
    if (L < 1 || T < 2 || T > 1024 || ldp < T) return (int)hipErrorInvalidValue;
    int threads = (512 / T) * T;                       // G full row-groups of T threads
    if (threads < 64) threads = T < 512 ? T : 512;
    const int G = threads / T > 0 ? threads / T : 1;
    const size_t lds = (size_t)(((T + 3) & ~3) + (size_t)G * T) * sizeof(float);
    hipLaunchKernelGGL(rollout_kernel, dim3(BH), dim3(threads), lds, stream, probs, L, layer_stride, T, ldp, scores);
    EDITOR_LAUNCH_CHECK();
    return 0;
}

extern "C" int editor_mask_or(const uint8_t* a, const uint8_t* b, const uint8_t* c, const uint8_t* d, uint8_t* out,
                              long n, hipStream_t stream)
{
    hipLaunchKernelGGL(mask_or4_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a, b, c, d, out, n);
    EDITOR_LAUNCH_CHECK();
    return 0;
}
