// Memory-bound row kernels of the EDITOR hot path, gfx950.
// All HBM-bound: one wavefront per token row, 16-byte vector accesses, fp32 statistics.
//   LayerNorm fwd/bwd         nn.LayerNorm in Block / BlockMask (vit_pytorch.py:206,211,215-220,268-297)
//   exact-erf GELU fwd/bwd    nn.GELU() in Mlp / MlpMasked     (vit_pytorch.py:130,141,149,163)
//   im2col for the 16x16 patch conv + token assembly (cls / pos / SIE)   (vit_pytorch.py:449-458,625-637)
//   SFTS mask apply + background-consistency loss fwd/bwd               (SFTS.py:208-225)
//   masked-mean pooling fwd/bwd                                          (make_model.py:186-203)
//   column sums (bias grads), partial-row reductions, dtype casts
#include "common.h"
#include "../../include/editor_hip.h"
#include <string.h>

namespace {

template <typename T> struct Vec4;
template <> struct Vec4<float> {
    static __device__ __forceinline__ float4 ld(const float* p) { return *reinterpret_cast<const float4*>(p); }
    static __device__ __forceinline__ void st(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
    static __device__ __forceinline__ float4 rnd(float4 v) { return v; }
};
template <> struct Vec4<bf16_t> {
    static __device__ __forceinline__ float4 ld(const bf16_t* p) {
        const uint2 u = *reinterpret_cast<const uint2*>(p);
        return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
                           __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
    }
    static __device__ __forceinline__ void st(bf16_t* p, float4 v) {
        uint2 u; u.x = pack_bf16x2(v.x, v.y); u.y = pack_bf16x2(v.z, v.w);
        *reinterpret_cast<uint2*>(p) = u;
    }
    static __device__ __forceinline__ float4 rnd(float4 v) {          // the values st() stores, back in fp32
        const float2_t_ a = H16<false>::unpack2(pack_bf16x2(v.x, v.y)), b = H16<false>::unpack2(pack_bf16x2(v.z, v.w));
        return make_float4(a.x, a.y, b.x, b.y);
    }
};

template <> struct Vec4<f16_t> {
    static __device__ __forceinline__ float4 ld(const f16_t* p) {
        const uint2 u = *reinterpret_cast<const uint2*>(p);
        const float2_t_ a = H16<true>::unpack2(u.x), b = H16<true>::unpack2(u.y);
        return make_float4(a.x, a.y, b.x, b.y);
    }
    static __device__ __forceinline__ void st(f16_t* p, float4 v) {
        uint2 u; u.x = pack_f16x2(v.x, v.y); u.y = pack_f16x2(v.z, v.w);
        *reinterpret_cast<uint2*>(p) = u;
    }
    static __device__ __forceinline__ float4 rnd(float4 v) {
        const float2_t_ a = H16<true>::unpack2(pack_f16x2(v.x, v.y)), b = H16<true>::unpack2(pack_f16x2(v.z, v.w));
        return make_float4(a.x, a.y, b.x, b.y);
    }
};

// split-precision pair (COMPUTE_DTYPE 'f16x2'): hi = round(v), lo = round(v - hi) (v - hi is exact in fp32)
template <typename T>
__device__ __forceinline__ void st_split(T* hi, T* lo, float4 v)
{
    const float4 h = Vec4<T>::rnd(v);
    Vec4<T>::st(hi, v);
    Vec4<T>::st(lo, make_float4(v.x - h.x, v.y - h.y, v.z - h.z, v.w - h.w));
}

constexpr int kMaxV = 4;   // float4 chunks per lane: D <= 64*4*4 = 1024

// ------------------------------------------------------------------------------------------------
// LayerNorm forward: y = (x-mean)*rstd*gamma+beta, optionally * rowmask (AttentionMask / MlpMasked zero
// the LN output of unselected tokens, vit_pytorch.py:245,162).  One wave per row.
// ------------------------------------------------------------------------------------------------
// RAGGED: D is a multiple of 4 but not of 256 (DeiT-small's 384): the last column group is guarded per lane.  A separate
// instantiation, so that the 768 / 1024-wide path keeps its code (and its bits).
template <typename T, bool SPLIT = false, bool RAGGED = false>     // SPLIT: y as the half pair (y, y_lo) of the split-precision forward
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
    const float* __restrict__ beta, float eps, long M, int D, const uint8_t* __restrict__ rowmask, int mask_period,
    T* __restrict__ y, float* __restrict__ mean_out, float* __restrict__ rstd_out, const int* __restrict__ m_live,
    T* __restrict__ y_lo = nullptr)
{
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m_live) M = min(M, (long)((*m_live + 63) & ~63));      // live rows, rounded up to the GEMM reduction tile (mask = 0 there)
    if (row >= M) return;
    const int nv = RAGGED ? (D + 255) >> 8 : D >> 8;        // column groups of 64 lanes x 4
    const float* xr = x + row * D;
    float4 v[kMaxV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxV; ++i) if (i < nv && (!RAGGED || (i * 64 + lane) * 4 < D)) {
        v[i] = *reinterpret_cast<const float4*>(xr + (i * 64 + lane) * 4);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxV; ++i) if (i < nv && (!RAGGED || (i * 64 + lane) * 4 < D)) {
        const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
        q += (a * a + b * b) + (c * c + d * d);
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
    const float keep = (rowmask && !rowmask[mask_period ? row % mask_period : row]) ? 0.f : 1.f;
    T* yr = y + row * D;
#pragma unroll
    for (int i = 0; i < kMaxV; ++i) if (i < nv && (!RAGGED || (i * 64 + lane) * 4 < D)) {
        const int c0 = (i * 64 + lane) * 4;
        const float4 g = *reinterpret_cast<const float4*>(gamma + c0);
        const float4 bt = *reinterpret_cast<const float4*>(beta + c0);
        float4 o;
        o.x = ((v[i].x - mean) * rstd * g.x + bt.x) * keep;
        o.y = ((v[i].y - mean) * rstd * g.y + bt.y) * keep;
        o.z = ((v[i].z - mean) * rstd * g.z + bt.z) * keep;
        o.w = ((v[i].w - mean) * rstd * g.w + bt.w) * keep;
        if constexpr (SPLIT) st_split(yr + c0, y_lo + row * D + c0, o);
        else Vec4<T>::st(yr + c0, o);
    }
    if (lane == 0) { if (mean_out) mean_out[row] = mean; if (rstd_out) rstd_out[row] = rstd; }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm forward onto COMPACTED rows (round 6: stochastic depth, vit_pytorch.py:52-69,218): a branch whose drop-path draw
// is 0 for a sample contributes x + 0 - its LayerNorm / fc1 / fc2 rows need not be computed.  editor_droppath_plan orders the
// token rows of a (block, branch) live samples first; this kernel writes row r's output to row perm[r] of y, so that the
// products that follow run on the live prefix (m_live) only.  A dead row (rowscale[r] == 0) writes ZEROS to its slot (rows
// [live, ...) of y are zero by contract: the partly live last tile and the weight gradient's last reduction tile read them) and
// copies its x row to copy_out - the block output of a dropped sample is its input (the fc2 epilogue scatters the live rows
// only).  mean / rstd stay indexed by the original row.  Same per-row arithmetic as layernorm_fwd_kernel.  D % 256 == 0.
// ------------------------------------------------------------------------------------------------
template <typename T, bool SPLIT = false>       // SPLIT: y as the half pair (y, y_lo) of the split-precision forward
__global__ __launch_bounds__(256) void layernorm_fwd_perm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
    const float* __restrict__ beta, float eps, long M, int D, T* __restrict__ y, float* __restrict__ mean_out,
    float* __restrict__ rstd_out, const int* __restrict__ perm, const float* __restrict__ rowscale, float* __restrict__ copy_out,
    T* __restrict__ y_lo = nullptr)
{
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nv = D >> 8;
    const float* xr = x + row * D;
    const bool dead = rowscale[row] == 0.f;
    float4 v[kMaxV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxV; ++i) if (i < nv) {
        v[i] = *reinterpret_cast<const float4*>(xr + (i * 64 + lane) * 4);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxV; ++i) if (i < nv) {
        const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
        q += (a * a + b * b) + (c * c + d * d);
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
    const long slot = perm[row];
    T* yr = y + slot * D;
    if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
    if (dead) {                                       // (wave-uniform: one wave per row)
#pragma unroll
        for (int i = 0; i < kMaxV; ++i) if (i < nv) {
            const int c0 = (i * 64 + lane) * 4;
            Vec4<T>::st(yr + c0, make_float4(0.f, 0.f, 0.f, 0.f));
            if constexpr (SPLIT) Vec4<T>::st(y_lo + slot * D + c0, make_float4(0.f, 0.f, 0.f, 0.f));
            *reinterpret_cast<float4*>(copy_out + row * D + c0) = v[i];
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < kMaxV; ++i) if (i < nv) {
        const int c0 = (i * 64 + lane) * 4;
        const float4 g = *reinterpret_cast<const float4*>(gamma + c0);
        const float4 bt = *reinterpret_cast<const float4*>(beta + c0);
        float4 o;
        o.x = (v[i].x - mean) * rstd * g.x + bt.x;
        o.y = (v[i].y - mean) * rstd * g.y + bt.y;
        o.z = (v[i].z - mean) * rstd * g.z + bt.z;
        o.w = (v[i].w - mean) * rstd * g.w + bt.w;
        if constexpr (SPLIT) st_split(yr + c0, y_lo + slot * D + c0, o);
        else Vec4<T>::st(yr + c0, o);
    }
}

// ------------------------------------------------------------------------------------------------
// Residual add + LayerNorm forward in one pass (round 4, bf16 mode):  x_out = x + rowscale[row] * branch,  y = LN(x_out).
// `branch` is the 16-bit output of the projection / fc2 product written with the PLAIN epilogue: the residual add leaves the
// GEMM's fp32 epilogue - 512 KiB of HBM-miss traffic per 256 x 256 tile at the ~10 B/clk a CU that is owned by one workgroup can
// pull - for this streaming kernel (30 waves per CU).  Measured at M = 49 536 (tools/residual_pricing.py): -17 us per
// projection + LayerNorm pair, -13 .. -16 us per fc2 + LayerNorm pair.  Price: the branch is rounded to 16 bits before the add
// (cls4t at B = 128: x 1.068, tools/branch16_accuracy.py) - cfg.MODEL.BRANCH16, on in bf16 mode only.  Dense rows (no row mask,
// no live-row count), D a multiple of 256; same per-row arithmetic as layernorm_fwd_kernel on x_out.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void resid_add_layernorm_fwd_kernel(const float* __restrict__ x, const T* __restrict__ branch,
    const float* __restrict__ rowscale, const float* __restrict__ gamma, const float* __restrict__ beta, float eps, long M, int D,
    float* __restrict__ x_out, T* __restrict__ y, float* __restrict__ mean_out, float* __restrict__ rstd_out)
{
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nv = D >> 8;
    const float rs = rowscale ? rowscale[row] : 1.f;
    const float* xr = x + row * D;
    const T* br = branch + row * D;
    float4 v[kMaxV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxV; ++i) if (i < nv) {
        const int c0 = (i * 64 + lane) * 4;
        const float4 a = *reinterpret_cast<const float4*>(xr + c0);
        const float4 b = Vec4<T>::ld(br + c0);
        v[i] = make_float4(a.x + rs * b.x, a.y + rs * b.y, a.z + rs * b.z, a.w + rs * b.w);
        *reinterpret_cast<float4*>(x_out + row * D + c0) = v[i];
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxV; ++i) if (i < nv) {
        const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
        q += (a * a + b * b) + (c * c + d * d);
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
    T* yr = y + row * D;
#pragma unroll
    for (int i = 0; i < kMaxV; ++i) if (i < nv) {
        const int c0 = (i * 64 + lane) * 4;
        const float4 g = *reinterpret_cast<const float4*>(gamma + c0);
        const float4 bt = *reinterpret_cast<const float4*>(beta + c0);
        float4 o;
        o.x = (v[i].x - mean) * rstd * g.x + bt.x;
        o.y = (v[i].y - mean) * rstd * g.y + bt.y;
        o.z = (v[i].z - mean) * rstd * g.z + bt.z;
        o.w = (v[i].w - mean) * rstd * g.w + bt.w;
        Vec4<T>::st(yr + c0, o);
    }
    if (lane == 0) { if (mean_out) mean_out[row] = mean; if (rstd_out) rstd_out[row] = rstd; }
}

__device__ __forceinline__ float round_like(float v, float) { return v; }
__device__ __forceinline__ float round_like(float v, bf16_t) { return bf16_to_f32(f32_to_bf16(v)); }
__device__ __forceinline__ float round_like(float v, f16_t) { return f16_to_f32(f32_to_f16(v)); }

// ------------------------------------------------------------------------------------------------
// LayerNorm backward.  dx_out = dx_in (optional residual-branch gradient) + d/dx LN;  per-block partial
// dgamma/dbeta rows go to `partials` [gridDim.x][2][D] and are folded by reduce_rows_kernel (deterministic).
// Optional second output (cast_out != NULL): the 16-bit, row-scaled copy of dx_out the NEXT linear layer's backward
// consumes - cast_out = round(dx_out * cast_rowscale[row] * cast_scale), with the column sums of the rounded values
// (that layer's bias gradient) as per-block partial rows in cast_partials [gridDim.x][D] - i.e. what
// cast_rows_colsum_kernel would produce from dx_out in a second pass over it (152 MB re-read per call at D = 768).
// ------------------------------------------------------------------------------------------------
// NV = D / 256 (float4 column groups per lane) and CAST (the second output) are compile-time: with worst-case-sized arrays
// and the cast accumulators always present the kernel needed 150 registers (3 waves per SIMD) instead of <= 128.
// PERM (round 6, stochastic-depth compaction - see layernorm_fwd_perm_kernel): dy may live on compacted rows (row r's gradient at row
// dy_perm[r]; a row whose slot is >= *dy_live belongs to a dropped sample: its gradient is zero and is NOT read - the products
// behind it skipped those tiles), and the cast output may go to compacted rows of the consumer branch (cast_perm[r]; a dropped
// row writes the zero its row scale makes of it).  A separate instantiation: the dense path keeps its code.
template <typename T, int NV, bool CAST, bool RAGGED = false, bool PERM = false>       // RAGGED: NV = ceil(D / 256), last group guarded (see the forward)
__global__ __launch_bounds__(256, NV <= 3 ? 4 : 3) void layernorm_bwd_kernel(const T* __restrict__ dy, const float* __restrict__ x,
    const float* __restrict__ gamma, const float* __restrict__ mean_in, const float* __restrict__ rstd_in, long M, int D,
    const uint8_t* __restrict__ rowmask, int mask_period, const float* __restrict__ dx_in, float* __restrict__ dx_out,
    float* __restrict__ partials, const int* __restrict__ m_live, float dy_scale, T* __restrict__ cast_out,
    const float* __restrict__ cast_rowscale, float cast_scale, float* __restrict__ cast_partials,
    const int* __restrict__ dy_perm = nullptr, const int* __restrict__ dy_live = nullptr, const int* __restrict__ cast_perm = nullptr)
{
    __shared__ float red[4][2][1024];
    if (m_live) M = min(M, (long)((*m_live + 63) & ~63));
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float4 g[NV], dg[NV], db[NV], ccs[CAST ? NV : 1];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        dg[i] = make_float4(0.f, 0.f, 0.f, 0.f); db[i] = dg[i];
        if (CAST) ccs[i] = dg[i];
        g[i] = (!RAGGED || (i * 64 + lane) * 4 < D) ? *reinterpret_cast<const float4*>(gamma + (i * 64 + lane) * 4) : dg[i];
    }
    const long dlive = (PERM && dy_live) ? (long)*dy_live : 0L;
    for (long row = (long)blockIdx.x * 4 + w; row < M; row += (long)gridDim.x * 4) {
        bool keep = !(rowmask && !rowmask[mask_period ? row % mask_period : row]);
        long drow = row, crow = row;
        if constexpr (PERM) {
            if (dy_perm) { drow = dy_perm[row]; keep = drow < dlive; }
            if (cast_perm) crow = cast_perm[row];
        }
        const float mean = mean_in[row], rstd = rstd_in[row];
        float4 xh[NV], d[NV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c0 = (i * 64 + lane) * 4;
            if (RAGGED && c0 >= D) { xh[i] = make_float4(0.f, 0.f, 0.f, 0.f); d[i] = xh[i]; continue; }   // (adds zeros to every sum)
            const float4 xv = *reinterpret_cast<const float4*>(x + row * D + c0);
            d[i] = keep ? Vec4<T>::ld(dy + drow * D + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
            d[i].x *= dy_scale; d[i].y *= dy_scale; d[i].z *= dy_scale; d[i].w *= dy_scale;   // (loss-scaled f16 gradients)
            xh[i] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
            dg[i].x += d[i].x * xh[i].x; dg[i].y += d[i].y * xh[i].y; dg[i].z += d[i].z * xh[i].z; dg[i].w += d[i].w * xh[i].w;
            db[i].x += d[i].x; db[i].y += d[i].y; db[i].z += d[i].z; db[i].w += d[i].w;
            d[i].x *= g[i].x; d[i].y *= g[i].y; d[i].z *= g[i].z; d[i].w *= g[i].w;
            s1 += (d[i].x + d[i].y) + (d[i].z + d[i].w);
            s2 += (d[i].x * xh[i].x + d[i].y * xh[i].y) + (d[i].z * xh[i].z + d[i].w * xh[i].w);
        }
        s1 = wave_sum(s1) / (float)D;
        s2 = wave_sum(s2) / (float)D;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c0 = (i * 64 + lane) * 4;
            if (RAGGED && c0 >= D) continue;
            float4 o;
            o.x = rstd * (d[i].x - s1 - xh[i].x * s2);
            o.y = rstd * (d[i].y - s1 - xh[i].y * s2);
            o.z = rstd * (d[i].z - s1 - xh[i].z * s2);
            o.w = rstd * (d[i].w - s1 - xh[i].w * s2);
            if (dx_in) {
                const float4 r = *reinterpret_cast<const float4*>(dx_in + row * D + c0);
                o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
            }
            *reinterpret_cast<float4*>(dx_out + row * D + c0) = o;
            if constexpr (CAST) {
                const float r = (cast_rowscale ? cast_rowscale[row] : 1.f) * cast_scale;
                o.x *= r; o.y *= r; o.z *= r; o.w *= r;
                Vec4<T>::st(cast_out + crow * D + c0, o);
                ccs[i].x += round_like(o.x, T{}); ccs[i].y += round_like(o.y, T{});
                ccs[i].z += round_like(o.z, T{}); ccs[i].w += round_like(o.w, T{});
            }
        }
    }
    if (CAST && cast_partials) {                         // (block-uniform)
#pragma unroll
        for (int i = 0; i < NV; ++i) *reinterpret_cast<float4*>(&red[w][0][(i * 64 + lane) * 4]) = ccs[CAST ? i : 0];
        __syncthreads();
        for (int c = threadIdx.x; c < D; c += 256)
            cast_partials[(long)blockIdx.x * D + c] = (red[0][0][c] + red[1][0][c]) + (red[2][0][c] + red[3][0][c]);
        __syncthreads();
    }
    if (!partials) return;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c0 = (i * 64 + lane) * 4;
        *reinterpret_cast<float4*>(&red[w][0][c0]) = dg[i];
        *reinterpret_cast<float4*>(&red[w][1][c0]) = db[i];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * D; c += 256) {
        const int which = c / D, col = c % D;
        partials[((long)blockIdx.x * 2 + which) * D + col] =
            (red[0][which][col] + red[1][which][col]) + (red[2][which][col] + red[3][which][col]);
    }
}

// out[c] (+)= scale * sum_p partials[p][c].   Block = 64 columns x 16 row-groups, eight independent partial sums per
// thread (the kernel is pure load latency: P = 1024 rows through 4 row-groups x 4 sums took 19 us per call, 87 calls
// per step); fixed summation order, so the result is run-to-run deterministic.
__global__ __launch_bounds__(1024) void reduce_rows_kernel(const float* __restrict__ partials, int P, long ncol,
                                                           float* __restrict__ out, int accumulate, float scale)
{
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const long c = (long)blockIdx.x * 64 + lane;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c < ncol) {
        int p = rg;
        for (; p + 112 < P; p += 128) {
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u] += partials[(long)(p + 16 * u) * ncol + c];
        }
        for (; p < P; p += 16) s[0] += partials[(long)p * ncol + c];
    }
    red[rg][lane] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    __syncthreads();
    if (rg == 0 && c < ncol) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += red[r][lane];
        t *= scale;
        out[c] = accumulate ? out[c] + t : t;
    }
}

// column sums of a (M,N) activation-gradient matrix -> partials [gridDim.y][N]  (bias gradients).
// Block = 256 columns (64 lanes x 4) x 4 row-groups; grid.y row chunks of `rows_per` rows.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ dy, long M, int N, long ld, int rows_per,
                                                     float* __restrict__ partials)
{
    __shared__ float4 red[4][64];
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c0 = (blockIdx.x * 64 + lane) * 4;
    const long r0 = (long)blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (c0 < N) {
        long r = r0 + rg;
        for (; r + 4 < r1; r += 8) {
            const float4 u = Vec4<T>::ld(dy + r * ld + c0), v = Vec4<T>::ld(dy + (r + 4) * ld + c0);
            a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
            b.x += v.x; b.y += v.y; b.z += v.z; b.w += v.w;
        }
        if (r < r1) { const float4 u = Vec4<T>::ld(dy + r * ld + c0); a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w; }
    }
    red[rg][lane] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    __syncthreads();
    if (rg == 0 && c0 < N) {
        const float4 p = red[0][lane], q = red[1][lane], r = red[2][lane], t = red[3][lane];
        *reinterpret_cast<float4*>(partials + (long)blockIdx.y * N + c0) =
            make_float4((p.x + q.x) + (r.x + t.x), (p.y + q.y) + (r.y + t.y), (p.z + q.z) + (r.z + t.z), (p.w + q.w) + (r.w + t.w));
    }
}

// ------------------------------------------------------------------------------------------------
// GELU (exact erf) forward / backward, elementwise on 4-element vectors
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float gelu_f(float a) { return 0.5f * a * (1.f + erff(a * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float a) {
    return 0.5f * (1.f + erff(a * 0.70710678118654752f)) + a * 0.3989422804014327f * __expf(-0.5f * a * a);
}
template <typename T>
__global__ void gelu_fwd_kernel(const T* __restrict__ a, T* __restrict__ g, long n4)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 v = Vec4<T>::ld(a + i * 4);
        v.x = gelu_f(v.x); v.y = gelu_f(v.y); v.z = gelu_f(v.z); v.w = gelu_f(v.w);
        Vec4<T>::st(g + i * 4, v);
    }
}
template <typename T>
__global__ void gelu_bwd_kernel(const T* __restrict__ a, const T* __restrict__ dg, T* __restrict__ da, long n4)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 v = Vec4<T>::ld(a + i * 4), d = Vec4<T>::ld(dg + i * 4);
        Vec4<T>::st(da + i * 4, make_float4(d.x * gelu_grad_f(v.x), d.y * gelu_grad_f(v.y),
                                           d.z * gelu_grad_f(v.z), d.w * gelu_grad_f(v.w)));
    }
}

// dtype casts (fp32 master weights -> bf16 MFMA operands, fp32 grads -> bf16 GEMM operands)
template <typename TI, typename TO>
__global__ void cast_kernel(const TI* __restrict__ in, TO* __restrict__ out, long n4)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x)
        Vec4<TO>::st(out + i * 4, Vec4<TI>::ld(in + i * 4));
}

// out[m,:] = in[m,:] * rowscale[m]  (fp32 -> activation dtype): gradient of a drop-path-scaled branch
template <typename TO>
__global__ void cast_rows_kernel(const float* __restrict__ in, const float* __restrict__ rowscale, long M, int D,
                                 TO* __restrict__ out, const int* __restrict__ m_live, float scale)
{
    const int d4 = D >> 2;
    if (m_live) M = min(M, (long)((*m_live + 63) & ~63));
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < M * d4; e += (long)gridDim.x * blockDim.x) {
        const float r = (rowscale ? rowscale[(uint32_t)e / (uint32_t)d4] : 1.f) * scale;      // (M * D / 4 < 2^32: host check; 32-bit division)
        float4 v = *reinterpret_cast<const float4*>(in + e * 4);
        v.x *= r; v.y *= r; v.z *= r; v.w *= r;
        Vec4<TO>::st(out + e * 4, v);
    }
}

// The same cast with the COLUMN SUMS of its (rounded) output on the side: the bias gradient of the nn.Linear this
// gradient feeds (db = colsum(dy)) otherwise costs one more pass over dy.  One wave per row, lanes own fixed columns,
// per-block partial rows -> reduce_rows_kernel (fixed order).  D a multiple of 256, D <= 256 * kMaxV.
template <typename TO>
__global__ __launch_bounds__(256) void cast_rows_colsum_kernel(const float* __restrict__ in, const float* __restrict__ rowscale,
    long M, int D, TO* __restrict__ out, float* __restrict__ partials, float scale, const int* __restrict__ perm = nullptr)
{
    __shared__ float red[4][256 * kMaxV];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nv = D >> 8;
    float4 cs[kMaxV];
#pragma unroll
    for (int i = 0; i < kMaxV; ++i) cs[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long row = (long)blockIdx.x * 4 + w; row < M; row += (long)gridDim.x * 4) {
        const float r = (rowscale ? rowscale[row] : 1.f) * scale;
        const long orow = perm ? (long)perm[row] : row;          // (compacted consumer rows, see layernorm_fwd_perm_kernel)
#pragma unroll
        for (int i = 0; i < kMaxV; ++i) if (i < nv) {
            const int c0 = (i * 64 + lane) * 4;
            float4 v = *reinterpret_cast<const float4*>(in + row * D + c0);
            v.x *= r; v.y *= r; v.z *= r; v.w *= r;
            Vec4<TO>::st(out + orow * D + c0, v);
            cs[i].x += round_like(v.x, TO{}); cs[i].y += round_like(v.y, TO{});
            cs[i].z += round_like(v.z, TO{}); cs[i].w += round_like(v.w, TO{});
        }
    }
#pragma unroll
    for (int i = 0; i < kMaxV; ++i) if (i < nv)
        *reinterpret_cast<float4*>(&red[w][(i * 64 + lane) * 4]) = cs[i];
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256)
        partials[(long)blockIdx.x * D + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

// ------------------------------------------------------------------------------------------------
// Patch embedding glue.  im2col of non-overlapping 16x16 patches: row (b*N+p), col (c*256 + i*16 + j)
// == Conv2d(k=16,s=16) weight layout (768,3,16,16) flattened (vit_pytorch.py:438,455-457).
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void im2col16_kernel(const float* __restrict__ img, int B, int C, int H, int W, T* __restrict__ out,
                                T* __restrict__ out_lo = nullptr)
{
    const int px = W >> 4, py = H >> 4;
    const long total4 = (long)B * py * px * C * 64;                 // float4 groups
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total4; e += (long)gridDim.x * blockDim.x) {
        const int j4 = (int)(e & 3);
        // (32-bit index arithmetic: the 64-bit divisions this replaces were most of the kernel - round 6: 55 us per modality at B = 128
        //  for 75 MB; the host checks that the (patch, channel, row) count fits)
        uint32_t r = (uint32_t)(e >> 2);
        const int i = (int)(r & 15u); r >>= 4;
        const int c = (int)(r % (uint32_t)C); r /= (uint32_t)C;
        const int p = (int)(r % (uint32_t)(py * px));
        const int b = (int)(r / (uint32_t)(py * px));
        const int y = (p / px) * 16 + i, x0 = (p % px) * 16 + j4 * 4;
        const float4 v = *reinterpret_cast<const float4*>(img + (((long)b * C + c) * H + y) * W + x0);
        const long o = ((long)b * py * px + p) * (C * 256) + c * 256 + i * 16 + j4 * 4;
        if (out_lo) st_split(out + o, out_lo + o, v);
        else Vec4<T>::st(out + o, v);
    }
}

// fp32 -> split-precision half pair of in * scale (scale a power of two: GEMM weights are pre-scaled so that the low-order
// parts of |w| ~ 0.02 stay in half's normal range; the product's alpha divides it out)
__global__ void split_f32_kernel(const float* __restrict__ in, f16_t* __restrict__ hi, f16_t* __restrict__ lo, long n4, float scale)
{
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (long)gridDim.x * blockDim.x) {
        float4 v = *reinterpret_cast<const float4*>(in + e * 4);
        v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
        st_split(hi + e * 4, lo + e * 4, v);
    }
}

// x[b,0,:] = cls + pos[0] + coef*sie[cam[b]] ; x[b,1+p,:] = patch[b*N+p,:] + pos[1+p] + coef*sie[cam[b]]
// (vit_pytorch.py:627-637).  patch already contains the conv bias.  nmod: the batch holds nmod modality
// copies stacked along the sample axis that share cam labels (cam index = b % Bcam).
template <typename T>
__global__ void embed_assemble_kernel(const T* __restrict__ patch, const float* __restrict__ cls,
    const float* __restrict__ pos, const float* __restrict__ sie, const long* __restrict__ cam, int Bcam, float coef,
    long Btot, int Tn, int D, float* __restrict__ x)
{
    const int d4 = D >> 2;
    const long total = Btot * Tn * d4;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        // (32-bit index arithmetic - total < 2^32 is checked by the host; 64-bit divisions were most of these glue kernels)
        const uint32_t e32 = (uint32_t)e;
        const int c0 = (int)(e32 % (uint32_t)d4) * 4;
        const long rt = e32 / (uint32_t)d4;
        const int tk = (int)((uint32_t)rt % (uint32_t)Tn);
        const long b = (uint32_t)rt / (uint32_t)Tn;
        float4 v = tk == 0 ? *reinterpret_cast<const float4*>(cls + c0)
                           : Vec4<T>::ld(patch + (b * (Tn - 1) + tk - 1) * D + c0);
        const float4 p = *reinterpret_cast<const float4*>(pos + (long)tk * D + c0);
        v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
        if (sie) {
            const float4 s = *reinterpret_cast<const float4*>(sie + cam[(uint32_t)b % (uint32_t)Bcam] * D + c0);
            v.x += coef * s.x; v.y += coef * s.y; v.z += coef * s.z; v.w += coef * s.w;
        }
        *reinterpret_cast<float4*>(x + rt * D + c0) = v;
    }
}

// backward of the assembly: dpatch = dx[:,1:,:] (cast), dpos[t] = sum_b dx[b,t], rowsum[b] = sum_t dx[b,t]
template <typename T>
__global__ void embed_bwd_patch_kernel(const float* __restrict__ dx, long Btot, int Tn, int D, T* __restrict__ dpatch, float scale)
{
    const int d4 = D >> 2;
    const long total = Btot * (Tn - 1) * d4;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const uint32_t e32 = (uint32_t)e;
        const int c0 = (int)(e32 % (uint32_t)d4) * 4;
        const long rp = e32 / (uint32_t)d4;
        const long b = (uint32_t)rp / (uint32_t)(Tn - 1);
        const int p = (int)((uint32_t)rp % (uint32_t)(Tn - 1));
        float4 v = *reinterpret_cast<const float4*>(dx + (b * Tn + p + 1) * D + c0);
        v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
        Vec4<T>::st(dpatch + rp * D + c0, v);
    }
}
// dpos[t,:] = sum_b dx[b,t,:].  Stage 1: grid (Tn, ceil(D/1024), S) - split s sums samples b = s, s + S, ... with EIGHT
// 16-byte loads in flight per thread into partial[s][t][:] (one block per token with a 384-deep serial load chain was 150 us
// for 152 MB); stage 2 = reduce_rows_kernel over the S partial rows (fixed order: deterministic).
__global__ __launch_bounds__(256) void embed_bwd_pos_kernel(const float* __restrict__ dx, long Btot, int Tn, int D, int S,
                                                            float* __restrict__ partial)
{
    const int tk = blockIdx.x, c0 = (blockIdx.y * 256 + threadIdx.x) * 4, sp = blockIdx.z;
    if (c0 >= D) return;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long b0 = sp; b0 < Btot; b0 += 8L * S) {
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const long b = b0 + (long)j * S;
            v[j] = b < Btot ? *reinterpret_cast<const float4*>(dx + (b * Tn + tk) * D + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { a.x += v[j].x; a.y += v[j].y; a.z += v[j].z; a.w += v[j].w; }
    }
    *reinterpret_cast<float4*>(partial + ((long)sp * Tn + tk) * D + c0) = a;
}
// rowsum[b,:] = sum_t dx[b,t,:]      grid (Btot, ceil(D/1024))
__global__ void embed_bwd_rowsum_kernel(const float* __restrict__ dx, int Tn, int D, float* __restrict__ rowsum)
{
    const long b = blockIdx.x;
    const int c0 = (blockIdx.y * 256 + threadIdx.x) * 4;
    if (c0 >= D) return;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), a2 = a;
    int tk = 0;
    for (; tk + 1 < Tn; tk += 2) {
        const float4 v = *reinterpret_cast<const float4*>(dx + (b * Tn + tk) * D + c0);
        const float4 u = *reinterpret_cast<const float4*>(dx + (b * Tn + tk + 1) * D + c0);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        a2.x += u.x; a2.y += u.y; a2.z += u.z; a2.w += u.w;
    }
    if (tk < Tn) {
        const float4 v = *reinterpret_cast<const float4*>(dx + (b * Tn + tk) * D + c0);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    *reinterpret_cast<float4*>(rowsum + b * D + c0) = make_float4(a.x + a2.x, a.y + a2.y, a.z + a2.z, a.w + a2.w);
}
// dsie[c,:] = coef * sum_{b: cam[b%Bcam]==c} rowsum[b,:].  grid (ncam, ceil(D/64)), 256 threads: lane = column, the four
// waves take samples b = w, w + 4, ... (eight predicated loads in flight each) and fold through LDS in wave order - four
// blocks of 192 threads walking 384 samples one dependent load at a time took 70 us.
__global__ __launch_bounds__(256) void embed_bwd_sie_kernel(const float* __restrict__ rowsum, const long* __restrict__ cam, int Bcam, long Btot,
                                                            int D, float coef, float* __restrict__ dsie)
{
    __shared__ float red[4][64];
    const int c = blockIdx.x, col = blockIdx.y * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
    float a = 0.f;
    if (col < D) {
        for (long b0 = w; b0 < Btot; b0 += 32) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const long b = b0 + 4L * j;
                v[j] = (b < Btot && cam[b % Bcam] == c) ? rowsum[b * D + col] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) a += v[j];
        }
    }
    red[w][threadIdx.x & 63] = a;
    __syncthreads();
    if (w == 0 && col < D) dsie[(long)c * D + col] = coef * (((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x]);
}

// ------------------------------------------------------------------------------------------------
// K10  SFTS mask apply + BCC loss (SFTS.py:208-225).  feat: (nmod, B, T, D) fp32 (final-LN tokens of each
// modality), index (B, T-1) uint8.  out = feat with unselected patch rows zeroed; loss partial sums of
// sum_{pairs} (bg_a - bg_b)^2 over unselected rows -> partials[gridDim.x].
// ------------------------------------------------------------------------------------------------
// Block = D/4 threads (one 16-byte column group each), SFTS_ROWS consecutive token rows per block, the nmod loads of FOUR rows
// requested before the first is used (the grid-stride form did two 64-bit divisions per 16 bytes and kept at most nmod
// loads in flight: 3.7 TB/s on 304 MB).
constexpr int SFTS_ROWS = 8;
template <int NMOD>
__global__ __launch_bounds__(256) void sfts_apply_kernel(const float* __restrict__ feat, const uint8_t* __restrict__ index,
    long rows, int Tn, int D, long mstride, float* __restrict__ out, float* __restrict__ partials)
{
    __shared__ float red[16];
    const int c0 = threadIdx.x * 4;
    const long r0 = (long)blockIdx.x * SFTS_ROWS;
    float acc = 0.f;
#pragma unroll
    for (int rr = 0; rr < SFTS_ROWS; rr += 4) {
        float4 v[4][NMOD];
        bool sel[4], live[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long rt = r0 + rr + u;
            live[u] = rt < rows;
            const int tk = (int)(rt % Tn);
            sel[u] = live[u] && (tk == 0 || index[(rt / Tn) * (Tn - 1) + tk - 1]);
#pragma unroll
            for (int m = 0; m < NMOD; ++m)
                v[u][m] = live[u] ? *reinterpret_cast<const float4*>(feat + m * mstride + rt * D + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (!live[u]) continue;
            const long rt = r0 + rr + u;
#pragma unroll
            for (int m = 0; m < NMOD; ++m)
                *reinterpret_cast<float4*>(out + m * mstride + rt * D + c0) = sel[u] ? v[u][m] : make_float4(0.f, 0.f, 0.f, 0.f);
            if (!sel[u] && partials) {
#pragma unroll
                for (int i = 0; i < NMOD; ++i)
#pragma unroll
                    for (int j = i + 1; j < NMOD; ++j) {
                        const float a = v[u][i].x - v[u][j].x, b2 = v[u][i].y - v[u][j].y, c = v[u][i].z - v[u][j].z, d = v[u][i].w - v[u][j].w;
                        acc += (a * a + b2 * b2) + (c * c + d * d);
                    }
            }
        }
    }
    if (partials) {
        acc = block_sum(acc, red);
        if (threadIdx.x == 0) partials[blockIdx.x] = acc;
    }
}
// backward: dfeat_m = dout_m on selected rows; on unselected patch rows dfeat_m = gscale * sum_{j!=m} (f_m - f_j)
// with gscale = dloss * 2 / (B*(T-1)*D)
__global__ __launch_bounds__(256) void sfts_apply_bwd_kernel(const float* __restrict__ feat, const uint8_t* __restrict__ index,
    const float* __restrict__ dout, const float* __restrict__ dloss, float gnorm, int nmod, long B, int Tn, int D,
    float* __restrict__ dfeat)
{
    const int d4 = D >> 2;
    const long total = B * Tn * d4;
    const long mstride = B * Tn * (long)D;
    const float gs = dloss ? dloss[0] * gnorm : 0.f;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int c0 = (int)(e % d4) * 4;
        const long rt = e / d4;
        const int tk = (int)(rt % Tn);
        const long b = rt / Tn;
        const bool sel = tk == 0 || index[b * (Tn - 1) + tk - 1];
        if (sel) {
            for (int m = 0; m < nmod; ++m)
                *reinterpret_cast<float4*>(dfeat + m * mstride + rt * D + c0) =
                    *reinterpret_cast<const float4*>(dout + m * mstride + rt * D + c0);
        } else {
            float4 v[4], s = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int m = 0; m < nmod; ++m) {
                v[m] = *reinterpret_cast<const float4*>(feat + m * mstride + rt * D + c0);
                s.x += v[m].x; s.y += v[m].y; s.z += v[m].z; s.w += v[m].w;
            }
            const float n = (float)nmod;
            for (int m = 0; m < nmod; ++m)
                *reinterpret_cast<float4*>(dfeat + m * mstride + rt * D + c0) =
                    make_float4(gs * (n * v[m].x - s.x), gs * (n * v[m].y - s.y), gs * (n * v[m].z - s.z), gs * (n * v[m].w - s.w));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K12  pooling of the fused tokens (make_model.py:186-203): per (modality m, sample b):
//   cls = x[b, m*T, :] ; patch = sum_{t>=1} x[b, m*T+t, :] / num[b],  num[b] = #rows of modality 0 with row-sum != 0
// x: (B, nmod*T, D) fp32.  out: (nmod, B, 2D) = [cls, patch].  One block per sample.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pool_fwd_kernel(const float* __restrict__ x, long B, int nmod, int Tn, int D,
                                                       float* __restrict__ out, float* __restrict__ num_out)
{
    __shared__ int cnt;
    const long b = blockIdx.x;
    const float* xb = x + b * (long)nmod * Tn * D;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    // num: rows of modality 0 (RGB) whose sum over D is non-zero (make_model.py:197-198)
    for (int tk = 1 + (threadIdx.x >> 6); tk < Tn; tk += 4) {
        float s = 0.f;
        for (int c = (threadIdx.x & 63); c < D; c += 64) s += xb[(long)tk * D + c];
        s = wave_sum(s);
        if ((threadIdx.x & 63) == 0 && s != 0.f) atomicAdd(&cnt, 1);
    }
    __syncthreads();
    const float num = (float)cnt;
    if (threadIdx.x == 0) num_out[b] = num;
    for (int m = 0; m < nmod; ++m) {
        const float* xm = xb + (long)m * Tn * D;
        float* o = out + ((long)m * B + b) * 2 * D;
        for (int c = threadIdx.x; c < D; c += 256) {
            float s = 0.f;
            for (int tk = 1; tk < Tn; ++tk) s += xm[(long)tk * D + c];
            o[c] = xm[c];
            o[D + c] = s / num;
        }
    }
}
// dx[b, m*T, :] = dout[m,b,:D] ; dx[b, m*T+t, :] = dout[m,b,D:] / num[b]  (num is a count: no gradient)
__global__ void pool_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ num, long B, int nmod, int Tn,
                                int D, float* __restrict__ dx)
{
    const long total = B * nmod * Tn * (long)D;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e % D);
        long r = e / D;
        const int tk = (int)(r % Tn); r /= Tn;
        const int m = (int)(r % nmod);
        const long b = r / nmod;
        const float* o = dout + ((long)m * B + b) * 2 * D;
        dx[e] = tk == 0 ? o[c] : o[D + c] / num[b];
    }
}

// y = x * rowmask (BlockMask's final re-mask, vit_pytorch.py:331-332) in place / or backward of it
__global__ void rowmask_mul_kernel(float* __restrict__ x, const uint8_t* __restrict__ rowmask, int period, long M, int D)
{
    const int d4 = D >> 2;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < M * d4; e += (long)gridDim.x * blockDim.x) {
        const long row = e / d4;
        if (!rowmask[period ? row % period : row])
            *reinterpret_cast<float4*>(x + e * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

inline unsigned grid_for(long n, int block = 256, long cap = 256L * 16) {
    long g = (n + block - 1) / block;
    return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

// dtype code: 0 = fp32, 1 = bf16, 2 = f16
#define DISPATCH_T(dt, CALL) do { if ((dt) == 1) { using TT = bf16_t; CALL; } else if ((dt) == 2) { using TT = f16_t; CALL; } \
                                  else if ((dt) == 0) { using TT = float; CALL; } else return (int)hipErrorInvalidValue; } while (0)

extern "C" int editor_layernorm_fwd(const float* x, const float* gamma, const float* beta, float eps, long M, int D,
    const uint8_t* rowmask, int mask_period, void* y, int y_bf16, float* mean, float* rstd, const int* m_live,
    hipStream_t stream)
{
    if (D % 4 || D > 1024 || M <= 0) return (int)hipErrorInvalidValue;
    if (D % 256) {
        DISPATCH_T(y_bf16, hipLaunchKernelGGL((layernorm_fwd_kernel<TT, false, true>), dim3((unsigned)((M + 3) / 4)), dim3(256), 0,
                   stream, x, gamma, beta, eps, M, D, rowmask, mask_period, (TT*)y, mean, rstd, m_live, (TT*)nullptr));
    } else {
        DISPATCH_T(y_bf16, hipLaunchKernelGGL(layernorm_fwd_kernel<TT>, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, stream,
                   x, gamma, beta, eps, M, D, rowmask, mask_period, (TT*)y, mean, rstd, m_live));
    }
    EDITOR_LAUNCH_CHECK();
    return 0;
}

extern "C" int editor_resid_add_layernorm_fwd(const float* x, const void* branch, int b16, const float* rowscale, const float* gamma,
    const float* beta, float eps, long M, int D, float* x_out, void* y, float* mean, float* rstd, hipStream_t stream)
{
    if (D % 256 || D > 256 * kMaxV || M <= 0 || (b16 != 1 && b16 != 2) || !x || !branch || !x_out || !y) return (int)hipErrorInvalidValue;
    if (b16 == 1)
        hipLaunchKernelGGL(resid_add_layernorm_fwd_kernel<bf16_t>, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, stream, x,
                           (const bf16_t*)branch, rowscale, gamma, beta, eps, M, D, x_out, (bf16_t*)y, mean, rstd);
    else
        hipLaunchKernelGGL(resid_add_layernorm_fwd_kernel<f16_t>, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, stream, x,
                           (const f16_t*)branch, rowscale, gamma, beta, eps, M, D, x_out, (f16_t*)y, mean, rstd);
    EDITOR_LAUNCH_CHECK();
    return 0;
}

namespace {
int layernorm_bwd_impl(const void* dy, int dy_bf16, float dy_scale, const float* x, const float* gamma, const float* mean,
    const float* rstd, long M, int D, const uint8_t* rowmask, int mask_period, const float* dx_in, float* dx_out,
    float* dgamma, float* dbeta, float* workspace, int ws_rows, const int* m_live, void* cast_out, const float* cast_rowscale,
    float cast_scale, float* cast_colsum, float cast_colsum_scale, hipStream_t stream, int* nparts = nullptr,
    const int* dy_perm = nullptr, const int* dy_live = nullptr, const int* cast_perm = nullptr)
{
    // nparts != NULL: the "_parts" form - the partial rows stay in the workspace ([P][2][D] at its start, the cast output's column
    // sums [P][D] behind ws_rows*2*D floats), *nparts = P, and the caller folds them (editor_reduce_rows_multi)
    if (D % 4 || D > 1024 || M <= 0 || ws_rows < 1) return (int)hipErrorInvalidValue;
    if (cast_out && (dy_bf16 == 0 || m_live || rowmask || D % 256)) return (int)hipErrorInvalidValue;    // dense 16-bit rows only
    long blocks = (M + 3) / 4;
    if (blocks > ws_rows) blocks = ws_rows;
    float* cast_partials = (cast_out && cast_colsum) ? workspace + (long)ws_rows * 2 * D : nullptr;   // third [ws_rows][D] region
#define LN_BWD_LAUNCH(NVv, CASTv) DISPATCH_T(dy_bf16, hipLaunchKernelGGL((layernorm_bwd_kernel<TT, NVv, CASTv>), dim3((unsigned)blocks), \
        dim3(256), 0, stream, (const TT*)dy, x, gamma, mean, rstd, M, D, rowmask, mask_period, dx_in, dx_out,                            \
        dgamma ? workspace : nullptr, m_live, dy_scale, (TT*)cast_out, cast_rowscale, cast_scale, cast_partials))
    if (dy_perm || cast_perm) {          // compacted rows (stochastic depth): the cast form on dense 16-bit rows, D = 768 / 1024
        if (!cast_out || (dy_perm && !dy_live)) return (int)hipErrorInvalidValue;
#define LN_BWD_PERM(NVv) DISPATCH_T(dy_bf16, hipLaunchKernelGGL((layernorm_bwd_kernel<TT, NVv, true, false, true>), dim3((unsigned)blocks), \
        dim3(256), 0, stream, (const TT*)dy, x, gamma, mean, rstd, M, D, rowmask, mask_period, dx_in, dx_out,                            \
        dgamma ? workspace : nullptr, m_live, dy_scale, (TT*)cast_out, cast_rowscale, cast_scale, cast_partials, dy_perm, dy_live, cast_perm))
        switch (D >> 8) {
            case 1: LN_BWD_PERM(1); break;
            case 2: LN_BWD_PERM(2); break;
            case 3: LN_BWD_PERM(3); break;
            default: LN_BWD_PERM(4); break;
        }
#undef LN_BWD_PERM
    } else
    if (D % 256) {                       // ragged width (384): guarded instantiations, no cast output
#define LN_BWD_RAGGED(NVv) DISPATCH_T(dy_bf16, hipLaunchKernelGGL((layernorm_bwd_kernel<TT, NVv, false, true>), dim3((unsigned)blocks), \
        dim3(256), 0, stream, (const TT*)dy, x, gamma, mean, rstd, M, D, rowmask, mask_period, dx_in, dx_out,                            \
        dgamma ? workspace : nullptr, m_live, dy_scale, (TT*)nullptr, (const float*)nullptr, 1.f, (float*)nullptr))
        switch ((D + 255) >> 8) {
            case 1: LN_BWD_RAGGED(1); break;
            case 2: LN_BWD_RAGGED(2); break;
            case 3: LN_BWD_RAGGED(3); break;
            default: LN_BWD_RAGGED(4); break;
        }
#undef LN_BWD_RAGGED
    } else
    switch ((D >> 8) * 2 + (cast_out ? 1 : 0)) {
        case 2: LN_BWD_LAUNCH(1, false); break;
        case 3: LN_BWD_LAUNCH(1, true); break;
        case 4: LN_BWD_LAUNCH(2, false); break;
        case 5: LN_BWD_LAUNCH(2, true); break;
        case 6: LN_BWD_LAUNCH(3, false); break;
        case 7: LN_BWD_LAUNCH(3, true); break;
        case 8: LN_BWD_LAUNCH(4, false); break;
        default: LN_BWD_LAUNCH(4, true); break;
    }
#undef LN_BWD_LAUNCH
    EDITOR_LAUNCH_CHECK();
    if (nparts) {
        if (dgamma && dbeta != dgamma + D) return (int)hipErrorInvalidValue;
        *nparts = (int)blocks;
        return 0;
    }
    if (cast_partials) {
        hipLaunchKernelGGL(reduce_rows_kernel, dim3((D + 63) / 64), dim3(1024), 0, stream, cast_partials, (int)blocks, (long)D,
                           cast_colsum, 0, cast_colsum_scale);
        EDITOR_LAUNCH_CHECK();
    }
    if (dgamma) {
        // workspace rows are [block][2][D] = P rows of 2D columns; dgamma and dbeta must be ONE (2,D) buffer
        // (dbeta == dgamma + D) so the reduction writes both without extra copies
        if (dbeta != dgamma + D) return (int)hipErrorInvalidValue;
        hipLaunchKernelGGL(reduce_rows_kernel, dim3((2 * D + 63) / 64), dim3(1024), 0, stream, workspace, (int)blocks,
                           (long)2 * D, dgamma, 0, 1.f);
        EDITOR_LAUNCH_CHECK();
    }
    return 0;
}
}  // namespace

extern "C" int editor_layernorm_bwd(const void* dy, int dy_bf16, float dy_scale, const float* x, const float* gamma, const float* mean,
    const float* rstd, long M, int D, const uint8_t* rowmask, int mask_period, const float* dx_in, float* dx_out,
    float* dgamma, float* dbeta, float* workspace, int ws_rows, const int* m_live, hipStream_t stream)
{
    return layernorm_bwd_impl(dy, dy_bf16, dy_scale, x, gamma, mean, rstd, M, D, rowmask, mask_period, dx_in, dx_out, dgamma, dbeta,
                              workspace, ws_rows, m_live, nullptr, nullptr, 1.f, nullptr, 1.f, stream);
}

extern "C" int editor_layernorm_bwd_cast(const void* dy, int dy_bf16, float dy_scale, const float* x, const float* gamma,
    const float* mean, const float* rstd, long M, int D, const float* dx_in, float* dx_out, float* dgamma, float* dbeta,
    float* workspace, int ws_rows, void* cast_out, const float* cast_rowscale, float cast_scale, float* cast_colsum,
    float cast_colsum_scale, hipStream_t stream)
{
    if (!cast_out) return (int)hipErrorInvalidValue;
    return layernorm_bwd_impl(dy, dy_bf16, dy_scale, x, gamma, mean, rstd, M, D, nullptr, 0, dx_in, dx_out, dgamma, dbeta, workspace,
                              ws_rows, nullptr, cast_out, cast_rowscale, cast_scale, cast_colsum, cast_colsum_scale, stream);
}

namespace {
int colsum_impl(const void* dy, int dy_bf16, long M, int N, long ld, float* out, float* workspace, int ws_rows, float scale,
                hipStream_t stream, int* nparts)
{
    if (N % 4 || ws_rows < 1) return (int)hipErrorInvalidValue;
    int rows_per = 64;                                   // 16 rows per wave-group pass x 4
    while ((M + rows_per - 1) / rows_per > ws_rows) rows_per *= 2;
    const int gy = (int)((M + rows_per - 1) / rows_per);
    DISPATCH_T(dy_bf16, hipLaunchKernelGGL(colsum_kernel<TT>, dim3((N / 4 + 63) / 64, gy), dim3(256), 0, stream,
               (const TT*)dy, M, N, ld, rows_per, workspace));
    EDITOR_LAUNCH_CHECK();
    if (nparts) { *nparts = gy; return 0; }
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((N + 63) / 64), dim3(1024), 0, stream, workspace, gy, (long)N, out, 0, scale);
    EDITOR_LAUNCH_CHECK();
    return 0;
}

// Up to eight fixed-order folds out_j[c] = scale_j * sum_p partials_j[p][c] as ONE launch (blockIdx.y = job): a transformer
// block's backward leaves six sets of partial rows (two LayerNorms' dgamma / dbeta, three bias gradients, one column sum from a
// dgrad epilogue) whose totals nothing needs before the block ends - one launch per block instead of six (round 4: 95 -> 31
// reduce launches per step).  Same summation order per element as reduce_rows_kernel: bit-identical.
constexpr int kMaxReduce = 8;
struct ReduceJobs { const float* partials[kMaxReduce]; float* out[kMaxReduce]; long ncol[kMaxReduce]; int P[kMaxReduce]; float scale[kMaxReduce]; };

__global__ __launch_bounds__(1024) void reduce_rows_multi_kernel(ReduceJobs j)
{
    __shared__ float red[16][64];
    const int job = blockIdx.y;
    const long ncol = j.ncol[job];
    if ((long)blockIdx.x * 64 >= ncol) return;                          // (whole block: before the barrier)
    const float* __restrict__ partials = j.partials[job];
    const int P = j.P[job];
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const long c = (long)blockIdx.x * 64 + lane;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c < ncol) {
        int p = rg;
        for (; p + 112 < P; p += 128) {
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u] += partials[(long)(p + 16 * u) * ncol + c];
        }
        for (; p < P; p += 16) s[0] += partials[(long)p * ncol + c];
    }
    red[rg][lane] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    __syncthreads();
    if (rg == 0 && c < ncol) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += red[r][lane];
        j.out[job][c] = t * j.scale[job];
    }
}
}  // namespace

extern "C" int editor_colsum(const void* dy, int dy_bf16, long M, int N, long ld, float* out, float* workspace,
                             int ws_rows, float scale, hipStream_t stream)
{
    return colsum_impl(dy, dy_bf16, M, N, ld, out, workspace, ws_rows, scale, stream, nullptr);
}

extern "C" int editor_colsum_parts(const void* dy, int dy_bf16, long M, int N, long ld, float* workspace, int ws_rows, int* nparts,
                                   hipStream_t stream)
{
    if (!nparts) return (int)hipErrorInvalidValue;
    return colsum_impl(dy, dy_bf16, M, N, ld, nullptr, workspace, ws_rows, 1.f, stream, nparts);
}

extern "C" int editor_layernorm_bwd_parts(const void* dy, int dy_bf16, float dy_scale, const float* x, const float* gamma,
    const float* mean, const float* rstd, long M, int D, const uint8_t* rowmask, int mask_period, const float* dx_in, float* dx_out,
    float* workspace, int ws_rows, const int* m_live, int* nparts, hipStream_t stream)
{
    if (!nparts || !workspace) return (int)hipErrorInvalidValue;
    // (dgamma / dbeta: any non-NULL pair tells the kernel to write its partial rows; the totals are the caller's to fold)
    return layernorm_bwd_impl(dy, dy_bf16, dy_scale, x, gamma, mean, rstd, M, D, rowmask, mask_period, dx_in, dx_out, workspace,
                              workspace + D, workspace, ws_rows, m_live, nullptr, nullptr, 1.f, nullptr, 1.f, stream, nparts);
}

extern "C" int editor_layernorm_bwd_cast_parts(const void* dy, int dy_bf16, float dy_scale, const float* x, const float* gamma,
    const float* mean, const float* rstd, long M, int D, const float* dx_in, float* dx_out, float* workspace, int ws_rows,
    void* cast_out, const float* cast_rowscale, float cast_scale, int want_colsum, int* nparts, hipStream_t stream)
{
    if (!cast_out || !nparts || !workspace) return (int)hipErrorInvalidValue;
    return layernorm_bwd_impl(dy, dy_bf16, dy_scale, x, gamma, mean, rstd, M, D, nullptr, 0, dx_in, dx_out, workspace, workspace + D,
                              workspace, ws_rows, nullptr, cast_out, cast_rowscale, cast_scale, want_colsum ? workspace : nullptr,
                              1.f, stream, nparts);
}

// editor_layernorm_bwd_cast_parts on COMPACTED rows (stochastic depth, editor_droppath_plan): dy_perm / dy_live - dy lives on the
// compacted rows of this LayerNorm's branch (NULL: dense); cast_perm - the cast output goes to the compacted rows of the branch
// that consumes it (NULL: dense).
extern "C" int editor_layernorm_bwd_cast_perm_parts(const void* dy, int dy_bf16, float dy_scale, const float* x, const float* gamma,
    const float* mean, const float* rstd, long M, int D, const float* dx_in, float* dx_out, float* workspace, int ws_rows,
    void* cast_out, const float* cast_rowscale, float cast_scale, int want_colsum, const int* dy_perm, const int* dy_live,
    const int* cast_perm, int* nparts, hipStream_t stream)
{
    if (!cast_out || !nparts || !workspace || (!dy_perm && !cast_perm)) return (int)hipErrorInvalidValue;
    return layernorm_bwd_impl(dy, dy_bf16, dy_scale, x, gamma, mean, rstd, M, D, nullptr, 0, dx_in, dx_out, workspace, workspace + D,
                              workspace, ws_rows, nullptr, cast_out, cast_rowscale, cast_scale, want_colsum ? workspace : nullptr,
                              1.f, stream, nparts, dy_perm, dy_live, cast_perm);
}

extern "C" int editor_layernorm_fwd_perm(const float* x, const float* gamma, const float* beta, float eps, long M, int D, void* y,
    int y_bf16, float* mean, float* rstd, const int* perm, const float* rowscale, float* copy_out, hipStream_t stream)
{
    if ((D & 255) || D > 256 * kMaxV || M <= 0 || !perm || !rowscale || !copy_out || !mean || !rstd || y_bf16 == 0)
        return (int)hipErrorInvalidValue;
    DISPATCH_T(y_bf16, hipLaunchKernelGGL(layernorm_fwd_perm_kernel<TT>, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, stream,
               x, gamma, beta, eps, M, D, (TT*)y, mean, rstd, perm, rowscale, copy_out));
    EDITOR_LAUNCH_CHECK();
    return 0;
}

extern "C" int editor_layernorm_fwd_perm_f16x2(const float* x, const float* gamma, const float* beta, float eps, long M, int D,
    uint16_t* y_hi, uint16_t* y_lo, float* mean, float* rstd, const int* perm, const float* rowscale, float* copy_out,
    hipStream_t stream)
{
    if ((D & 255) || D > 256 * kMaxV || M <= 0 || !perm || !rowscale || !copy_out || !mean || !rstd || !y_hi || !y_lo)
        return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL((layernorm_fwd_perm_kernel<f16_t, true>), dim3((unsigned)((M + 3) / 4)), dim3(256), 0, stream,
                       x, gamma, beta, eps, M, D, (f16_t*)y_hi, mean, rstd, perm, rowscale, copy_out, (f16_t*)y_lo);
    EDITOR_LAUNCH_CHECK();
    return 0;
}

extern "C" int editor_reduce_rows_multi(int count, const float* const* partials, const int* P, const long* ncol, float* const* out,
                                        const float* scale, hipStream_t stream)
{
    if (count < 1 || count > kMaxReduce || !partials || !P || !ncol || !out || !scale) return (int)hipErrorInvalidValue;
    ReduceJobs j;
    memset(&j, 0, sizeof(j));
    long maxcol = 0;
    for (int i = 0; i < count; ++i) {
        if (!partials[i] || !out[i] || P[i] < 1 || ncol[i] < 1) return (int)hipErrorInvalidValue;
        j.partials[i] = partials[i]; j.out[i] = out[i]; j.ncol[i] = ncol[i]; j.P[i] = P[i]; j.scale[i] = scale[i];
        if (ncol[i] > maxcol) maxcol = ncol[i];
    }
    hipLaunchKernelGGL(reduce_rows_multi_kernel, dim3((unsigned)((maxcol + 63) / 64), (unsigned)count), dim3(1024), 0, stream, j);
    EDITOR_LAUNCH_CHECK();
    return 0;
}

extern "C" int editor_reduce_rows(const float* partials, int P, long ncol, float* out, int accumulate, float scale,
                                  hipStream_t stream)
{
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((unsigned)((ncol + 63) / 64)), dim3(1024), 0, stream, partials, P, ncol,
                       out, accumulate, scale);
    EDITOR_LAUNCH_CHECK();
    return 0;
}

extern "C" int editor_gelu_fwd(const void* a, void* g, long n, int bf16, hipStream_t stream)
{
    if (n % 4) return (int)hipErrorInvalidValue;
    DISPATCH_T(bf16, hipLaunchKernelGGL(gelu_fwd_kernel<TT>, dim3(grid_for(n / 4)), dim3(256), 0, stream, (const TT*)a, (TT*)g, n / 4));
    EDITOR_LAUNCH_CHECK();
    return 0;
}
extern "C" int editor_gelu_bwd(const void* a, const void* dg, void* da, long n, int bf16, hipStream_t stream)
{
    if (n % 4) return (int)hipErrorInvalidValue;
    DISPATCH_T(bf16, hipLaunchKernelGGL(gelu_bwd_kernel<TT>, dim3(grid_for(n / 4)), dim3(256), 0, stream,
               (const TT*)a, (const TT*)dg, (TT*)da, n / 4));
    EDITOR_LAUNCH_CHECK();
    return 0;
}

extern "C" int editor_cast_f32_to_bf16(const float* in, uint16_t* out, long n, hipStream_t stream)
{
    if (n % 4) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL((cast_kernel<float, bf16_t>), dim3(grid_for(n / 4)), dim3(256), 0, stream, in, out, n / 4);
    EDITOR_LAUNCH_CHECK();
    return 0;
}
extern "C" int editor_cast_f32_to_f16(const float* in, uint16_t* out, long n, hipStream_t stream)
{
    if (n % 4) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL((cast_kernel<float, f16_t>), dim3(grid_for(n / 4)), dim3(256), 0, stream, in, (f16_t*)out, n / 4);
    EDITOR_LAUNCH_CHECK();
    return 0;
}
extern "C" int editor_cast_f16_to_f32(const uint16_t* in, float* out, long n, hipStream_t stream)
{
    if (n % 4) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL((cast_kernel<f16_t, float>), dim3(grid_for(n / 4)), dim3(256), 0, stream, (const f16_t*)in, out, n / 4);
    EDITOR_LAUNCH_CHECK();
    return 0;
}
extern "C" int editor_cast_bf16_to_f32(const uint16_t* in, float* out, long n, hipStream_t stream)
{
    if (n % 4) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL((cast_kernel<bf16_t, float>), dim3(grid_for(n / 4)), dim3(256), 0, stream, in, out, n / 4);
    EDITOR_LAUNCH_CHECK();
    return 0;
}

extern "C" int editor_cast_rows(const float* in, const float* rowscale, long M, int D, void* out, int out_bf16,
                                const int* m_live, float scale, hipStream_t stream)
{
    if (D % 4 || M * (D / 4) >= (1L << 32)) return (int)hipErrorInvalidValue;      // (32-bit element index in the kernel)
    DISPATCH_T(out_bf16, hipLaunchKernelGGL(cast_rows_kernel<TT>, dim3(grid_for(M * (D / 4))), dim3(256), 0, stream,
               in, rowscale, M, D, (TT*)out, m_live, scale));
    EDITOR_LAUNCH_CHECK();
    return 0;
}

extern "C" int editor_cast_rows_colsum_perm_parts(const float* in, const float* rowscale, long M, int D, void* out, int out_bf16,
                                                  float* workspace, int ws_rows, float scale, const int* perm, int* nparts,
                                                  hipStream_t stream)
{
    if ((D & 255) || D > 256 * kMaxV || ws_rows < 1 || !nparts || !workspace || !perm) return (int)hipErrorInvalidValue;
    long blocks = (M + 3) / 4;
    if (blocks > ws_rows) blocks = ws_rows;
    DISPATCH_T(out_bf16, hipLaunchKernelGGL(cast_rows_colsum_kernel<TT>, dim3((unsigned)blocks), dim3(256), 0, stream,
               in, rowscale, M, D, (TT*)out, workspace, scale, perm));
    EDITOR_LAUNCH_CHECK();
    *nparts = (int)blocks;
    return 0;
}

extern "C" int editor_cast_rows_colsum_parts(const float* in, const float* rowscale, long M, int D, void* out, int out_bf16,
                                             float* workspace, int ws_rows, float scale, int* nparts, hipStream_t stream)
{
    if ((D & 255) || D > 256 * kMaxV || ws_rows < 1 || !nparts || !workspace) return (int)hipErrorInvalidValue;
    long blocks = (M + 3) / 4;
    if (blocks > ws_rows) blocks = ws_rows;
    DISPATCH_T(out_bf16, hipLaunchKernelGGL(cast_rows_colsum_kernel<TT>, dim3((unsigned)blocks), dim3(256), 0, stream,
               in, rowscale, M, D, (TT*)out, workspace, scale));
    EDITOR_LAUNCH_CHECK();
    *nparts = (int)blocks;
    return 0;
}

extern "C" int editor_cast_rows_colsum(const float* in, const float* rowscale, long M, int D, void* out, int out_bf16,
                                       float* colsum, float* workspace, int ws_rows, float scale, float colsum_scale,
                                       hipStream_t stream)
{
    if ((D & 255) || D > 256 * kMaxV || ws_rows < 1 || !colsum || !workspace) return (int)hipErrorInvalidValue;
    long blocks = (M + 3) / 4;
    if (blocks > ws_rows) blocks = ws_rows;
    DISPATCH_T(out_bf16, hipLaunchKernelGGL(cast_rows_colsum_kernel<TT>, dim3((unsigned)blocks), dim3(256), 0, stream,
               in, rowscale, M, D, (TT*)out, workspace, scale));
    EDITOR_LAUNCH_CHECK();
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((D + 63) / 64), dim3(1024), 0, stream, workspace, (int)blocks, (long)D, colsum,
                       0, colsum_scale);
    EDITOR_LAUNCH_CHECK();
    return 0;
}

extern "C" int editor_layernorm_fwd_f16x2(const float* x, const float* gamma, const float* beta, float eps, long M, int D,
    const uint8_t* rowmask, int mask_period, uint16_t* y_hi, uint16_t* y_lo, float* mean, float* rstd, const int* m_live,
    hipStream_t stream)
{
    if (D % 4 || D > 1024 || M <= 0 || !y_hi || !y_lo) return (int)hipErrorInvalidValue;
    if (D % 256)
        hipLaunchKernelGGL((layernorm_fwd_kernel<f16_t, true, true>), dim3((unsigned)((M + 3) / 4)), dim3(256), 0, stream,
                           x, gamma, beta, eps, M, D, rowmask, mask_period, (f16_t*)y_hi, mean, rstd, m_live, (f16_t*)y_lo);
    else
        hipLaunchKernelGGL((layernorm_fwd_kernel<f16_t, true>), dim3((unsigned)((M + 3) / 4)), dim3(256), 0, stream,
                           x, gamma, beta, eps, M, D, rowmask, mask_period, (f16_t*)y_hi, mean, rstd, m_live, (f16_t*)y_lo);
    EDITOR_LAUNCH_CHECK();
    return 0;
}

extern "C" int editor_im2col16_f16x2(const float* img, int B, int C, int H, int W, uint16_t* out_hi, uint16_t* out_lo,
                                     hipStream_t stream)
{
    if ((H & 15) || (W & 15) || !out_hi || !out_lo) return (int)hipErrorInvalidValue;
    const long total4 = (long)B * (H >> 4) * (W >> 4) * C * 64;
    if (total4 >= (1L << 33)) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(im2col16_kernel<f16_t>, dim3(grid_for(total4)), dim3(256), 0, stream, img, B, C, H, W, (f16_t*)out_hi,
                       (f16_t*)out_lo);
    EDITOR_LAUNCH_CHECK();
    return 0;
}

extern "C" int editor_split_f32(const float* in, uint16_t* hi, uint16_t* lo, long n, float scale, hipStream_t stream)
{
    if (n <= 0 || (n & 3) || !hi || !lo) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(split_f32_kernel, dim3(grid_for(n / 4)), dim3(256), 0, stream, in, (f16_t*)hi, (f16_t*)lo, n / 4, scale);
    EDITOR_LAUNCH_CHECK();
    return 0;
}

extern "C" int editor_im2col16(const float* img, int B, int C, int H, int W, void* out, int out_bf16, hipStream_t stream)
{
    if ((H & 15) || (W & 15)) return (int)hipErrorInvalidValue;
    const long total4 = (long)B * (H >> 4) * (W >> 4) * C * 64;
    if (total4 >= (1L << 33)) return (int)hipErrorInvalidValue;          // (the kernel's row index is 32-bit: total4 / 4 < 2^31)
    DISPATCH_T(out_bf16, hipLaunchKernelGGL(im2col16_kernel<TT>, dim3(grid_for(total4)), dim3(256), 0, stream, img, B, C, H, W, (TT*)out));
    EDITOR_LAUNCH_CHECK();
    return 0;
}

extern "C" int editor_embed_assemble(const void* patch, int patch_bf16, const float* cls, const float* pos,
    const float* sie, const long* cam, int Bcam, float coef, long Btot, int T, int D, float* x, hipStream_t stream)
{
    if (D % 4) return (int)hipErrorInvalidValue;
    const long total = Btot * T * (D / 4);
    if (total >= (1L << 32)) return (int)hipErrorInvalidValue;                       // (32-bit element index in the kernel)
    DISPATCH_T(patch_bf16, hipLaunchKernelGGL(embed_assemble_kernel<TT>, dim3(grid_for(total)), dim3(256), 0, stream,
               (const TT*)patch, cls, pos, sie, cam, Bcam, coef, Btot, T, D, x));
    EDITOR_LAUNCH_CHECK();
    return 0;
}

extern "C" int editor_embed_assemble_bwd(const float* dx, const long* cam, int Bcam, int ncam, float coef, long Btot,
    int T, int D, void* dpatch, int dpatch_bf16, float dpatch_scale, float* dpos, float* dsie, float* workspace,
    hipStream_t stream)
{
    if (D % 4) return (int)hipErrorInvalidValue;
    const long total = Btot * (T - 1) * (D / 4);
    if (total >= (1L << 32)) return (int)hipErrorInvalidValue;                       // (32-bit element index in the kernel)
    DISPATCH_T(dpatch_bf16, hipLaunchKernelGGL(embed_bwd_patch_kernel<TT>, dim3(grid_for(total)), dim3(256), 0, stream,
               dx, Btot, T, D, (TT*)dpatch, dpatch_scale));
    EDITOR_LAUNCH_CHECK();
    // workspace: max(EDITOR_EMBED_POS_SPLITS * T * D, Btot * D) floats - the dpos partial rows, then (re-used) the row sums
    if (!workspace) return (int)hipErrorInvalidValue;
    const int S = (int)(Btot < EDITOR_EMBED_POS_SPLITS ? Btot : EDITOR_EMBED_POS_SPLITS);
    hipLaunchKernelGGL(embed_bwd_pos_kernel, dim3(T, (D + 1023) / 1024, S), dim3(256), 0, stream, dx, Btot, T, D, S, workspace);
    EDITOR_LAUNCH_CHECK();
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((unsigned)(((long)T * D + 63) / 64)), dim3(1024), 0, stream, workspace, S, (long)T * D,
                       dpos, 0, 1.f);
    EDITOR_LAUNCH_CHECK();
    if (dsie) {
        hipLaunchKernelGGL(embed_bwd_rowsum_kernel, dim3((unsigned)Btot, (D + 1023) / 1024), dim3(256), 0, stream, dx, T, D,
                           workspace);
        EDITOR_LAUNCH_CHECK();
        hipLaunchKernelGGL(embed_bwd_sie_kernel, dim3(ncam, (D + 63) / 64), dim3(256), 0, stream, workspace, cam, Bcam,
                           Btot, D, coef, dsie);
        EDITOR_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int editor_sfts_apply(const float* feat, const uint8_t* index, int nmod, long B, int T, int D, float* out,
                                 float* loss, float* workspace, int ws_len, hipStream_t stream)
{
    if (D % 4 || D > 1024 || nmod < 2 || nmod > 4) return (int)hipErrorInvalidValue;
    const long rows = B * T;
    const unsigned g = (unsigned)((rows + SFTS_ROWS - 1) / SFTS_ROWS);       // one partial sum per block
    if (loss && (!workspace || (long)g > ws_len)) return (int)hipErrorInvalidValue;
    const long mstride = rows * (long)D;
    float* part = loss ? workspace : nullptr;
    switch (nmod) {
        case 2: hipLaunchKernelGGL(sfts_apply_kernel<2>, dim3(g), dim3(D / 4), 0, stream, feat, index, rows, T, D, mstride, out, part); break;
        case 3: hipLaunchKernelGGL(sfts_apply_kernel<3>, dim3(g), dim3(D / 4), 0, stream, feat, index, rows, T, D, mstride, out, part); break;
        default: hipLaunchKernelGGL(sfts_apply_kernel<4>, dim3(g), dim3(D / 4), 0, stream, feat, index, rows, T, D, mstride, out, part); break;
    }
    EDITOR_LAUNCH_CHECK();
    if (loss) {   // MSELoss mean over B*(T-1)*D elements, summed over modality pairs (SFTS.py:221)
        hipLaunchKernelGGL(reduce_rows_kernel, dim3(1), dim3(1024), 0, stream, workspace, (int)g, 1L, loss, 0,
                           1.f / ((float)B * (T - 1) * D));
        EDITOR_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int editor_sfts_apply_bwd(const float* feat, const uint8_t* index, const float* dout, const float* dloss,
                                     int nmod, long B, int T, int D, float* dfeat, hipStream_t stream)
{
    if (D % 4 || nmod < 2 || nmod > 4) return (int)hipErrorInvalidValue;
    const long total = B * T * (D / 4);
    hipLaunchKernelGGL(sfts_apply_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, stream, feat, index, dout, dloss,
                       2.f / ((float)B * (T - 1) * D), nmod, B, T, D, dfeat);
    EDITOR_LAUNCH_CHECK();
    return 0;
}

extern "C" int editor_pool_fwd(const float* x, long B, int nmod, int T, int D, float* out, float* num, hipStream_t stream)
{
    hipLaunchKernelGGL(pool_fwd_kernel, dim3((unsigned)B), dim3(256), 0, stream, x, B, nmod, T, D, out, num);
    EDITOR_LAUNCH_CHECK();
    return 0;
}
extern "C" int editor_pool_bwd(const float* dout, const float* num, long B, int nmod, int T, int D, float* dx,
                               hipStream_t stream)
{
    hipLaunchKernelGGL(pool_bwd_kernel, dim3(grid_for(B * nmod * T * (long)D)), dim3(256), 0, stream, dout, num, B, nmod, T, D, dx);
    EDITOR_LAUNCH_CHECK();
    return 0;
}
extern "C" int editor_rowmask_mul(float* x, const uint8_t* rowmask, int period, long M, int D, hipStream_t stream)
{
    if (D % 4) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(rowmask_mul_kernel, dim3(grid_for(M * (D / 4))), dim3(256), 0, stream, x, rowmask, period, M, D);
    EDITOR_LAUNCH_CHECK();
    return 0;
}
