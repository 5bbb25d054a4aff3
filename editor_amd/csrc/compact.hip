// Compacted (variable-length) form of the HMA head, gfx950.
// After SFTS the unselected patch rows are exactly zero and STAY zero through every masked block of
// BlockMask.forward (vit_pytorch.py:240-258,158-168,309-337: no biases, -65504 fill underflows to 0, query mask) -
// SURVEY.md 5 "HMA exact-zero invariant".  So the head runs on the kept rows only: per sample b a sequence of
// L_b = 1 + #selected tokens, packed back to back (cu = exclusive prefix sum), about half the rows and a quarter of
// the attention of the dense form.  These kernels build the packing plan and move rows between the layouts:
//   A  modality-major  [m][cu[b]+l]            (the three per-modality blocks: rows of one modality contiguous)
//   B  sample-major    [3*cu[b] + m*L_b + l]   (the joint block attends over the 3*L_b rows of a sample)
#include "common.h"
#include "../../include/editor_hip.h"

namespace {

// one block, thread per sample: L_b = 1 + popcount(index[b,:]); cu = exclusive scan; tok = kept token ids (0 = cls)
__global__ __launch_bounds__(1024) void compact_plan_kernel(const uint8_t* __restrict__ index, int B, int N,
                                                            int* __restrict__ cu, int* __restrict__ tok)
{
    __shared__ int scan[1024];
    const int b = threadIdx.x;
    int cnt = 0;
    if (b < B) {
        cnt = 1;
        for (int n = 0; n < N; ++n) cnt += index[(long)b * N + n] ? 1 : 0;
    }
    scan[b] = cnt;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {          // Hillis-Steele inclusive scan
        const int v = b >= off ? scan[b - off] : 0;
        __syncthreads();
        scan[b] += v;
        __syncthreads();
    }
    if (b < B) {
        const int start = scan[b] - cnt;
        cu[b] = start;
        if (b == B - 1) cu[B] = scan[b];
        int w = start;
        tok[w++] = 0;
        for (int n = 0; n < N; ++n)
            if (index[(long)b * N + n]) tok[w++] = n + 1;
    }
}

// row maps.  thread per (sample b, position l < L_b)
__global__ void compact_maps_kernel(const int* __restrict__ cu, const int* __restrict__ tok, int B, int T, int nmod,
    long MA, long MB, int* __restrict__ mapA, int* __restrict__ mapB, int* __restrict__ mapCls,
    uint8_t* __restrict__ maskA, uint8_t* __restrict__ maskB, int* __restrict__ cu3)
{
    const int b = blockIdx.x;
    const int start = cu[b], len = cu[b + 1] - start;
    if (threadIdx.x == 0) {
        cu3[b] = nmod * start;
        if (b == B - 1) cu3[B] = nmod * cu[B];
        for (int m = 0; m < nmod; ++m) mapCls[m * B + b] = (int)(m * MA + start);
    }
    for (int l = threadIdx.x; l < len; l += blockDim.x) {
        maskA[start + l] = 1;
        for (int m = 0; m < nmod; ++m) {
            mapA[m * MA + start + l] = (m * B + b) * T + tok[start + l];           // dense (nmod,B,T,D) row
            const long pb = (long)nmod * start + (long)m * len + l;
            mapB[pb] = (int)(m * MA + start + l);                                  // layout-A row
            maskB[pb] = 1;
        }
    }
}

// out[r,:] = src[r] >= 0 ? in[src[r],:] : 0
__global__ void gather_rows_kernel(const float* __restrict__ in, const int* __restrict__ src, long R, int D,
                                   float* __restrict__ out, const int* __restrict__ r_live, int live_mul, long live_stride)
{
    const int d4 = D >> 2;
    // rows beyond the live extent (rounded up to 64) of each `live_stride` segment are never read downstream
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < R * d4; e += (long)gridDim.x * blockDim.x) {
        const long r = e / d4;
        const int c0 = (int)(e % d4) * 4;
        if (r_live && (r % live_stride) >= ((live_mul * *r_live + 63) & ~63)) continue;
        const int s = src[r];
        *reinterpret_cast<float4*>(out + r * D + c0) =
            s >= 0 ? *reinterpret_cast<const float4*>(in + (long)s * D + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
// dx[src[r],:] = dy[r,:]  (dx zero-filled first; every source row is gathered at most once)
__global__ void scatter_rows_kernel(const float* __restrict__ dy, const int* __restrict__ src, long R, int D,
                                    float* __restrict__ dx)
{
    const int d4 = D >> 2;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < R * d4; e += (long)gridDim.x * blockDim.x) {
        const long r = e / d4;
        const int c0 = (int)(e % d4) * 4;
        const int s = src[r];
        if (s >= 0) *reinterpret_cast<float4*>(dx + (long)s * D + c0) = *reinterpret_cast<const float4*>(dy + r * D + c0);
    }
}

// pooling on layout B (make_model.py:186-203): block per (sample, modality).  Both row loops keep EIGHT independent loads in
// flight per thread (one dependent load per trip over ~80 rows made this 116 us for 30 MB).
__global__ __launch_bounds__(256) void pool_packed_fwd_kernel(const float* __restrict__ x, const int* __restrict__ cu,
    long B, int nmod, int D, float* __restrict__ out, float* __restrict__ num_out)
{
    __shared__ int cnt;
    const long b = blockIdx.x / nmod;
    const int m = blockIdx.x % nmod;
    const int start = cu[b], len = cu[b + 1] - start;
    const float* x0 = x + ((long)nmod * start) * D;                  // modality 0 (RGB) rows of this sample
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    for (int l = 1 + (threadIdx.x >> 6); l < len; l += 4) {          // num = #RGB patch rows with non-zero row sum
        float s = 0.f;
        for (int c0 = lane; c0 < D; c0 += 512) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = c0 + 64 * j < D ? x0[(long)l * D + c0 + 64 * j] : 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[j];
        }
        s = wave_sum(s);
        if (lane == 0 && s != 0.f) atomicAdd(&cnt, 1);
    }
    __syncthreads();
    const float num = (float)cnt;
    if (threadIdx.x == 0 && m == 0) num_out[b] = num;
    const float* xm = x0 + (long)m * len * D;
    float* o = out + ((long)m * B + b) * 2 * D;
    for (int c = threadIdx.x; c < D; c += 256) {
        float s = 0.f;
        for (int l0 = 1; l0 < len; l0 += 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = l0 + j < len ? xm[(long)(l0 + j) * D + c] : 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[j];                   // (row order, as before)
        }
        o[c] = xm[c];
        o[D + c] = s / num;
    }
}
__global__ __launch_bounds__(256) void pool_packed_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ num,
    const int* __restrict__ cu, long B, int nmod, int D, float* __restrict__ dx)
{
    const long b = blockIdx.x / nmod;
    const int m = blockIdx.x % nmod;
    const int start = cu[b], len = cu[b + 1] - start;
    float* xm = dx + ((long)nmod * start + (long)m * len) * D;
    const float* o = dout + ((long)m * B + b) * 2 * D;
    const float inv = 1.f / num[b];
    for (int c = threadIdx.x; c < D; c += 256) {
        const float gp = o[D + c] * inv;
        xm[c] = o[c];
        for (int l = 1; l < len; ++l) xm[(long)l * D + c] = gp;
    }
}

inline unsigned grid_for(long n, int block = 256, long cap = 256L * 16) {
    long g = (n + block - 1) / block;
    return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" int editor_compact_plan(const uint8_t* index, int B, int N, int* cu, int* tok, hipStream_t stream)
{
    if (B < 1 || B > 1024 || N < 1) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(compact_plan_kernel, dim3(1), dim3(1024), 0, stream, index, B, N, cu, tok);
    EDITOR_LAUNCH_CHECK();
    return 0;
}

extern "C" int editor_compact_maps(const int* cu, const int* tok, int B, int T, int nmod, long MA, long MB, int* mapA,
    int* mapB, int* mapCls, uint8_t* maskA, uint8_t* maskB, int* cu3, hipStream_t stream)
{
    hipError_t e;
    if ((e = hipMemsetAsync(mapA, 0xff, sizeof(int) * nmod * MA, stream)) != hipSuccess) return (int)e;   // -1
    if ((e = hipMemsetAsync(mapB, 0xff, sizeof(int) * MB, stream)) != hipSuccess) return (int)e;
    if ((e = hipMemsetAsync(maskA, 0, MA, stream)) != hipSuccess) return (int)e;
    if ((e = hipMemsetAsync(maskB, 0, MB, stream)) != hipSuccess) return (int)e;
    hipLaunchKernelGGL(compact_maps_kernel, dim3(B), dim3(256), 0, stream, cu, tok, B, T, nmod, MA, MB, mapA, mapB, mapCls,
                       maskA, maskB, cu3);
    EDITOR_LAUNCH_CHECK();
    return 0;
}

extern "C" int editor_gather_rows(const float* in, const int* src, long R, int D, float* out, const int* r_live,
                                  int live_mul, long live_stride, hipStream_t stream)
{
    if (D % 4 || (r_live && live_stride <= 0)) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(R * (D / 4))), dim3(256), 0, stream, in, src, R, D, out, r_live,
                       live_mul, live_stride);
    EDITOR_LAUNCH_CHECK();
    return 0;
}
// rows [live, roundup64(live)) of a packed (rows x row_bytes) buffer <- 0: the contract of the live-row kernels (reductions over
// token rows read whole 64-row K-tiles) without zero-filling the worst-case-sized buffer (4 x 150 MB of fills per step)
__global__ void zero_tail_rows_kernel(char* __restrict__ buf, long row_bytes, long rows, const int* __restrict__ live)
{
    const long r0 = *live, r1 = min(rows, (r0 + 63) & ~63L);
    const long n16 = (r1 - r0) * (row_bytes >> 4);
    uint4* p = reinterpret_cast<uint4*>(buf + r0 * row_bytes);
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n16; e += (long)gridDim.x * blockDim.x) p[e] = make_uint4(0u, 0u, 0u, 0u);
}

extern "C" int editor_zero_tail_rows(void* buf, long row_bytes, long rows, const int* live, hipStream_t stream)
{
    if (!buf || !live || (row_bytes & 15) || rows < 1) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(zero_tail_rows_kernel, dim3(64), dim3(256), 0, stream, (char*)buf, row_bytes, rows, live);
    EDITOR_LAUNCH_CHECK();
    return 0;
}

extern "C" int editor_scatter_rows(const float* dy, const int* src, long R, int D, long rows_out, float* dx,
                                   hipStream_t stream)
{
    if (D % 4) return (int)hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(dx, 0, sizeof(float) * rows_out * D, stream);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(grid_for(R * (D / 4))), dim3(256), 0, stream, dy, src, R, D, dx);
    EDITOR_LAUNCH_CHECK();
    return 0;
}

// The same without the zero fill of dx (round 4: 152 MB of hipMemset per call - 103 us - for rows nobody reads): the caller either
// guarantees that the rows no source index names are never read (the gradient of the layout-A gather goes straight into
// editor_sfts_apply_bwd, which reads selected rows only) or zeroes what the live-row kernels DO read beyond the live extent
// with editor_zero_tail_rows (rows [live, roundup64(live)) of every segment).
extern "C" int editor_scatter_rows_nofill(const float* dy, const int* src, long R, int D, float* dx, hipStream_t stream)
{
    if (D % 4) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(grid_for(R * (D / 4))), dim3(256), 0, stream, dy, src, R, D, dx);
    EDITOR_LAUNCH_CHECK();
    return 0;
}

extern "C" int editor_pool_packed_bwd_nofill(const float* dout, const float* num, const int* cu, long B, int nmod, int D, float* dx,
                                             hipStream_t stream)
{
    hipLaunchKernelGGL(pool_packed_bwd_kernel, dim3((unsigned)(B * nmod)), dim3(256), 0, stream, dout, num, cu, B, nmod, D, dx);
    EDITOR_LAUNCH_CHECK();
    return 0;
}

extern "C" int editor_pool_packed_fwd(const float* x, const int* cu, long B, int nmod, int D, float* out, float* num,
                                      hipStream_t stream)
{
    hipLaunchKernelGGL(pool_packed_fwd_kernel, dim3((unsigned)(B * nmod)), dim3(256), 0, stream, x, cu, B, nmod, D, out, num);
    EDITOR_LAUNCH_CHECK();
    return 0;
}
extern "C" int editor_pool_packed_bwd(const float* dout, const float* num, const int* cu, long B, int nmod, int D,
                                      long rows, float* dx, hipStream_t stream)
{
    hipError_t e = hipMemsetAsync(dx, 0, sizeof(float) * rows * D, stream);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(pool_packed_bwd_kernel, dim3((unsigned)(B * nmod)), dim3(256), 0, stream, dout, num, cu, B, nmod, D, dx);
    EDITOR_LAUNCH_CHECK();
    return 0;
}
