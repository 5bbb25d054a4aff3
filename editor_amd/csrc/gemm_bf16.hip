// bf16 MFMA GEMM of the EDITOR hot path, gfx950 / CDNA4.
//
//     C[m,n] = alpha * sum_k opA(A)[m,k] * opB(B)[k,n]  (+ bias[n])  (* rowscale[m])  (+ beta * C[m,n])
//
// bf16 operands, fp32 accumulation on v_mfma_f32_16x16x32_bf16, C in bf16 or fp32.  One kernel family
// covers the three contractions of a linear layer WITHOUT materialising transposes:
//     forward   y  = x W^T        A (M,K) k-major,  B = W (N,K) k-major
//     dgrad     dx = dy W         A (M,N) k-major,  B = W (N,K): reduction index is the ROW of W  -> "row-major-k"
//     wgrad     dW = dy^T x       A = dy (M,N) and B = x (M,K): reduction index is the row of both
// k-major operands are fed to the matrix core with ds_read_b128; operands whose reduction index is the row
// index are staged untransposed and read with the CDNA4 LDS transpose load ds_read_b64_tr_b16.
//
// Tiling: 128x128x64 per workgroup, 4 wavefronts (2x2), each 64x64 = 4x4 MFMA tiles; LDS double buffered
// (64 KiB -> 2 workgroups/CU), register-staged global->LDS with XOR-swizzled images; XCD-aware tile order
// (consecutive n-tiles of one A row-panel stay on one XCD's L2); optional split-K with fp32 atomics.
// The MFMA is issued with its operands swapped (D^T tile) so each lane owns 4 CONSECUTIVE output columns:
// 8/16-byte epilogue stores and float4 bias loads.
#include "common.h"
#include "../../include/editor_hip.h"

typedef __attribute__((ext_vector_type(4))) short short4_t;
typedef __attribute__((ext_vector_type(8))) short short8_t;
typedef __attribute__((ext_vector_type(4))) float float4_t;

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * 64 * 2;          // 16 KiB per operand tile per stage

struct GemmB16Args {
    const bf16_t* A; const bf16_t* B; void* C;
    int M, N, K;
    long lda, ldb, ldc;
    float alpha, beta;
    const float* bias; const float* rowscale;
    int splitk, tiles_m, tiles_n;
    int epilogue; void* aux; long ldaux;     // EDITOR_EPI_* (editor_hip.h)
};

__device__ __forceinline__ float gelu_f(float a) { return 0.5f * a * (1.f + erff(a * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float a) {
    return 0.5f * (1.f + erff(a * 0.70710678118654752f)) + a * 0.3989422804014327f * __expf(-0.5f * a * a);
}

// ---- LDS images -------------------------------------------------------------------------------------
// k-major tile  [128 rows][64 k]  : byte = row*128 + ((chunk ^ (row&7)) * 16), chunk = k/8
// row-k tile    [64 k][128 cols]  : byte = k*256 + ((blk ^ swz(k)) * 32) + within, blk = col/16
__device__ __forceinline__ int swz_rowk(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }

template <bool KMAJOR>
struct Stage {
    uint4 v[4];
    // issue the global loads of one 128x64 (or 64x128) tile; rows/cols beyond the matrix read zeros
    __device__ __forceinline__ void load(const bf16_t* __restrict__ P, long ld, int r0, int k0, int R, int K) {
        const int tid = threadIdx.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int chunk = tid + 256 * j;
            long off; bool ok;
            if (KMAJOR) {
                const int row = chunk >> 3, c = chunk & 7;
                ok = (r0 + row < R) && (k0 + c * 8 < K);
                off = (long)(r0 + row) * ld + k0 + c * 8;
            } else {
                const int kr = chunk >> 4, c = chunk & 15;
                ok = (k0 + kr < K) && (r0 + c * 8 < R);
                off = (long)(k0 + kr) * ld + r0 + c * 8;
            }
            v[j] = ok ? *reinterpret_cast<const uint4*>(P + off) : make_uint4(0u, 0u, 0u, 0u);
        }
    }
    __device__ __forceinline__ void store(char* lds) const {
        const int tid = threadIdx.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int chunk = tid + 256 * j;
            int byte;
            if (KMAJOR) {
                const int row = chunk >> 3, c = chunk & 7;
                byte = row * 128 + ((c ^ (row & 7)) << 4);
            } else {
                const int kr = chunk >> 4, c = chunk & 15;
                byte = kr * 256 + (((c >> 1) ^ swz_rowk(kr)) << 5) + ((c & 1) << 4);
            }
            *reinterpret_cast<uint4*>(lds + byte) = v[j];
        }
    }
};

// fragment of sub-tile `sub` (16 rows / cols starting at base16) for k-step s: lane (i = l&15, g = l>>4) gets
// element (base16 + i, k = s*32 + g*8 + e), e = 0..7
template <bool KMAJOR>
__device__ __forceinline__ short8_t load_frag(const char* lds, int base16, int s, int lane)
{
    const int i = lane & 15, g = lane >> 4;
    if (KMAJOR) {
        const int row = base16 + i, c = s * 4 + g;
        return *reinterpret_cast<const short8_t*>(lds + row * 128 + ((c ^ (row & 7)) << 4));
    } else {
        // ds_read_b64_tr_b16: the 16 lanes of a group present the sixteen 8-byte pieces of a [4 k][16 col] block
        // (lane i -> k-row i>>2, piece i&3) and lane i receives column i of it (4 consecutive k).
        const int blk = base16 >> 4;
        const int k0 = s * 32 + g * 8 + (i >> 2);
        const int k1 = k0 + 4;
        const int a0 = k0 * 256 + ((blk ^ swz_rowk(k0)) << 5) + ((i & 3) << 3);
        const int a1 = k1 * 256 + ((blk ^ swz_rowk(k1)) << 5) + ((i & 3) << 3);
        typedef __attribute__((address_space(3))) short4_t* lds_p;
        const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(lds + a0));
        const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(lds + a1));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
}

template <bool A_KMAJOR, bool B_KMAJOR, bool C_F32>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(GemmB16Args g)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 stages][A tile | B tile]
    // ---- XCD-aware tile order: workgroup b runs on XCD b%8; give each XCD a contiguous run of tiles --------
    const int nwg = g.tiles_m * g.tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int tile_n = wgid % g.tiles_n, tile_m = wgid / g.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    // ---- K range of this split ------------------------------------------------------------------------------
    const int ktiles = (g.K + BK - 1) / BK;
    const int per = (ktiles + g.splitk - 1) / g.splitk;
    const int kt0 = blockIdx.y * per, kt1 = min(ktiles, kt0 + per);
    if (kt0 >= kt1) return;

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wm = (w >> 1) * 64, wn = (w & 1) * 64;
    float4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = float4_t{0.f, 0.f, 0.f, 0.f};

    Stage<A_KMAJOR> sa;
    Stage<B_KMAJOR> sb;
    sa.load(g.A, g.lda, m0, kt0 * BK, g.M, g.K);
    sb.load(g.B, g.ldb, n0, kt0 * BK, g.N, g.K);
    sa.store(smem);
    sb.store(smem + TILE_BYTES);
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
        const int cur = (kt - kt0) & 1;
        const char* la = smem + cur * 2 * TILE_BYTES;
        const char* lb = la + TILE_BYTES;
        const bool more = kt + 1 < kt1;
        if (more) {                                   // issue next tile's HBM loads under this tile's MFMAs
            sa.load(g.A, g.lda, m0, (kt + 1) * BK, g.M, g.K);
            sb.load(g.B, g.ldb, n0, (kt + 1) * BK, g.N, g.K);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            short8_t fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = load_frag<A_KMAJOR>(la, wm + i * 16, s, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = load_frag<B_KMAJOR>(lb, wn + j * 16, s, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)   // operands swapped: D^T tile, lane owns 4 consecutive n
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
        if (more) {
            char* na = smem + (cur ^ 1) * 2 * TILE_BYTES;
            sa.store(na);
            sb.store(na + TILE_BYTES);
        }
        __syncthreads();
    }
    // ---- epilogue: lane (i = l&15, g = l>>4): row m = .. + i, cols n = .. + g*4 + {0..3} -----------------------
    const int li = lane & 15, lg = lane >> 4;
    const bool first_split = blockIdx.y == 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm + i * 16 + li;
        if (m >= g.M) continue;
        const float rs = g.rowscale ? g.rowscale[m] : 1.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn + j * 16 + lg * 4;
            if (n >= g.N) continue;                    // N is a multiple of 4 (checked on the host)
            float4 v = make_float4(g.alpha * acc[i][j][0], g.alpha * acc[i][j][1], g.alpha * acc[i][j][2],
                                   g.alpha * acc[i][j][3]);
            if (g.bias && first_split) {
                const float4 bv = *reinterpret_cast<const float4*>(g.bias + n);
                v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
            }
            v.x *= rs; v.y *= rs; v.z *= rs; v.w *= rs;
            if (g.epilogue == EDITOR_EPI_RESIDUAL) {          // C = v + aux (fp32 residual stream)
                const float4 o = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(g.aux) + (long)m * g.ldaux + n);
                v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
            } else if (g.epilogue == EDITOR_EPI_GELU) {       // aux = v (pre-activation, bf16), C = gelu(v)
                uint2 pre; pre.x = pack_bf16x2(v.x, v.y); pre.y = pack_bf16x2(v.z, v.w);
                *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(g.aux) + (long)m * g.ldaux + n) = pre;
                v.x = gelu_f(__uint_as_float(pre.x << 16)); v.y = gelu_f(__uint_as_float(pre.x & 0xffff0000u));
                v.z = gelu_f(__uint_as_float(pre.y << 16)); v.w = gelu_f(__uint_as_float(pre.y & 0xffff0000u));
            } else if (g.epilogue == EDITOR_EPI_GELU_BWD) {   // C = v * gelu'(aux), aux = saved pre-activation
                const uint2 pre = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(g.aux) + (long)m * g.ldaux + n);
                v.x *= gelu_grad_f(__uint_as_float(pre.x << 16)); v.y *= gelu_grad_f(__uint_as_float(pre.x & 0xffff0000u));
                v.z *= gelu_grad_f(__uint_as_float(pre.y << 16)); v.w *= gelu_grad_f(__uint_as_float(pre.y & 0xffff0000u));
            }
            if (C_F32) {
                float* c = reinterpret_cast<float*>(g.C) + (long)m * g.ldc + n;
                if (g.splitk > 1) {
                    atomicAdd(c, v.x); atomicAdd(c + 1, v.y); atomicAdd(c + 2, v.z); atomicAdd(c + 3, v.w);
                } else {
                    if (g.beta != 0.f) {
                        const float4 o = *reinterpret_cast<const float4*>(c);
                        v.x += g.beta * o.x; v.y += g.beta * o.y; v.z += g.beta * o.z; v.w += g.beta * o.w;
                    }
                    *reinterpret_cast<float4*>(c) = v;
                }
            } else {
                bf16_t* c = reinterpret_cast<bf16_t*>(g.C) + (long)m * g.ldc + n;
                if (g.beta != 0.f) {
                    const uint2 o = *reinterpret_cast<const uint2*>(c);
                    v.x += g.beta * __uint_as_float(o.x << 16); v.y += g.beta * __uint_as_float(o.x & 0xffff0000u);
                    v.z += g.beta * __uint_as_float(o.y << 16); v.w += g.beta * __uint_as_float(o.y & 0xffff0000u);
                }
                uint2 o; o.x = pack_bf16x2(v.x, v.y); o.y = pack_bf16x2(v.z, v.w);
                *reinterpret_cast<uint2*>(c) = o;
            }
        }
    }
}

__global__ void scale_c_kernel(float* C, long rows, int cols, long ld, float beta)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * cols) return;
    float* c = C + (e / cols) * ld + (e % cols);
    *c = beta == 0.f ? 0.f : *c * beta;
}

template <bool AK, bool BK_, bool CF>
int launch(const GemmB16Args& g, hipStream_t stream)
{
    auto kern = gemm_bf16_kernel<AK, BK_, CF>;
    static bool attr_done = false;                       // per-instantiation; idempotent
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(g.tiles_m * g.tiles_n, g.splitk), dim3(256), 4 * TILE_BYTES, stream, g);
    EDITOR_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" int editor_gemm_bf16(const uint16_t* A, const uint16_t* B, void* C, int c_f32, int M, int N, int K, long lda,
    long ldb, long ldc, int transA, int transB, float alpha, float beta, const float* bias, const float* rowscale,
    int splitk, int epilogue, void* aux, long ldaux, hipStream_t stream)
{
    if (epilogue != EDITOR_EPI_NONE && (!aux || (ldaux & 3) || splitk > 1)) return (int)hipErrorInvalidValue;
    if (M <= 0 || N <= 0 || K <= 0) return (int)hipErrorInvalidValue;
    // 16-byte vector accesses: leading dimensions and the contiguous extents must be multiples of 8 bf16
    if ((lda & 7) || (ldb & 7) || (N & 3) || (ldc & 3)) return (int)hipErrorInvalidValue;
    if (!transA && (K & 7)) return (int)hipErrorInvalidValue;
    if (transA && (M & 7)) return (int)hipErrorInvalidValue;
    if (!transB && (K & 7)) return (int)hipErrorInvalidValue;
    if (transB && (N & 7)) return (int)hipErrorInvalidValue;
    if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B) | reinterpret_cast<uintptr_t>(C)) & 15)
        return (int)hipErrorInvalidValue;
    if (splitk < 1) splitk = 1;
    const int ktiles = (K + BK - 1) / BK;
    if (splitk > ktiles) splitk = ktiles;
    if (splitk > 1) {
        if (!c_f32) return (int)hipErrorInvalidValue;
        if (beta != 1.f) {
            hipLaunchKernelGGL(scale_c_kernel, dim3((unsigned)(((long)M * N + 255) / 256)), dim3(256), 0, stream,
                               (float*)C, (long)M, N, ldc, beta);
            EDITOR_LAUNCH_CHECK();
        }
    }
    GemmB16Args g{(const bf16_t*)A, (const bf16_t*)B, C, M, N, K, lda, ldb, ldc, alpha, beta, bias, rowscale, splitk,
                  (M + BM - 1) / BM, (N + BN - 1) / BN, epilogue, aux, ldaux};
    const int sel = (transA ? 0 : 4) | (transB ? 0 : 2) | (c_f32 ? 1 : 0);
    switch (sel) {
        case 7: return launch<true, true, true>(g, stream);
        case 6: return launch<true, true, false>(g, stream);
        case 5: return launch<true, false, true>(g, stream);
        case 4: return launch<true, false, false>(g, stream);
        case 3: return launch<false, true, true>(g, stream);
        case 2: return launch<false, true, false>(g, stream);
        case 1: return launch<false, false, true>(g, stream);
        default: return launch<false, false, false>(g, stream);
    }
}
