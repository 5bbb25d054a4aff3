// Generic fp32 GEMM on the exact-f32 matrix cores (v_mfma_f32_16x16x4_f32), gfx950.
// This is the PARITY-mode contraction (cfg.MODEL.COMPUTE_DTYPE='f32') and the kernel for every small /
// odd-shaped product of the hot path (classifier heads, REDUCE layers, per-head attention products in
// parity mode).  Any M,N,K; A and B in either storage order; two-level batching; split-K.
//     C[m,n] = alpha * sum_k opA(A)[m,k] * opB(B)[k,n]  (+ bias[n])  (+ beta * C[m,n])  (* rowscale[m])
// Replaces F.linear / torch.matmul call sites of vit_pytorch.py:139-145,184-198,240-258 in fp32.
#include "common.h"
#include "../../include/editor_hip.h"

typedef __attribute__((ext_vector_type(4))) float float4_t;

namespace {

constexpr int BM = 64, BN = 64, BK = 16, LDT = 80;   // LDT: LDS row stride (floats); 80 % 32 == 16 -> conflict-free

struct GemmArgs {
    const float* A; const float* B; float* C;
    int M, N, K;
    long lda, ldb, ldc;
    int transA, transB;
    int batch2; long sA1, sB1, sC1, sA2, sB2, sC2;
    float alpha, beta;
    const float* bias; const float* rowscale;
    int splitk;
    int epilogue; float* aux; long ldaux;
};

__device__ __forceinline__ float gelu_f(float a) { return 0.5f * a * (1.f + erff(a * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float a) {
    return 0.5f * (1.f + erff(a * 0.70710678118654752f)) + a * 0.3989422804014327f * __expf(-0.5f * a * a);
}

// One (rows x BK) tile of a row-major operand P[r][k] (ld) or of its transpose P[k][r], as two float4 per thread.  LOAD and STORE are
// separate steps (round 4): the global loads of K-tile t+1 are issued before the MFMAs of K-tile t and written to LDS after them.
// They used to be one step: every 16-deep K-tile paid a full global-memory latency in front of its 8 MFMAs.  (BK = 32 with the same
// split measured SLOWER - 0.67 vs 0.54 ms per step over the 22 head / REDUCE products: 8-way instead of 4-way LDS write conflicts.)
constexpr int NH = BK / 16;                        // float4 chunks per thread and operand tile
struct TileRegs { float v[NH][4]; };

__device__ __forceinline__ void load_tile(const float* __restrict__ P, long ld, int trans, int r0, int k0, int R, int kend, TileRegs& t)
{
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        const int c = threadIdx.x + 256 * h;            // float4 chunk of the 64 x BK tile
        float* v = t.v[h];
        v[0] = v[1] = v[2] = v[3] = 0.f;
        if (!trans) {                                   // k contiguous: chunk -> row, 4 consecutive k
            const int r = c / (BK / 4), kk = (c % (BK / 4)) * 4;
            const int gr = r0 + r, gk = k0 + kk;
            if (gr < R) {
                const float* src = P + (long)gr * ld + gk;
                if (gk + 3 < kend && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) {
                    const float4 q = *reinterpret_cast<const float4*>(src);
                    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) if (gk + i < kend) v[i] = src[i];
                }
            }
        } else {                                        // r contiguous: chunk -> k c/16, 4 consecutive rows
            const int kk = c >> 4, r = (c & 15) * 4;
            const int gk = k0 + kk, gr = r0 + r;
            if (gk < kend) {
                const float* src = P + (long)gk * ld + gr;
                if (gr + 3 < R && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) {
                    const float4 q = *reinterpret_cast<const float4*>(src);
                    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) if (gr + i < R) v[i] = src[i];
                }
            }
        }
    }
}
__device__ __forceinline__ void store_tile(const TileRegs& t, int trans, float (*S)[LDT])
{
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        const int c = threadIdx.x + 256 * h;
        const float* v = t.v[h];
        if (!trans) {
            const int r = c / (BK / 4), kk = (c % (BK / 4)) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) S[kk + i][r] = v[i];
        } else {
            const int kk = c >> 4, r = (c & 15) * 4;
            *reinterpret_cast<float4*>(&S[kk][r]) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void gemm_f32_kernel(GemmArgs g)
{
    __shared__ __attribute__((aligned(16))) float As[2][BK][LDT];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][LDT];
    const int z = blockIdx.z;
    const int split = z % g.splitk, bz = z / g.splitk;
    const int b1 = bz / g.batch2, b2 = bz % g.batch2;
    const float* A = g.A + b1 * g.sA1 + b2 * g.sA2;
    const float* B = g.B + b1 * g.sB1 + b2 * g.sB2;
    float* C = g.C + b1 * g.sC1 + b2 * g.sC2;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    // K range of this split (multiples of BK)
    const int ktiles = (g.K + BK - 1) / BK;
    const int per = (ktiles + g.splitk - 1) / g.splitk;
    const int kt0 = split * per, kt1 = min(ktiles, kt0 + per);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wm = (w >> 1) * 32, wn = (w & 1) * 32;
    float4_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = float4_t{0.f, 0.f, 0.f, 0.f};

    // opB(B)[k][n]: transB==0 means B is stored [N][K] (nn.Linear weight) -> "rows"=n, k contiguous
    TileRegs ra, rb;
    if (kt0 < kt1) {
        load_tile(A, g.lda, g.transA, m0, kt0 * BK, g.M, g.K, ra);
        load_tile(B, g.ldb, g.transB, n0, kt0 * BK, g.N, g.K, rb);
        store_tile(ra, g.transA, As[0]);
        store_tile(rb, g.transB, Bs[0]);
    }
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
        const int cur = (kt - kt0) & 1;
        const bool more = kt + 1 < kt1;
        if (more) {                                     // next K-tile's loads in flight under this tile's MFMAs
            load_tile(A, g.lda, g.transA, m0, (kt + 1) * BK, g.M, g.K, ra);
            load_tile(B, g.ldb, g.transB, n0, (kt + 1) * BK, g.N, g.K, rb);
        }
#pragma unroll
        for (int ks = 0; ks < BK; ks += 4) {
            const int kr = ks + (lane >> 4);
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = As[cur][kr][wm + i * 16 + (lane & 15)];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = Bs[cur][kr][wn + j * 16 + (lane & 15)];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (more) {
            store_tile(ra, g.transA, As[cur ^ 1]);
            store_tile(rb, g.transB, Bs[cur ^ 1]);
        }
        __syncthreads();
    }
    // epilogue: C/D map col = lane&15, row = (lane>>4)*4 + r
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm + i * 16 + (lane >> 4) * 4 + r;
                const int n = n0 + wn + j * 16 + (lane & 15);
                if (m >= g.M || n >= g.N) continue;
                float v = g.alpha * acc[i][j][r];
                if (g.bias && split == 0) v += g.bias[n];
                if (g.rowscale) v *= g.rowscale[m];
                if (g.epilogue == EDITOR_EPI_RESIDUAL) v += g.aux[(long)m * g.ldaux + n];
                else if (g.epilogue == EDITOR_EPI_GELU) { g.aux[(long)m * g.ldaux + n] = v; v = gelu_f(v); }
                else if (g.epilogue == EDITOR_EPI_GELU_BWD) v *= gelu_grad_f(g.aux[(long)m * g.ldaux + n]);
                float* c = C + (long)m * g.ldc + n;
                if (g.splitk > 1) atomicAdd(c, v);
                else *c = g.beta != 0.f ? v + g.beta * *c : v;
            }
}

__global__ void scale_inplace_kernel(float* C, long rows, int cols, long ld, float beta)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * cols) return;
    float* c = C + (e / cols) * ld + (e % cols);
    *c = beta == 0.f ? 0.f : *c * beta;
}

}  // namespace

extern "C" int editor_gemm_f32(const float* A, const float* B, float* C, int M, int N, int K, long lda, long ldb,
    long ldc, int transA, int transB, int batch1, long sA1, long sB1, long sC1, int batch2, long sA2, long sB2, long sC2,
    float alpha, float beta, const float* bias, const float* rowscale, int splitk, int epilogue, float* aux, long ldaux,
    hipStream_t stream)
{
    if (epilogue != EDITOR_EPI_NONE && (!aux || splitk > 1 || batch1 * batch2 != 1)) return (int)hipErrorInvalidValue;
    if (M <= 0 || N <= 0 || K <= 0 || batch1 < 1 || batch2 < 1) return (int)hipErrorInvalidValue;
    if (splitk < 1) splitk = 1;
    const int ktiles = (K + BK - 1) / BK;
    if (splitk > ktiles) splitk = ktiles;
    if (splitk > 1) {
        if (rowscale) return (int)hipErrorInvalidValue;
        // pre-scale C by beta (0 -> clear), then every split accumulates atomically
        for (int b1 = 0; b1 < batch1; ++b1)
            for (int b2 = 0; b2 < batch2; ++b2) {
                float* c = C + b1 * sC1 + b2 * sC2;
                if (beta != 1.f) {
                    hipLaunchKernelGGL(scale_inplace_kernel, dim3((unsigned)(((long)M * N + 255) / 256)), dim3(256), 0,
                                       stream, c, (long)M, N, ldc, beta);
                    EDITOR_LAUNCH_CHECK();
                }
            }
    }
    GemmArgs g{A, B, C, M, N, K, lda, ldb, ldc, transA, transB, batch2, sA1, sB1, sC1, sA2, sB2, sC2,
               alpha, beta, bias, rowscale, splitk, epilogue, aux, ldaux};
    dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM, batch1 * batch2 * splitk);
    hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, stream, g);
    EDITOR_LAUNCH_CHECK();
    return 0;
}
