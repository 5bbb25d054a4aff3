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
// Three tilings, one MFMA / LDS-image vocabulary (the MFMA is issued with its operands swapped - a D^T tile - so each lane
// owns 4 CONSECUTIVE output columns):
//   gemm_bf16_pp_kernel    256x256x64 "ping-pong", 8 waves as two groups one barrier apart, LDS-DMA staging, counted vmcnt:
//                          THE path's kernel - every forward, dgrad and (one round of split-K workgroups) wgrad product of
//                          the transformer blocks, 208-row short tiles for the 768-wide outputs, and the split-precision
//                          (hi.hi + hi.lo + lo.hi) forward of the 'f16x2' mode
//   gemm_bf16_pipe_kernel  256x128x64, three LDS-DMA stages: shapes the ping-pong kernel does not take
//   gemm_bf16_kernel       128x128x64, 4 waves, register- or DMA-staged: small / ragged problems (K % 64 != 0, N < 128)
// XCD-aware tile order (consecutive n-tiles of one A row-panel stay on one XCD's L2); split-K through per-split slabs and a
// fixed-order reduction (deterministic), fp32 atomics only on the generic kernel.
#include "common.h"
#include "../../include/editor_hip.h"
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <type_traits>

typedef __attribute__((ext_vector_type(4))) short short4_t;
typedef __attribute__((ext_vector_type(8))) short short8_t;
typedef __attribute__((ext_vector_type(4))) float float4_t;

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));

namespace {

// F16 = false: bfloat16 operands (v_mfma_f32_16x16x32_bf16); true: IEEE half (v_mfma_f32_16x16x32_f16, same rate, 3 more
// mantissa bits - the reference's own autocast dtype, engine/processor.py:79).  Fragments travel as raw 16-bit lanes.
template <bool F16>
__device__ __forceinline__ float4_t mfma16(short8_t a, short8_t b, float4_t c)
{
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, a), __builtin_bit_cast(half8_t, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// 32x32x16 form (EDITOR_PP_MI32, round 6): the same FLOPs per matrix-core cycle in half the MFMA issues; operand lane l holds row
// l & 31, k-slice (l >> 5) * 8 .. + 7; D: col = l & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (l >> 5)
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
template <bool F16>
__device__ __forceinline__ f32x16_t mfma32(short8_t a, short8_t b, f32x16_t c)
{
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, a), __builtin_bit_cast(half8_t, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * 64 * 2;          // 16 KiB per operand tile per stage

#ifndef EDITOR_GEMM_GM
#define EDITOR_GEMM_GM 4
#endif
#ifndef EDITOR_PP_MI32
#define EDITOR_PP_MI32 0        // 1: full-tile forward / dgrad products with 16-bit staged epilogues run on v_mfma_f32_32x32x16 (A/B: tools/gemm_alt_ab.py)
#endif
#ifndef EDITOR_PP_MI32_PHASES
#define EDITOR_PP_MI32_PHASES 4 // the MI32 instantiations' K-tile: 4 phases (8 MFMAs on TWO accumulators each) or 2 (16 MFMAs round-robin over FOUR)
#endif
#ifndef EDITOR_PP_PHASES
#define EDITOR_PP_PHASES 0      // phases per K-tile of the ping-pong kernel: 4 (16-MFMA clusters), 2 (32-MFMA clusters, round 5), 0 = per
#endif                          // operand layout as measured (tools/gemm_alt_ab.py): 2 for the weight gradients, 4 for everything else

struct GemmB16Args {
    const bf16_t* A; const bf16_t* B; void* C;
    int M, N, K;
    long lda, ldb, ldc;
    float alpha, beta;
    const float* bias; const float* rowscale;
    int splitk, tiles_m, tiles_n;
    int epilogue; void* aux; long ldaux;     // EDITOR_EPI_* (editor_hip.h)
    int slabs;                               // split-K partial tiles go to per-split slabs of C (= workspace)
    const int* m_live;                       // device scalar: only the first *m_live token rows are live (NULL: all)
    int live_is_k;                           // the token-row extent is the reduction (wgrad) instead of M
    int pp_staged;                           // 256x256 kernel: LDS-staged epilogue (full-line stores) instead of the direct one
    float* colsum;                           // EDITOR_EPI_COLSUM: [tiles_m][N] column sums of the rounded output tile rows
    unsigned long long* trace;               // debug (EDITOR_GEMM_TRACE): per-workgroup s_memtime stamps
    int force_pp;                            // EDITOR_EPI_FORCE_PP: the 256x256 kernel whatever the heuristic says
    int aux_grad;                            // EDITOR_EPI_AUX_GRAD: aux holds gelu'(pre-activation), not the pre-activation
    int tile_frags;                          // EDITOR_EPI_TILE_ROWS: 16-row fragments per tile of the ping-pong kernel (13; 0 = 16)
    // split-precision forward (editor_gemm_f16x2): every operand is a PAIR of half matrices x = hi + lo of the same shape and
    // leading dimension; the product is hi.hi + lo.hi + hi.lo in one fp32 accumulator (three passes over K)
    const bf16_t* A_lo; const bf16_t* B_lo;
    void* C_lo;                              // 16-bit outputs leave as a pair too (C = hi, C_lo = lo); NULL for fp32 outputs
    int linear_ids;                          // ping-pong kernel: `bid` already is the position in the grouped tile order (the
                                             // grouped weight-gradient launch does its own XCD mapping)
    int prefer_pipe;                         // EDITOR_EPI_PIPE128
    int stagger;                             // EDITOR_EPI_STAGGER(c): the first round's workgroups start spread over c * 2048 cycles (ping-pong kernel)
    int ablate;                              // debug build only (EDITOR_GEMM_ABLATE, tools/gemm_bound_probe.py): 1 = no LDS-DMA inside the
                                             // K loop, 2 = no MFMAs, 4 = no fragment reads - what each costs, by deletion
    const int* rowmap;                       // EDITOR_EPI_RESIDUAL, fp32 C (editor_gemm_h16_rows): output row m lands in row rowmap[m] of
                                             // C / aux / rowscale - the product ran on COMPACTED token rows (stochastic-depth: live samples
                                             // first, editor_droppath_plan) and its residual epilogue scatters them back
};

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute of a kernel: remember it per device index, so a
// second GPU in the same process (cuda:1 tensors) does not launch > 64 KiB kernels without it
template <auto KERN>
int ensure_lds(int bytes)
{
    static bool done[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= 64 || !done[dev]) {
        e = hipFuncSetAttribute((const void*)KERN, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return (int)e;
        if (dev >= 0 && dev < 64) done[dev] = true;
    }
    return 0;
}

// compute units of the current device (cached per device index): the stagger below is laid out for 256 CUs in 8 XCDs
inline int device_cus()
{
    static int cus[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (!cus[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        cus[dev] = n;
    }
    return cus[dev];
}

// exact-erf GELU in fp32 (split-precision forward: the activation is evaluated on the UNROUNDED pre-activation, as
// nn.GELU does on the reference's fp32 CPU path, vit_pytorch.py:139-145)
__device__ __forceinline__ float gelu_exact(float a) { return 0.5f * a * (1.f + erff(a * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_exact(float a)
{
    return 0.5f * (1.f + erff(a * 0.70710678118654752f)) + a * 0.3989422804014327f * __expf(-0.5f * a * a);
}

// exact-erf GELU (nn.GELU default) for bf16 outputs, two values per instruction (v_pk_fma_f32 / v_pk_mul_f32):
// Phi(a) = 0.5 erfc(-a/sqrt2) with erfc(z) ~ (1 + a1 z + ... + a6 z^6)^-16 for z >= 0 (Abramowitz-Stegun 7.1.28,
// |abs err| <= 3e-7 -> 8e-7 on gelu in fp32 arithmetic, four orders below bf16 resolution): six packed FMAs, four packed
// squarings and ONE reciprocal per value - no exponential, where libm's erff is branchy and the 7.1.26 form needs
// v_exp + v_rcp (quarter-rate instructions; the GELU epilogue was VALU-bound on them).  The f32 parity kernels keep erff.
typedef float v2f_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f_t phi2(v2f_t a)
{
    const v2f_t z = __builtin_elementwise_abs(a) * 0.70710678118654752f;
    v2f_t p = __builtin_elementwise_fma(z, (v2f_t)(0.0000430638f), (v2f_t)(0.0002765672f));
    p = __builtin_elementwise_fma(p, z, (v2f_t)(0.0001520143f));
    p = __builtin_elementwise_fma(p, z, (v2f_t)(0.0092705272f));
    p = __builtin_elementwise_fma(p, z, (v2f_t)(0.0422820123f));
    p = __builtin_elementwise_fma(p, z, (v2f_t)(0.0705230784f));
    p = __builtin_elementwise_fma(p, z, (v2f_t)(1.0f));
    p = p * p; p = p * p; p = p * p; p = p * p;                              // overflow -> inf -> rcp 0: the tails are exact
    v2f_t h, r;
    h.x = 0.5f * __builtin_amdgcn_rcpf(p.x); h.y = 0.5f * __builtin_amdgcn_rcpf(p.y);     // 0.5 erfc(|a|/sqrt2)
    r.x = a.x > 0.f ? 1.f - h.x : h.x; r.y = a.y > 0.f ? 1.f - h.y : h.y;
    return r;
}
__device__ __forceinline__ v2f_t gelu2(v2f_t a) { return a * phi2(a); }
// d/da gelu = Phi(a) + a phi(a), phi(a) = exp(-a^2/2)/sqrt(2 pi)
__device__ __forceinline__ v2f_t gelu_grad2_phi(v2f_t a, v2f_t ph)             // ph = phi2(a)
{
    const v2f_t q = a * a * -0.72134752044448170f;                           // -a^2/2 * log2(e)
    v2f_t e; e.x = __builtin_amdgcn_exp2f(q.x); e.y = __builtin_amdgcn_exp2f(q.y);
    return __builtin_elementwise_fma(a * 0.3989422804014327f, e, ph);
}
__device__ __forceinline__ v2f_t gelu_grad2(v2f_t a) { return gelu_grad2_phi(a, phi2(a)); }
// Both at once, Phi evaluated ONCE (round 6: the f16 fc1 epilogue - no table there - called gelu2 and gelu_grad2 in different basic
// blocks, so the compiler kept two copies of the erfc polynomial + reciprocal per element).  The same operations in the same order
// as the two functions above: bit-identical results.
__device__ __forceinline__ void gelu_both2(v2f_t a, bool want_grad, v2f_t& gv, v2f_t& dv)
{
    const v2f_t ph = phi2(a);
    if (want_grad) dv = gelu_grad2_phi(a, ph);
    gv = a * ph;
}
// GELU lookup for bf16 pre-activations (fc1's forward epilogue, ping-pong kernel).  The epilogue evaluates gelu and gelu' on
// values that have ALREADY been rounded to bf16 - 16 bits of input - and was VALU-bound on it (~33 instructions + 4
// transcendentals per pair of elements, ~20 k cycles per 256x256 tile against 29 k for the K = 768 main loop).  A table over
// the bf16 values with 2^-12 <= |x| < 16 (16 exponents x 128 mantissas x 2 signs = 4 096 entries of {gelu, gelu'} packed in 32
// bits, 16 KiB of LDS) is filled per tile BY THE SAME FUNCTIONS (8 entries per thread), so a lookup returns exactly the bits
// the arithmetic would; the rare chunk holding a value outside that range takes the arithmetic path.
constexpr uint32_t kLutBase = 115u << 7, kLutSpan = 2048u, kLutBytes = 2 * kLutSpan * 4;   // bf16 exponent field 115 = 2^-12
__device__ __forceinline__ uint32_t gelu_lut_bits(uint32_t j) { return ((j & (kLutSpan - 1)) + kLutBase) | ((j >> 11) << 15); }

__device__ __forceinline__ float gelu_f(float a) { return gelu2(v2f_t{a, a}).x; }
__device__ __forceinline__ float gelu_grad_f(float a) { return gelu_grad2(v2f_t{a, a}).x; }

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

// Direct HBM->LDS staging (global_load_lds_dwordx4: LDS address = wave-uniform base + lane*16, so the image is
// written linearly and the XOR swizzle is applied to the per-lane SOURCE address; the read side applies the same
// involution).  Each wave issues 4 x 1 KiB pieces per operand tile.  Requires the reduction extent to be a multiple of
// BK (no zero fill possible); free-dimension tails are clamped to a valid address (their outputs are never stored).
template <bool KMAJOR>
__device__ __forceinline__ void stage_glds(const bf16_t* __restrict__ P, long ld, int r0, int k0, int R, char* lds,
                                           int wave, int lane)
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int piece = wave * 4 + j;                       // 1 KiB piece of the 16 KiB tile
        long off;
        if (KMAJOR) {
            const int row = piece * 8 + (lane >> 3);
            const int c = (lane & 7) ^ (row & 7);
            const int gr = min(r0 + row, R - 1);
            off = (long)gr * ld + k0 + c * 8;
        } else {
            const int kr = piece * 4 + (lane >> 4);
            const int p = lane & 15;
            const int col = ((((p >> 1) ^ swz_rowk(kr)) << 4) | ((p & 1) << 3));
            const int gc = min(r0 + col, R - 8);
            off = (long)(k0 + kr) * ld + gc;
        }
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(P + off),
                                         (__attribute__((address_space(3))) void*)(lds + piece * 1024), 16, 0, 0);
    }
}

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

template <bool F16, bool C_F32, int MT>
__device__ __forceinline__ void epilogue_store(const GemmB16Args& g, float4_t (&acc)[MT][4], int mbase, int nbase, int lane,
                                               bool first_split)
{
    // lane (i = l&15, g = l>>4): row m = mbase + 16*i' + i, cols n = nbase + 16*j + g*4 + {0..3}
    const int li = lane & 15, lg = lane >> 4;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = mbase + i * 16 + li;
        if (m >= g.M) continue;
        const float rs = g.rowscale ? g.rowscale[m] : 1.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = nbase + j * 16 + lg * 4;
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
                uint2 pre; pre.x = H16<F16>::pack2(v.x, v.y); pre.y = H16<F16>::pack2(v.z, v.w);
                uint2 sv = pre;
                if (g.aux_grad) {
                    const v2f_t d0 = gelu_grad2(H16<F16>::unpack2(pre.x)), d1 = gelu_grad2(H16<F16>::unpack2(pre.y));
                    sv.x = H16<F16>::pack2(d0.x, d0.y); sv.y = H16<F16>::pack2(d1.x, d1.y);
                }
                if (g.aux) *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(g.aux) + (long)m * g.ldaux + n) = sv;
                const v2f_t g0 = gelu2(H16<F16>::unpack2(pre.x)), g1 = gelu2(H16<F16>::unpack2(pre.y));
                v.x = g0.x; v.y = g0.y; v.z = g1.x; v.w = g1.y;
            } else if (g.epilogue == EDITOR_EPI_GELU_BWD) {   // C = v * gelu'(aux), aux = saved pre-activation
                const uint2 pre = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(g.aux) + (long)m * g.ldaux + n);
                const v2f_t u0 = H16<F16>::unpack2(pre.x), u1 = H16<F16>::unpack2(pre.y);
                const v2f_t g0 = g.aux_grad ? u0 : gelu_grad2(u0), g1 = g.aux_grad ? u1 : gelu_grad2(u1);
                v.x *= g0.x; v.y *= g0.y; v.z *= g1.x; v.w *= g1.y;
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
                    const v2f_t o0 = H16<F16>::unpack2(o.x), o1 = H16<F16>::unpack2(o.y);
                    v.x += g.beta * o0.x; v.y += g.beta * o0.y; v.z += g.beta * o1.x; v.w += g.beta * o1.y;
                }
                uint2 o; o.x = H16<F16>::pack2(v.x, v.y); o.y = H16<F16>::pack2(v.z, v.w);
                *reinterpret_cast<uint2*>(c) = o;
            }
        }
    }
}

template <bool F16, bool A_KMAJOR, bool B_KMAJOR, bool C_F32, bool GLDS>
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

    const int wu = __builtin_amdgcn_readfirstlane(w);
    Stage<A_KMAJOR> sa;
    Stage<B_KMAJOR> sb;
    if (GLDS) {
        stage_glds<A_KMAJOR>(g.A, g.lda, m0, kt0 * BK, g.M, smem, wu, lane);
        stage_glds<B_KMAJOR>(g.B, g.ldb, n0, kt0 * BK, g.N, smem + TILE_BYTES, wu, lane);
    } else {
        sa.load(g.A, g.lda, m0, kt0 * BK, g.M, g.K);
        sb.load(g.B, g.ldb, n0, kt0 * BK, g.N, g.K);
        sa.store(smem);
        sb.store(smem + TILE_BYTES);
    }
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
        const int cur = (kt - kt0) & 1;
        const char* la = smem + cur * 2 * TILE_BYTES;
        const char* lb = la + TILE_BYTES;
        const bool more = kt + 1 < kt1;
        if (more) {                                   // issue next tile's HBM loads under this tile's MFMAs
            if (GLDS) {
                char* na = smem + (cur ^ 1) * 2 * TILE_BYTES;
                stage_glds<A_KMAJOR>(g.A, g.lda, m0, (kt + 1) * BK, g.M, na, wu, lane);
                stage_glds<B_KMAJOR>(g.B, g.ldb, n0, (kt + 1) * BK, g.N, na + TILE_BYTES, wu, lane);
            } else {
                sa.load(g.A, g.lda, m0, (kt + 1) * BK, g.M, g.K);
                sb.load(g.B, g.ldb, n0, (kt + 1) * BK, g.N, g.K);
            }
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
                    acc[i][j] = mfma16<F16>(fb[j], fa[i], acc[i][j]);
        }
        if (more && !GLDS) {
            char* na = smem + (cur ^ 1) * 2 * TILE_BYTES;
            sa.store(na);
            sb.store(na + TILE_BYTES);
        }
        __syncthreads();                               // (GLDS: also drains the in-flight LDS-DMA, vmcnt(0))
    }
    epilogue_store<F16, C_F32, 4>(g, acc, m0 + wm, n0 + wn, lane, blockIdx.y == 0);
}

// ---------------------------------------------------------------------------------------------------------
// LDS-staged epilogue.  Measured (tools/gemm_bench.py with the stores removed): the direct epilogue above - every
// lane storing its own 8 / 16 bytes, i.e. 32-byte pieces of 16 different rows per wave instruction - costs 30-45 % of
// the kernel (590 -> 890-1040 TFLOP/s without it): the store path is issue bound on partial lines.  Here the finished
// tile is written to LDS in the output dtype (padded rows, conflict-free) and leaves as 16 bytes per lane with the
// 16/32 lanes of a row covering 256/512 CONTIGUOUS bytes: full 128-byte lines, 4x fewer store instructions.
// Split-K partial tiles go to per-split slabs of a workspace (plain stores, reduced by a second tiny kernel) instead
// of fp32 atomics.
// ---------------------------------------------------------------------------------------------------------
// The accumulators are parked in LDS as raw fp32 (padded rows), then a ROLLED loop - 8 output columns per lane per
// trip - applies alpha / bias / row scale / residual / GELU and stores 16 bytes (bf16) or 2 x 16 bytes (fp32) per lane,
// the 16 lanes of a tile row covering 256 / 512 contiguous bytes.  Keeping the math in a rolled loop matters: the fully
// unrolled per-fragment epilogue is ~10k straight-line instructions executed once per tile, i.e. always instruction-cache
// cold (measured: ~5.5 us per tile even with the global stores removed).
template <bool F16, bool C_F32, int EPI, int PBM, int PBN, int NTHREADS, bool SPLIT = false>
__device__ __forceinline__ void epilogue_copy_out(const GemmB16Args& g, const char* lds, int m0, int n0, int split, int mlim)
{
    constexpr int RBP = PBN * 4 + 16;
    constexpr int GPR = PBN / 8;                                // 8-column groups per tile row
    constexpr int ITERS = PBM * GPR / NTHREADS;                 // trips per thread (8 for 256x128 / 512 threads)
    static_assert(PBM * GPR % NTHREADS == 0, "tile must divide evenly");
    const bool add_bias = g.bias && split == 0;
    float* Cf = C_F32 ? reinterpret_cast<float*>(g.C) + (g.splitk > 1 ? (long)split * g.M * g.ldc : 0L) : nullptr;
    bf16_t* Cb = reinterpret_cast<bf16_t*>(g.C);
    // row scatter (compacted rows): only the LIVE rows are written - the rows of the last live tile behind *m_live belong to dropped
    // samples, whose output rows the producer of the compacted operand has already filled (and whose A rows may lie beyond the tiles
    // the product before this one computed: with a different tile height there, 0 * garbage would be NaN)
    if (EPI == EDITOR_EPI_RESIDUAL && g.rowmap && g.m_live) mlim = min(mlim, *g.m_live);
    // branch-free body per epilogue kind, all LDS reads of a half issued before the first use
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        float4 lo[ITERS / 2], hi[ITERS / 2];
        // the epilogue's global operands (residual rows / saved pre-activations) are requested for the whole half up
        // front, so that their latency is paid once and under the LDS reads instead of once per item
        float4 ra[EPI == EDITOR_EPI_RESIDUAL ? ITERS / 2 : 1], rb[EPI == EDITOR_EPI_RESIDUAL ? ITERS / 2 : 1];
        uint4 pa[EPI == EDITOR_EPI_GELU_BWD ? ITERS / 2 : 1];
        if (EPI == EDITOR_EPI_RESIDUAL || EPI == EDITOR_EPI_GELU_BWD) {
#pragma unroll
            for (int it = 0; it < ITERS / 2; ++it) {
                const int c = threadIdx.x + (half * (ITERS / 2) + it) * NTHREADS;
                int m = min(m0 + c / GPR, g.M - 1);
                const int n = min(n0 + (c % GPR) * 8, g.N - 8);
                if (EPI == EDITOR_EPI_RESIDUAL) {
                    if (g.rowmap) m = g.rowmap[m];
                    const float* r = reinterpret_cast<const float*>(g.aux) + (long)m * g.ldaux + n;
                    ra[it] = *reinterpret_cast<const float4*>(r); rb[it] = *reinterpret_cast<const float4*>(r + 4);
                } else {
                    pa[it] = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(g.aux) + (long)m * g.ldaux + n);
                }
            }
        }
#pragma unroll
        for (int it = 0; it < ITERS / 2; ++it) {
            const int c = threadIdx.x + (half * (ITERS / 2) + it) * NTHREADS;
            const int row = c / GPR, cg = c % GPR;
            lo[it] = *reinterpret_cast<const float4*>(lds + row * RBP + cg * 32);
            hi[it] = *reinterpret_cast<const float4*>(lds + row * RBP + cg * 32 + 16);
        }
#pragma unroll
        for (int it = 0; it < ITERS / 2; ++it) {
            const int c = threadIdx.x + (half * (ITERS / 2) + it) * NTHREADS;
            const int row = c / GPR, cg = c % GPR;
            int m = m0 + row;
            const int n = n0 + cg * 8;
            if (m >= mlim || n >= g.N) continue;                 // N is a multiple of 8 on this path (checked on the host)
            if (EPI == EDITOR_EPI_RESIDUAL && g.rowmap) m = g.rowmap[m];     // (compacted rows: scatter back, see GemmB16Args)
            float x[8] = {lo[it].x, lo[it].y, lo[it].z, lo[it].w, hi[it].x, hi[it].y, hi[it].z, hi[it].w};
            const float rs = g.rowscale ? g.rowscale[m] : 1.f;
            float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (add_bias) {                                      // n is a multiple of 8: two aligned 16-byte loads
                const float4 b0 = *reinterpret_cast<const float4*>(g.bias + n), b1 = *reinterpret_cast<const float4*>(g.bias + n + 4);
                bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w; bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = (x[e] * g.alpha + bv[e]) * rs;
            if (EPI == EDITOR_EPI_RESIDUAL) {
                const float4 r0 = ra[it], r1 = rb[it];
                x[0] += r0.x; x[1] += r0.y; x[2] += r0.z; x[3] += r0.w; x[4] += r1.x; x[5] += r1.y; x[6] += r1.z; x[7] += r1.w;
            } else if (EPI == EDITOR_EPI_GELU && SPLIT) {        // aux <- gelu'(x) (half, for the 16-bit backward), C <- gelu(x) in fp32
                uint4 p;
                p.x = H16<F16>::pack2(gelu_grad_exact(x[0]), gelu_grad_exact(x[1])); p.y = H16<F16>::pack2(gelu_grad_exact(x[2]), gelu_grad_exact(x[3]));
                p.z = H16<F16>::pack2(gelu_grad_exact(x[4]), gelu_grad_exact(x[5])); p.w = H16<F16>::pack2(gelu_grad_exact(x[6]), gelu_grad_exact(x[7]));
                if (g.aux) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(g.aux) + (long)m * g.ldaux + n) = p;
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = gelu_exact(x[e]);
            } else if (EPI == EDITOR_EPI_GELU) {                 // aux <- pre-activation (bf16), C <- gelu(rounded pre-activation)
                uint4 p;
                p.x = H16<F16>::pack2(x[0], x[1]); p.y = H16<F16>::pack2(x[2], x[3]); p.z = H16<F16>::pack2(x[4], x[5]); p.w = H16<F16>::pack2(x[6], x[7]);
                const uint32_t pw[4] = {p.x, p.y, p.z, p.w};
                const bool want_grad = g.aux_grad && g.aux;      // save gelu'(rounded pre-activation) for the backward instead
                uint32_t dw_[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v2f_t gv, dv = v2f_t{0.f, 0.f};
                    gelu_both2(H16<F16>::unpack2(pw[e]), want_grad, gv, dv);
                    dw_[e] = H16<F16>::pack2(dv.x, dv.y);
                    x[2 * e] = gv.x; x[2 * e + 1] = gv.y;
                }
                if (want_grad) p = make_uint4(dw_[0], dw_[1], dw_[2], dw_[3]);
                if (g.aux) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(g.aux) + (long)m * g.ldaux + n) = p;
            } else if (EPI == EDITOR_EPI_GELU_BWD) {
                const uint4 p = pa[it];
                const uint32_t pw[4] = {p.x, p.y, p.z, p.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const v2f_t uv = H16<F16>::unpack2(pw[e]);
                    const v2f_t gv = g.aux_grad ? uv : gelu_grad2(uv);
                    x[2 * e] *= gv.x; x[2 * e + 1] *= gv.y;
                }
            }
            if (C_F32) {
                float* o = Cf + (long)m * g.ldc + n;
                *reinterpret_cast<float4*>(o) = make_float4(x[0], x[1], x[2], x[3]);
                *reinterpret_cast<float4*>(o + 4) = make_float4(x[4], x[5], x[6], x[7]);
            } else {
                uint4 o;
                o.x = H16<F16>::pack2(x[0], x[1]); o.y = H16<F16>::pack2(x[2], x[3]); o.z = H16<F16>::pack2(x[4], x[5]); o.w = H16<F16>::pack2(x[6], x[7]);
                *reinterpret_cast<uint4*>(Cb + (long)m * g.ldc + n) = o;
                if constexpr (SPLIT) {                           // low-order half: x - hi is exact in fp32, then rounded once
                    const uint32_t ow[4] = {o.x, o.y, o.z, o.w};
                    uint32_t lw[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const v2f_t h = H16<F16>::unpack2(ow[e]);
                        lw[e] = H16<F16>::pack2(x[2 * e] - h.x, x[2 * e + 1] - h.y);
                    }
                    *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(g.C_lo) + (long)m * g.ldc + n) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
                }
            }
        }
    }
}

template <bool F16, bool C_F32, int MT, int PBM, int PBN, int NTHREADS>
__device__ __forceinline__ void epilogue_staged(const GemmB16Args& g, float4_t (&acc)[MT][4], char* lds, int m0, int n0,
                                                int wm, int wn, int lane, int split)
{
    constexpr int RBP = PBN * 4 + 16;                          // padded fp32 row: consecutive rows shift by 4 banks
    const int li = lane & 15, lg = lane >> 4;
    __syncthreads();                                            // every wave is done with the operand stages
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<float4_t*>(lds + (wm + i * 16 + li) * RBP + (wn + j * 16 + lg * 4) * 4) = acc[i][j];
    __syncthreads();
    switch (g.epilogue) {
        case EDITOR_EPI_RESIDUAL: epilogue_copy_out<F16, C_F32, EDITOR_EPI_RESIDUAL, PBM, PBN, NTHREADS>(g, lds, m0, n0, split, g.M); break;
        case EDITOR_EPI_GELU:     epilogue_copy_out<F16, C_F32, EDITOR_EPI_GELU, PBM, PBN, NTHREADS>(g, lds, m0, n0, split, g.M); break;
        case EDITOR_EPI_GELU_BWD: epilogue_copy_out<F16, C_F32, EDITOR_EPI_GELU_BWD, PBM, PBN, NTHREADS>(g, lds, m0, n0, split, g.M); break;
        default:                  epilogue_copy_out<F16, C_F32, EDITOR_EPI_NONE, PBM, PBN, NTHREADS>(g, lds, m0, n0, split, g.M); break;
    }
}

// =====================================================================================================
// Pipelined large-tile variant (round 1's performance path; since round 2 the ping-pong kernel below takes every large
// product - forward, dgrad and, with a one-round split, wgrad - and this one the shapes it does not).  A 128x128 tile moves 64 FLOP per byte staged into LDS and is
// bound by L2->CU bandwidth at ~25 % of the MFMA peak (measured: 49 % of wave cycles in s_waitcnt/barrier, 0 LDS bank
// conflicts); this kernel uses 256 x BN tiles, BN = 256 (128 FLOP/B) or 128 (85 FLOP/B, for N = 768 problems whose
// 256-wide tiling would leave the last wave of workgroups a quarter full):
//   * 8 wavefronts, one workgroup per CU; BN=256: waves 2(M) x 4(N), 128x64 each; BN=128: 4 x 2, 64x64 each
//   * LDS stages filled by global_load_lds (LDS-DMA) D = STAGES-1 K-tiles ahead; ONE raw s_barrier per K-tile and a
//     COUNTED s_waitcnt vmcnt, so the DMA of later tiles stays in flight across the barrier
//   * grouped tile order inside each XCD's tile range (GM tile-rows x all tile-columns, column-major) so the resident
//     workgroups of an XCD share few A and B panels (both fit its 4 MiB L2)
// =====================================================================================================

template <bool KMAJOR, int ROWS_OR_COLS>
__device__ __forceinline__ void pstage_glds(const bf16_t* __restrict__ P, long ld, int r0, int k0, int R, char* lds,
                                            int piece0, int npieces, int lane)
{
#pragma unroll
    for (int j = 0; j < npieces; ++j) {
        const int piece = piece0 + j;
        long off;
        if (KMAJOR) {
            const int row = piece * 8 + (lane >> 3);
            const int c = (lane & 7) ^ (row & 7);
            off = (long)min(r0 + row, R - 1) * ld + k0 + c * 8;
        } else {
            constexpr int LPR = ROWS_OR_COLS / 8;              // lanes (16-byte units) per k-row
            constexpr int RPP = 64 / LPR;                      // k-rows per 1 KiB piece
            const int kr = piece * RPP + lane / LPR;
            const int p = lane % LPR;
            const int col = ((((p >> 1) ^ swz_rowk(kr)) << 4) | ((p & 1) << 3));
            off = (long)(k0 + kr) * ld + min(r0 + col, R - 8);
        }
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(P + off),
                                         (__attribute__((address_space(3))) void*)(lds + piece * 1024), 16, 0, 0);
    }
}

template <bool KMAJOR, int COLS>
__device__ __forceinline__ short8_t pload_frag(const char* lds, int base16, int s, int lane)
{
    const int i = lane & 15, g = lane >> 4;
    if (KMAJOR) {
        const int row = base16 + i, c = s * 4 + g;
        return *reinterpret_cast<const short8_t*>(lds + row * 128 + ((c ^ (row & 7)) << 4));
    } else {
        const int blk = base16 >> 4;
        const int k0 = s * 32 + g * 8 + (i >> 2);
        const int k1 = k0 + 4;
        const int a0 = k0 * (COLS * 2) + ((blk ^ swz_rowk(k0)) << 5) + ((i & 3) << 3);
        const int a1 = k1 * (COLS * 2) + ((blk ^ swz_rowk(k1)) << 5) + ((i & 3) << 3);
        typedef __attribute__((address_space(3))) short4_t* lds_p;
        const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(lds + a0));
        const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(lds + a1));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
}

template <bool F16, bool A_KMAJOR, bool B_KMAJOR, bool C_F32, int PBM, int PBN, int STAGES, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64, 2) void gemm_bf16_pipe_kernel(GemmB16Args g)
{
    constexpr int PA_BYTES = PBM * BK * 2, PB_BYTES = PBN * BK * 2, PSTAGE = PA_BYTES + PB_BYTES;
    constexpr int WAVES_N = PBN / 64, WAVES_M = NWAVES / WAVES_N, WM = PBM / WAVES_M, MT = WM / 16;
    constexpr int A_PIECES = PA_BYTES / 1024 / NWAVES, B_PIECES = PB_BYTES / 1024 / NWAVES, PIECES = A_PIECES + B_PIECES;
    constexpr int GM = PBN == 256 ? 4 : 4;                     // tile-rows per group
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // ---- tile order: XCD-contiguous ranges, grouped (GM rows x all columns, column-major) inside --------------
    const int nwg = g.tiles_m * g.tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int grp = GM * g.tiles_n;
    const int gm0 = (wgid / grp) * GM, rem = wgid % grp;
    const int gsz = min(GM, g.tiles_m - gm0);
    const int tile_m = gm0 + rem % gsz, tile_n = rem / gsz;
    const int m0 = tile_m * PBM, n0 = tile_n * PBN;
    // compacted HMA: the host launches for the worst-case row count, the live extent is a device scalar
    int ktiles = g.K / BK;
    if (g.m_live) {
        const int live = *g.m_live;
        if (g.live_is_k) ktiles = min(ktiles, (live + BK - 1) / BK);      // rows in [live, roundup) are zero by contract
        else if (m0 >= live) return;                                        // tile of dead rows: nobody reads them
    }
    const int per = (ktiles + g.splitk - 1) / g.splitk;
    const int kt0 = blockIdx.y * per, kt1 = min(ktiles, kt0 + per);
    const int nk = max(kt1 - kt0, 0);                                       // (an empty split still writes its zero slab)

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wu = __builtin_amdgcn_readfirstlane(w);
    const int wm = (w / WAVES_N) * WM, wn = (w % WAVES_N) * 64;
    float4_t acc[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = float4_t{0.f, 0.f, 0.f, 0.f};

    auto issue = [&](int t) {
        char* st = smem + (t % STAGES) * PSTAGE;
        const int k0 = (kt0 + t) * BK;
        pstage_glds<A_KMAJOR, PBM>(g.A, g.lda, m0, k0, g.M, st, wu * A_PIECES, A_PIECES, lane);
        pstage_glds<B_KMAJOR, PBN>(g.B, g.ldb, n0, k0, g.N, st + PA_BYTES, wu * B_PIECES, B_PIECES, lane);
    };
    // Software pipeline.  All STAGES buffers are kept full (tile t+STAGES is issued as soon as tile t's buffer is
    // released), and the MFMA operands are double-buffered in REGISTERS across the barrier: while the matrix core
    // works on k-half s, the ds_reads of the next k-half (possibly of the next tile) are already in flight, so a
    // wave never sits between "barrier released" and "first MFMA" waiting for LDS.
    //   step 2t  : read F1(t)            | MFMA F0(t)
    //   step 2t+1: wait tile t+1, barrier, DMA tile t+STAGES -> buffer of tile t, read F0(t+1) | MFMA F1(t)
    // Fragment reads are issued as inline asm so that hipcc does not track them: its own bookkeeping inserts a
    // conservative s_waitcnt lgkmcnt(0) at the loop header (it cannot count reads that are in flight across the
    // back edge), which would serialise "read F1 -> MFMA F0".  The waits below are placed by hand instead:
    //   lgkmcnt(NF): the NF reads just issued may stay in flight, everything older (the operands of the next MFMA
    //                group) has landed;   lgkmcnt(0) before the barrier.
    // Every wait is followed by sched_barrier(0) (an MFMA is register-only and would otherwise be hoisted above it).
    short8_t fa0[MT], fb0[4], fa1[MT], fb1[4];
    // loop-invariant per-lane byte offsets inside a stage, for k-half 0; k-half 1 = XOR 64 (k-major: chunk index bit 2)
    // or + 32 k-rows (row-k image)
    uint32_t offA[MT], offB[4];
    {
        const int li = lane & 15, lg = lane >> 4;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            if (A_KMAJOR) { const int row = wm + i * 16 + li; offA[i] = row * 128 + ((lg ^ (row & 7)) << 4); }
            else { const int k0 = lg * 8 + (li >> 2); offA[i] = k0 * (PBM * 2) + ((((wm >> 4) + i) ^ swz_rowk(k0)) << 5) + ((li & 3) << 3); }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (B_KMAJOR) { const int row = wn + j * 16 + li; offB[j] = PA_BYTES + row * 128 + ((lg ^ (row & 7)) << 4); }
            else { const int k0 = lg * 8 + (li >> 2); offB[j] = PA_BYTES + k0 * (PBN * 2) + ((((wn >> 4) + j) ^ swz_rowk(k0)) << 5) + ((li & 3) << 3); }
        }
    }
    const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    constexpr int NF = (A_KMAJOR ? MT : 2 * MT) + (B_KMAJOR ? 4 : 8);     // LDS reads per fragment set
    constexpr int NFW = NF > 15 ? 15 : NF;
    auto rd128 = [](uint32_t addr) {
        short8_t v;
        asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
        return v;
    };
    auto rdtr = [](uint32_t addr, auto rowskip) {
        short4_t lo, hi;
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(addr));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "i"(decltype(rowskip)::value));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto load_frags = [&](short8_t (&fa)[MT], short8_t (&fb)[4], int t, int s) {
        const uint32_t st = smem_base + (t % STAGES) * PSTAGE;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (B_KMAJOR) fb[j] = rd128(st + (offB[j] ^ (s << 6)));
            else fb[j] = rdtr(st + offB[j] + s * 32 * (PBN * 2), std::integral_constant<int, 4 * PBN * 2>{});
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            if (A_KMAJOR) fa[i] = rd128(st + (offA[i] ^ (s << 6)));
            else fa[i] = rdtr(st + offA[i] + s * 32 * (PBM * 2), std::integral_constant<int, 4 * PBM * 2>{});
        }
    };
    auto mma = [&](short8_t (&fa)[MT], short8_t (&fb)[4]) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = mfma16<F16>(fb[j], fa[i], acc[i][j]);
    };
    // one LDS-DMA piece of tile t (j < A_PIECES: A, else B), used to spread the DMA issue slots between MFMAs
    auto issue_piece = [&](int t, int j) {
        char* st = smem + (t % STAGES) * PSTAGE;
        const int k0 = (kt0 + t) * BK;
        if (j < A_PIECES) pstage_glds<A_KMAJOR, PBM>(g.A, g.lda, m0, k0, g.M, st, wu * A_PIECES + j, 1, lane);
        else pstage_glds<B_KMAJOR, PBN>(g.B, g.ldb, n0, k0, g.N, st + PA_BYTES, wu * B_PIECES + (j - A_PIECES), 1, lane);
    };
    // MFMA group with the PIECES DMA instructions of tile `t_issue` interleaved (one after every other MFMA): an
    // LDS-DMA issue costs the wave ~60-180 cycles of issue time (MI355X_MICROARCH.md), which is free in the shadow
    // of a 16-cycle MFMA but serialises when the six of them are issued back to back right after the barrier.
    auto mma_issue = [&](short8_t (&fa)[MT], short8_t (&fb)[4], int t_issue) {
        int n = 0;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[i][j] = mfma16<F16>(fb[j], fa[i], acc[i][j]);
                if ((n & 1) == 1 && (n >> 1) < PIECES) {
                    __builtin_amdgcn_sched_barrier(0);
                    issue_piece(t_issue, n >> 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                ++n;
            }
    };
#define WAIT_LGKM(n) do { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#pragma unroll
    for (int d = 0; d < STAGES; ++d)
        if (d < nk) issue(d);
    if (nk > 0) {   // tile 0 landed (later tiles may still be in flight)
        const int later = min(nk - 1, STAGES - 1);
        if (later >= 2)      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PIECES) : "memory");
        else if (later == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
        else                 asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    if (nk > 0) load_frags(fa0, fb0, 0, 0);
    int t = 0;
    // steady state (tile t+STAGES exists)
    for (; t + STAGES < nk; ++t) {
        load_frags(fa1, fb1, t, 1);
        WAIT_LGKM(NFW);                                               // F0(t) landed, F1(t) in flight
        mma(fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);                            // the waits below stay BEHIND these MFMAs
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * PIECES) : "memory");   // tile t+1 landed
        WAIT_LGKM(0);                                                 // my reads of tile t's buffer are done
        __builtin_amdgcn_s_barrier();                                 // ... everyone's: the buffer can be refilled
        load_frags(fa0, fb0, t + 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        mma_issue(fa1, fb1, t + STAGES);                              // + DMA of tile t+STAGES into tile t's buffer
        __builtin_amdgcn_sched_barrier(0);
    }
    // drain: no more tiles to request
    for (; t < nk; ++t) {
        load_frags(fa1, fb1, t, 1);
        WAIT_LGKM(NFW);
        mma(fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        WAIT_LGKM(0);
        if (t + 1 < nk) {
            __builtin_amdgcn_s_barrier();
            load_frags(fa0, fb0, t + 1, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        mma(fa1, fb1);
    }
#undef WAIT_LGKM
    // Measured per-launch averages in the training step (rocprofv3): forward products (both operands k-major: qkv, fc1+GELU,
    // proj/fc2+residual) are faster with the direct register epilogue (372 / 229 us vs 443 / 249 us staged), dgrad and
    // wgrad (split-K slabs) with the LDS-staged one (252 vs 262 us; 170 vs 247 us).
    constexpr bool kStaged = PBM * (PBN * 4 + 16) <= STAGES * PSTAGE && !(A_KMAJOR && B_KMAJOR);
    if (kStaged && g.beta == 0.f && (g.splitk == 1 || g.slabs) && (g.N & 7) == 0 && (g.ldc & 7) == 0 && (g.ldaux & 7) == 0)
        epilogue_staged<F16, C_F32, MT, PBM, PBN, NWAVES * 64>(g, acc, smem, m0, n0, wm, wn, lane, blockIdx.y);
    else
        epilogue_store<F16, C_F32, MT>(g, acc, m0 + wm, n0 + wn, lane, blockIdx.y == 0);
}

// =====================================================================================================
// 256 x 256 "ping-pong" variant.  Eight wavefronts as two groups of four (wr = 0 / 1, one wave of each group per SIMD)
// that run ONE BARRIER APART: every s_barrier flips the roles, so that while one group feeds the matrix core with a
// 16-MFMA cluster (one 64x32 quadrant of its 128x64 output over the whole K-tile) the other group issues its LDS
// fragment reads and its share of the LDS-DMA for a later K-tile.  Per K-tile a wave runs four such phases
//   P1: read A0,B0 (12 x ds_read_b128)  DMA B1(t+1)  | MFMA A0 x B0
//   P2: read A1    (8)                  DMA A0(t+2)  | MFMA A1 x B0
//   P3: read B1    (4, over B0)         DMA B0(t+2)  | MFMA A1 x B1
//   P4:                                 DMA A1(t+2), s_waitcnt vmcnt(6)  | MFMA A0 x B1
// LDS = 2 K-tile buffers x 4 units of 16 KiB.  A "unit" is the part of an operand tile that ONE phase reads: A0 = the
// first 64 rows of each 128-row wave-row (rows 0-63 and 128-191 of the tile), A1 the others; B0 = the first 32 columns
// of each 64-column wave-column, B1 the others - so a unit is free again right after its phase and is refilled two
// K-tiles ahead, 5-7 phases before it is read.  One counted vmcnt per K-tile (three units stay in flight across it).
// Fragments are read one phase after the wait that retires their unit; a unit is restaged no earlier than the phase
// after its last read, whose lgkmcnt(0) precedes the barrier (so the other group's reads are retired too).
// =====================================================================================================

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

template <int IMM>
__device__ __forceinline__ short8_t lds_rd128(uint32_t addr)
{
    short8_t v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(IMM));
    return v;
}
template <int IMM>
__device__ __forceinline__ short8_t lds_rdtr(uint32_t addr)
{
    short4_t lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(addr), "n"(IMM));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "n"(IMM + 1024));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// Per-lane BYTE offsets (32-bit) of the two 1 KiB pieces this wave stages of a unit, relative to the operand's
// k0 = 0 position.  SPAN = 64 (A: rows per wave-row half) or 32 (B); image index rho in [0,128) -> tile-local index
// (rho / SPAN) * 2*SPAN + sub*SPAN + rho % SPAN.  The K-tile position is a wave-uniform offset added at issue time.
template <bool KMAJOR, int SPAN, bool SWZ32 = false>
__device__ __forceinline__ void pp_unit_offsets(long ld, int r0, int R, int sub, int wu, int lane, uint32_t (&vo)[2], int g1base = 128)
{
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int piece = wu * 2 + j;
        if (KMAJOR) {
            const int rho = piece * 8 + (lane >> 3);
            int loc = (rho / SPAN) * (2 * SPAN) + sub * SPAN + (rho % SPAN);
            if (loc >= 128) loc += g1base - 128;                // (A only: short tiles, see the kernel's F0 / F1)
            const int c = (lane & 7) ^ (SWZ32 ? (rho >> 1) & 7 : rho & 7);   // (SWZ32: 32-row fragments, see pp_body's MI32)
            vo[j] = (uint32_t)(((long)min(r0 + loc, R - 1) * ld + c * 8) * 2);
        } else {
            const int kr = piece * 4 + (lane >> 4);
            const int p = lane & 15;
            const int col = ((((p >> 1) ^ swz_rowk(kr)) << 4) | ((p & 1) << 3));
            const int loc = (col / SPAN) * (2 * SPAN) + sub * SPAN + (col % SPAN);
            vo[j] = (uint32_t)(((long)kr * ld + min(r0 + loc, R - 8)) * 2);
        }
    }
}

// SHORT TILES (F0 + F1 < 16).  The hot path's M = 49 536 token rows make 194 x 3 = 582 tiles of a 768-wide output: 2.27
// rounds of 256 workgroups, i.e. three rounds with the last one 27 % full.  F0 / F1 = number of live 16-row fragments of
// wave group 0 / 1 (8 / 8 = the full 256-row tile): with 7 / 6 the tile is 208 rows, 239 x 3 = 717 tiles = 2.8 rounds of
// 0.81-size tiles - 2.44 full-tile rounds of work and of epilogue traffic instead of 3.  Group 1's rows start at F0*16
// (its LDS image rows still at 128); the dead fragments of the A1 units are neither read nor multiplied, the two groups
// keep alternating on the matrix core with 16+16+12+12 / 16+16+8+8 MFMAs per K-tile.  Staged epilogues only.
// Measured: a 208-row tile takes 0.95 of a full tile's time, not 0.81 - each of the eight barrier intervals of a K-tile
// costs ~300 cycles whatever the MFMA count of its phase (16 MFMAs = 256) - so this buys +4-7 % on the 768-wide products
// (7 / 6 is the only short shape instantiated; 8 / 7 = 240 rows for N = 3072 measured 4-8 % SLOWER than full tiles).
// SPLIT (editor_gemm_f16x2, forward products of the 'f16x2' mode): operands are pairs x = hi + lo of half matrices; the
// K loop runs three segments over the SAME tile pipeline - lo.hi, hi.lo, then hi.hi (small terms first) - by switching the
// LDS-DMA source per K-tile; the accumulator, the phases and the barriers are unchanged.  Dropped: lo.lo (2^-22 relative).
// MI32 (round 6, EDITOR_PP_MI32): the SAME tile pipeline - units, phases, barriers, LDS-DMA pieces, 24 ds_read_b128 per K-tile - on
// v_mfma_f32_32x32x16: a phase's 64 x 32 quadrant is 2 tiles x 4 k-steps = 8 MFMAs of 32 matrix-core cycles instead of 16 of 16.
// A fragment is 32 rows x 16 k (lane: row l & 31, 16-byte chunk 2 ks + (l >> 5)); ds_read_b128 serves lanes {0-3, 12-15, 20-27} together,
// i.e. 16 rows of one chunk column whose numbers repeat mod 8 - so the image's XOR swizzle is over (row >> 1) & 7 instead of row & 7
// (both halves of the bank space, eight distinct chunk positions each: conflict-free; the LDS-DMA source permutation follows).
// Only launched with both operands k-major, a 16-bit output and the one-pass staged epilogue (launch_pp checks): a lane holds 4 rows
// x 32 columns in groups of four consecutive columns, the same 8-byte staging writes at other coordinates.  Sums 16 k per MFMA
// instead of 32: same value up to fp32 rounding, not bit-identical to the 16x16x32 form.
template <bool F16, bool A_KMAJOR, bool B_KMAJOR, bool C_F32, int F0 = 8, int F1 = 8, bool SPLIT = false, bool MI32 = false>
__device__ __forceinline__ void pp_body(const GemmB16Args& g, const int bid, const int by)
{
    static_assert(!MI32 || (A_KMAJOR && B_KMAJOR && !C_F32 && !SPLIT && F0 == 8 && F1 == 8), "32x32x16: full tiles, k-major operands, 16-bit output");
    static_assert(!SPLIT || (F16 && A_KMAJOR && B_KMAJOR), "split precision: half operands, forward layout");
    static_assert(F0 >= 5 && F0 <= 8 && F1 >= 5 && F1 <= F0, "live fragments per wave group");
    static_assert((F0 == 8 && F1 == 8) || A_KMAJOR, "short tiles: k-major A only");
    constexpr int TH = (F0 + F1) * 16;                          // tile height
    constexpr int UNIT = 16384, KTB = 4 * UNIT;                 // per K-tile buffer: A0 | A1 | B0 | B1
    constexpr int UA0 = 0, UA1 = UNIT, UB0 = 2 * UNIT, UB1 = 3 * UNIT;
    constexpr int GM = EDITOR_GEMM_GM;                          // tile rows per group (measured: tools/gm_sweep.sh)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // live rows (device scalar): the launch is sized for M, only the tiles of the first *m_live rows do work.  The tile order is built
    // over the LIVE tile rows - the first tiles_m_live * tiles_n workgroups of the launch take them, spread over the XCDs like a launch
    // of that size, the rest exit - not over all of them with the dead ones skipped: the XCD-contiguous ranges below would leave every
    // dead tile on the last XCDs and the others with their full share (measured, round 6: 5 % fewer live tiles, same launch time).
    int tiles_m = g.tiles_m;
    int ktiles = g.K / BK;
    if (g.m_live) {
        const int live = *g.m_live;
        if (g.live_is_k) ktiles = min(ktiles, (live + BK - 1) / BK);
        else if (!g.linear_ids) tiles_m = min(tiles_m, (live + TH - 1) / TH);
    }
    const int nwg = tiles_m * g.tiles_n;
    if (!g.linear_ids && bid >= nwg) {
        // a workgroup beyond the live tiles: nothing to compute; the dead tile rows' column-sum partials are part of the fold, so
        // these workgroups zero them ((g.tiles_m - tiles_m) * N floats, a few KiB)
        if (g.colsum) {
            const long n_dead = (long)(g.tiles_m - tiles_m) * g.N;
            const long first = (long)tiles_m * g.N;
            for (long e = (long)(bid - nwg) * 512 + threadIdx.x; e < n_dead; e += (long)(g.tiles_m * g.tiles_n - nwg) * 512)
                g.colsum[first + e] = 0.f;
        }
        return;
    }
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int wgid = g.linear_ids ? bid : (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int grp = GM * g.tiles_n;
    const int gm0 = (wgid / grp) * GM, rem = wgid % grp;
    const int gsz = min(GM, tiles_m - gm0);
    const int tile_m = gm0 + rem % gsz;
    const int tile_n = rem / gsz;
    const int m0 = tile_m * TH, n0 = tile_n * 256;
    if (g.m_live && !g.live_is_k && g.linear_ids && m0 >= *g.m_live) return;     // (grouped launches: static tile ranges per problem)
    // EDITOR_EPI_STAGGER: every tile of a launch takes the same time, so the 256 CUs run in LOCKSTEP - all in their K loops (matrix
    // cores busy, HBM idle), then all in their epilogues (HBM saturated: every epilogue kind measures ~12 B per clock and CU = the
    // chip's HBM rate, matrix cores idle).  The first round's workgroups (one per CU) therefore start spread over `stagger` cycles,
    // 32 phases per XCD; the hardware hands every later tile to the CU that frees up, so the phases persist, some CUs are in their
    // epilogue while the others multiply, and the late starters simply take one tile less of the (partly filled) last round.
    if (g.stagger && by == 0 && bid < 256 && (int)gridDim.x > 256) {
        const unsigned wait = ((unsigned)g.stagger >> 5) * (unsigned)((bid >> 3) & 31);
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        while ((unsigned)(__builtin_amdgcn_s_memtime() - t0) < wait) __builtin_amdgcn_s_sleep(8);
    }
    const int nkb = ktiles;                                     // K-tiles of one operand half (SPLIT)
    if constexpr (SPLIT) ktiles *= 3;
    const int per = (ktiles + g.splitk - 1) / g.splitk;
    const int kt0 = by * per, kt1 = min(ktiles, kt0 + per);
    const int nk = max(kt1 - kt0, 0);

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wu = __builtin_amdgcn_readfirstlane(w);
    const int wr = wu >> 2, wc = wu & 3;
    const int gb = wr ? F0 * 16 : 0;                            // tile row of this wave group's first row
    const int fg = wr ? F1 : F0;                                // ... and its live fragments
    const int li = lane & 15, lg = lane >> 4;
#define PP_STAMP(k) do { if (g.trace && threadIdx.x == 0) g.trace[(long)(by * gridDim.x + bid) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
    PP_STAMP(0);

    float4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = float4_t{0.f, 0.f, 0.f, 0.f};
    f32x16_t acc32[4][2];                                       // MI32: [32-row tile of the wave's 128 rows][32-column tile of its 64]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc32[i][j][e] = 0.f;
    const int l31 = lane & 31, lh = lane >> 5;

    // per-lane fragment addresses inside a unit (buffer 0); [s] for k-major images, [i] / [j] for row-k images
    const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    uint32_t adA[4], adB[4];
    if constexpr (MI32) {                                       // [ks]: k-step of 16 inside the K-tile
        const int ra = wr * 64 + l31, rb = wc * 32 + l31;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            adA[ks] = smem_base + ra * 128 + (((ks * 2 + lh) ^ ((ra >> 1) & 7)) << 4);
            adB[ks] = smem_base + rb * 128 + (((ks * 2 + lh) ^ ((rb >> 1) & 7)) << 4);
        }
    } else {
    adB[2] = adB[3] = 0;
    if (A_KMAJOR) {
        const int row = wr * 64 + li;
#pragma unroll
        for (int s = 0; s < 2; ++s) adA[s] = smem_base + row * 128 + (((s * 4 + lg) ^ (row & 7)) << 4);
        adA[2] = adA[3] = 0;
    } else {
        const int k0 = lg * 8 + (li >> 2);
#pragma unroll
        for (int i = 0; i < 4; ++i) adA[i] = smem_base + k0 * 256 + (((wr * 4 + i) ^ swz_rowk(k0)) << 5) + ((li & 3) << 3);
    }
    if (B_KMAJOR) {
        const int row = wc * 32 + li;
#pragma unroll
        for (int s = 0; s < 2; ++s) adB[s] = smem_base + row * 128 + (((s * 4 + lg) ^ (row & 7)) << 4);
    } else {
        const int k0 = lg * 8 + (li >> 2);
#pragma unroll
        for (int j = 0; j < 2; ++j) adB[j] = smem_base + k0 * 256 + (((wc * 2 + j) ^ swz_rowk(k0)) << 5) + ((li & 3) << 3);
    }
    }
    // LDS-DMA source offsets of this wave's two pieces of each unit (A0, A1, B0, B1)
    uint32_t voA[2][2], voB[2][2];
    pp_unit_offsets<A_KMAJOR, 64, MI32>(g.lda, m0, g.M, 0, wu, lane, voA[0], F0 * 16);
    pp_unit_offsets<A_KMAJOR, 64, MI32>(g.lda, m0, g.M, 1, wu, lane, voA[1], F0 * 16);
    pp_unit_offsets<B_KMAJOR, 32, MI32>(g.ldb, n0, g.N, 0, wu, lane, voB[0]);
    pp_unit_offsets<B_KMAJOR, 32, MI32>(g.ldb, n0, g.N, 1, wu, lane, voB[1]);
    const long kstepA = A_KMAJOR ? (long)BK * 2 : (long)BK * g.lda * 2;       // bytes per K-tile
    const long kstepB = B_KMAJOR ? (long)BK * 2 : (long)BK * g.ldb * 2;
    const char* baseA = reinterpret_cast<const char*>(g.A) + kt0 * kstepA;
    const char* baseB = reinterpret_cast<const char*>(g.B) + kt0 * kstepB;
    short8_t fa[4][2], fb0[2][2], fb1[2][2];

    // (DS immediate offsets are 16 bits: the second K-tile buffer, 64 KiB up, goes through the address register)
    auto read_a = [&](auto U, auto BUF, auto LIVE) {           // LIVE: fragments of the unit this wave group uses
        constexpr int base = decltype(U)::value, bo = decltype(BUF)::value * KTB;
        if constexpr (MI32) {                                  // 32-row fragment I2, k-step KS -> fa[I2 * 2 + KS / 2][KS % 2]
            static_for<0, decltype(LIVE)::value / 2>([&](auto i) {
                static_for<0, 4>([&](auto s) {
                    constexpr int I2 = decltype(i)::value, KS = decltype(s)::value;
                    fa[I2 * 2 + (KS >> 1)][KS & 1] = lds_rd128<base + I2 * 4096>(adA[KS] + bo);
                });
            });
        } else
        static_for<0, decltype(LIVE)::value>([&](auto i) {
            static_for<0, 2>([&](auto s) {
                constexpr int I = decltype(i)::value, S = decltype(s)::value;
                if constexpr (A_KMAJOR) fa[I][S] = lds_rd128<base + I * 2048>(adA[S] + bo);
                else fa[I][S] = lds_rdtr<base + S * 8192>(adA[I] + bo);
            });
        });
    };
    auto read_b = [&](short8_t (&fbv)[2][2], auto U, auto BUF) {
        constexpr int base = decltype(U)::value, bo = decltype(BUF)::value * KTB;
        if constexpr (MI32) {                                  // the unit's 32 columns are ONE fragment per k-step: fbv[KS / 2][KS % 2]
            static_for<0, 4>([&](auto s) {
                constexpr int KS = decltype(s)::value;
                fbv[KS >> 1][KS & 1] = lds_rd128<base>(adB[KS] + bo);
            });
        } else
        static_for<0, 2>([&](auto j) {
            static_for<0, 2>([&](auto s) {
                constexpr int J = decltype(j)::value, S = decltype(s)::value;
                if constexpr (B_KMAJOR) fbv[J][S] = lds_rd128<base + J * 2048>(adB[S] + bo);
                else fbv[J][S] = lds_rdtr<base + S * 8192>(adB[J] + bo);
            });
        });
    };
    auto mma_q = [&](short8_t (&fbv)[2][2], auto MI, auto NJ, auto LIVE) {
        constexpr int mi = decltype(MI)::value, nj = decltype(NJ)::value;
        if constexpr (MI32) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int i2 = 0; i2 < 2; ++i2)
                    acc32[mi * 2 + i2][nj] = mfma32<F16>(fbv[ks >> 1][ks & 1], fa[i2 * 2 + (ks >> 1)][ks & 1], acc32[mi * 2 + i2][nj]);
        } else
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < decltype(LIVE)::value; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[mi * 4 + i][nj * 2 + j] =
                        mfma16<F16>(fbv[j][s], fa[i][s], acc[mi * 4 + i][nj * 2 + j]);
    };
    // MI32, two-phase K-tile: both 32-column halves of row group MI in one cluster, the four accumulators taken round-robin - a
    // v_mfma_f32_32x32x16 that depends on the one issued two slots earlier waits for it (measured, round 6: the four-phase form's
    // clusters of 8 MFMAs on two accumulators take ~165 cycles more than their 256)
    auto mma_pair32 = [&](auto MI) {
        constexpr int mi = decltype(MI)::value;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int nj = 0; nj < 2; ++nj)
#pragma unroll
                for (int i2 = 0; i2 < 2; ++i2)
                    acc32[mi * 2 + i2][nj] = mfma32<F16>(nj ? fb1[ks >> 1][ks & 1] : fb0[ks >> 1][ks & 1], fa[i2 * 2 + (ks >> 1)][ks & 1],
                                                         acc32[mi * 2 + i2][nj]);
    };
    auto dma2 = [&](const char* src, const uint32_t (&vo)[2], char* unit) {
        // The K-tile's source position stays an OPAQUE scalar pair: left visible, loop strength reduction folds it into the per-lane
        // offsets - eight 64-bit VGPR pointer pairs carried and incremented through the K loop (16 registers, eight v_lshl_add_u64 per
        // K-tile) and the `off` addressing form - instead of global_load_lds_dwordx4 v_offset32, s[base:base+1].
        asm volatile("" : "+s"(src));
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + vo[j]),
                                             (__attribute__((address_space(3))) void*)(unit + (wu * 2 + j) * 1024), 16, 0, 0);
    };
    // short tiles: the two pieces this wave stages of an A1 unit are rows (wu & 3) * 16 .. + 15 of its group's second 64 rows;
    // beyond the group's live fragments nobody reads them (waves 3, 6, 7 of the 7 + 6 tile) - and a wave that issues none
    // still counts right: vmcnt(6) always spans the A0 / B0 / B1 pieces issued after the A1 ones
    const bool a1_live = (wu & 3) < (wr ? F1 : F0) - 4;
    const char* baseAlo = reinterpret_cast<const char*>(g.A_lo);
    const char* baseBlo = reinterpret_cast<const char*>(g.B_lo);
    auto stage_a = [&](int t, int sub) {   // unit A<sub> of K-tile t
        if (sub && !a1_live) return;
        const char* src = baseA + t * kstepA;
        if constexpr (SPLIT)               // segments: [0,nkb) A_lo | [nkb,2nkb) A_hi | [2nkb,3nkb) A_hi
            src = t < nkb ? baseAlo + t * kstepA : baseA + (t - (t < 2 * nkb ? nkb : 2 * nkb)) * kstepA;
        dma2(src, voA[sub], smem + (t & 1) * KTB + (sub ? UA1 : UA0));
    };
    auto stage_b = [&](int t, int sub) {
        const char* src = baseB + t * kstepB;
        if constexpr (SPLIT)               //           [0,nkb) B_hi | [nkb,2nkb) B_lo | [2nkb,3nkb) B_hi
            src = t < nkb ? baseB + t * kstepB : (t < 2 * nkb ? baseBlo + (t - nkb) * kstepB : baseB + (t - 2 * nkb) * kstepB);
        dma2(src, voB[sub], smem + (t & 1) * KTB + (sub ? UB1 : UB0));
    };
    using c0 = std::integral_constant<int, 0>;
    using c1 = std::integral_constant<int, 1>;
#define PP_WAIT_LGKM0() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define PP_BAR() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
#ifdef EDITOR_DEBUG_TRACE
#define PP_ABL(bit) (g.ablate & (bit))
#else
#define PP_ABL(bit) false
#endif
#define PP_MMA(fbv, MI, NJ, LIVE) do { if (!PP_ABL(2)) { __builtin_amdgcn_s_setprio(1); mma_q(fbv, MI{}, NJ{}, LIVE{}); __builtin_amdgcn_s_setprio(0); } } while (0)

    // one K-tile (four phases) out of buffer BUF.  Units: A0,B0 are read in P1, B1 in P2, A1 in P3; each is refilled
    // for K-tile t+2 in the phase after (A1: two phases after, at P1 of t+1).
    using c4 = std::integral_constant<int, 4>;
    // FG: live fragments of this wave group (4 in A0 + FG-4 in A1).  (always_inline: an out-of-line copy would take the
    // accumulators by reference, i.e. through scratch memory)
    auto ktile4 = [&](int t, auto BUF, auto FG) __attribute__((always_inline)) {
        using B = decltype(BUF);
        using L1 = std::integral_constant<int, decltype(FG)::value - 4>;
        // P1
        if (!PP_ABL(4)) read_b(fb0, std::integral_constant<int, UB0>{}, B{});
        __builtin_amdgcn_sched_barrier(0);
        if (!PP_ABL(4)) read_a(std::integral_constant<int, UA0>{}, B{}, c4{});
        if (t + 1 < nk && !PP_ABL(1)) stage_a(t + 1, 1);
        PP_WAIT_LGKM0(); PP_BAR();
        PP_MMA(fb0, c0, c0, c4);
        PP_BAR();
        // P2
        if (!PP_ABL(4)) read_b(fb1, std::integral_constant<int, UB1>{}, B{});
        if (t + 2 < nk && !PP_ABL(1)) stage_a(t + 2, 0);
        PP_WAIT_LGKM0(); PP_BAR();
        PP_MMA(fb1, c0, c1, c4);
        PP_BAR();
        // P3
        if (!PP_ABL(4)) read_a(std::integral_constant<int, UA1>{}, B{}, L1{});
        if (t + 2 < nk && !PP_ABL(1)) stage_b(t + 2, 0);
        PP_WAIT_LGKM0(); PP_BAR();
        PP_MMA(fb1, c1, c1, L1);
        PP_BAR();
        // P4
        if (t + 2 < nk) {
            if (!PP_ABL(1)) stage_b(t + 2, 1);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");            // through A1(t+1); three units stay in flight
        } else if (t + 1 < nk) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        PP_BAR();
        PP_MMA(fb0, c1, c0, L1);
        PP_BAR();
    };
    // TWO phases per K-tile (round 5): 32-MFMA clusters, four barrier intervals per K-tile instead of eight.  The four-phase form above
    // spends ~335 cycles per interval against a 16-MFMA cluster's 256: the other group's fragment reads + lgkmcnt(0) + its two LDS-DMA
    // issues take longer than the cluster they are meant to hide behind (profiles/r04_gemm_bound_probe.txt: MFMAs + barriers 1.02 us per
    // K-tile, complete 1.35).  Here a wave runs
    //   R1: read B0, B1, A0 (16 x ds_read_b128)   DMA A1(t+1)                      | the other group: MFMA A1 x B0, A1 x B1 of its tile
    //   M1: MFMA A0 x B0, A0 x B1 (32)
    //   R2: read A1 (8, over A0's registers)       DMA A0(t+2), B0(t+2), B1(t+2)   | the other group: MFMA A0 x B0, A0 x B1
    //   M2: MFMA A1 x B0, A1 x B1 (32)
    // with the same units, buffers and register budget (fa is overwritten once M1 is done).  Unit lifetimes (slots = barrier intervals,
    // group 0 starts tile t at slot 0, group 1 at slot 1): A0 / B0 / B1(t) are read in slots 0 and 1, refilled for t+2 from slot 2 on
    // (both groups' R2(t)), read again in slot 8; A1(t) is read in slots 2 and 3, refilled from slot 4 on (R1(t+1)), read in slot 10.
    // A unit's pieces are awaited (own vmcnt, then the barrier) one full phase before group 0 reads it: A1(t) at the end of R1(t),
    // A0 / B0 / B1(t+1) at the end of R2(t) - always with the newest eight (six for a wave that stages no A1 rows) still in flight.
    // Every accumulator still receives one MFMA chain per K-tile in the same k order: bit-identical to the four-phase form.
    // Measured (profiles/r05_gemm_alt_ab.txt, M = 49 536): 2 - 10 % SLOWER than four phases on the forward / dgrad products (k-major
    // operands: its R2 carries six LDS-DMA issues, ~150 cycles each beside fragment reads, and outlasts the 512-cycle cluster), 3 - 4 %
    // FASTER on the weight gradients (both operands through ds_read_b64_tr_b16: twice the fragment-read instructions per phase, which
    // the longer clusters cover) - so the choreography follows the layout.
    auto ktile2 = [&](int t, auto BUF, auto FG) __attribute__((always_inline)) {
        using B = decltype(BUF);
        using L1 = std::integral_constant<int, decltype(FG)::value - 4>;
        const bool counted = t + 2 < nk;
        // R1
        if (!PP_ABL(4)) { read_b(fb0, std::integral_constant<int, UB0>{}, B{}); read_b(fb1, std::integral_constant<int, UB1>{}, B{}); }
        __builtin_amdgcn_sched_barrier(0);
        if (!PP_ABL(4)) read_a(std::integral_constant<int, UA0>{}, B{}, c4{});
        if (t + 1 < nk && !PP_ABL(1)) stage_a(t + 1, 1);
        if (counted) { if (a1_live) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PP_WAIT_LGKM0(); PP_BAR();
        // M1
        if constexpr (MI32) { if (!PP_ABL(2)) { __builtin_amdgcn_s_setprio(1); mma_pair32(c0{}); __builtin_amdgcn_s_setprio(0); } }
        else { PP_MMA(fb0, c0, c0, c4); PP_MMA(fb1, c0, c1, c4); }
        PP_BAR();
        // R2
        if (!PP_ABL(4)) read_a(std::integral_constant<int, UA1>{}, B{}, L1{});
        if (counted && !PP_ABL(1)) {     // (one unit at a time: six 64-bit source addresses computed up front cost 8 more registers)
            __builtin_amdgcn_sched_barrier(0); stage_a(t + 2, 0);
            __builtin_amdgcn_sched_barrier(0); stage_b(t + 2, 0);
            __builtin_amdgcn_sched_barrier(0); stage_b(t + 2, 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (counted) { if (a1_live) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PP_WAIT_LGKM0(); PP_BAR();
        // M2
        if constexpr (MI32) { if (!PP_ABL(2)) { __builtin_amdgcn_s_setprio(1); mma_pair32(c1{}); __builtin_amdgcn_s_setprio(0); } }
        else { PP_MMA(fb0, c1, c0, L1); PP_MMA(fb1, c1, c1, L1); }
        PP_BAR();
    };
    constexpr bool kTwoPhase = MI32 ? EDITOR_PP_MI32_PHASES == 2 : (EDITOR_PP_PHASES == 2 || (EDITOR_PP_PHASES == 0 && !A_KMAJOR && !B_KMAJOR));
    auto ktile = [&](int t, auto BUF, auto FG) __attribute__((always_inline)) {
        if constexpr (kTwoPhase) ktile2(t, BUF, FG); else ktile4(t, BUF, FG);
    };

    if (nk > 0) {
        stage_a(0, 0); stage_b(0, 0); stage_b(0, 1); stage_a(0, 1);
        if (nk > 1) {
            stage_a(1, 0); stage_b(1, 0); stage_b(1, 1);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    PP_BAR();
    PP_STAMP(1);
    if (wr == 1) PP_BAR();                                              // group 1 runs one barrier behind group 0
    auto run = [&](auto FG) __attribute__((always_inline)) {
        int t = 0;
        for (; t + 1 < nk; t += 2) { ktile(t, c0{}, FG); ktile(t + 1, c1{}, FG); }
        if (t < nk) ktile(t, c0{}, FG);
    };
    if constexpr (F0 == F1) run(std::integral_constant<int, F0>{});
    else if (wr == 0) run(std::integral_constant<int, F0>{});
    else run(std::integral_constant<int, F1>{});
    if (wr == 0) PP_BAR();                                              // re-align the two groups
    PP_STAMP(2);
#undef PP_MMA
#undef PP_WAIT_LGKM0

    // A raw s_barrier does not wait for the wave's own outstanding LDS operations (gfx950 backs off barriers instead of
    // draining before them) and LDS queues are per SIMD pair: another wave's read can overtake a write that was issued
    // before the barrier.  The epilogue's barriers are therefore preceded by lgkmcnt(0).
#define PP_EBAR() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); PP_BAR(); } while (0)
    const bool staged = g.pp_staged && g.beta == 0.f && (g.splitk == 1 || g.slabs) && (g.N & 7) == 0 && (g.ldc & 7) == 0 &&
                        (g.ldaux & 7) == 0;
    if (MI32 || (!SPLIT && staged && !C_F32 && (g.epilogue == EDITOR_EPI_NONE || g.epilogue == EDITOR_EPI_GELU || g.epilogue == EDITOR_EPI_GELU_BWD))) {
        // bf16 outputs whose epilogue is per-element: scale / bias / row scale in registers, ONE pass of the whole
        // 256x256 tile through LDS as bf16 (rows padded to 528 B), then 16-byte row-contiguous stores.  GELU: the
        // staged value is the (rounded) pre-activation, which is an output anyway; the activation is computed from it
        // on the way out, as the unfused form would.
        constexpr int RB = 256 * 2 + 16;
        // a lane's accumulators as NI row slots x NJ groups of four consecutive columns: 8 x 4 (16x16 tiles: row li of fragment i, columns
        // lg * 4 of fragment j) or 4 x 8 (MI32: row l31 of 32-row tile i, columns (j & 3) * 8 + lh * 4 of 32-column tile j >> 2)
        constexpr int NI = MI32 ? 4 : 8, NJ = MI32 ? 8 : 4;
        auto e_row = [&](int i) { return MI32 ? gb + i * 32 + l31 : gb + i * 16 + li; };
        auto e_col = [&](int j) { return MI32 ? wc * 64 + (j >> 2) * 32 + (j & 3) * 8 + lh * 4 : wc * 64 + j * 16 + lg * 4; };
        auto e_acc = [&](int i, int j, int e) -> float {
            if constexpr (MI32) return acc32[i][j >> 2][(j & 3) * 4 + e]; else return acc[i][j][e];
        };
        auto e_live = [&](int i) { return MI32 || i < fg; };
        PP_EBAR();
        const bool add_bias = g.bias && by == 0;
        float4 bv[NJ];                                         // this lane's column groups: loaded once, not per row
        float rsv[NI];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int nl = e_col(j);
            bv[j] = (add_bias && n0 + nl < g.N) ? *reinterpret_cast<const float4*>(g.bias + n0 + nl) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int ml = e_row(i);                             // tile row (= row of the staged image)
            rsv[i] = (g.rowscale && e_live(i) && m0 + ml < g.M) ? g.rowscale[m0 + ml] : 1.f;
        }
        if (g.epilogue == EDITOR_EPI_GELU_BWD) {
            // C = value * gelu'(saved pre-activation): the operand is fetched in the ACCUMULATOR layout (8 bytes per lane
            // and fragment, all 32 requests first - 64 registers are free once the operand fragments are dead), the
            // product is rounded once, and the tile leaves through the same one-pass bf16 staging as the plain
            // epilogue (the two-pass fp32 staging + per-item operand loads took 38 k cycles per tile against 30 k for the
            // K = 768 main loop).
            const bf16_t* Ab = reinterpret_cast<const bf16_t*>(g.aux);
            uint2 pre[NI][NJ];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                if (!e_live(i)) continue;
                const int m = min(m0 + e_row(i), g.M - 1);
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int n = min(n0 + e_col(j), g.N - 4);
                    pre[i][j] = *reinterpret_cast<const uint2*>(Ab + (long)m * g.ldaux + n);
                }
            }
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                if (!e_live(i)) continue;
                const int ml = e_row(i);
                const float rs = rsv[i];
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int nl = e_col(j);
                    const v2f_t u0 = H16<F16>::unpack2(pre[i][j].x), u1 = H16<F16>::unpack2(pre[i][j].y);
                    const v2f_t g0 = g.aux_grad ? u0 : gelu_grad2(u0), g1 = g.aux_grad ? u1 : gelu_grad2(u1);
                    uint2 o;
                    o.x = H16<F16>::pack2((e_acc(i, j, 0) * g.alpha + bv[j].x) * rs * g0.x, (e_acc(i, j, 1) * g.alpha + bv[j].y) * rs * g0.y);
                    o.y = H16<F16>::pack2((e_acc(i, j, 2) * g.alpha + bv[j].z) * rs * g1.x, (e_acc(i, j, 3) * g.alpha + bv[j].w) * rs * g1.y);
                    *reinterpret_cast<uint2*>(smem + ml * RB + nl * 2) = o;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                if (!e_live(i)) continue;
                const int ml = e_row(i);
                const float rs = rsv[i];
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int nl = e_col(j);
                    uint2 o;
                    o.x = H16<F16>::pack2((e_acc(i, j, 0) * g.alpha + bv[j].x) * rs, (e_acc(i, j, 1) * g.alpha + bv[j].y) * rs);
                    o.y = H16<F16>::pack2((e_acc(i, j, 2) * g.alpha + bv[j].z) * rs, (e_acc(i, j, 3) * g.alpha + bv[j].w) * rs);
                    *reinterpret_cast<uint2*>(smem + ml * RB + nl * 2) = o;
                }
            }
        }
        const bool gelu = g.epilogue == EDITOR_EPI_GELU;
        uint32_t* lut = reinterpret_cast<uint32_t*>(smem + 256 * RB);          // behind the staged image
        const bool use_lut = !F16 && gelu;
        if (use_lut) {
#pragma unroll 1
            for (uint32_t j = threadIdx.x * 2; j < 2 * kLutSpan; j += 1024) {
                const v2f_t x = H16<false>::unpack2(gelu_lut_bits(j) | (gelu_lut_bits(j + 1) << 16));
                const v2f_t gv = gelu2(x), dv = gelu_grad2(x);
                lut[j] = H16<false>::pack2(gv.x, dv.x);
                lut[j + 1] = H16<false>::pack2(gv.y, dv.y);
            }
        }
        PP_EBAR();
        PP_STAMP(3);
        bf16_t* Cb = reinterpret_cast<bf16_t*>(g.C);
        bf16_t* Ab = reinterpret_cast<bf16_t*>(g.aux);
        float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // column sums of this thread's 8 columns (same for all its rows)
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
            const int c = threadIdx.x + it * 512;
            const int row = c >> 5, cg = c & 31;
            const int m = m0 + row, n = n0 + cg * 8;
            uint4 p = *reinterpret_cast<const uint4*>(smem + row * RB + cg * 16);
            if (row >= TH || m >= g.M || n >= g.N) continue;
            bool looked_up = false;
            if (use_lut) {
                const uint32_t pw[4] = {p.x, p.y, p.z, p.w};
                uint32_t ent[8], bad = 0u;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t lo = pw[e] & 0xFFFFu, hi = pw[e] >> 16;
                    const uint32_t i0 = (lo & 0x7FFFu) - kLutBase, i1 = (hi & 0x7FFFu) - kLutBase;
                    bad |= (uint32_t)(i0 >= kLutSpan) | (uint32_t)(i1 >= kLutSpan);
                    ent[2 * e] = lut[(i0 & (kLutSpan - 1)) + ((lo >> 15) << 11)];
                    ent[2 * e + 1] = lut[(i1 & (kLutSpan - 1)) + ((hi >> 15) << 11)];
                }
                if (__builtin_amdgcn_ballot_w64(bad != 0u) == 0ull) {       // (wave-uniform; ~1 chunk in 10 goes the long way)
                    uint32_t gw[4], dw_[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        gw[e] = __builtin_amdgcn_perm(ent[2 * e + 1], ent[2 * e], 0x05040100u);     // {gelu(lo), gelu(hi)}
                        dw_[e] = __builtin_amdgcn_perm(ent[2 * e + 1], ent[2 * e], 0x07060302u);    // {gelu'(lo), gelu'(hi)}
                    }
                    if (Ab) *reinterpret_cast<uint4*>(Ab + (long)m * g.ldaux + n) = g.aux_grad ? make_uint4(dw_[0], dw_[1], dw_[2], dw_[3]) : p;
                    p = make_uint4(gw[0], gw[1], gw[2], gw[3]);
                    looked_up = true;
                }
            }
            if (gelu && !looked_up) {
                // the backward only needs gelu'(pre-activation): save THAT (one multiply there instead of an erfc + exponential per
                // element in the dgrad epilogue); a no-grad forward (Ab NULL) saves nothing.  Phi is evaluated once for both outputs.
                uint32_t pw[4] = {p.x, p.y, p.z, p.w}, dw_[4];
                const bool want_grad = Ab && g.aux_grad;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v2f_t gv, dv = v2f_t{0.f, 0.f};
                    gelu_both2(H16<F16>::unpack2(pw[e]), want_grad, gv, dv);
                    dw_[e] = H16<F16>::pack2(dv.x, dv.y);
                    pw[e] = H16<F16>::pack2(gv.x, gv.y);
                }
                if (Ab) *reinterpret_cast<uint4*>(Ab + (long)m * g.ldaux + n) = want_grad ? make_uint4(dw_[0], dw_[1], dw_[2], dw_[3]) : p;
                p = make_uint4(pw[0], pw[1], pw[2], pw[3]);
            }
            if (g.colsum) {
                const uint32_t pw[4] = {p.x, p.y, p.z, p.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) { const v2f_t u = H16<F16>::unpack2(pw[e]); cs[2 * e] += u.x; cs[2 * e + 1] += u.y; }
            }
            *reinterpret_cast<uint4*>(Cb + (long)m * g.ldc + n) = p;
        }
        if (g.colsum) {       // 16 threads share each column group: fold through LDS (the tile image is no longer needed)
            PP_EBAR();
            float* red = reinterpret_cast<float*>(smem);
#pragma unroll
            for (int e = 0; e < 8; ++e) red[(threadIdx.x >> 5) * 256 + (threadIdx.x & 31) * 8 + e] = cs[e];
            PP_EBAR();
            if (threadIdx.x < 256 && n0 + (int)threadIdx.x < g.N) {
                float t = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) t += red[r * 256 + threadIdx.x];
                g.colsum[(long)tile_m * g.N + n0 + threadIdx.x] = t;
            }
        }
        PP_STAMP(4);
    } else if (staged) {
        // two passes of 128 rows through the (now free) operand buffers: fp32 image, padded rows
        constexpr int RBP = 256 * 4 + 16;
#pragma unroll 1
        for (int pass = 0; pass < 2; ++pass) {
            PP_EBAR();
            if (wr == pass) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (i >= fg) continue;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        *reinterpret_cast<float4_t*>(smem + (i * 16 + li) * RBP + (wc * 64 + j * 16 + lg * 4) * 4) = acc[i][j];
                }
            }
            PP_EBAR();
            const int mp = m0 + (pass ? F0 * 16 : 0);                      // the pass's wave group: first row, row limit
            const int ml = min(g.M, mp + (pass ? F1 : F0) * 16);
            if constexpr (SPLIT) {                                        // (the launcher admits NONE / RESIDUAL / GELU only)
                switch (g.epilogue) {
                    case EDITOR_EPI_RESIDUAL: epilogue_copy_out<F16, C_F32, EDITOR_EPI_RESIDUAL, 128, 256, 512, !C_F32>(g, smem, mp, n0, by, ml); break;
                    case EDITOR_EPI_GELU:     epilogue_copy_out<F16, C_F32, EDITOR_EPI_GELU, 128, 256, 512, !C_F32>(g, smem, mp, n0, by, ml); break;
                    default:                  epilogue_copy_out<F16, C_F32, EDITOR_EPI_NONE, 128, 256, 512, !C_F32>(g, smem, mp, n0, by, ml); break;
                }
            } else
            switch (g.epilogue) {
                case EDITOR_EPI_RESIDUAL: epilogue_copy_out<F16, C_F32, EDITOR_EPI_RESIDUAL, 128, 256, 512>(g, smem, mp, n0, by, ml); break;
                case EDITOR_EPI_GELU:     epilogue_copy_out<F16, C_F32, EDITOR_EPI_GELU, 128, 256, 512>(g, smem, mp, n0, by, ml); break;
                case EDITOR_EPI_GELU_BWD: epilogue_copy_out<F16, C_F32, EDITOR_EPI_GELU_BWD, 128, 256, 512>(g, smem, mp, n0, by, ml); break;
                default:                  epilogue_copy_out<F16, C_F32, EDITOR_EPI_NONE, 128, 256, 512>(g, smem, mp, n0, by, ml); break;
            }
            PP_STAMP(3 + pass);
        }
    } else {
        epilogue_store<F16, C_F32, 8>(g, acc, m0 + gb, n0 + wc * 64, lane, by == 0);   // (full tiles only: host)
    }
    if (g.trace) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); PP_STAMP(5); }
#undef PP_STAMP
#undef PP_EBAR
#undef PP_BAR
}

template <bool F16, bool A_KMAJOR, bool B_KMAJOR, bool C_F32, int F0 = 8, int F1 = 8, bool SPLIT = false, bool MI32 = false>
__global__ __launch_bounds__(512) void gemm_bf16_pp_kernel(GemmB16Args g)
{
    pp_body<F16, A_KMAJOR, B_KMAJOR, C_F32, F0, F1, SPLIT, MI32>(g, blockIdx.x, blockIdx.y);
}

// GROUPED weight gradients: the four dW = dy^T x products of one transformer block (qkv, proj, fc1, fc2: 27 + 9 + 36 + 36 =
// 108 output tiles of 256 x 256, reduction over all token rows) as ONE launch - grid.x = the tiles of all problems back to
// back, grid.y = the split of the reduction shared by all - so that the split comes from tile parallelism: 108 x 7 = 756
// workgroups = 2.95 rounds of 110-K-tile workgroups, where four launches had 9 / 28 / 7 / 7 splits, one partly filled round
// each, and the 768 x 768 product ran at 692 TFLOP/s behind 28 slabs.  Per-problem arguments ride in the kernel argument.
constexpr int kMaxGroup = 4;
struct GemmGroupArgs { GemmB16Args p[kMaxGroup]; int start[kMaxGroup + 1]; int n; };

template <bool F16>
__global__ __launch_bounds__(512) void gemm_bf16_pp_group_kernel(GemmGroupArgs ga)
{
    // 1-D grid of tiles x splits.  Consecutive workgroups go to consecutive XCDs; each XCD takes a CONTIGUOUS range of the
    // split-major order, i.e. (almost) one split = one range of token rows: the tiles resident on an XCD then read the same
    // rows of dy and x at the same time and share them through its L2.  (Tile-major across XCDs - round 3's first version -
    // had every (tile, split) stream its own panels: 5.5 GB fetched per block for 1.3 GB of operands.)
    const int ntile = ga.start[ga.n], total = (int)gridDim.x;
    const int L = (int)blockIdx.x, xcd = L & 7, q = total >> 3, r = total & 7;
    const int Lp = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
    const int split = Lp / ntile, t = Lp % ntile;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < kMaxGroup; ++i) pi += (i < ga.n && t >= ga.start[i]) ? 1 : 0;
    pi = __builtin_amdgcn_readfirstlane(pi);
    pp_body<F16, false, false, true, 8, 8, false>(ga.p[pi], t - ga.start[pi], split);
}

// ---------------------------------------------------------------------------------------------------------------------------
// A MEMORY-BOUND ROLE inside the weight-gradient launch (round 4, last experiment; DESIGN 9, tools/hetero_probe.py: an HBM-bound and
// an MFMA-bound kernel do not overlap across queues on this runtime, but two ROLES of one launch do - a CU streams ~55 GB/s, so a
// quarter of the chip carries 3 TB/s while the rest multiplies).  The LayerNorm-1 backward of a block (with the 16-bit gradient copy
// for the block below, norm.hip: layernorm_bwd_kernel<T, NV, CAST = true>) is independent of the block's grouped weight gradients
// and adjacent to them: here its rows are taken by the first `nmem` workgroups of the launch - wave per row as in the stand-alone
// kernel, the same arithmetic per row, so dx and the 16-bit copy are bit-identical; the partial sums (dgamma / dbeta / column sums of
// the copy) come out as one row per memory workgroup instead of one per block of the stand-alone grid.
struct LnRoleArgs {
    const uint16_t* dy; const float* x; const float* gamma; const float* mean; const float* rstd; long M; int D;
    const float* dx_in; float* dx_out; float* partials; float dy_scale;
    uint16_t* cast_out; const float* cast_rowscale; float cast_scale; float* cast_partials;
    int nmem;
};
template <bool F16, int NV>
__device__ __forceinline__ void ln_bwd_cast_role(const LnRoleArgs& a, const int wg, float* red /* [8][3][1024] floats */)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int D = a.D;
    float4 g[NV], dg[NV], db[NV], ccs[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        dg[i] = make_float4(0.f, 0.f, 0.f, 0.f); db[i] = dg[i]; ccs[i] = dg[i];
        g[i] = *reinterpret_cast<const float4*>(a.gamma + (i * 64 + lane) * 4);
    }
    // R rows per wave in flight: a workgroup of the role has 8 waves on its CU where the stand-alone kernel has 16 workgroups' worth -
    // with one row per wave (two dependent memory phases per row) the role reached ~15 GB/s per CU; every load of R rows is issued
    // before the first is used
    constexpr int R = 2;
    for (long row0 = (long)wg * 8 * R + w; row0 < a.M; row0 += (long)a.nmem * 8 * R) {
        float4 xv[R][NV], rin[R][NV];
        uint2 dyv[R][NV];
        float mean[R], rstd[R], rsc[R];
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const long row = row0 + j * 8;
            const bool live = row < a.M;
            mean[j] = live ? a.mean[row] : 0.f;
            rstd[j] = live ? a.rstd[row] : 0.f;
            rsc[j] = (live && a.cast_rowscale) ? a.cast_rowscale[row] : 1.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c0 = (i * 64 + lane) * 4;
                xv[j][i] = live ? *reinterpret_cast<const float4*>(a.x + row * D + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
                dyv[j][i] = live ? *reinterpret_cast<const uint2*>(a.dy + row * D + c0) : make_uint2(0u, 0u);
                rin[j][i] = (live && a.dx_in) ? *reinterpret_cast<const float4*>(a.dx_in + row * D + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const long row = row0 + j * 8;
            if (row >= a.M) break;                               // (wave-uniform)
            float4 xh[NV], d[NV];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const float2_t_ lo = H16<F16>::unpack2(dyv[j][i].x), hi = H16<F16>::unpack2(dyv[j][i].y);
                d[i] = make_float4(lo.x * a.dy_scale, lo.y * a.dy_scale, hi.x * a.dy_scale, hi.y * a.dy_scale);
                xh[i] = make_float4((xv[j][i].x - mean[j]) * rstd[j], (xv[j][i].y - mean[j]) * rstd[j], (xv[j][i].z - mean[j]) * rstd[j],
                                    (xv[j][i].w - mean[j]) * rstd[j]);
                dg[i].x += d[i].x * xh[i].x; dg[i].y += d[i].y * xh[i].y; dg[i].z += d[i].z * xh[i].z; dg[i].w += d[i].w * xh[i].w;
                db[i].x += d[i].x; db[i].y += d[i].y; db[i].z += d[i].z; db[i].w += d[i].w;
                d[i].x *= g[i].x; d[i].y *= g[i].y; d[i].z *= g[i].z; d[i].w *= g[i].w;
                s1 += (d[i].x + d[i].y) + (d[i].z + d[i].w);
                s2 += (d[i].x * xh[i].x + d[i].y * xh[i].y) + (d[i].z * xh[i].z + d[i].w * xh[i].w);
            }
            s1 = wave_sum(s1) / (float)D;
            s2 = wave_sum(s2) / (float)D;
            const float r = rsc[j] * a.cast_scale;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c0 = (i * 64 + lane) * 4;
                float4 o;
                o.x = rstd[j] * (d[i].x - s1 - xh[i].x * s2) + rin[j][i].x;
                o.y = rstd[j] * (d[i].y - s1 - xh[i].y * s2) + rin[j][i].y;
                o.z = rstd[j] * (d[i].z - s1 - xh[i].z * s2) + rin[j][i].z;
                o.w = rstd[j] * (d[i].w - s1 - xh[i].w * s2) + rin[j][i].w;
                *reinterpret_cast<float4*>(a.dx_out + row * D + c0) = o;
                o.x *= r; o.y *= r; o.z *= r; o.w *= r;
                uint2 pk; pk.x = H16<F16>::pack2(o.x, o.y); pk.y = H16<F16>::pack2(o.z, o.w);
                *reinterpret_cast<uint2*>(a.cast_out + row * D + c0) = pk;
                const float2_t_ rl = H16<F16>::unpack2(pk.x), rh = H16<F16>::unpack2(pk.y);
                ccs[i].x += rl.x; ccs[i].y += rl.y; ccs[i].z += rh.x; ccs[i].w += rh.y;
            }
        }
    }
    // this workgroup's partial rows: [which = dgamma, dbeta, column sums of the copy][D], its eight waves folded in a fixed order
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c0 = (i * 64 + lane) * 4;
        *reinterpret_cast<float4*>(&red[(w * 3 + 0) * 1024 + c0]) = dg[i];
        *reinterpret_cast<float4*>(&red[(w * 3 + 1) * 1024 + c0]) = db[i];
        *reinterpret_cast<float4*>(&red[(w * 3 + 2) * 1024 + c0]) = ccs[i];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 3 * D; c += 512) {
        const int which = c / D, col = c % D;
        float t = 0.f;
#pragma unroll
        for (int ww = 0; ww < 8; ++ww) t += red[(ww * 3 + which) * 1024 + col];
        if (which < 2) a.partials[((long)wg * 2 + which) * D + col] = t;
        else if (a.cast_partials) a.cast_partials[(long)wg * D + col] = t;
    }
}

template <bool F16>
__global__ __launch_bounds__(512) void gemm_bf16_pp_group_ln_kernel(GemmGroupArgs ga, LnRoleArgs ln)
{
    if ((int)blockIdx.x < ln.nmem) {                              // (workgroup-uniform: the memory role never meets the tiles' barriers)
        extern __shared__ __attribute__((aligned(16))) char smem_ln[];
        if (ln.D == 768) ln_bwd_cast_role<F16, 3>(ln, (int)blockIdx.x, reinterpret_cast<float*>(smem_ln));
        else ln_bwd_cast_role<F16, 4>(ln, (int)blockIdx.x, reinterpret_cast<float*>(smem_ln));
        return;
    }
    // (the weight-gradient group's own order, gemm_bf16_pp_group_kernel, over the remaining workgroups; nmem is a multiple of 8)
    const int ntile = ga.start[ga.n], total = (int)gridDim.x - ln.nmem;
    const int L = (int)blockIdx.x - ln.nmem, xcd = L & 7, q = total >> 3, r = total & 7;
    const int Lp = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
    const int split = Lp / ntile, t = Lp % ntile;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < kMaxGroup; ++i) pi += (i < ga.n && t >= ga.start[i]) ? 1 : 0;
    pi = __builtin_amdgcn_readfirstlane(pi);
    pp_body<F16, false, false, true, 8, 8, false>(ga.p[pi], t - ga.start[pi], split);
}

// GROUPED forward / dgrad products (round 4): up to four products of IDENTICAL shape, layout and epilogue kind with different
// operands - the three per-modality blocks of the HMA head, whose ~7 400 live token rows make 87 tiles of a 768-wide output each
// (a third of a round of 256 CUs per launch) - as ONE launch: grid = the problems' tiles back to back.  Each workgroup runs the
// plain kernel's body on its problem's arguments: same tiles, same arithmetic, bit-identical to the separate launches.
template <bool F16, bool C_F32>
__global__ __launch_bounds__(512) void gemm_bf16_pp_fgroup_kernel(GemmGroupArgs ga)
{
    const int ntile = ga.start[1];                               // tiles per problem (all equal)
    int pi = (int)blockIdx.x / ntile;
    pi = __builtin_amdgcn_readfirstlane(pi);
    pp_body<F16, true, true, C_F32, 8, 8, false>(ga.p[pi], (int)blockIdx.x - pi * ntile, 0);
}

__global__ void scale_c_kernel(float* C, long rows, int cols, long ld, float beta)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * cols) return;
    float* c = C + (e / cols) * ld + (e % cols);
    *c = beta == 0.f ? 0.f : *c * beta;
}

template <bool F16, bool AK, bool BK_, bool CF, bool GL>
int launch(const GemmB16Args& g, hipStream_t stream)
{
    auto kern = gemm_bf16_kernel<F16, AK, BK_, CF, GL>;
    if (int e = ensure_lds<gemm_bf16_kernel<F16, AK, BK_, CF, GL>>(4 * TILE_BYTES)) return e;
    hipLaunchKernelGGL(kern, dim3(g.tiles_m * g.tiles_n, g.splitk), dim3(256), 4 * TILE_BYTES, stream, g);
    EDITOR_LAUNCH_CHECK();
    return 0;
}

// out[e] = beta*out[e] + sum_s slab[s][e]   (split-K reduction, fixed order s = 0, 1, ...).  Eight slabs are requested
// before the first is added: with a rolled loop (one load, one dependent add per trip) the 9 - 28 slabs of a weight
// gradient were nine to twenty-eight serialised HBM latencies per thread (37 us for 71 MB).
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slabs, int nsplit, long n4, float* __restrict__ out,
                                                          float beta)
{
    const float4* __restrict__ sl = reinterpret_cast<const float4*>(slabs);
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (long)gridDim.x * blockDim.x) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int s0 = 0; s0 < nsplit; s0 += 8) {
            float4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                v[j] = s0 + j < nsplit ? sl[(long)(s0 + j) * n4 + e] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (s0 + j < nsplit) { a.x += v[j].x; a.y += v[j].y; a.z += v[j].z; a.w += v[j].w; }
            }
        }
        if (beta != 0.f) {
            const float4 o = reinterpret_cast<float4*>(out)[e];
            a.x += beta * o.x; a.y += beta * o.y; a.z += beta * o.z; a.w += beta * o.w;
        }
        reinterpret_cast<float4*>(out)[e] = a;
    }
}

// The same fold for the up-to-four weight gradients of a grouped launch, as ONE launch (blockIdx.y = problem): 65 -> 17 launches
// per step; every element is summed in the same fixed order as slab_reduce_kernel does it (bit-identical).
struct SlabJobs { const float* slabs[kMaxGroup]; float* out[kMaxGroup]; long n4[kMaxGroup]; int nsplit; };
__global__ __launch_bounds__(256) void slab_reduce_multi_kernel(SlabJobs j)
{
    const int p = blockIdx.y;
    const long n4 = j.n4[p];
    const float4* __restrict__ sl = reinterpret_cast<const float4*>(j.slabs[p]);
    float4* __restrict__ out = reinterpret_cast<float4*>(j.out[p]);
    const int nsplit = j.nsplit;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (long)gridDim.x * blockDim.x) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int s0 = 0; s0 < nsplit; s0 += 8) {
            float4 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q)
                v[q] = s0 + q < nsplit ? sl[(long)(s0 + q) * n4 + e] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (s0 + q < nsplit) { a.x += v[q].x; a.y += v[q].y; a.z += v[q].z; a.w += v[q].w; }
            }
        }
        out[e] = a;
    }
}

template <bool F16, bool AK, bool BK_, bool CF, int PBM, int PBN, int STAGES, int NWAVES>
int launch_pipe_t(GemmB16Args g, hipStream_t stream)
{
    constexpr int LDS = STAGES * (PBM + PBN) * BK * 2;
    auto kern = gemm_bf16_pipe_kernel<F16, AK, BK_, CF, PBM, PBN, STAGES, NWAVES>;
    if (int e = ensure_lds<gemm_bf16_pipe_kernel<F16, AK, BK_, CF, PBM, PBN, STAGES, NWAVES>>(LDS)) return e;
    g.tiles_m = (g.M + PBM - 1) / PBM;
    g.tiles_n = (g.N + PBN - 1) / PBN;
    hipLaunchKernelGGL(kern, dim3(g.tiles_m * g.tiles_n, g.splitk), dim3(NWAVES * 64), LDS, stream, g);
    EDITOR_LAUNCH_CHECK();
    return 0;
}

template <bool F16, bool AK, bool BK_, bool CF, int F0, int F1, bool SPLIT = false, bool MI32 = false>
int launch_pp_t(GemmB16Args g, hipStream_t stream)
{
    constexpr int LDS = 256 * (256 * 2 + 16) + (int)kLutBytes;  // >= 2 K-tile buffers, the fp32 half-tile image, the bf16 tile image + GELU table
    auto kern = gemm_bf16_pp_kernel<F16, AK, BK_, CF, F0, F1, SPLIT, MI32>;
    if (int e = ensure_lds<gemm_bf16_pp_kernel<F16, AK, BK_, CF, F0, F1, SPLIT, MI32>>(LDS)) return e;
    g.tiles_m = (g.M + (F0 + F1) * 16 - 1) / ((F0 + F1) * 16);
    g.tiles_n = (g.N + 255) / 256;
    // measured: the LDS-staged epilogue (full-line 16-byte stores) beats the direct one on every layout here
    // (fwd N=2304,K=768: 701 vs 530 TFLOP/s; dgrad 974 vs 905)
    g.pp_staged = 1;
    // EDITOR_EPI_STAGGER's phase table ((bid >> 3) & 31 over the first 256 workgroups, one per CU) is MI355X's 256 CUs in 8 XCDs: on a
    // part with another CU count the spin would only delay workgroups - there the launch runs without it (same bits)
    if (g.stagger && device_cus() != 256) g.stagger = 0;
#ifdef EDITOR_DEBUG_TRACE
    // Debug build only (libeditor_debug.so, tools/gemm_bench.py): experiment switches and the per-workgroup phase
    // timeline.  The product library has no environment lookups, allocations or synchronisation in its entry points.
    if (const char* e = getenv("EDITOR_GEMM_PP_STAGED")) g.pp_staged = atoi(e);
    if (const char* e = getenv("EDITOR_GEMM_ABLATE")) g.ablate = atoi(e);
    if (getenv("EDITOR_GEMM_TRACE")) {
        const int nwg = g.tiles_m * g.tiles_n * g.splitk;
        static unsigned long long* buf = nullptr;
        if (!buf && hipMalloc(&buf, sizeof(unsigned long long) * 8 * 65536) != hipSuccess) return 1;
        if (nwg <= 65536) {
            hipError_t e1 = hipMemsetAsync(buf, 0, sizeof(unsigned long long) * 8 * nwg, stream);
            if (e1 != hipSuccess) return (int)e1;
            g.trace = buf;
            hipLaunchKernelGGL(kern, dim3(g.tiles_m * g.tiles_n, g.splitk), dim3(512), LDS, stream, g);
            if ((e1 = hipStreamSynchronize(stream)) != hipSuccess) return (int)e1;
            static unsigned long long host[8 * 65536];
            if ((e1 = hipMemcpy(host, buf, sizeof(unsigned long long) * 8 * nwg, hipMemcpyDeviceToHost)) != hipSuccess) return (int)e1;
            double seg[5] = {0, 0, 0, 0, 0};
            unsigned long long t_min = ~0ull, t_max = 0;
            for (int i = 0; i < nwg; ++i) {
                const unsigned long long* r = host + (long)i * 8;
                for (int k = 0; k < 5; ++k) seg[k] += (double)(r[k == 4 ? 5 : k + 1] - r[k == 4 ? 4 : k]);
                if (r[0] < t_min) t_min = r[0];
                if (r[5] > t_max) t_max = r[5];
            }
            fprintf(stderr, "[pp trace] M=%d N=%d K=%d epi=%d cf32=%d wgs=%d span=%.0f | per-wg mean ticks: prologue %.0f main %.0f "
                    "epi0 %.0f epi1 %.0f drain %.0f\n", g.M, g.N, g.K, g.epilogue, (int)CF, nwg, (double)(t_max - t_min),
                    seg[0] / nwg, seg[1] / nwg, seg[2] / nwg, seg[3] / nwg, seg[4] / nwg);
            return 0;
        }
    }
#endif
    hipLaunchKernelGGL(kern, dim3(g.tiles_m * g.tiles_n, g.splitk), dim3(512), LDS, stream, g);
    EDITOR_LAUNCH_CHECK();
    return 0;
}

template <bool F16, bool AK, bool BK_, bool CF>
int launch_pp(const GemmB16Args& g, hipStream_t stream)
{
    if constexpr (AK && BK_) {
        // short tiles (EDITOR_EPI_TILE_ROWS, see the kernel): both operands k-major and a staged epilogue (checked by the caller)
        if (g.tile_frags == 13) return launch_pp_t<F16, AK, BK_, CF, 7, 6>(g, stream);
#if EDITOR_PP_MI32
        // full tiles, 16-bit output through the one-pass staged epilogue (the conditions pp_body's `staged` path tests): 32x32x16 MFMAs
        if constexpr (!CF) {
            if (g.beta == 0.f && g.splitk == 1 && (g.N & 7) == 0 && (g.ldc & 7) == 0 && (g.ldaux & 7) == 0 &&
                (g.epilogue == EDITOR_EPI_NONE || g.epilogue == EDITOR_EPI_GELU || g.epilogue == EDITOR_EPI_GELU_BWD))
                return launch_pp_t<F16, AK, BK_, CF, 8, 8, false, true>(g, stream);
        }
#endif
    }
    return launch_pp_t<F16, AK, BK_, CF, 8, 8>(g, stream);
}

template <bool F16, bool AK, bool BK_, bool CF>
int launch_pipe(const GemmB16Args& g, hipStream_t stream)
{
    // 256x256 ping-pong kernel where it measures faster (tools/gemm_bench.py with GEMM_EPI=1, M = 49 536 token rows,
    // TFLOP/s 256x128 -> 256x256): qkv+bias 596 -> 828, fc1+bias+GELU 412 -> 516, proj+residual 344 -> 448, fc2+residual
    // 702 -> 865, dgrads 777 -> 896, 850 -> 974, 604 -> 624, 649 -> 747, fc2 dgrad+GELU' 460 -> 507.  wgrad (both operands
    // through ds_read_b64_tr_b16, few output tiles): the CALLER picks - at the 256x128 kernel's best split the ping-pong
    // kernel is slower (742 vs 792), at its own (one round of 256x256 tiles x splits) faster (979-1140 vs 729-862); see
    // editor_amd.functional._splitk_for, which passes EDITOR_EPI_FORCE_PP with that split.
    int pp_mode = (g.force_pp || g.tile_frags == 13) ? 1 : -1;          // 1 force, 0 off (debug build only)
#ifdef EDITOR_DEBUG_TRACE
    if (const char* e = getenv("EDITOR_GEMM_PP")) pp_mode = atoi(e);
#endif
    // EDITOR_EPI_PIPE128 (the caller's tile-count heuristic, editor_amd.ops.gemm): few token rows - the strong-scaling series'
    // B_local = 16 gives M = 6 192, i.e. 72 - 90 tiles of a 768-wide output on 256 CUs - take the 256x128 tiles instead
    const bool pp_auto = g.M >= 2048 && g.splitk == 1 && AK && g.N >= 512 && !g.prefer_pipe;
    if (g.N >= 256 && (pp_mode == 1 || g.colsum || (pp_mode < 0 && pp_auto))) return launch_pp<F16, AK, BK_, CF>(g, stream);
    return launch_pipe_t<F16, AK, BK_, CF, 256, 128, 3, 8>(g, stream);
}

template <bool F16>
int gemm_h16(const uint16_t* A, const uint16_t* B, void* C, int c_f32, int M, int N, int K, long lda,
    long ldb, long ldc, int transA, int transB, float alpha, float beta, const float* bias, const float* rowscale,
    int splitk, int epilogue, void* aux, long ldaux, float* splitk_ws, const int* m_live, hipStream_t stream,
    const int* rowmap = nullptr)
{
    if (M <= 0 || N <= 0 || K <= 0) return (int)hipErrorInvalidValue;
    // 16-byte vector accesses: leading dimensions and the contiguous extents must be multiples of 8 bf16
    if ((lda & 7) || (ldb & 7) || (N & 3) || (ldc & 3)) return (int)hipErrorInvalidValue;
    if (!transA && (K & 7)) return (int)hipErrorInvalidValue;
    if (transA && (M & 7)) return (int)hipErrorInvalidValue;
    if (!transB && (K & 7)) return (int)hipErrorInvalidValue;
    if (transB && (N & 7)) return (int)hipErrorInvalidValue;
    if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B) | reinterpret_cast<uintptr_t>(C)) & 15)
        return (int)hipErrorInvalidValue;
    const bool want_colsum = (epilogue & EDITOR_EPI_COLSUM) != 0;
    const bool force_pp = (epilogue & EDITOR_EPI_FORCE_PP) != 0;
    const bool aux_grad = (epilogue & EDITOR_EPI_AUX_GRAD) != 0;
    const int tile_frags = (epilogue >> 12) & 15;                       // EDITOR_EPI_TILE_ROWS(h): h / 16, 0 = full tiles
    if (tile_frags != 0 && tile_frags != 13) return (int)hipErrorInvalidValue;
    const bool prefer_pipe = (epilogue & EDITOR_EPI_PIPE128) != 0 && !want_colsum && !force_pp && tile_frags == 0;
    const int stagger = ((epilogue >> 17) & 63) * 2048;                 // EDITOR_EPI_STAGGER(c)
    epilogue &= ~(EDITOR_EPI_COLSUM | EDITOR_EPI_FORCE_PP | EDITOR_EPI_AUX_GRAD | EDITOR_EPI_PIPE128 | 0xF000 | EDITOR_EPI_STAGGER(63));
    if (aux_grad && epilogue != EDITOR_EPI_GELU && epilogue != EDITOR_EPI_GELU_BWD) return (int)hipErrorInvalidValue;
    if (want_colsum && (c_f32 || epilogue == EDITOR_EPI_RESIDUAL || splitk > 1 || !splitk_ws || transA || M < 2048 || N < 512 ||
                        (N & 7) || (ldc & 7) || (ldaux & 7) || (K % BK) || beta != 0.f))
        return (int)hipErrorInvalidValue;                        // the column sums exist in the one-pass 256x256 epilogue only
    // (with m_live: the tiles beyond the live rows zero their partial row, the rows of the last live tile beyond *m_live are
    //  the caller's zero rows by contract - stochastic-depth compaction, editor_droppath_plan)
    if (rowmap && (!c_f32 || (epilogue & 0xFF) != EDITOR_EPI_RESIDUAL || transA || transB || splitk > 1 || (N & 7) || (ldc & 7) ||
                   (ldaux & 7) || (K % BK) || beta != 0.f || M < 256 || N < 256))
        return (int)hipErrorInvalidValue;                        // row scatter: the staged fp32 residual epilogue of the 256x256 kernel
    // (aux may be NULL for EDITOR_EPI_GELU: a no-grad forward - model.eval() / do_inference - saves nothing for a backward)
    if (epilogue != EDITOR_EPI_NONE && ((!aux && epilogue != EDITOR_EPI_GELU) || (ldaux & 3) || splitk > 1)) return (int)hipErrorInvalidValue;
    if (splitk < 1) splitk = 1;
    const int ktiles = (K + BK - 1) / BK;
    if (splitk > ktiles) splitk = ktiles;
    if (splitk > 1 && !c_f32) return (int)hipErrorInvalidValue;
    // direct-to-LDS staging needs whole BK tiles along the reduction and >= 8 valid elements to clamp to
    const bool glds = (K % BK == 0) && M >= 8 && N >= 8;
    // large problems: 3-stage LDS-DMA pipeline (256x128 tiles)
    const bool pipe = glds && M >= 256 && N >= 128;
    const bool short_tiles = tile_frags == 13;
    if (short_tiles && (transA || transB || splitk != 1 || beta != 0.f || (N & 7) || (ldc & 7) || (ldaux & 7) || !pipe || N < 256))
        return (int)hipErrorInvalidValue;                       // (an error, not a fallback: the column-sum layout depends on h)
    // split-K: per-split slabs in the workspace + a reduction kernel (pipelined path), else fp32 atomics into C
    const bool slabs = splitk > 1 && pipe && (transA || transB) && splitk_ws && ldc == N && (((long)M * N) & 3) == 0 &&
                       (reinterpret_cast<uintptr_t>(splitk_ws) & 15) == 0;
    if (splitk > 1 && !slabs && beta != 1.f) {
        hipLaunchKernelGGL(scale_c_kernel, dim3((unsigned)(((long)M * N + 255) / 256)), dim3(256), 0, stream,
                           (float*)C, (long)M, N, ldc, beta);
        EDITOR_LAUNCH_CHECK();
    }
    GemmB16Args g{(const bf16_t*)A, (const bf16_t*)B, slabs ? (void*)splitk_ws : C, M, N, K, lda, ldb, ldc, alpha,
                  slabs ? 0.f : beta, bias, rowscale, splitk, (M + BM - 1) / BM, (N + BN - 1) / BN, epilogue, aux, ldaux,
                  slabs ? 1 : 0,
                  m_live, transA ? 1 : 0, 0, want_colsum ? splitk_ws : nullptr, nullptr, force_pp ? 1 : 0, aux_grad ? 1 : 0,
                  tile_frags, nullptr, nullptr, nullptr, 0, prefer_pipe ? 1 : 0, stagger, 0, rowmap};
    if (rowmap) g.force_pp = 1;
    if (m_live && (!pipe || (splitk > 1 && !slabs))) return (int)hipErrorInvalidValue;   // live-row form: pipelined path only
    const int sel = (transA ? 0 : 4) | (transB ? 0 : 2) | (c_f32 ? 1 : 0);
    int rc;
    if (pipe) {
        switch (sel) {
            case 7: rc = launch_pipe<F16, true, true, true>(g, stream); break;
            case 6: rc = launch_pipe<F16, true, true, false>(g, stream); break;
            case 5: rc = launch_pipe<F16, true, false, true>(g, stream); break;
            case 4: rc = launch_pipe<F16, true, false, false>(g, stream); break;
            case 3: rc = launch_pipe<F16, false, true, true>(g, stream); break;
            case 2: rc = launch_pipe<F16, false, true, false>(g, stream); break;
            case 1: rc = launch_pipe<F16, false, false, true>(g, stream); break;
            default: rc = launch_pipe<F16, false, false, false>(g, stream); break;
        }
        if (rc) return rc;
        if (slabs) {
            const long n4 = (long)M * N / 4;
            long blocks = (n4 + 255) / 256;
            if (blocks > 4096) blocks = 4096;
            hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, splitk_ws, splitk, n4,
                               (float*)C, beta);
            EDITOR_LAUNCH_CHECK();
        }
        return 0;
    }
#define GEMM_CASE(n, a, b, c) case n: return glds ? launch<F16, a, b, c, true>(g, stream) : launch<F16, a, b, c, false>(g, stream)
    switch (sel) {
        GEMM_CASE(7, true, true, true);
        GEMM_CASE(6, true, true, false);
        GEMM_CASE(5, true, false, true);
        GEMM_CASE(4, true, false, false);
        GEMM_CASE(3, false, true, true);
        GEMM_CASE(2, false, true, false);
        GEMM_CASE(1, false, false, true);
        default: return glds ? launch<F16, false, false, false, true>(g, stream) : launch<F16, false, false, false, false>(g, stream);
    }
#undef GEMM_CASE
}

// Forward product of the split-precision mode: C = alpha * (A_hi + A_lo)(B_hi + B_lo)^T (+bias) (*rowscale) (+aux residual /
// GELU), A (M,K) and B (N,K) k-major half pairs, always on the 256x256 ping-pong kernel (any M, N; K % 64 == 0).
int gemm_f16x2(const uint16_t* A_hi, const uint16_t* A_lo, const uint16_t* B_hi, const uint16_t* B_lo, void* C, void* C_lo,
    int c_f32, int M, int N, int K, long lda, long ldb, long ldc, float alpha, const float* bias, const float* rowscale,
    int epilogue, void* aux, long ldaux, const int* m_live, hipStream_t stream, const int* rowmap = nullptr)
{
    if (M <= 0 || N < 8 || K <= 0 || (K % BK) || (N & 7) || (lda & 7) || (ldb & 7) || (ldc & 7) || (ldaux & 7))
        return (int)hipErrorInvalidValue;
    if (!A_hi || !A_lo || !B_hi || !B_lo || !C || (!c_f32 && !C_lo) || (c_f32 && C_lo)) return (int)hipErrorInvalidValue;
    const uintptr_t al = reinterpret_cast<uintptr_t>(A_hi) | reinterpret_cast<uintptr_t>(A_lo) | reinterpret_cast<uintptr_t>(B_hi) |
                         reinterpret_cast<uintptr_t>(B_lo) | reinterpret_cast<uintptr_t>(C) | reinterpret_cast<uintptr_t>(C_lo);
    if (al & 15) return (int)hipErrorInvalidValue;
    const int tile_frags = (epilogue >> 12) & 15;
    if (tile_frags != 0 && tile_frags != 13) return (int)hipErrorInvalidValue;
    const bool aux_grad = (epilogue & EDITOR_EPI_AUX_GRAD) != 0;
    epilogue &= ~(EDITOR_EPI_FORCE_PP | EDITOR_EPI_AUX_GRAD | 0xF000);
    if (epilogue != EDITOR_EPI_NONE && epilogue != EDITOR_EPI_RESIDUAL && epilogue != EDITOR_EPI_GELU) return (int)hipErrorInvalidValue;
    if (epilogue != EDITOR_EPI_NONE && !aux && epilogue != EDITOR_EPI_GELU) return (int)hipErrorInvalidValue;   // (GELU, aux NULL: no-grad forward)
    if (epilogue == EDITOR_EPI_GELU && (c_f32 || !aux_grad)) return (int)hipErrorInvalidValue;   // aux = gelu'(x) for the backward
    if (epilogue == EDITOR_EPI_RESIDUAL && !c_f32) return (int)hipErrorInvalidValue;
    if (rowmap && (epilogue != EDITOR_EPI_RESIDUAL || M < 256 || N < 256)) return (int)hipErrorInvalidValue;   // (see GemmB16Args::rowmap)
    GemmB16Args g{(const bf16_t*)A_hi, (const bf16_t*)B_hi, C, M, N, K, lda, ldb, ldc, alpha, 0.f, bias, rowscale, 1, 0, 0,
                  epilogue, aux, ldaux, 0, m_live, 0, 1, nullptr, nullptr, 1, 1, tile_frags,
                  (const bf16_t*)A_lo, (const bf16_t*)B_lo, C_lo};
    g.rowmap = rowmap;
    if (tile_frags == 13)
        return c_f32 ? launch_pp_t<true, true, true, true, 7, 6, true>(g, stream) : launch_pp_t<true, true, true, false, 7, 6, true>(g, stream);
    return c_f32 ? launch_pp_t<true, true, true, true, 8, 8, true>(g, stream) : launch_pp_t<true, true, true, false, 8, 8, true>(g, stream);
}

// The weight gradients of one block in one launch (see gemm_bf16_pp_group_kernel).  Problem i: dW_i (N_i x K_i fp32, contiguous) =
// alpha * dy_i^T x_i with dy_i (M, N_i), x_i (M, K_i) 16-bit row-major, M token rows shared; N_i, K_i multiples of 256.
// ws: splitk * sum_i N_i K_i floats of slabs; fixed-order reduction per problem afterwards (deterministic).
template <bool F16>
int gemm_wgrad_group(int count, const uint16_t* const* dy, const uint16_t* const* x, float* const* dw, const int* N, const int* K,
                     int M, float alpha, int splitk, float* ws, const int* m_live, hipStream_t stream, const LnRoleArgs* ln = nullptr,
                     const int* const* m_live_each = nullptr)
{
    if (count < 1 || count > kMaxGroup || M < 64 || (M % 64) || splitk < 1 || !ws) return (int)hipErrorInvalidValue;
    GemmGroupArgs ga;
    memset(&ga, 0, sizeof(ga));
    ga.n = count;
    long wsoff = 0;
    int tiles = 0;
    const int ktiles = M / BK;
    if (splitk > ktiles) splitk = ktiles;
    float* slab[kMaxGroup];
    for (int i = 0; i < count; ++i) {
        if (N[i] % 256 || K[i] % 256 || !dy[i] || !x[i] || !dw[i]) return (int)hipErrorInvalidValue;
        if ((reinterpret_cast<uintptr_t>(dy[i]) | reinterpret_cast<uintptr_t>(x[i]) | reinterpret_cast<uintptr_t>(dw[i])) & 15) return (int)hipErrorInvalidValue;
        slab[i] = ws + wsoff;
        // output (N_i, K_i); "A" = dy stored (Kred = M, N_i) row-k, "B" = x stored (Kred = M, K_i) row-k
        GemmB16Args g{(const bf16_t*)dy[i], (const bf16_t*)x[i], (void*)slab[i], N[i], K[i], M, (long)N[i], (long)K[i], (long)K[i],
                      alpha, 0.f, nullptr, nullptr, splitk, N[i] / 256, K[i] / 256, EDITOR_EPI_NONE, nullptr, (long)K[i], 1,
                      m_live_each ? m_live_each[i] : m_live, 1, 1, nullptr, nullptr, 1, 0, 0, nullptr, nullptr, nullptr, 1};
        ga.p[i] = g;
        ga.start[i] = tiles;
        tiles += g.tiles_m * g.tiles_n;
        wsoff += (long)splitk * N[i] * K[i];
    }
    ga.start[count] = tiles;
    constexpr int LDS = 256 * (256 * 2 + 16) + (int)kLutBytes;
    if (ln) {
        if (int e = ensure_lds<gemm_bf16_pp_group_ln_kernel<F16>>(LDS)) return e;
        hipLaunchKernelGGL(gemm_bf16_pp_group_ln_kernel<F16>, dim3(ln->nmem + tiles * splitk), dim3(512), LDS, stream, ga, *ln);
    } else {
        if (int e = ensure_lds<gemm_bf16_pp_group_kernel<F16>>(LDS)) return e;
        hipLaunchKernelGGL(gemm_bf16_pp_group_kernel<F16>, dim3(tiles * splitk), dim3(512), LDS, stream, ga);
    }
    EDITOR_LAUNCH_CHECK();
    SlabJobs sj;
    memset(&sj, 0, sizeof(sj));
    sj.nsplit = splitk;
    long n4max = 0;
    for (int i = 0; i < count; ++i) {
        sj.slabs[i] = slab[i]; sj.out[i] = dw[i]; sj.n4[i] = (long)N[i] * K[i] / 4;
        if (sj.n4[i] > n4max) n4max = sj.n4[i];
    }
    long blocks = (n4max + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(slab_reduce_multi_kernel, dim3((unsigned)blocks, (unsigned)count), dim3(256), 0, stream, sj);
    EDITOR_LAUNCH_CHECK();
    return 0;
}

}  // namespace

template <bool F16>
int gemm_fwd_group(int count, const uint16_t* const* A, const uint16_t* const* B, void* const* C, int c_f32, int M, int N, int K,
                   long lda, long ldb, long ldc, float alpha, const float* const* bias, const float* const* rowscale, int epilogue,
                   void* const* aux, long ldaux, const int* m_live, hipStream_t stream)
{
    if (count < 1 || count > kMaxGroup || M < 256 || N < 256 || (N & 255) || K < BK || (K % BK)) return (int)hipErrorInvalidValue;
    if ((lda & 7) || (ldb & 7) || (ldc & 7) || (ldaux & 7)) return (int)hipErrorInvalidValue;
    const bool aux_grad = (epilogue & EDITOR_EPI_AUX_GRAD) != 0;
    epilogue &= ~(EDITOR_EPI_FORCE_PP | EDITOR_EPI_AUX_GRAD);
    if (epilogue != EDITOR_EPI_NONE && epilogue != EDITOR_EPI_RESIDUAL && epilogue != EDITOR_EPI_GELU && epilogue != EDITOR_EPI_GELU_BWD)
        return (int)hipErrorInvalidValue;
    if (aux_grad && epilogue != EDITOR_EPI_GELU && epilogue != EDITOR_EPI_GELU_BWD) return (int)hipErrorInvalidValue;
    GemmGroupArgs ga;
    memset(&ga, 0, sizeof(ga));
    ga.n = count;
    const int tiles_m = (M + 255) / 256, tiles_n = N / 256;
    for (int i = 0; i < count; ++i) {
        if (!A[i] || !B[i] || !C[i]) return (int)hipErrorInvalidValue;
        void* ax = aux ? aux[i] : nullptr;
        if (epilogue != EDITOR_EPI_NONE && !ax && epilogue != EDITOR_EPI_GELU) return (int)hipErrorInvalidValue;
        if ((reinterpret_cast<uintptr_t>(A[i]) | reinterpret_cast<uintptr_t>(B[i]) | reinterpret_cast<uintptr_t>(C[i])) & 15)
            return (int)hipErrorInvalidValue;
        GemmB16Args g{(const bf16_t*)A[i], (const bf16_t*)B[i], C[i], M, N, K, lda, ldb, ldc, alpha, 0.f, bias ? bias[i] : nullptr,
                      rowscale ? rowscale[i] : nullptr, 1, tiles_m, tiles_n, epilogue, ax, ldaux, 0, m_live, 0, 1, nullptr, nullptr, 1,
                      aux_grad ? 1 : 0, 0, nullptr, nullptr, nullptr, 0, 0};
        ga.p[i] = g;
        ga.start[i] = i * tiles_m * tiles_n;
    }
    ga.start[count] = count * tiles_m * tiles_n;
    for (int i = count + 1; i <= kMaxGroup; ++i) ga.start[i] = ga.start[count];
    constexpr int LDS = 256 * (256 * 2 + 16) + (int)kLutBytes;
    const dim3 grid(count * tiles_m * tiles_n);
    if (c_f32) {
        if (int e = ensure_lds<gemm_bf16_pp_fgroup_kernel<F16, true>>(LDS)) return e;
        hipLaunchKernelGGL((gemm_bf16_pp_fgroup_kernel<F16, true>), grid, dim3(512), LDS, stream, ga);
    } else {
        if (int e = ensure_lds<gemm_bf16_pp_fgroup_kernel<F16, false>>(LDS)) return e;
        hipLaunchKernelGGL((gemm_bf16_pp_fgroup_kernel<F16, false>), grid, dim3(512), LDS, stream, ga);
    }
    EDITOR_LAUNCH_CHECK();
    return 0;
}

extern "C" int editor_gemm_group(int dtype, int count, const uint16_t* const* A, const uint16_t* const* B, void* const* C, int c_f32,
    int M, int N, int K, long lda, long ldb, long ldc, float alpha, const float* const* bias, const float* const* rowscale,
    int epilogue, void* const* aux, long ldaux, const int* m_live, hipStream_t stream)
{
    if (dtype == 2) return gemm_fwd_group<true>(count, A, B, C, c_f32, M, N, K, lda, ldb, ldc, alpha, bias, rowscale, epilogue, aux, ldaux, m_live, stream);
    if (dtype == 1) return gemm_fwd_group<false>(count, A, B, C, c_f32, M, N, K, lda, ldb, ldc, alpha, bias, rowscale, epilogue, aux, ldaux, m_live, stream);
    return (int)hipErrorInvalidValue;
}

#ifdef EDITOR_DEBUG_TRACE
// Probe (debug build only, tools/hetero_probe.py): ONE launch whose first `nmem` workgroups stream memory (d = s0 + s1 over
// n4 float4, grid-stride over those workgroups - a stand-in for an HBM-bound pass such as a LayerNorm backward) while the others run
// the ping-pong kernel's body on the tiles of a forward product.  Same resource footprint for both roles (one workgroup per CU), so
// the first nmem workgroups to be dispatched hold nmem CUs for as long as their stream lasts and the tiles cycle over the rest:
// does an MFMA-bound and an HBM-bound role overlap INSIDE a launch, where two queues do not (4.1e)?
struct HeteroArgs { GemmB16Args g; const float4_t* s0; const float4_t* s1; float4_t* d; long n4; int nmem; };
template <bool F16, int U>                                      // U: 2 U x 16-byte loads in flight per thread (U = 8: 128 KiB per workgroup)
__global__ __launch_bounds__(512) void gemm_hetero_probe_kernel(HeteroArgs h)
{
    if ((int)blockIdx.x < h.nmem) {
        const long stride = (long)h.nmem * 512;
        long i = (long)blockIdx.x * 512 + threadIdx.x;
        for (; i + (U - 1) * stride < h.n4; i += U * stride) {
            float4_t a[U], b[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { a[u] = __builtin_nontemporal_load(h.s0 + i + u * stride); b[u] = __builtin_nontemporal_load(h.s1 + i + u * stride); }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                __builtin_nontemporal_store(a[u] + b[u], h.d + i + u * stride);
            }
        }
        for (; i < h.n4; i += stride) {
            h.d[i] = h.s0[i] + h.s1[i];
        }
        return;
    }
    pp_body<F16, true, true, false, 8, 8, false>(h.g, (int)blockIdx.x - h.nmem, 0);
}

// the same with the REAL LayerNorm-backward role (ln_bwd_cast_role) as the memory-bound part: is it the role or its partner that
// made editor_gemm_wgrad_group_ln slow?
template <bool F16>
__global__ __launch_bounds__(512) void gemm_hetero_ln_probe_kernel(GemmB16Args g, LnRoleArgs ln)
{
    if ((int)blockIdx.x < ln.nmem) {
        extern __shared__ __attribute__((aligned(16))) char smem_ln[];
        ln_bwd_cast_role<F16, 3>(ln, (int)blockIdx.x, reinterpret_cast<float*>(smem_ln));
        return;
    }
    pp_body<F16, true, true, false, 8, 8, false>(g, (int)blockIdx.x - ln.nmem, 0);
}
extern "C" int editor_probe_gemm_hetero_ln(const uint16_t* A, const uint16_t* B, void* C, int M, int N, int K, const float* bias,
    int with_tiles, const uint16_t* ln_dy, const float* ln_x, const float* gamma, const float* mean, const float* rstd, long ln_M,
    const float* dx_in, float* dx_out, float* partials, uint16_t* cast_out, float* cast_partials, int nmem, hipStream_t stream)
{
    if (M < 256 || N < 256 || (N & 255) || K < BK || (K % BK) || nmem < 8 || (nmem & 7)) return (int)hipErrorInvalidValue;
    const int tiles_m = (M + 255) / 256, tiles_n = N / 256;
    GemmB16Args g{(const bf16_t*)A, (const bf16_t*)B, C, M, N, K, (long)K, (long)K, (long)N, 1.f, 0.f, bias, nullptr, 1, tiles_m, tiles_n,
                  EDITOR_EPI_NONE, nullptr, (long)N, 0, nullptr, 0, 1, nullptr, nullptr, 1, 0, 0, nullptr, nullptr, nullptr, 0, 0};
    LnRoleArgs ln{ln_dy, ln_x, gamma, mean, rstd, ln_M, 768, dx_in, dx_out, partials, 1.f, cast_out, nullptr, 1.f, cast_partials, nmem};
    constexpr int LDS = 256 * (256 * 2 + 16) + (int)kLutBytes;
    if (int e = ensure_lds<gemm_hetero_ln_probe_kernel<false>>(LDS)) return e;
    const int grid = nmem + (with_tiles ? tiles_m * tiles_n : 0);
    hipLaunchKernelGGL((gemm_hetero_ln_probe_kernel<false>), dim3(grid), dim3(512), LDS, stream, g, ln);
    EDITOR_LAUNCH_CHECK();
    return 0;
}

// tiles == 0: only the memory role (nmem workgroups); n4 == 0: only the product (the memory workgroups return at once)
extern "C" int editor_probe_gemm_hetero(const uint16_t* A, const uint16_t* B, void* C, int M, int N, int K, const float* bias,
                                        int with_tiles, const float* s0, const float* s1, float* d, long n4, int nmem,
                                        int unroll, hipStream_t stream)
{
    if (M < 256 || N < 256 || (N & 255) || K < BK || (K % BK) || nmem < 0 || (nmem & 7)) return (int)hipErrorInvalidValue;
    const int tiles_m = (M + 255) / 256, tiles_n = N / 256;
    HeteroArgs h;
    memset(&h, 0, sizeof(h));
    GemmB16Args g{(const bf16_t*)A, (const bf16_t*)B, C, M, N, K, (long)K, (long)K, (long)N, 1.f, 0.f, bias, nullptr, 1, tiles_m, tiles_n,
                  EDITOR_EPI_NONE, nullptr, (long)N, 0, nullptr, 0, 1, nullptr, nullptr, 1, 0, 0, nullptr, nullptr, nullptr, 0, 0};
    h.g = g;
    h.s0 = reinterpret_cast<const float4_t*>(s0); h.s1 = reinterpret_cast<const float4_t*>(s1); h.d = reinterpret_cast<float4_t*>(d);
    h.n4 = n4; h.nmem = nmem;
    constexpr int LDS = 256 * (256 * 2 + 16) + (int)kLutBytes;
    const int grid = nmem + (with_tiles ? tiles_m * tiles_n : 0);
    if (grid < 1) return (int)hipErrorInvalidValue;
#define HETERO_CASE(Uv) case Uv: { if (int e = ensure_lds<gemm_hetero_probe_kernel<false, Uv>>(LDS)) return e;                 \
        hipLaunchKernelGGL((gemm_hetero_probe_kernel<false, Uv>), dim3(grid), dim3(512), LDS, stream, h); break; }
    switch (unroll) { HETERO_CASE(4) HETERO_CASE(8) HETERO_CASE(16) default: return (int)hipErrorInvalidValue; }
#undef HETERO_CASE
    EDITOR_LAUNCH_CHECK();
    return 0;
}
#endif

extern "C" int editor_gemm_wgrad_group(int dtype, int count, const uint16_t* const* dy, const uint16_t* const* x, float* const* dw,
    const int* N, const int* K, int M, float alpha, int splitk, float* ws, const int* m_live, hipStream_t stream)
{
    if (dtype == 2) return gemm_wgrad_group<true>(count, dy, x, dw, N, K, M, alpha, splitk, ws, m_live, stream);
    if (dtype == 1) return gemm_wgrad_group<false>(count, dy, x, dw, N, K, M, alpha, splitk, ws, m_live, stream);
    return (int)hipErrorInvalidValue;
}

// editor_gemm_wgrad_group with a live-row count PER PROBLEM (m_live: host array of `count` device scalars, NULL entries = all M rows):
// a block whose MLP branch ran on compacted rows (stochastic depth, editor_droppath_plan) next to its dense attention branch
extern "C" int editor_gemm_wgrad_group_live(int dtype, int count, const uint16_t* const* dy, const uint16_t* const* x, float* const* dw,
    const int* N, const int* K, int M, float alpha, int splitk, float* ws, const int* const* m_live, hipStream_t stream)
{
    if (!m_live) return (int)hipErrorInvalidValue;
    if (dtype == 2) return gemm_wgrad_group<true>(count, dy, x, dw, N, K, M, alpha, splitk, ws, nullptr, stream, nullptr, m_live);
    if (dtype == 1) return gemm_wgrad_group<false>(count, dy, x, dw, N, K, M, alpha, splitk, ws, nullptr, stream, nullptr, m_live);
    return (int)hipErrorInvalidValue;
}

extern "C" int editor_gemm_wgrad_group_ln(int dtype, int count, const uint16_t* const* dy, const uint16_t* const* x, float* const* dw,
    const int* N, const int* K, int M, float alpha, int splitk, float* ws,
    const uint16_t* ln_dy, float ln_dy_scale, const float* ln_x, const float* gamma, const float* mean, const float* rstd, long ln_M,
    int D, const float* dx_in, float* dx_out, float* partials, uint16_t* cast_out, const float* cast_rowscale, float cast_scale,
    float* cast_partials, int nmem, hipStream_t stream)
{
    if ((D != 768 && D != 1024) || nmem < 8 || (nmem & 7) || nmem > 256 || ln_M < 1 || !ln_dy || !ln_x || !gamma || !mean || !rstd ||
        !dx_out || !partials || !cast_out)
        return (int)hipErrorInvalidValue;
    if ((reinterpret_cast<uintptr_t>(ln_dy) & 7) || ((reinterpret_cast<uintptr_t>(ln_x) | reinterpret_cast<uintptr_t>(dx_out) |
         reinterpret_cast<uintptr_t>(dx_in) | reinterpret_cast<uintptr_t>(gamma)) & 15) || (reinterpret_cast<uintptr_t>(cast_out) & 7))
        return (int)hipErrorInvalidValue;
    LnRoleArgs ln{ln_dy, ln_x, gamma, mean, rstd, ln_M, D, dx_in, dx_out, partials, ln_dy_scale, cast_out, cast_rowscale, cast_scale,
                  cast_partials, nmem};
    if (dtype == 2) return gemm_wgrad_group<true>(count, dy, x, dw, N, K, M, alpha, splitk, ws, nullptr, stream, &ln);
    if (dtype == 1) return gemm_wgrad_group<false>(count, dy, x, dw, N, K, M, alpha, splitk, ws, nullptr, stream, &ln);
    return (int)hipErrorInvalidValue;
}

extern "C" int editor_gemm_f16x2(const uint16_t* A_hi, const uint16_t* A_lo, const uint16_t* B_hi, const uint16_t* B_lo, void* C,
    void* C_lo, int c_f32, int M, int N, int K, long lda, long ldb, long ldc, float alpha, const float* bias,
    const float* rowscale, int epilogue, void* aux, long ldaux, const int* m_live, hipStream_t stream)
{
    return gemm_f16x2(A_hi, A_lo, B_hi, B_lo, C, C_lo, c_f32, M, N, K, lda, ldb, ldc, alpha, bias, rowscale, epilogue, aux, ldaux,
                      m_live, stream);
}

// editor_gemm_f16x2 (fp32 C, EDITOR_EPI_RESIDUAL) with an output ROW MAP: the split-precision fc2 of a stochastic-depth-compacted MLP
extern "C" int editor_gemm_f16x2_rows(const uint16_t* A_hi, const uint16_t* A_lo, const uint16_t* B_hi, const uint16_t* B_lo, void* C,
    int M, int N, int K, long lda, long ldb, long ldc, float alpha, const float* bias, const float* rowscale, int epilogue, void* aux,
    long ldaux, const int* m_live, const int* rowmap, hipStream_t stream)
{
    if (!rowmap) return (int)hipErrorInvalidValue;
    return gemm_f16x2(A_hi, A_lo, B_hi, B_lo, C, nullptr, 1, M, N, K, lda, ldb, ldc, alpha, bias, rowscale, epilogue, aux, ldaux,
                      m_live, stream, rowmap);
}

extern "C" int editor_gemm_bf16(const uint16_t* A, const uint16_t* B, void* C, int c_f32, int M, int N, int K, long lda,
    long ldb, long ldc, int transA, int transB, float alpha, float beta, const float* bias, const float* rowscale,
    int splitk, int epilogue, void* aux, long ldaux, float* splitk_ws, const int* m_live, hipStream_t stream)
{
    return gemm_h16<false>(A, B, C, c_f32, M, N, K, lda, ldb, ldc, transA, transB, alpha, beta, bias, rowscale, splitk, epilogue,
                           aux, ldaux, splitk_ws, m_live, stream);
}

// IEEE-half operands (cfg.MODEL.COMPUTE_DTYPE = 'f16'): same kernels, v_mfma_f32_16x16x32_f16, half outputs
extern "C" int editor_gemm_f16(const uint16_t* A, const uint16_t* B, void* C, int c_f32, int M, int N, int K, long lda,
    long ldb, long ldc, int transA, int transB, float alpha, float beta, const float* bias, const float* rowscale,
    int splitk, int epilogue, void* aux, long ldaux, float* splitk_ws, const int* m_live, hipStream_t stream)
{
    return gemm_h16<true>(A, B, C, c_f32, M, N, K, lda, ldb, ldc, transA, transB, alpha, beta, bias, rowscale, splitk, epilogue,
                          aux, ldaux, splitk_ws, m_live, stream);
}

// editor_gemm_bf16 / _f16 (dtype 1 / 2) with an output ROW MAP (EDITOR_EPI_RESIDUAL, fp32 C, both operands k-major): see GemmB16Args::rowmap
extern "C" int editor_gemm_h16_rows(int dtype, const uint16_t* A, const uint16_t* B, void* C, int M, int N, int K, long lda, long ldb,
    long ldc, float alpha, const float* bias, const float* rowscale, int epilogue, void* aux, long ldaux, const int* m_live,
    const int* rowmap, hipStream_t stream)
{
    if (!rowmap) return (int)hipErrorInvalidValue;
    if (dtype == 2) return gemm_h16<true>(A, B, C, 1, M, N, K, lda, ldb, ldc, 0, 0, alpha, 0.f, bias, rowscale, 1, epilogue, aux, ldaux,
                                          nullptr, m_live, stream, rowmap);
    if (dtype == 1) return gemm_h16<false>(A, B, C, 1, M, N, K, lda, ldb, ldc, 0, 0, alpha, 0.f, bias, rowscale, 1, epilogue, aux, ldaux,
                                           nullptr, m_live, stream, rowmap);
    return (int)hipErrorInvalidValue;
}
