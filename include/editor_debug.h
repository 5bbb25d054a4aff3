/* Debug / bring-up entry points.  NOT part of libeditor_hip.so: they live in separate shared objects that only tests
 * and tools load, so that the product library keeps its contract (include/editor_hip.h: no allocation, no
 * synchronisation, no environment lookups inside an entry point).
 *
 *   editor_amd/libeditor_probe.so   csrc/probe.hip - hardware-semantics probes used by tests/test_gpu_select.py
 *   editor_amd/libeditor_gemm_trace.so   csrc/gemm_bf16.hip compiled with -DEDITOR_DEBUG_TRACE: the same
 *       editor_gemm_bf16 / editor_gemm_f16 symbols, plus (EDITOR_GEMM_TRACE=1 in the environment) a per-workgroup
 *       s_memtime timeline printed per launch - this build allocates a device buffer and synchronises; the
 *       EDITOR_GEMM_PP / EDITOR_GEMM_PP_STAGED experiment switches exist only here.  Built on demand by
 *       `python -m editor_amd.build --trace`; used by tools/gemm_bench.py. */
#ifndef EDITOR_DEBUG_H
#define EDITOR_DEBUG_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct ihipStream_t* editor_stream_t;
/* what ds_read_b64_tr_b16 returns for a known LDS image (lds[i] = i as u16): lane l reads at byte address addr[l] */
int editor_probe_tr16(const int* addr, uint16_t* out, editor_stream_t stream);
/* lane -> element maps of v_mfma_f32_16x16x32_bf16: D (16x16) = A (16x32) B (32x16), operands given as fp32 */
int editor_probe_mfma16(const float* A, const float* B, float* D, editor_stream_t stream);
/* D (16x16 fp32) = A (16x32) B (32x16) with the operands given as RAW 16-bit patterns (f16 != 0: IEEE half, else bf16):
 * pins the matrix core's treatment of subnormal half operands (tests/test_gpu_kernels.py) */
int editor_probe_mfma16_raw(const uint16_t* A, const uint16_t* B, float* D, int f16, editor_stream_t stream);
/* n (s_memtime, s_memrealtime) pairs (out: 2n u64), one every ~sleep x 8 k shader cycles, from a single resident wave:
 * shader-clock cycles against the constant 100 MHz reference = the clock the CUs really run at (tools/clock_probe.py) */
int editor_probe_clock_trace(unsigned long long* out, int n, int sleep, editor_stream_t stream);
/* grid x 4 wavefronts each issuing iters x 8 independent v_mfma_f32_16x16x32_bf16 on register operands (no memory traffic):
 * the dense bf16 rate the chip sustains under its power management.  FLOPs = grid * 4 * iters * 8 * 16384. */
int editor_probe_mfma_peak(float* out, int grid, int iters, int zero, editor_stream_t stream);
/* pricing experiment (tools/residual_pricing.py; not on the product path): x_out = x + rowscale[row] * branch (bf16), y =
 * LayerNorm(x_out) in bf16, D = 768 - the residual add moved from the GEMM's fp32 epilogue into the LayerNorm that follows */
int editor_probe_resid_add_layernorm(const float* x, const uint16_t* branch, const float* rowscale, const float* gamma,
                                     const float* beta, float eps, long M, int D, float* x_out, uint16_t* y, float* mean,
                                     float* rstd, editor_stream_t stream);
/* bring-up of the four-wave 256 x 256 GEMM tile (csrc/gemm_w4.hip: one wave per SIMD, 128 x 128 per wave, accumulators in the
 * accumulator file, one workgroup barrier per K-tile): C (M,N) 16-bit = A (M,K) B (N,K)^T, N % 256 == 0, K % 64 == 0 */
int editor_probe_gemm_w4(const uint16_t* A, const uint16_t* B, uint16_t* C, int f16, int M, int N, int K, long lda, long ldb,
                         long ldc, int ablate /* 1 no LDS-DMA in the loop, 2 no MFMAs, 4 no fragment reads */, editor_stream_t stream);
/* libeditor_gemm_trace.so only (csrc/gemm_bf16.hip built with -DEDITOR_DEBUG_TRACE; tools/hetero_probe.py): ONE launch in which the
 * first nmem workgroups (a multiple of 8) stream memory - d = s0 + s1 over n4 float4 - and the others run the ping-pong kernel's body
 * on the tiles of C (M,N) bf16 = A (M,K) B (N,K)^T + bias: do an HBM-bound and an MFMA-bound role overlap inside a launch?
 * with_tiles = 0: the memory role alone; n4 = 0: the product alone (on the CUs the idle memory workgroups free at once). */
int editor_probe_gemm_hetero(const uint16_t* A, const uint16_t* B, void* C, int M, int N, int K, const float* bias, int with_tiles,
                             const float* s0, const float* s1, float* d, long n4, int nmem,
                             int unroll /* 4, 8, 16: 2 x unroll 16-byte loads in flight per thread */, editor_stream_t stream);
/* ... and with the real LayerNorm-backward role of editor_gemm_wgrad_group_ln (D = 768, bf16) in place of the plain stream */
int editor_probe_gemm_hetero_ln(const uint16_t* A, const uint16_t* B, void* C, int M, int N, int K, const float* bias, int with_tiles,
                                const uint16_t* ln_dy, const float* ln_x, const float* gamma, const float* mean, const float* rstd,
                                long ln_M, const float* dx_in, float* dx_out, float* partials, uint16_t* cast_out,
                                float* cast_partials, int nmem, editor_stream_t stream);
#ifdef __cplusplus
}
#endif
#endif
