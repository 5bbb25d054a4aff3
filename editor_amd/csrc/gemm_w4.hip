// 256 x 256 x 64 bf16 / f16 GEMM tile on FOUR wavefronts - one per SIMD, each owning a 128 x 128 quadrant and the whole
// 512-register file (round 4; the structural alternative to the eight-wave ping-pong kernel of gemm_bf16.hip).
//
// Why: tools/gemm_bound_probe.py --ablate (profiles/r04_gemm_bound_probe.txt) takes the ping-pong kernel's K loop apart at the
// boost clock the step runs at: 1.43 us per K-tile against 0.86 us of matrix-core work - its sixteen workgroup barriers per
// K-tile alone cost 0.43 us, MFMAs + barriers 1.02 us, and the other wave group's fragment reads and LDS-DMA issue slow the
// multiplying group down further.  Here ONE instruction stream per SIMD interleaves its own fragment reads and LDS-DMA pieces
// between its MFMAs (no partner to hand the matrix core to, one workgroup barrier per K-tile), and a wave reads
// (128 + 128) x 64 operand elements per K-tile instead of (128 + 64): 128 KiB of LDS reads per K-tile and CU instead of 192.
//
//     C[m,n] = sum_k A[m,k] B[n,k]     A (M,K) and B (N,K) k-major 16-bit, fp32 accumulation on v_mfma_f32_16x16x32
//
// Same LDS images, fragment maps and per-element summation order as gemm_bf16_pp_kernel (K-tiles in order, k-step 0 then 1,
// one MFMA per 32-deep step): results are bit-identical to it.
#include "common.h"
#include "../../include/editor_debug.h"
#include <type_traits>

typedef __attribute__((ext_vector_type(8))) short short8_t;
typedef __attribute__((ext_vector_type(4))) float float4_t;
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));

namespace {

template <bool F16>
__device__ __forceinline__ float4_t mfma16(short8_t a, short8_t b, float4_t c)
{
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, a), __builtin_bit_cast(half8_t, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

// (inline asm: the compiler must not count these against the LDS-DMA in flight - it would drain vmcnt before every read)
template <int IMM>
__device__ __forceinline__ short8_t lds_rd128(uint32_t addr)
{
    short8_t v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(IMM));
    return v;
}

// accumulator fragment (I, J) of the wave's 128 x 128 quadrant: a[32 I + 4 J .. + 3], in place (D = C)
template <bool F16, int I, int J>
__device__ __forceinline__ void w4_mfma(short8_t b, short8_t a)
{
    constexpr int R = (I * 8 + J) * 4;
    if constexpr (F16)
        asm volatile("v_mfma_f32_16x16x32_f16 a[%0:%1], %2, %3, a[%0:%1]" : : "n"(R), "n"(R + 3), "v"(b), "v"(a));
    else
        asm volatile("v_mfma_f32_16x16x32_bf16 a[%0:%1], %2, %3, a[%0:%1]" : : "n"(R), "n"(R + 3), "v"(b), "v"(a));
}
template <int I, int J>
__device__ __forceinline__ float4_t w4_acc_read()
{
    constexpr int R = (I * 8 + J) * 4;
    float4_t v;
    asm volatile("v_accvgpr_read_b32 %0, a[%4]\n\tv_accvgpr_read_b32 %1, a[%5]\n\tv_accvgpr_read_b32 %2, a[%6]\n\tv_accvgpr_read_b32 %3, a[%7]"
                 : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]) : "n"(R), "n"(R + 1), "n"(R + 2), "n"(R + 3));
    return v;
}
#define W4_ACC_CLOBBERS "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255"
#define W4_ZERO_ACC "v_accvgpr_write_b32 a0, 0\n\tv_accvgpr_write_b32 a1, 0\n\tv_accvgpr_write_b32 a2, 0\n\tv_accvgpr_write_b32 a3, 0\n\tv_accvgpr_write_b32 a4, 0\n\tv_accvgpr_write_b32 a5, 0\n\tv_accvgpr_write_b32 a6, 0\n\tv_accvgpr_write_b32 a7, 0\n\tv_accvgpr_write_b32 a8, 0\n\tv_accvgpr_write_b32 a9, 0\n\tv_accvgpr_write_b32 a10, 0\n\tv_accvgpr_write_b32 a11, 0\n\tv_accvgpr_write_b32 a12, 0\n\tv_accvgpr_write_b32 a13, 0\n\tv_accvgpr_write_b32 a14, 0\n\tv_accvgpr_write_b32 a15, 0\n\tv_accvgpr_write_b32 a16, 0\n\tv_accvgpr_write_b32 a17, 0\n\tv_accvgpr_write_b32 a18, 0\n\tv_accvgpr_write_b32 a19, 0\n\tv_accvgpr_write_b32 a20, 0\n\tv_accvgpr_write_b32 a21, 0\n\tv_accvgpr_write_b32 a22, 0\n\tv_accvgpr_write_b32 a23, 0\n\tv_accvgpr_write_b32 a24, 0\n\tv_accvgpr_write_b32 a25, 0\n\tv_accvgpr_write_b32 a26, 0\n\tv_accvgpr_write_b32 a27, 0\n\tv_accvgpr_write_b32 a28, 0\n\tv_accvgpr_write_b32 a29, 0\n\tv_accvgpr_write_b32 a30, 0\n\tv_accvgpr_write_b32 a31, 0\n\tv_accvgpr_write_b32 a32, 0\n\tv_accvgpr_write_b32 a33, 0\n\tv_accvgpr_write_b32 a34, 0\n\tv_accvgpr_write_b32 a35, 0\n\tv_accvgpr_write_b32 a36, 0\n\tv_accvgpr_write_b32 a37, 0\n\tv_accvgpr_write_b32 a38, 0\n\tv_accvgpr_write_b32 a39, 0\n\tv_accvgpr_write_b32 a40, 0\n\tv_accvgpr_write_b32 a41, 0\n\tv_accvgpr_write_b32 a42, 0\n\tv_accvgpr_write_b32 a43, 0\n\tv_accvgpr_write_b32 a44, 0\n\tv_accvgpr_write_b32 a45, 0\n\tv_accvgpr_write_b32 a46, 0\n\tv_accvgpr_write_b32 a47, 0\n\tv_accvgpr_write_b32 a48, 0\n\tv_accvgpr_write_b32 a49, 0\n\tv_accvgpr_write_b32 a50, 0\n\tv_accvgpr_write_b32 a51, 0\n\tv_accvgpr_write_b32 a52, 0\n\tv_accvgpr_write_b32 a53, 0\n\tv_accvgpr_write_b32 a54, 0\n\tv_accvgpr_write_b32 a55, 0\n\tv_accvgpr_write_b32 a56, 0\n\tv_accvgpr_write_b32 a57, 0\n\tv_accvgpr_write_b32 a58, 0\n\tv_accvgpr_write_b32 a59, 0\n\tv_accvgpr_write_b32 a60, 0\n\tv_accvgpr_write_b32 a61, 0\n\tv_accvgpr_write_b32 a62, 0\n\tv_accvgpr_write_b32 a63, 0\n\tv_accvgpr_write_b32 a64, 0\n\tv_accvgpr_write_b32 a65, 0\n\tv_accvgpr_write_b32 a66, 0\n\tv_accvgpr_write_b32 a67, 0\n\tv_accvgpr_write_b32 a68, 0\n\tv_accvgpr_write_b32 a69, 0\n\tv_accvgpr_write_b32 a70, 0\n\tv_accvgpr_write_b32 a71, 0\n\tv_accvgpr_write_b32 a72, 0\n\tv_accvgpr_write_b32 a73, 0\n\tv_accvgpr_write_b32 a74, 0\n\tv_accvgpr_write_b32 a75, 0\n\tv_accvgpr_write_b32 a76, 0\n\tv_accvgpr_write_b32 a77, 0\n\tv_accvgpr_write_b32 a78, 0\n\tv_accvgpr_write_b32 a79, 0\n\tv_accvgpr_write_b32 a80, 0\n\tv_accvgpr_write_b32 a81, 0\n\tv_accvgpr_write_b32 a82, 0\n\tv_accvgpr_write_b32 a83, 0\n\tv_accvgpr_write_b32 a84, 0\n\tv_accvgpr_write_b32 a85, 0\n\tv_accvgpr_write_b32 a86, 0\n\tv_accvgpr_write_b32 a87, 0\n\tv_accvgpr_write_b32 a88, 0\n\tv_accvgpr_write_b32 a89, 0\n\tv_accvgpr_write_b32 a90, 0\n\tv_accvgpr_write_b32 a91, 0\n\tv_accvgpr_write_b32 a92, 0\n\tv_accvgpr_write_b32 a93, 0\n\tv_accvgpr_write_b32 a94, 0\n\tv_accvgpr_write_b32 a95, 0\n\tv_accvgpr_write_b32 a96, 0\n\tv_accvgpr_write_b32 a97, 0\n\tv_accvgpr_write_b32 a98, 0\n\tv_accvgpr_write_b32 a99, 0\n\tv_accvgpr_write_b32 a100, 0\n\tv_accvgpr_write_b32 a101, 0\n\tv_accvgpr_write_b32 a102, 0\n\tv_accvgpr_write_b32 a103, 0\n\tv_accvgpr_write_b32 a104, 0\n\tv_accvgpr_write_b32 a105, 0\n\tv_accvgpr_write_b32 a106, 0\n\tv_accvgpr_write_b32 a107, 0\n\tv_accvgpr_write_b32 a108, 0\n\tv_accvgpr_write_b32 a109, 0\n\tv_accvgpr_write_b32 a110, 0\n\tv_accvgpr_write_b32 a111, 0\n\tv_accvgpr_write_b32 a112, 0\n\tv_accvgpr_write_b32 a113, 0\n\tv_accvgpr_write_b32 a114, 0\n\tv_accvgpr_write_b32 a115, 0\n\tv_accvgpr_write_b32 a116, 0\n\tv_accvgpr_write_b32 a117, 0\n\tv_accvgpr_write_b32 a118, 0\n\tv_accvgpr_write_b32 a119, 0\n\tv_accvgpr_write_b32 a120, 0\n\tv_accvgpr_write_b32 a121, 0\n\tv_accvgpr_write_b32 a122, 0\n\tv_accvgpr_write_b32 a123, 0\n\tv_accvgpr_write_b32 a124, 0\n\tv_accvgpr_write_b32 a125, 0\n\tv_accvgpr_write_b32 a126, 0\n\tv_accvgpr_write_b32 a127, 0\n\tv_accvgpr_write_b32 a128, 0\n\tv_accvgpr_write_b32 a129, 0\n\tv_accvgpr_write_b32 a130, 0\n\tv_accvgpr_write_b32 a131, 0\n\tv_accvgpr_write_b32 a132, 0\n\tv_accvgpr_write_b32 a133, 0\n\tv_accvgpr_write_b32 a134, 0\n\tv_accvgpr_write_b32 a135, 0\n\tv_accvgpr_write_b32 a136, 0\n\tv_accvgpr_write_b32 a137, 0\n\tv_accvgpr_write_b32 a138, 0\n\tv_accvgpr_write_b32 a139, 0\n\tv_accvgpr_write_b32 a140, 0\n\tv_accvgpr_write_b32 a141, 0\n\tv_accvgpr_write_b32 a142, 0\n\tv_accvgpr_write_b32 a143, 0\n\tv_accvgpr_write_b32 a144, 0\n\tv_accvgpr_write_b32 a145, 0\n\tv_accvgpr_write_b32 a146, 0\n\tv_accvgpr_write_b32 a147, 0\n\tv_accvgpr_write_b32 a148, 0\n\tv_accvgpr_write_b32 a149, 0\n\tv_accvgpr_write_b32 a150, 0\n\tv_accvgpr_write_b32 a151, 0\n\tv_accvgpr_write_b32 a152, 0\n\tv_accvgpr_write_b32 a153, 0\n\tv_accvgpr_write_b32 a154, 0\n\tv_accvgpr_write_b32 a155, 0\n\tv_accvgpr_write_b32 a156, 0\n\tv_accvgpr_write_b32 a157, 0\n\tv_accvgpr_write_b32 a158, 0\n\tv_accvgpr_write_b32 a159, 0\n\tv_accvgpr_write_b32 a160, 0\n\tv_accvgpr_write_b32 a161, 0\n\tv_accvgpr_write_b32 a162, 0\n\tv_accvgpr_write_b32 a163, 0\n\tv_accvgpr_write_b32 a164, 0\n\tv_accvgpr_write_b32 a165, 0\n\tv_accvgpr_write_b32 a166, 0\n\tv_accvgpr_write_b32 a167, 0\n\tv_accvgpr_write_b32 a168, 0\n\tv_accvgpr_write_b32 a169, 0\n\tv_accvgpr_write_b32 a170, 0\n\tv_accvgpr_write_b32 a171, 0\n\tv_accvgpr_write_b32 a172, 0\n\tv_accvgpr_write_b32 a173, 0\n\tv_accvgpr_write_b32 a174, 0\n\tv_accvgpr_write_b32 a175, 0\n\tv_accvgpr_write_b32 a176, 0\n\tv_accvgpr_write_b32 a177, 0\n\tv_accvgpr_write_b32 a178, 0\n\tv_accvgpr_write_b32 a179, 0\n\tv_accvgpr_write_b32 a180, 0\n\tv_accvgpr_write_b32 a181, 0\n\tv_accvgpr_write_b32 a182, 0\n\tv_accvgpr_write_b32 a183, 0\n\tv_accvgpr_write_b32 a184, 0\n\tv_accvgpr_write_b32 a185, 0\n\tv_accvgpr_write_b32 a186, 0\n\tv_accvgpr_write_b32 a187, 0\n\tv_accvgpr_write_b32 a188, 0\n\tv_accvgpr_write_b32 a189, 0\n\tv_accvgpr_write_b32 a190, 0\n\tv_accvgpr_write_b32 a191, 0\n\tv_accvgpr_write_b32 a192, 0\n\tv_accvgpr_write_b32 a193, 0\n\tv_accvgpr_write_b32 a194, 0\n\tv_accvgpr_write_b32 a195, 0\n\tv_accvgpr_write_b32 a196, 0\n\tv_accvgpr_write_b32 a197, 0\n\tv_accvgpr_write_b32 a198, 0\n\tv_accvgpr_write_b32 a199, 0\n\tv_accvgpr_write_b32 a200, 0\n\tv_accvgpr_write_b32 a201, 0\n\tv_accvgpr_write_b32 a202, 0\n\tv_accvgpr_write_b32 a203, 0\n\tv_accvgpr_write_b32 a204, 0\n\tv_accvgpr_write_b32 a205, 0\n\tv_accvgpr_write_b32 a206, 0\n\tv_accvgpr_write_b32 a207, 0\n\tv_accvgpr_write_b32 a208, 0\n\tv_accvgpr_write_b32 a209, 0\n\tv_accvgpr_write_b32 a210, 0\n\tv_accvgpr_write_b32 a211, 0\n\tv_accvgpr_write_b32 a212, 0\n\tv_accvgpr_write_b32 a213, 0\n\tv_accvgpr_write_b32 a214, 0\n\tv_accvgpr_write_b32 a215, 0\n\tv_accvgpr_write_b32 a216, 0\n\tv_accvgpr_write_b32 a217, 0\n\tv_accvgpr_write_b32 a218, 0\n\tv_accvgpr_write_b32 a219, 0\n\tv_accvgpr_write_b32 a220, 0\n\tv_accvgpr_write_b32 a221, 0\n\tv_accvgpr_write_b32 a222, 0\n\tv_accvgpr_write_b32 a223, 0\n\tv_accvgpr_write_b32 a224, 0\n\tv_accvgpr_write_b32 a225, 0\n\tv_accvgpr_write_b32 a226, 0\n\tv_accvgpr_write_b32 a227, 0\n\tv_accvgpr_write_b32 a228, 0\n\tv_accvgpr_write_b32 a229, 0\n\tv_accvgpr_write_b32 a230, 0\n\tv_accvgpr_write_b32 a231, 0\n\tv_accvgpr_write_b32 a232, 0\n\tv_accvgpr_write_b32 a233, 0\n\tv_accvgpr_write_b32 a234, 0\n\tv_accvgpr_write_b32 a235, 0\n\tv_accvgpr_write_b32 a236, 0\n\tv_accvgpr_write_b32 a237, 0\n\tv_accvgpr_write_b32 a238, 0\n\tv_accvgpr_write_b32 a239, 0\n\tv_accvgpr_write_b32 a240, 0\n\tv_accvgpr_write_b32 a241, 0\n\tv_accvgpr_write_b32 a242, 0\n\tv_accvgpr_write_b32 a243, 0\n\tv_accvgpr_write_b32 a244, 0\n\tv_accvgpr_write_b32 a245, 0\n\tv_accvgpr_write_b32 a246, 0\n\tv_accvgpr_write_b32 a247, 0\n\tv_accvgpr_write_b32 a248, 0\n\tv_accvgpr_write_b32 a249, 0\n\tv_accvgpr_write_b32 a250, 0\n\tv_accvgpr_write_b32 a251, 0\n\tv_accvgpr_write_b32 a252, 0\n\tv_accvgpr_write_b32 a253, 0\n\tv_accvgpr_write_b32 a254, 0\n\tv_accvgpr_write_b32 a255, 0"

struct W4Args {
    const bf16_t* A; const bf16_t* B; bf16_t* C;
    int M, N, K;
    long lda, ldb, ldc;
    int tiles_m, tiles_n;
    int ablate;                             // bring-up: 1 = no LDS-DMA inside the K loop, 2 = no MFMAs, 4 = no fragment reads
};

constexpr int BK = 64;
constexpr int OPB = 256 * BK * 2;            // one operand tile of a stage: 256 rows x 128 bytes = 32 KiB
constexpr int STAGE = 2 * OPB;               // A | B
constexpr int LDS_BYTES = 2 * STAGE;         // two stages: 128 KiB

template <bool F16, int ABL = 0>            // ABL: bring-up ablations (compile-time: a run-time switch costs registers)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_w4_kernel(W4Args g)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // XCD-aware tile order (as the ping-pong kernel: groups of 4 tile rows x all tile columns stay on one XCD)
    const int nwg = g.tiles_m * g.tiles_n, bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    constexpr int GM = 4;
    const int grp = GM * g.tiles_n;
    const int gm0 = (wgid / grp) * GM, rem = wgid % grp;
    const int gsz = min(GM, g.tiles_m - gm0);
    const int tile_m = gm0 + rem % gsz, tile_n = rem / gsz;
    const int m0 = tile_m * 256, n0 = tile_n * 256;
    const int nk = g.K / BK;
    constexpr int abl = ABL;

    const int lane = threadIdx.x & 63;
    const int wu = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wu >> 1, wc = wu & 1;
    const int li = lane & 15, lg = lane >> 4;

    // The 256 x 256 fp32 accumulators live in the accumulator file, a[0:255], OWNED by the asm statements below (fragment (i, j)
    // = a[32 i + 4 j .. + 3]): left to the register allocator the 64 tiles wander between the two files (one v_accvgpr_write
    // quadruple in front of every MFMA, measured in the first build).  The clobber list makes the kernel descriptor allocate
    // them; the compiler must not touch AGPRs itself (audit: no v_accvgpr_* outside ASMSTART / ASMEND, vgpr_spill_count 0).
    asm volatile(W4_ZERO_ACC ::: W4_ACC_CLOBBERS);

    // fragment addresses: A rows wr*128 + i*16 + li, B rows wc*128 + j*16 + li; k-step s chunk s*4 + lg, XOR-swizzled by row & 7
    const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    uint32_t adA[2], adB[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        adA[s] = smem_base + (wr * 128 + li) * 128 + (((s * 4 + lg) ^ (li & 7)) << 4);
        adB[s] = smem_base + OPB + (wc * 128 + li) * 128 + (((s * 4 + lg) ^ (li & 7)) << 4);
    }
    // LDS-DMA: a K-tile is 64 pieces of 1 KiB (8 rows x 128 bytes); wave w stages pieces 16w .. 16w + 15, i.e. waves 0, 1 the A
    // tile and waves 2, 3 the B tile.  Lane l of a piece: row 8p + (l >> 3), source chunk (l & 7) ^ (row & 7).
    const bool isB = wu >= 2;
    const char* opbase = reinterpret_cast<const char*>(isB ? g.B : g.A);
    const long ldo = isB ? g.ldb : g.lda;
    const int r0 = isB ? n0 : m0, R = isB ? g.N : g.M;
    uint32_t vo[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int row = ((wu & 1) * 16 + j) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ (row & 7);
        vo[j] = (uint32_t)(((long)min(r0 + row, R - 1) * ldo + c * 8) * 2);
    }
    char* dst0 = smem + (isB ? OPB : 0) + (wu & 1) * 16 * 1024;
    // (raw buffer form: the per-lane offset of a piece is loop-invariant (VGPR), the K-tile position a scalar offset - no vector
    //  address arithmetic per piece, where the global_load_lds form needed two 64-bit adds in front of each of the 16)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(opbase), 0, 0x7fffffff, 0x00020000);
    auto dma = [&](int t, auto J) {                     // piece J of K-tile t
        constexpr int j = decltype(J)::value;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(dst0 + (t & 1) * STAGE + j * 1024), 16,
                                                 (int)vo[j], t * (BK * 2), 0, 0);
    };
    short8_t fa[8], fb[8], ga[8], gb[8];
#define W4_BAR() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define W4_LGKM0() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define W4_VM(n) do { asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

    // ---- prologue: K-tiles 0 and 1 on their way, tile 0 landed, its first k-step's fragments requested ---------------------
    static_for<0, 16>([&](auto J) { dma(0, J); });
    if (nk > 1) {
        static_for<0, 16>([&](auto J) { dma(1, J); });
        W4_VM(16);
    } else {
        W4_VM(0);
    }
    W4_BAR();
    static_for<0, 8>([&](auto I) { fa[decltype(I)::value] = lds_rd128<decltype(I)::value * 2048>(adA[0]); });
    static_for<0, 8>([&](auto I) { fb[decltype(I)::value] = lds_rd128<decltype(I)::value * 2048>(adB[0]); });

    // one K-tile out of stage ST (M2 / M1: K-tiles t+2 / t+1 exist - compile-time, the last two tiles are peeled).
    //   block 0   the 64 MFMAs of k-step 0, the 16 fragment reads of k-step 1 between them
    //   barrier A every wave has read the whole stage -> it may be refilled
    //   block 1a  32 MFMAs of k-step 1, the 16 LDS-DMA pieces of tile t+2 (into this stage) between them
    //   barrier B this wave's pieces of tile t+1 (issued 1 1/4 tiles ago) have landed (counted vmcnt: the 16 of t+2 stay in
    //             flight) -> tile t+1 is visible
    //   block 1b  32 MFMAs of k-step 1, the 16 fragment reads of tile t+1's k-step 0 between them
    auto ktile = [&](int t, auto ST, auto M2c, auto M1c) __attribute__((always_inline)) {
        constexpr int so = decltype(ST)::value * STAGE, sn = (1 - decltype(ST)::value) * STAGE;
        constexpr bool M2 = decltype(M2c)::value, M1 = decltype(M1c)::value;
        W4_LGKM0();
        static_for<0, 8>([&](auto I) {
            constexpr int i = decltype(I)::value;
            if constexpr (!(abl & 4)) {
                ga[i] = lds_rd128<i * 2048>(adA[1] + so);
                gb[i] = lds_rd128<i * 2048>(adB[1] + so);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!(abl & 2)) static_for<0, 8>([&](auto Jc) { w4_mfma<F16, i, decltype(Jc)::value>(fb[decltype(Jc)::value], fa[i]); });
            __builtin_amdgcn_sched_barrier(0);
        });
        W4_LGKM0();
        if constexpr (M2) W4_BAR();
        static_for<0, 4>([&](auto I) {
            constexpr int i = decltype(I)::value;
            static_for<0, 4>([&](auto P) {              // one piece in front of every pair of MFMAs
                constexpr int pq = decltype(P)::value;
                if constexpr (M2 && !(abl & 1)) dma(t + 2, std::integral_constant<int, 4 * i + pq>{});
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (!(abl & 2)) {
                    w4_mfma<F16, i, 2 * pq>(gb[2 * pq], ga[i]);
                    w4_mfma<F16, i, 2 * pq + 1>(gb[2 * pq + 1], ga[i]);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        });
        if constexpr (M1) {
            if constexpr (M2) W4_VM(16); else W4_VM(0);
            W4_BAR();
        }
        static_for<4, 8>([&](auto I) {
            constexpr int i = decltype(I)::value;
            if constexpr (M1 && !(abl & 4)) {
                fa[2 * (i - 4)] = lds_rd128<(2 * (i - 4)) * 2048>(adA[0] + sn);
                fa[2 * (i - 4) + 1] = lds_rd128<(2 * (i - 4) + 1) * 2048>(adA[0] + sn);
                fb[2 * (i - 4)] = lds_rd128<(2 * (i - 4)) * 2048>(adB[0] + sn);
                fb[2 * (i - 4) + 1] = lds_rd128<(2 * (i - 4) + 1) * 2048>(adB[0] + sn);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!(abl & 2)) static_for<0, 8>([&](auto Jc) { w4_mfma<F16, i, decltype(Jc)::value>(gb[decltype(Jc)::value], ga[i]); });
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    {
        using T_ = std::true_type; using F_ = std::false_type;
        using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
        int t = 0;
        for (; t + 3 < nk; t += 2) { ktile(t, S0{}, T_{}, T_{}); ktile(t + 1, S1{}, T_{}, T_{}); }
        // t is even, 1 .. 3 tiles left
        if (t + 3 == nk) { ktile(t, S0{}, T_{}, T_{}); ktile(t + 1, S1{}, F_{}, T_{}); ktile(t + 2, S0{}, F_{}, F_{}); }
        else if (t + 2 == nk) { ktile(t, S0{}, F_{}, T_{}); ktile(t + 1, S1{}, F_{}, F_{}); }
        else if (t + 1 == nk) { ktile(t, S0{}, F_{}, F_{}); }
    }
    W4_LGKM0();
#undef W4_BAR
#undef W4_LGKM0
#undef W4_VM
    // ---- probe epilogue: direct 8-byte stores (lane (i', g): row i*16 + i', columns j*16 + 4g .. + 3) ----------------------
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");           // (the last MFMAs' results before the accumulator reads below)
    static_for<0, 8>([&](auto Ic) {
        constexpr int i = decltype(Ic)::value;
        const int m = m0 + wr * 128 + i * 16 + li;
        static_for<0, 8>([&](auto Jc) {
            constexpr int j = decltype(Jc)::value;
            const float4_t v = w4_acc_read<i, j>();
            const int n = n0 + wc * 128 + j * 16 + lg * 4;
            if (m < g.M && n < g.N) {
                uint2 o;
                o.x = H16<F16>::pack2(v[0], v[1]);
                o.y = H16<F16>::pack2(v[2], v[3]);
                *reinterpret_cast<uint2*>(g.C + (long)m * g.ldc + n) = o;
            }
        });
    });
}

template <bool F16, int ABL>
int launch_w4(const W4Args& g, hipStream_t stream)
{
    hipError_t e = hipFuncSetAttribute((const void*)gemm_w4_kernel<F16, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((gemm_w4_kernel<F16, ABL>), dim3(g.tiles_m * g.tiles_n), dim3(256), LDS_BYTES, stream, g);
    EDITOR_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" int editor_probe_gemm_w4(const uint16_t* A, const uint16_t* B, uint16_t* C, int f16, int M, int N, int K, long lda,
                                    long ldb, long ldc, int ablate, hipStream_t stream)
{
    if (M < 1 || N < 256 || (N & 255) || K < 64 || (K & 63) || (lda & 7) || (ldb & 7) || (ldc & 3)) return (int)hipErrorInvalidValue;
    W4Args g{A, B, C, M, N, K, lda, ldb, ldc, (M + 255) / 256, N / 256, ablate};
    if (f16) return ablate ? (int)hipErrorInvalidValue : launch_w4<true, 0>(g, stream);
    switch (ablate) {
        case 0: return launch_w4<false, 0>(g, stream);
        case 1: return launch_w4<false, 1>(g, stream);
        case 2: return launch_w4<false, 2>(g, stream);
        case 3: return launch_w4<false, 3>(g, stream);
        case 4: return launch_w4<false, 4>(g, stream);
        case 5: return launch_w4<false, 5>(g, stream);
        case 6: return launch_w4<false, 6>(g, stream);
        case 7: return launch_w4<false, 7>(g, stream);
        default: return (int)hipErrorInvalidValue;
    }
}
