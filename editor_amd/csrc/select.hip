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

// ---- round 4: 4 x 4 pixels per lane ------------------------------------------------------------------------------------
// The kernel above is VALU-bound, not shuffle- or HBM-bound: a lane owns 2 x 2 pixels and levels 2-4 are computed by EVERY lane
// of the 2x2 / 4x4 / 8x8 lane groups that share a coefficient block (4x, 16x, 64x redundant), plus 39 IEEE divisions by the
// modality count per lane - ~1 470 VALU instructions per 4 pixels, i.e. ~7.7 bytes per clock and CU at full issue rate, under
// the ~10 B/clk a CU can pull from HBM (measured: 0.47 of HBM warm, 0.29 inside the step).
// Here a lane owns 4 x 4 pixels of a plane (four 16-byte loads: full 256-byte row segments per wave instruction), levels 1 and 2
// are register-local, only levels 3 and 4 cross lanes (xor 1 / 4 and 2 / 8 inside a row of 16 lanes: DPP quad permutes and
// ds_swizzle, no address registers), a wave covers FOUR patches, and x / 3 is the three-instruction correctly-rounded form
// (q = RN(x * RN(1/3)); r = fma(-q, 3, x) exact; RN(q + r * RN(1/3)) = RN(x / 3): Markstein) with the IEEE sequence kept for
// the inputs it does not cover (denormals, infinities: a wave-uniform, never-taken branch on image data).  ~1 600 instructions
// per 16 pixels: 3.7x fewer per pixel, same arithmetic per element (same haar_fwd / haar_inv expressions, same summation
// orders) - bit-identical counts (tests/test_gpu_select.py against the reference's goldens and against the kernel above).
template <int XM> __device__ __forceinline__ float lane_xor16(float v)
{
    // value of lane (l ^ XM), XM < 16: inside a row of 16 lanes
    if constexpr (XM == 1) return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    else if constexpr (XM == 2) return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    else return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x1F | (XM << 10)));
}
template <int BX, int BY> __device__ __forceinline__ Quad gather_level16(float v, int lane)
{
    const float vx = lane_xor16<BX>(v), vy = lane_xor16<BY>(v), vxy = lane_xor16<BX | BY>(v);
    const bool cx = lane & BX, cy = lane & BY;
    const float x00 = cy ? (cx ? vxy : vy) : (cx ? vx : v);
    const float x01 = cy ? (cx ? vy : vxy) : (cx ? v : vx);
    const float x10 = cy ? (cx ? vx : v) : (cx ? vxy : vy);
    const float x11 = cy ? (cx ? v : vx) : (cx ? vy : vxy);
    return haar_fwd(x00, x01, x10, x11);
}
// x / DIV, correctly rounded.  `rare` collects the inputs the short form does not cover.
template <int DIV> __device__ __forceinline__ float div_small(float x, bool& rare)
{
    if constexpr (DIV == 4 || DIV == 2 || DIV == 1) return x * (1.0f / DIV);     // exact scaling (true division rounds the same way)
    else {
        constexpr float c = (float)DIV, rc = 1.0f / (float)DIV;                  // RN(1 / DIV)
        rare |= __builtin_amdgcn_classf(x, 0x010 | 0x080 | 0x004 | 0x200);   // +-denormal, +-infinity
        const float q = __fmul_rn(x, rc);
        const float r = __fmaf_rn(-q, c, x);
        return __fmaf_rn(r, rc, q);
    }
}

template <int NMOD, int NC>
__global__ __launch_bounds__(256) void freq_counts4_kernel(
    const float* __restrict__ m0, const float* __restrict__ m1, const float* __restrict__ m2, const float* __restrict__ m3,
    int B, int H, int W, int32_t* __restrict__ counts)
{
    const int lane = threadIdx.x & 63;
    const int tiles_x = W >> 4, tiles_y = H >> 4, ntile = tiles_x * tiles_y;
    const long total = (long)B * ntile;
    const long tile_raw = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);     // four patches per wave
    const long tile = tile_raw < total ? tile_raw : total - 1;                                // (a clamped slot never writes)
    const int b = (int)(tile / ntile), p = (int)(tile % ntile);
    const int ty = p / tiles_x, tx = p % tiles_x;
    const int lx = lane & 3, ly = (lane >> 2) & 3;                                            // 4 x 4 lanes per patch
    const int y0 = ty * 16 + ly * 4, x0 = tx * 16 + lx * 4;
    const float* mods[4] = {m0, m1, m2, m3};

    float sum[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) sum[i][j] = 0.f;
    bool rare = false;

    float4 cur[NMOD][4], nxt[NMOD][4];
    auto load = [&](float4 (&dst)[NMOD][4], int c) {
#pragma unroll
        for (int m = 0; m < NMOD; ++m) {
            const float* base = mods[m] + (((long)b * NC + c) * H + y0) * W + x0;
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[m][r] = *reinterpret_cast<const float4*>(base + (long)r * W);
        }
    };
    constexpr bool AHEAD = NMOD <= 3;                      // (four modalities: 128 more registers would leave one wave per SIMD)
    load(cur, 0);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if (AHEAD && c + 1 < NC) load(nxt, c + 1);         // the next channel's rows are in flight under this channel's arithmetic
        // ---- analysis, modality by modality, coefficients summed in modality order (Frequency.py:71-74)
        Quad a1[2][2], a2, a3, a4;                         // level-1 quads of the lane's four 2x2 blocks, levels 2-4
        float ll4 = 0.f;
#pragma unroll
        for (int m = 0; m < NMOD; ++m) {
            const float4 r0 = cur[m][0], r1 = cur[m][1], r2 = cur[m][2], r3 = cur[m][3];
            const Quad q00 = haar_fwd(r0.x, r0.y, r1.x, r1.y), q01 = haar_fwd(r0.z, r0.w, r1.z, r1.w);
            const Quad q10 = haar_fwd(r2.x, r2.y, r3.x, r3.y), q11 = haar_fwd(r2.z, r2.w, r3.z, r3.w);
            const Quad q2 = haar_fwd(q00.ll, q01.ll, q10.ll, q11.ll);
            const Quad q3 = gather_level16<1, 4>(q2.ll, lane);
            const Quad q4 = gather_level16<2, 8>(q3.ll, lane);
            if (m == 0) { a1[0][0] = q00; a1[0][1] = q01; a1[1][0] = q10; a1[1][1] = q11; a2 = q2; a3 = q3; a4 = q4; ll4 = q4.ll; }
            else {
                a1[0][0].lh += q00.lh; a1[0][0].hl += q00.hl; a1[0][0].hh += q00.hh;
                a1[0][1].lh += q01.lh; a1[0][1].hl += q01.hl; a1[0][1].hh += q01.hh;
                a1[1][0].lh += q10.lh; a1[1][0].hl += q10.hl; a1[1][0].hh += q10.hh;
                a1[1][1].lh += q11.lh; a1[1][1].hl += q11.hl; a1[1][1].hh += q11.hh;
                a2.lh += q2.lh; a2.hl += q2.hl; a2.hh += q2.hh;
                a3.lh += q3.lh; a3.hl += q3.hl; a3.hh += q3.hh;
                a4.lh += q4.lh; a4.hl += q4.hl; a4.hh += q4.hh;
                ll4 += q4.ll;
            }
        }
        // ---- mean over the modalities
        float* dv[22] = {&a1[0][0].lh, &a1[0][0].hl, &a1[0][0].hh, &a1[0][1].lh, &a1[0][1].hl, &a1[0][1].hh,
                         &a1[1][0].lh, &a1[1][0].hl, &a1[1][0].hh, &a1[1][1].lh, &a1[1][1].hl, &a1[1][1].hh,
                         &a2.lh, &a2.hl, &a2.hh, &a3.lh, &a3.hl, &a3.hh, &a4.lh, &a4.hl, &a4.hh, &ll4};
        float raw[22];
        bool rare_c = false;
#pragma unroll
        for (int i = 0; i < 22; ++i) { raw[i] = *dv[i]; *dv[i] = div_small<NMOD>(raw[i], rare_c); }
        if (__builtin_amdgcn_ballot_w64(rare_c) != 0ull) {           // denormal / infinite coefficient somewhere in the wave
            const float fnm = (float)NMOD;
#pragma unroll
            for (int i = 0; i < 22; ++i) *dv[i] = raw[i] / fnm;
        }
        a4.ll = ll4;
        // ---- synthesis: every lane rebuilds its own LL chain (no communication), then its 16 pixels
        a3.ll = haar_inv(a4, (lane >> 3) & 1, (lane >> 1) & 1);
        a2.ll = haar_inv(a3, (lane >> 2) & 1, lane & 1);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                a1[i][j].ll = haar_inv(a2, i, j);
#pragma unroll
                for (int ry = 0; ry < 2; ++ry)
#pragma unroll
                    for (int rx = 0; rx < 2; ++rx) sum[2 * i + ry][2 * j + rx] += haar_inv(a1[i][j], ry, rx);
            }
        if (c + 1 < NC) {
            if constexpr (AHEAD) {
#pragma unroll
                for (int m = 0; m < NMOD; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) cur[m][r] = nxt[m][r];
            } else {
                load(cur, c + 1);
            }
        }
    }
    // sign(mean over channels): torch.mean then .gt(0) (Frequency.py:44,54) - the division is kept (a sum that underflows to
    // zero when divided is not positive)
    float mean[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) mean[i] = div_small<NC>(sum[i >> 2][i & 3], rare);
    if (__builtin_amdgcn_ballot_w64(rare) != 0ull) {
#pragma unroll
        for (int i = 0; i < 16; ++i) mean[i] = sum[i >> 2][i & 3] / (float)NC;
    }
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) cnt += mean[i] > 0.f;
    // sum over the 16 lanes of the patch
    cnt += __builtin_amdgcn_mov_dpp(cnt, 0xB1, 0xF, 0xF, true);
    cnt += __builtin_amdgcn_mov_dpp(cnt, 0x4E, 0xF, 0xF, true);
    cnt += __builtin_amdgcn_ds_swizzle(cnt, 0x1F | (4 << 10));
    cnt += __builtin_amdgcn_ds_swizzle(cnt, 0x1F | (8 << 10));
    if ((lane & 15) == 0 && tile_raw < total) counts[tile] = cnt;
}

static int freq_counts_launch(const float* m0, const float* m1, const float* m2, const float* m3, int nmod, int B, int C, int H, int W,
                              int32_t* counts, hipStream_t stream, int variant = 0)
{
    const long tiles = (long)B * (H >> 4) * (W >> 4);
    const dim3 grid((unsigned)((tiles + 3) / 4)), block(256);
    const dim3 grid4((unsigned)((tiles + 15) / 16));            // 4 x 4 pixels per lane: four patches per wave, 16 per block
    const bool al16 = ((reinterpret_cast<uintptr_t>(m0) | reinterpret_cast<uintptr_t>(m1) | reinterpret_cast<uintptr_t>(m2) |
                        reinterpret_cast<uintptr_t>(m3)) & 15) == 0;
    if (nmod == 3 && C == 3 && al16 && variant == 0)
        hipLaunchKernelGGL((freq_counts4_kernel<3, 3>), grid4, block, 0, stream, m0, m1, m2, m3, B, H, W, counts);
    else if (nmod == 4 && C == 3 && al16 && variant == 0)
        hipLaunchKernelGGL((freq_counts4_kernel<4, 3>), grid4, block, 0, stream, m0, m1, m2, m3, B, H, W, counts);
    else if (nmod == 3 && C == 3)
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
//         pairs).
// The two algorithms are serial walks whose every step depends on the previous one, so a row is ONE lane's work - but lanes
// of a wave that walk different rows diverge at every data-dependent branch and the wave pays the union of their paths
// (rounds 1-3: 64 rows per wave, 222 us for the 128 frequency rows on TWO wavefronts of the whole chip, 91 us for the 4 608
// attention rows).  Here a row is one WAVE's: its 64 lanes load the row (coalesced), lane 0 walks it in LDS with each
// (value, index) pair packed into ONE 8-byte word (a move is one ds_read_b64 + one ds_write_b64), and the lanes share the
// mask write.  128 rows -> 128 waves on 32+ CUs, 4 608 rows -> 1 152 workgroups: the kernel takes one row's serial time.
// ------------------------------------------------------------------------------------------------
template <typename V> struct Pair { V v; uint32_t i; };
static_assert(sizeof(Pair<float>) == 8 && sizeof(Pair<int>) == 8, "one LDS word per pair");
template <typename V> struct Row {
    Pair<V>* q;
    __device__ __forceinline__ V v(int j) const { return q[j].v; }
    __device__ __forceinline__ uint32_t i(int j) const { return q[j].i; }
};

__device__ __forceinline__ bool before(float x, float y) { return (isnan(x) && !isnan(y)) || (x > y); }
__device__ __forceinline__ bool before(int x, int y) { return x > y; }

template <typename V> __device__ __forceinline__ Pair<V> get(Row<V>& r, int j) { return r.q[j]; }
template <typename V> __device__ __forceinline__ void put(Row<V>& r, int j, Pair<V> p) { r.q[j] = p; }
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
    // the scan: eight candidates are requested together (independent LDS reads), the heap top lives in a register and is
    // re-read only after a replacement
    V top = r.v(base);
    int i = middle;
    for (; i + 8 <= last; i += 8) {
        Pair<V> c[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) c[e] = get(r, base + i + e);
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (before(c[e].v, top)) {
                put(r, base + i + e, get(r, base));
                adjust_heap_(r, base, 0, middle, c[e]);
                top = r.v(base);
            }
    }
    for (; i < last; ++i)
        if (before(r.v(base + i), top)) {
            Pair<V> val = get(r, base + i);
            put(r, base + i, get(r, base));
            adjust_heap_(r, base, 0, middle, val);
            top = r.v(base);
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
__global__ __launch_bounds__(256) void topk_mask_kernel(const V* __restrict__ vals, int rows, int n, int k,
                                                        int group, uint8_t* __restrict__ mask)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long row = (long)blockIdx.x * (blockDim.x >> 6) + wave;
    if (row >= rows) return;                                     // (whole wave; no block-level barrier below)
    Pair<V>* q = reinterpret_cast<Pair<V>*>(smem) + (size_t)wave * n;
    for (int j = lane; j < n; j += 64) q[j] = Pair<V>{vals[row * n + j], (uint32_t)j};
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        Row<V> r{q};
        if ((long)k * 64 <= n) heap_select_(r, 0, k, n);             // std::partial_sort's selection half
        else {
            int lg = 0; for (int t = n; t > 1; t >>= 1) ++lg;
            introselect_(r, 0, k - 1, n, 2 * lg);                    // std::nth_element
        }
    }
    __builtin_amdgcn_wave_barrier();
    uint8_t* out = mask + (row / group) * (long)n;
    for (int j = lane; j < k; j += 64) out[q[j].i] = 1;          // benign same-value races across heads
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
    // one wave per row, 4 rows per workgroup while that fits 32 KiB of LDS (8 bytes per element), else one
    const int wpb = (size_t)4 * n * 8 <= 32 * 1024 ? 4 : 1;
    const size_t lds = (size_t)wpb * n * 8;
    hipLaunchKernelGGL(topk_mask_kernel<V>, dim3((rows + wpb - 1) / wpb), dim3(64 * wpb), lds, stream, vals, rows, n, k, group, mask);
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
