// Fused bf16 multi-head attention (forward + backward) for the EDITOR hot path, gfx950 / CDNA4.
// Restates Attention.forward (vit_pytorch.py:184-198) and AttentionMask.forward (:240-258) on packed
// qkv rows (B*T, 3*heads*64); sequences are short (T = 129 / 193 backbone, 387 joint HMA block), so the
// whole key range of one (sample, head) lives in LDS, every pass streams over 16x16 score tiles with the row
// log-sum-exp known in advance (no running rescale of the output), and the softmax output can be emitted
// (the backbone returns it, vit_pytorch.py:638-644; the rollout consumes it).
//
// One workgroup per (sample, head); each wavefront owns 16-row "own" tiles and sweeps all "other" rows:
//   FWD   own = queries, LDS = K,V      S^T = K Q^T -> softmax over keys -> O^T = V^T P^T
//   DQ    own = queries, LDS = K,V      P^T = exp(S^T - lse), dP^T = V dO^T, dS^T, dQ^T = K^T dS^T ; writes delta
//   DKV   own = keys,    LDS = Q,dO     P = exp(S - lse), dP = dO V^T, dS ; dV^T = dO^T P, dK^T = Q^T dS
// Every product is a v_mfma_f32_16x16x32_bf16 whose accumulator layout (lane = own column, 4 consecutive
// other-rows) is directly the next product's B operand after a bf16 pack - no cross-lane movement - using a
// permuted reduction order that the transposed operand (ds_read_b64_tr_b16 from the row-major LDS image)
// follows.  Outputs are written as 8-byte (4 x bf16) stores.
#include "common.h"
#include "../../include/editor_hip.h"
#include <type_traits>

#include "attn_common.h"
#include <string.h>

namespace {

// Round 6, the all-padding last tile of a 16 n + 1 token image (tools/attn_ab.py: same-box A/B against libeditor_attn_alt.so, which is
// this source with BOTH switches flipped; every output bit-identical either way):
#ifndef ATTN_ROLLOUT_SKIP
#define ATTN_ROLLOUT_SKIP 1            // rollout step: the query loop ends at the last populated tile, the one-hot first step reads ONE tile:
#endif                                 //   two layers 74.7 -> 66.2 us (T = 129), 124.4 -> 103.1 (T = 193) - shipped
#ifndef ATTN_PAIR_SKIP
#define ATTN_PAIR_SKIP 0               // q / kv passes: the last tile PAIR taken with one tile.  Less arithmetic, not less time: forward 77.5 ->
#endif                                 //   77.8 us, backward 223.3 -> 225.3 (T = 129), 381.1 -> 393.7 (T = 193) - measured, NOT shipped
#ifndef ATTN_UNROLL_BWD
#define ATTN_UNROLL_BWD 1              // dense backward passes with compile-time trip counts (measured: tools/attn_bench.py)
#endif

// Output tiles leave through LDS.  The accumulators hold a TRANSPOSED 16 x 64 tile (lane (i, g): element (row i, column
// dt*16 + 4g + r)); stored directly, one instruction writes 8-byte pieces of 16 different rows (32 contiguous bytes per row)
// and a 128-byte row needs four instructions - the kernels were bound by the number of partial-line requests, not by bytes
// (block size, occupancy and 20 % less arithmetic all left their time unchanged; without the global loads / stores of the
// own side they ran 15-25 % faster).  Here the tile is parked in 2.5 KiB of the wave's own LDS (rows padded to 160 bytes:
// the 8-byte column writes of 8 rows hit disjoint banks) and leaves as 16 bytes per lane, 8 lanes per full 128-byte row.
constexpr int STG_ROW = ROWB + 32, STG_BYTES = 16 * STG_ROW;      // (HD = 64: rows of 160 bytes)
constexpr int STG_LPR = CPR <= 4 ? 4 : (CPR <= 8 ? 8 : 16);      // lanes per staged row on the way out (16 bytes each)
template <bool F16>
__device__ __forceinline__ void store_tile(char* stg, const float4_t (&o)[ND], bf16_t* dst, long ldo, int nrows, int lane)
{
    const int li = lane & 15, lg = lane >> 4;
    if (!stg) {                                      // no LDS left for staging (608-token images), or forward (see launch_all)
        if (li < nrows) {
#pragma unroll
            for (int dt = 0; dt < ND; ++dt)
                *reinterpret_cast<uint2*>(dst + (long)li * ldo + dt * 16 + 4 * lg) = pack4<F16>(o[dt][0], o[dt][1], o[dt][2], o[dt][3]);
        }
        return;
    }
#pragma unroll
    for (int dt = 0; dt < ND; ++dt)
        *reinterpret_cast<uint2*>(stg + li * STG_ROW + dt * 32 + lg * 8) = pack4<F16>(o[dt][0], o[dt][1], o[dt][2], o[dt][3]);
    // (same wave wrote and reads: the compiler's lgkmcnt wait orders the two; no barrier)
    constexpr int RPP = 64 / STG_LPR;                // rows per pass (HD = 64: 8 rows of 8 lanes, two passes)
#pragma unroll
    for (int pass = 0; pass < 16 / RPP; ++pass) {
        const int row = pass * RPP + lane / STG_LPR, ch = lane % STG_LPR;
        if (CPR == STG_LPR || ch < CPR) {
            const uint4 v = *reinterpret_cast<const uint4*>(stg + row * STG_ROW + ch * 16);
            if (row < nrows) *reinterpret_cast<uint4*>(dst + (long)row * ldo + ch * 8) = v;
        }
    }
}


// ---------------------------------------------------------------------------------------------------------
// Column sums of the backward's output tiles (the qkv bias gradient), taken from the accumulators on their way out:
// editor_attention_bwd_colsum_*.  A 16-row tile sits as o[dt][r] = (row li, column dt*16 + 4*lg + r): the sum over its
// rows is a sum over the 16 lanes of a DPP row.  Reduce-scatter butterfly - partners i <-> 15-i (row_mirror), i <-> 7-i
// (row_half_mirror), i <-> 3-i and i <-> i^1 (quad_perm) - in which a lane keeps the half of the values its own lane bit
// selects and hands the other half over: 15 DPP additions + 30 selects for 16 values, and lane li ends up with the total of
// value li (dt = li / 4, r = li % 4).  Fixed order: deterministic.  (colsum_kernel read the 16-bit tensor back: 228 MB per
// backbone layer.)  CS_SLOTS accumulators per lane: values 0-15 in slot 0; head widths 32 / 96: the 8 values of the odd half
// in their own slot on lanes li < 8 (three butterfly steps and one plain exchange with lane li ^ 8).
// ---------------------------------------------------------------------------------------------------------
constexpr int CS_SLOTS = (ND * 4 + 15) / 16;
// (96-wide heads beyond 160 tokens: the dK / dV pass is at its register budget - the column sums are not built there)
template <int NT> constexpr bool kCols = HD <= 64 || NT <= 10;
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
constexpr int DPP_QUAD_XOR1 = 0xB1, DPP_QUAD_REV = 0x1B, DPP_ROW_MIRROR = 0x140, DPP_ROW_HALF_MIRROR = 0x141, DPP_ROW_ROR8 = 0x128;
// 8 values -> lane (li & 7) of each half row holds the total of value (li & 7) over the 8 lanes of its half
__device__ __forceinline__ float scatter8(const float (&u)[8], bool b2, bool b1, bool b0)
{
    float w[4], x[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = (b2 ? u[4 + j] : u[j]) + dpp_f<DPP_ROW_HALF_MIRROR>(b2 ? u[j] : u[4 + j]);
#pragma unroll
    for (int j = 0; j < 2; ++j) x[j] = (b1 ? w[2 + j] : w[j]) + dpp_f<DPP_QUAD_REV>(b1 ? w[j] : w[2 + j]);
    return (b0 ? x[1] : x[0]) + dpp_f<DPP_QUAD_XOR1>(b0 ? x[0] : x[1]);
}
// adds the column sums of the tile's first `nrows` rows to the lane's accumulators
__device__ __forceinline__ void tile_colsum(const float4_t (&o)[ND], int nrows, int lane, float (&mine)[CS_SLOTS])
{
    const int li = lane & 15;
    const bool hi = li & 8, b2 = li & 4, b1 = li & 2, b0 = li & 1;
    const bool live = li < nrows;
#pragma unroll
    for (int c = 0; c < CS_SLOTS; ++c) {
        if (c * 4 + 4 <= ND) {                      // sixteen values: o[4c .. 4c+3]
            float u[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float lo = live ? o[c * 4 + (j >> 2)][j & 3] : 0.f, up = live ? o[c * 4 + 2 + (j >> 2)][j & 3] : 0.f;
                u[j] = (hi ? up : lo) + dpp_f<DPP_ROW_MIRROR>(hi ? lo : up);
            }
            mine[c] += scatter8(u, b2, b1, b0);
        } else {                                    // eight values: o[4c], o[4c+1]
            float u[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) u[j] = live ? o[c * 4 + (j >> 2)][j & 3] : 0.f;
            const float t = scatter8(u, b2, b1, b0);
            mine[c] += t + dpp_f<DPP_ROW_ROR8>(t);                       // (lane i ^ 8 holds the same value index for the other half row)
        }
    }
}
// the workgroup's waves fold their accumulators through LDS (`red`: nw x HD floats, free once everyone has left the tile loop -
// the caller's barrier) in wave order; threads c < HD then write column c of `dst`
__device__ __forceinline__ void wave_colsum_park(const float (&mine)[CS_SLOTS], float* red, int lane)
{
    const int li = lane & 15, lg = lane >> 4;
#pragma unroll
    for (int c = 0; c < CS_SLOTS; ++c) {
        const bool full = c * 4 + 4 <= ND;
        if (full || li < 8) {
            const int idx = c * 16 + (full ? li : (li & 7));
            red[(idx >> 2) * 16 + 4 * lg + (idx & 3)] = mine[c];
        }
    }
}

struct AttnArgs {
    const bf16_t* qkv; const bf16_t* dout; const bf16_t* out_fwd;
    bf16_t* out; bf16_t* dqkv; float* probs; float* lse; float* delta;
    const uint8_t* mask;
    int T, heads; float scale;
    int ldp;                              // row stride (floats) of the probability output, multiple of 4
    const int* cu;                        // varlen: sequence b owns packed rows [cu[b], cu[b+1]); NULL = dense (b*T)
    long Mtot;                            // total packed rows (lse / delta are laid out [heads][Mtot])
    int stage_out;                        // outputs leave through per-wave LDS staging (store_tile); 0: direct 8-byte stores
    float* colparts;                      // backward: [B][3 * heads * HD] column sums of dqkv per sequence (or NULL), see tile_colsum
};

// ---------------------------------------------------------------------------------------------------------
// FWD and DQ passes (own = queries; LDS holds the K and V images).  Both STREAM over key tiles with the row
// log-sum-exp known in advance (FWD computes it in a first sweep over the score tiles - an online max/sum that
// costs 2 extra MFMAs per tile but no registers; DQ reads the value FWD saved), so a wave holds one score tile at
// a time: ~64-90 VGPRs, 4+ waves/SIMD, instead of a full 16xT score block (256 VGPRs, one workgroup per CU).
// ---------------------------------------------------------------------------------------------------------
// the four validity bits of key tile t (16 tiles per 64-bit word; NT <= 38 tiles -> three words: a two-word form
// silently aliased tiles >= 32, i.e. keys >= 512 of the 513..608-token sequences)
struct KeyBits { unsigned long long w[3]; };
__device__ __forceinline__ uint32_t nibble_of(const KeyBits& kb, int t)
{
    const unsigned long long word = t < 16 ? kb.w[0] : (t < 32 ? kb.w[1] : kb.w[2]);
    return (uint32_t)((word >> (4 * (t & 15))) & 0xfull);
}

// FULL: every sequence fills all NT key tiles (dense backbone calls): the tile loops then have compile-time trip counts,
// so the compiler batches the LDS fragment reads of many tiles ahead of the MFMAs instead of serialising
// "read -> wait -> MFMA" behind a branch per tile (that serialisation made the forward LDS-latency-bound).
// MASKED: a token mask is present (dense-masked HMA form); false for the backbone and the packed sequences, where a key
// is valid iff it lies inside the sequence - then validity is one integer compare per key against a per-lane limit
// instead of a per-lane bitmap (whose 64-bit shifts and SGPR traffic were ~1/3 of the forward's VALU + SALU work).
template <int NT, bool BWD, bool FULL, bool F16, bool MASKED>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(HD <= 64 ? 4 : 3, 8))) void attn_q_pass_kernel(AttnArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int Tp = NT * 16;
    char* kimg = smem;
    char* vimg = smem + Tp * ROWB;
    const int D = a.heads * HD;
    const int b = blockIdx.x / a.heads, hh = blockIdx.x % a.heads;
    const long ld = 3L * D;
    const long row0 = a.cu ? (long)a.cu[b] : (long)b * a.T;         // first packed row of this sequence
    const int T = a.cu ? a.cu[b + 1] - a.cu[b] : a.T;               // its length
    const int nt = FULL ? NT : min(NT, ((T + 31) >> 5) << 1);       // key tiles actually populated (even count)
    const bf16_t* qbase = a.qkv + row0 * ld + hh * HD;
    load_image(kimg, qbase + D, ld, T, nt * 16);
    load_image(vimg, qbase + 2 * D, ld, T, nt * 16);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    // the wave's first own-side (query) fragments are requested together with the images, the next tile's at the top of each
    // tile: their HBM latency used to be paid in front of every 16-query tile, after the images had already been waited for
    short8_t qnext[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) qnext[s] = frag_own(qbase, ld, w * 16, T, s, lane);
    images_ready();

    const int li = lane & 15, lg = lane >> 4;
    const uint8_t* mk = (MASKED && a.mask) ? a.mask + (long)b * T : nullptr;
    // validity of this lane's keys: key(t, r) = 16t + 4g + r
    KeyBits kb{{0ull, 0ull, 0ull}};
    if constexpr (MASKED) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = 16 * t + 4 * lg + r;
                const bool ok = key < T && (!mk || mk[key]);
                if (ok) kb.w[t >> 4] |= 1ull << (4 * (t & 15) + r);
            }
    }
    const int klim = T - 4 * lg;                                    // unmasked: key(t, r) valid  <=>  16t + r < klim
    auto vbits = [&](int t) -> uint32_t {
        if constexpr (MASKED) return nibble_of(kb, t);
        else {
            const int rem = klim - 16 * t;                           // valid keys of this lane in tile t (<= 0: none, >= 4: all)
            return rem >= 4 ? 0xfu : (rem <= 0 ? 0u : ((1u << rem) - 1u));
        }
    };
    const float sc = a.scale * kLog2e;
    // 16 n + 1 tokens (every ViT sequence: patches + the class token): the LAST key tile of the even-sized image lies entirely beyond
    // the sequence end - its scores are masked to -inf, its probabilities are exact zeros.  With ATTN_PAIR_SKIP the dense (FULL) forms
    // skip that half of their last pair of key tiles: two to four MFMAs, four exponentials and their selects per own tile, same bits
    // (round 6) - and the same time: a measured non-win, compiled out by default (the branches below fold away).
    const bool odd = ATTN_PAIR_SKIP && FULL && !MASKED && NT <= 14 && T <= 16 * (NT - 1);   // (NT <= 14: the backbone sequences; the longer forms spill with the extra copy of the pair)
    const long row_idx0 = (long)hh * a.Mtot + row0;                // lse / delta index of row 0
    const long prow0 = ((long)b * a.heads + hh) * a.T;             // probability rows (dense mode only)
    char* stg = a.stage_out ? smem + 2 * Tp * ROWB + w * STG_BYTES : nullptr;     // this wave's output staging (store_tile)
    float csum[CS_SLOTS];                                          // BWD: column sums of this wave's dQ tiles (a.colparts)
#pragma unroll
    for (int c = 0; c < CS_SLOTS; ++c) csum[c] = 0.f;

    for (int q0 = w * 16; q0 < T; q0 += nw * 16) {
        const int q = q0 + li;
        const bool qok = q < T && (!mk || mk[q]);
        short8_t qf[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) qf[s] = qnext[s];
        if (q0 + nw * 16 < T) {
#pragma unroll
            for (int s = 0; s < KS; ++s) qnext[s] = frag_own(qbase, ld, q0 + nw * 16, T, s, lane);
        }
        float lse;                                        // log2-sum-exp2 of the scaled scores of row q
        if constexpr (!BWD && NT <= 14) {
            // ---- short sequences (backbone: T = 129 / 193): the lane's 4*NT scores stay in registers, so S^T = K Q^T
            // is computed ONCE and each key costs one exponential (the streaming form below recomputes the scores in
            // its second sweep and pays the online-rescale exponentials on top: 9 instead of 4 per 4 keys) -----------
            float4_t sreg[NT];
            float m = -INFINITY;
            if constexpr (FULL) {
                // two key tiles at a time: four fragment reads in flight, then four MFMAs (two independent accumulators);
                // the scheduling barrier keeps the compiler from hoisting ALL reads up front (163 VGPRs, half the occupancy)
#pragma unroll
                for (int t = 0; t < NT; t += 2) {
                    float4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                    if (t == NT - 2 && odd) {                 // (wave-uniform) the pair's second tile holds no key
#pragma unroll
                        for (int s = 0; s < KS; ++s) acc0 = mfma16<F16>(frag_k(kimg, t * 16, s, lane), qf[s], acc0);
                    } else if constexpr (KS == 2) {
                        const short8_t a0 = frag_k(kimg, t * 16, 0, lane), a1 = frag_k(kimg, t * 16, 1, lane);
                        const short8_t b0 = frag_k(kimg, t * 16 + 16, 0, lane), b1 = frag_k(kimg, t * 16 + 16, 1, lane);
                        acc0 = mfma16<F16>(a0, qf[0], acc0);
                        acc1 = mfma16<F16>(b0, qf[0], acc1);
                        acc0 = mfma16<F16>(a1, qf[1], acc0);
                        acc1 = mfma16<F16>(b1, qf[1], acc1);
                    } else {                                  // other head widths: KS k-steps of 32
                        short8_t ka[KS], kb2[KS];
#pragma unroll
                        for (int s = 0; s < KS; ++s) { ka[s] = frag_k(kimg, t * 16, s, lane); kb2[s] = frag_k(kimg, t * 16 + 16, s, lane); }
#pragma unroll
                        for (int s = 0; s < KS; ++s) {
                            acc0 = mfma16<F16>(ka[s], qf[s], acc0);
                            acc1 = mfma16<F16>(kb2[s], qf[s], acc1);
                        }
                    }
                    // FULL launches are unmasked and populate all NT tiles (T > 16 (NT - 2)): only the last two tiles can
                    // hold keys beyond the sequence end; the others take no compare / select at all
                    const uint32_t vb0 = (t < NT - 2) ? 0xfu : vbits(t), vb1 = (t + 1 < NT - 2) ? 0xfu : vbits(t + 1);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        sreg[t][r] = (vb0 >> r) & 1u ? acc0[r] * sc : -INFINITY;
                        sreg[t + 1][r] = (vb1 >> r) & 1u ? acc1[r] * sc : -INFINITY;
                        m = fmaxf(m, fmaxf(sreg[t][r], sreg[t + 1][r]));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                sreg[t] = float4_t{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                if (t < nt) {
                    float4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < KS; ++s)
                        acc = mfma16<F16>(frag_k(kimg, t * 16, s, lane), qf[s], acc);
                    const uint32_t vb = vbits(t);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        sreg[t][r] = (vb >> r) & 1u ? acc[r] * sc : -INFINITY;
                        m = fmaxf(m, sreg[t][r]);
                    }
                }
            }
            }
            const float M = group_max(m);
            const float Ms = M > -INFINITY ? M : 0.f;                  // (fully masked row: every exponent is -inf -> 0)
            float l = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (FULL || t < nt) {
                    if (t == NT - 1 && odd) { sreg[t] = float4_t{0.f, 0.f, 0.f, 0.f}; continue; }   // exp2(-inf - Ms), l += 0
#pragma unroll
                    for (int r = 0; r < 4; ++r) { sreg[t][r] = __builtin_amdgcn_exp2f(sreg[t][r] - Ms); l += sreg[t][r]; }
                }
            const float L = group_sum(l);
            const bool live = qok && L > 0.f;
            const float inv = live ? __builtin_amdgcn_rcpf(L) : 0.f;
            lse = live ? M + __builtin_amdgcn_logf(L) : INFINITY;
            if (a.lse && lg == 0 && q < T) a.lse[row_idx0 + q] = lse;
            float* pr = (a.probs && q < T) ? a.probs + (prow0 + q) * a.ldp : nullptr;
            float4_t o[ND];
#pragma unroll
            for (int dt = 0; dt < ND; ++dt) o[dt] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s2 = 0; s2 < NT / 2; ++s2)
                if (FULL || 2 * s2 < nt) {
                    uint2 pk[2];
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        const int t = 2 * s2 + half;
                        const float p0 = sreg[t][0] * inv, p1 = sreg[t][1] * inv, p2 = sreg[t][2] * inv, p3 = sreg[t][3] * inv;
                        if (pr && 16 * t + 4 * lg < a.ldp) *reinterpret_cast<float4*>(pr + 16 * t + 4 * lg) = make_float4(p0, p1, p2, p3);
                        pk[half] = pack4<F16>(p0, p1, p2, p3);
                    }
                    const short8_t pf = join(pk[0], pk[1]);
#pragma unroll
                    for (int dt = 0; dt < ND; ++dt)
                        o[dt] = mfma16<F16>(frag_t(vimg, s2, dt, lane), pf, o[dt]);
                    if (FULL) __builtin_amdgcn_sched_barrier(0);
                }
            store_tile<F16>(stg, o, a.out + (row0 + q0) * D + hh * HD, D, T - q0, lane);
            continue;
        }
        if (!BWD) {
            // ---- sweep 1: online max / sum over this lane's keys, then across the 4 lane groups ------------------
            float m = -INFINITY, l = 0.f;
#pragma unroll 2
            for (int t = 0; t < nt; ++t) {
                float4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < KS; ++s)
                    acc = mfma16<F16>(frag_k(kimg, t * 16, s, lane), qf[s], acc);
                float sv[4], tm = -INFINITY;
                const uint32_t vb = vbits(t);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    sv[r] = (vb >> r) & 1u ? acc[r] * sc : -INFINITY;
                    tm = fmaxf(tm, sv[r]);
                }
                if (tm > m) { l *= __builtin_amdgcn_exp2f(m - tm); m = tm; }              // (m == -inf: l == 0, stays 0)
                if (m > -INFINITY) l += (__builtin_amdgcn_exp2f(sv[0] - m) + __builtin_amdgcn_exp2f(sv[1] - m)) + (__builtin_amdgcn_exp2f(sv[2] - m) + __builtin_amdgcn_exp2f(sv[3] - m));
            }
            const float M = group_max(m);
            const float L = group_sum(m > -INFINITY ? l * __builtin_amdgcn_exp2f(m - M) : 0.f);
            lse = (qok && L > 0.f) ? M + __builtin_amdgcn_logf(L) : INFINITY;          // +inf -> P == 0 (masked query)
            if (a.lse && lg == 0 && q < T) a.lse[row_idx0 + q] = lse;
        } else {
            lse = q < T ? a.lse[row_idx0 + q] : INFINITY;
        }

        float dl = 0.f;
        short8_t dof[KS];
        if (BWD) {   // delta[q] = sum_d dO[q,d] * O[q,d]
            const bf16_t* dobase = a.dout + row0 * D + hh * HD;
            const bf16_t* obase = a.out_fwd + row0 * D + hh * HD;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                dof[s] = frag_own(dobase, D, q0, T, s, lane);
                const short8_t of = frag_own(obase, D, q0, T, s, lane);
#pragma unroll
                for (int e = 0; e < 8; ++e) dl += H16<F16>::to_f32((uint16_t)dof[s][e]) * H16<F16>::to_f32((uint16_t)of[e]);
            }
            dl = group_sum(dl);
            if (lg == 0 && q < T) a.delta[row_idx0 + q] = dl;
        }
        // probability rows are padded to a multiple of 4 floats so that each lane's 4 consecutive keys are ONE 16-byte store
        float* pr = (!BWD && a.probs && q < T) ? a.probs + (prow0 + q) * a.ldp : nullptr;

        // ---- sweep 2: P^T tiles from lse; FWD: O^T += V^T P^T ; BWD: dS^T, dQ^T += K^T dS^T ----------------------
        float4_t o[ND];
#pragma unroll
        for (int dt = 0; dt < ND; ++dt) o[dt] = float4_t{0.f, 0.f, 0.f, 0.f};
        // one pair of key tiles; CHECK = false for pairs that cannot hold a key beyond the sequence end (unmasked sequences:
        // every pair but the last) - no compare / select per key there
        auto pair = [&](int s2, auto chk, auto one) {
            constexpr bool CHECK = decltype(chk)::value;
            constexpr int NH = decltype(one)::value ? 1 : 2;     // 1: the pair's second key tile is empty (`odd`)
            uint2 pk[2] = {make_uint2(0u, 0u), make_uint2(0u, 0u)};
#pragma unroll
            for (int half = 0; half < NH; ++half) {
                const int t = 2 * s2 + half;
                float4_t acc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    acc = mfma16<F16>(frag_k(kimg, t * 16, s, lane), qf[s], acc);
                    if (BWD) dp = mfma16<F16>(frag_k(vimg, t * 16, s, lane), dof[s], dp);
                }
                float pv[4];
                const uint32_t vb = CHECK ? vbits(t) : 0xfu;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    pv[r] = __builtin_amdgcn_exp2f((!CHECK || ((vb >> r) & 1u)) ? acc[r] * sc - lse : -INFINITY);
                if (!BWD) {
                    if (pr && 16 * t + 4 * lg < a.ldp)
                        *reinterpret_cast<float4*>(pr + 16 * t + 4 * lg) = make_float4(pv[0], pv[1], pv[2], pv[3]);
                    pk[half] = pack4<F16>(pv[0], pv[1], pv[2], pv[3]);
                } else {
                    // (the score scale is applied to the finished dQ tile, 16 multiplies per lane instead of 4 per key tile; for
                    // the power-of-two scales of 64-wide heads the result is bit-identical: rounding commutes with it)
                    pk[half] = pack4<F16>(pv[0] * (dp[0] - dl), pv[1] * (dp[1] - dl), pv[2] * (dp[2] - dl), pv[3] * (dp[3] - dl));
                }
            }
            const short8_t pf = join(pk[0], pk[1]);
            const char* timg = BWD ? kimg : vimg;
#pragma unroll
            for (int dt = 0; dt < ND; ++dt)
                o[dt] = mfma16<F16>(frag_t(timg, s2, dt, lane), pf, o[dt]);
        };
        const int npair = nt / 2;
        if constexpr (FULL && !MASKED && ATTN_UNROLL_BWD) {
            // dense backbone call: all NT tiles populated -> compile-time trip count, so that the fragment reads of the next
            // pair are issued under the MFMAs / exponentials of this one (the rolled loop serialises read -> wait -> MFMA)
#pragma unroll
            for (int s2 = 0; s2 < NT / 2 - 1; ++s2) { pair(s2, std::false_type{}, std::false_type{}); __builtin_amdgcn_sched_barrier(0); }
            if (odd) pair(NT / 2 - 1, std::true_type{}, std::true_type{});
            else pair(NT / 2 - 1, std::true_type{}, std::false_type{});
        } else if constexpr (MASKED) {
#pragma unroll 1
            for (int s2 = 0; s2 < npair; ++s2) pair(s2, std::true_type{}, std::false_type{});
        } else {
#pragma unroll 1
            for (int s2 = 0; s2 + 1 < npair; ++s2) pair(s2, std::false_type{}, std::false_type{});
            if (npair > 0) pair(npair - 1, std::true_type{}, std::false_type{});
        }
        if (BWD) {
#pragma unroll
            for (int dt = 0; dt < ND; ++dt) o[dt] *= a.scale;
            store_tile<F16>(stg, o, a.dqkv + (row0 + q0) * ld + hh * HD, ld, T - q0, lane);
            if (kCols<NT> && a.colparts) tile_colsum(o, 16, lane, csum);   // (query rows beyond the sequence end: P == 0, exact zeros)
        } else {
            store_tile<F16>(stg, o, a.out + (row0 + q0) * D + hh * HD, D, T - q0, lane);
        }
    }
    if constexpr (BWD && kCols<NT>) {
        if (a.colparts) {                                          // (workgroup-uniform)
            __syncthreads();                                       // everyone has left the tile loop: the K image is free
            float* red = reinterpret_cast<float*>(smem);
            wave_colsum_park(csum, red + w * HD, lane);
            __syncthreads();
            if (threadIdx.x < HD) {
                float t = 0.f;
                for (int i = 0; i < nw; ++i) t += red[i * HD + threadIdx.x];
                a.colparts[(long)b * 3 * D + hh * HD + threadIdx.x] = t;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// DKV pass (own = keys; LDS holds the Q and dO images + lse/delta of every query)
// ---------------------------------------------------------------------------------------------------------
template <int NT, bool F16, bool FULL>
// (amdgpu_waves_per_eu: with a lower bound of >= 2 waves per SIMD the register budget is <= 256 and LLVM selects the VGPR form of the
//  MFMAs; without it the results land in AGPRs and every score / dP tile costs four v_accvgpr_read_b32 in a VALU-bound kernel)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(HD <= 64 ? 3 : 2, 8))) void attn_kv_pass_kernel(AttnArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int Tp = NT * 16;
    char* qimg = smem;
    char* doimg = smem + Tp * ROWB;
    float* lse_s = reinterpret_cast<float*>(smem + 2 * Tp * ROWB);
    float* dl_s = lse_s + Tp;
    const int D = a.heads * HD;
    const int b = blockIdx.x / a.heads, hh = blockIdx.x % a.heads;
    const long ld = 3L * D;
    const long row0 = a.cu ? (long)a.cu[b] : (long)b * a.T;
    const int T = a.cu ? a.cu[b + 1] - a.cu[b] : a.T;
    const int nt = min(NT, ((T + 31) >> 5) << 1);
    const bf16_t* qbase = a.qkv + row0 * ld + hh * HD;
    load_image(qimg, qbase, ld, T, nt * 16);
    load_image(doimg, a.dout + row0 * D + hh * HD, D, T, nt * 16);
    const uint8_t* mk = a.mask ? a.mask + (long)b * T : nullptr;
    for (int t = threadIdx.x; t < Tp; t += blockDim.x) {
        const long idx = (long)hh * a.Mtot + row0 + t;
        const bool ok = t < T && (!mk || mk[t]);
        lse_s[t] = ok ? a.lse[idx] : INFINITY;
        dl_s[t] = ok ? a.delta[idx] : 0.f;
    }
    images_ready();

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const float sc = a.scale * kLog2e;
    // 16 n + 1 tokens: the last QUERY tile of the even-sized images is all padding (lse = +inf: P == 0 exactly) - the dense form skips
    // that half of its last pair (see attn_q_pass_kernel)
    const bool odd = ATTN_PAIR_SKIP && FULL && NT <= 14 && T <= 16 * (NT - 1);
    float csk[CS_SLOTS], csv[CS_SLOTS];                            // column sums of this wave's dK / dV tiles (a.colparts)
#pragma unroll
    for (int c = 0; c < CS_SLOTS; ++c) csk[c] = csv[c] = 0.f;

    for (int k0 = w * 16; k0 < T; k0 += nw * 16) {
        const int key = k0 + li;
        const bool kok = key < T && (!mk || mk[key]);
        // (requesting these one tile ahead, as the query pass does with its own fragments, was not faster: 143 us against 124-135)
        short8_t kf[KS], vf[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            kf[s] = frag_own(qbase + D, ld, k0, T, s, lane);
            vf[s] = frag_own(qbase + 2 * D, ld, k0, T, s, lane);
        }
        float4_t dv[ND], dk[ND];
#pragma unroll
        for (int dt = 0; dt < ND; ++dt) { dv[dt] = float4_t{0.f, 0.f, 0.f, 0.f}; dk[dt] = dv[dt]; }
        // k-major fragments of one pair of query tiles (8 x ds_read_b128)
        auto qload = [&](int u2, short8_t (&fq)[2][KS], short8_t (&fd)[2][KS]) {
#pragma unroll
            for (int half = 0; half < 2; ++half)
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    fq[half][s] = frag_k(qimg, (2 * u2 + half) * 16, s, lane);
                    fd[half][s] = frag_k(doimg, (2 * u2 + half) * 16, s, lane);
                }
        };
        auto qcompute = [&](int u2, short8_t (&fq)[2][KS], short8_t (&fd)[2][KS], auto one) {
            constexpr int NH = decltype(one)::value ? 1 : 2;     // 1: the pair's second query tile lies beyond the sequence end (`odd`)
            uint2 pk[2] = {make_uint2(0u, 0u), make_uint2(0u, 0u)}, dsk[2] = {make_uint2(0u, 0u), make_uint2(0u, 0u)};
#pragma unroll
            for (int half = 0; half < NH; ++half) {
                const int u = 2 * u2 + half;
                float4_t s_ = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    s_ = mfma16<F16>(fq[half][s], kf[s], s_);
                    dp = mfma16<F16>(fd[half][s], vf[s], dp);
                }
                // this lane's four queries are consecutive: one 16-byte LDS read each for lse and delta, and the mask goes
                // into the exponent (exp2(-inf) = 0) - the per-element "cond ? exp2f(..) : 0" form compiled into four
                // branch blocks, each with its own scalar LDS read and wait
                const float4 l4 = *reinterpret_cast<const float4*>(lse_s + 16 * u + 4 * lg);
                const float4 d4 = *reinterpret_cast<const float4*>(dl_s + 16 * u + 4 * lg);
                const float lq[4] = {l4.x, l4.y, l4.z, l4.w}, dq[4] = {d4.x, d4.y, d4.z, d4.w};
                float pv[4], dsv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // FULL (dense, unmasked): a key beyond the sequence end needs no select - its dK / dV rows are never
                    // stored and its K / V fragments are zeros (finite scores).  The score scale goes onto the finished dK tile.
                    const float pp = __builtin_amdgcn_exp2f((FULL || kok) ? s_[r] * sc - lq[r] : -INFINITY);   // lse == +inf -> 0
                    pv[r] = pp;
                    dsv[r] = pp * (dp[r] - dq[r]);
                }
                pk[half] = pack4<F16>(pv[0], pv[1], pv[2], pv[3]);
                dsk[half] = pack4<F16>(dsv[0], dsv[1], dsv[2], dsv[3]);
            }
            const short8_t pf = join(pk[0], pk[1]), df = join(dsk[0], dsk[1]);
#pragma unroll
            for (int dt = 0; dt < ND; ++dt) {
                dv[dt] = mfma16<F16>(frag_t(doimg, u2, dt, lane), pf, dv[dt]);
                dk[dt] = mfma16<F16>(frag_t(qimg, u2, dt, lane), df, dk[dt]);
            }
        };
        if constexpr (FULL && ATTN_UNROLL_BWD) {
            // dense backbone call: compile-time trip count and the fragments double-buffered in registers - the reads of
            // pair u2+1 are in flight under the MFMAs / exponentials of pair u2 (the rolled loop: read -> wait -> MFMA)
            short8_t fq[2][2][KS], fd[2][2][KS];
            qload(0, fq[0], fd[0]);
#pragma unroll
            for (int u2 = 0; u2 < NT / 2; ++u2) {
                if (u2 + 1 < NT / 2) qload(u2 + 1, fq[(u2 + 1) & 1], fd[(u2 + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                if (u2 == NT / 2 - 1 && odd) qcompute(u2, fq[u2 & 1], fd[u2 & 1], std::true_type{});
                else qcompute(u2, fq[u2 & 1], fd[u2 & 1], std::false_type{});
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll 1
            for (int u2 = 0; u2 < nt / 2; ++u2) {
                // the eight k-major fragments of this pair of query tiles are requested together, ahead of the eight MFMAs
                // (left to itself the compiler reads one fragment ahead: read -> wait -> MFMA, LDS latency each time)
                short8_t fq[2][KS], fd[2][KS];
                qload(u2, fq, fd);
                __builtin_amdgcn_sched_barrier(0);
                qcompute(u2, fq, fd, std::false_type{});
            }
        }
        {
            char* stg = a.stage_out ? smem + 2 * Tp * ROWB + 2 * Tp * sizeof(float) + w * STG_BYTES : nullptr;
            bf16_t* kdst = a.dqkv + (row0 + k0) * ld + D + hh * HD;
#pragma unroll
            for (int dt = 0; dt < ND; ++dt) dk[dt] *= a.scale;
            store_tile<F16>(stg, dk, kdst, ld, T - k0, lane);
            store_tile<F16>(stg, dv, kdst + D, ld, T - k0, lane);
            if (kCols<NT> && a.colparts) {
                // FULL: the tile rows of keys beyond the sequence end hold finite garbage that is never stored - dropped here (last
                // tile only); otherwise they are exact zeros (P == 0)
                if (FULL && T - k0 < 16) { tile_colsum(dk, T - k0, lane, csk); tile_colsum(dv, T - k0, lane, csv); }
                else { tile_colsum(dk, 16, lane, csk); tile_colsum(dv, 16, lane, csv); }
            }
        }
    }
    if (kCols<NT> && a.colparts) {
        __syncthreads();                                           // the Q image is free
        float* red = reinterpret_cast<float*>(smem);
        wave_colsum_park(csk, red + (w * 2) * HD, lane);
        wave_colsum_park(csv, red + (w * 2 + 1) * HD, lane);
        __syncthreads();
        if (threadIdx.x < 2 * HD) {
            const int which = threadIdx.x / HD, c = threadIdx.x % HD;
            float t = 0.f;
            for (int i = 0; i < nw; ++i) t += red[(i * 2 + which) * HD + c];
            a.colparts[(long)b * 3 * D + (1 + which) * D + hh * HD + c] = t;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Rollout step (SFTS.py:150-153, row-vector form):  r_out[k] = sum_q r_in[q] * P_l[q,k]  for one layer l, with the
// probabilities RECOMPUTED from that layer's saved q/k and row log-sum-exps (P[q,k] = exp2(s[q,k] - lse[q]), exactly
// the values the forward produced) instead of read from a materialised (3B,h,T,T) tensor: the forward then skips its
// 314 MB-per-layer probability output (201 -> 131 us) and the 3.7 GB buffer disappears.  Same skeleton as the DKV
// pass: own = keys, LDS holds the Q image, lse and r_in; the weighted sum over queries is fp32 VALU work.
// r_in == NULL: one-hot CLS row (the first step, l = L-1).  final: write r_out[1:] to a (BH, T-1) score tensor.
// ---------------------------------------------------------------------------------------------------------
template <int NT, bool F16>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void attn_rollout_step_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ lse,
    const float* __restrict__ r_in, int T, int heads, float scale, long Mtot, float* __restrict__ r_out, int final_step)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int Tp = NT * 16;
    char* qimg = smem;
    float* lse_s = reinterpret_cast<float*>(smem + Tp * ROWB);
    float* w_s = lse_s + Tp;
    const int D = heads * HD;
    const int b = blockIdx.x / heads, hh = blockIdx.x % heads;
    const long ld = 3L * D;
    const long row0 = (long)b * T;
    const int nt = min(NT, ((T + 31) >> 5) << 1);
    // query tiles that contribute: the populated ones (the image is an even number of tiles) - and only the FIRST when the incoming
    // vector is the one-hot class-token row (r_in == NULL, the first step of the rollout): every other query's weight is an exact zero
    const int nu = !ATTN_ROLLOUT_SKIP ? nt : (r_in ? min(nt, (T + 15) >> 4) : 1);
    const bf16_t* qbase = qkv + row0 * ld + hh * HD;
    load_image(qimg, qbase, ld, T, nt * 16);
    for (int t = threadIdx.x; t < Tp; t += blockDim.x) {
        lse_s[t] = t < T ? lse[(long)hh * Mtot + row0 + t] : INFINITY;
        w_s[t] = t < T ? (r_in ? r_in[(long)blockIdx.x * T + t] : (t == 0 ? 1.f : 0.f)) : 0.f;
    }
    // the wave's own key fragments come straight from global memory: the first tile's are requested BEFORE the wait for the Q
    // image and every later tile's one tile ahead, under the arithmetic of the current one (round 4: they used to be requested
    // at the top of each trip - two to three exposed HBM latencies per wave on top of the image's)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    short8_t kn[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) kn[s] = frag_own(qbase + D, ld, w * 16, T, s, lane);
    images_ready();
    const int li = lane & 15, lg = lane >> 4;
    const float sc = scale * kLog2e;
    for (int k0 = w * 16; k0 < T; k0 += nw * 16) {
        const int key = k0 + li;
        short8_t kf[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) kf[s] = kn[s];
        if (k0 + nw * 16 < T) {
#pragma unroll
            for (int s = 0; s < KS; ++s) kn[s] = frag_own(qbase + D, ld, k0 + nw * 16, T, s, lane);
        }
        float acc = 0.f;
#pragma unroll 2
        for (int u = 0; u < nu; ++u) {
            float4_t s_ = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS; ++s)
                s_ = mfma16<F16>(frag_k(qimg, u * 16, s, lane), kf[s], s_);
            const float4 l4 = *reinterpret_cast<const float4*>(lse_s + 16 * u + 4 * lg);
            const float4 w4 = *reinterpret_cast<const float4*>(w_s + 16 * u + 4 * lg);
            acc = fmaf(__builtin_amdgcn_exp2f(s_[0] * sc - l4.x), w4.x, acc);               // lse = +inf (pad rows) -> 0
            acc = fmaf(__builtin_amdgcn_exp2f(s_[1] * sc - l4.y), w4.y, acc);
            acc = fmaf(__builtin_amdgcn_exp2f(s_[2] * sc - l4.z), w4.z, acc);
            acc = fmaf(__builtin_amdgcn_exp2f(s_[3] * sc - l4.w), w4.w, acc);
        }
        acc = group_sum(acc);
        if (lg == 0 && key < T) {
            if (final_step) { if (key >= 1) r_out[(long)blockIdx.x * (T - 1) + key - 1] = acc; }
            else r_out[(long)blockIdx.x * T + key] = acc;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Long sequences (T > 608: the joint HMA block of the 4-modal 512-token configuration, up to 4 x 513 = 2052 tokens).
// The key range no longer fits the CU's LDS, so a workgroup owns 64 "own" rows (one 16-row tile per wave) and streams the
// "other" side through LDS in chunks of 256 rows (K,V or Q,dO: 64 KiB -> two workgroups per CU); grid = (B*heads,
// ceil(Tmax/64)).  Same tile algebra as the whole-sequence kernels above; key validity (sequence end, optional token mask
// of the dense-masked form) is computed per tile.  FWD sweeps the chunks twice (row statistics, then
// P V with the final log-sum-exp - no running rescale of the output accumulators); DQ and DKV sweep once.
// ---------------------------------------------------------------------------------------------------------
constexpr int LCH = 256;

template <bool BWD, bool F16>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void attn_q_long_kernel(AttnArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* kimg = smem;
    char* vimg = smem + LCH * ROWB;
    const int D = a.heads * HD;
    const int b = blockIdx.x / a.heads, hh = blockIdx.x % a.heads;
    const long ld = 3L * D;
    const long row0 = a.cu ? (long)a.cu[b] : (long)b * a.T;
    const int T = a.cu ? a.cu[b + 1] - a.cu[b] : a.T;
    const int qb0 = blockIdx.y * 64;
    if (qb0 >= T) return;                                             // (whole workgroup: before any barrier)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const bf16_t* qbase = a.qkv + row0 * ld + hh * HD;
    const uint8_t* mk = a.mask ? a.mask + (long)b * T : nullptr;     // dense-masked form only (packed rows are all live)
    const int q0 = qb0 + w * 16, q = q0 + li;
    const bool qin = q < T;
    const bool qok = qin && (!mk || mk[q]);
    short8_t qf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) qf[s] = frag_own(qbase, ld, q0, T, s, lane);
    const float sc = a.scale * kLog2e;
    const long row_idx0 = (long)hh * a.Mtot + row0;
    float lse;
    if (!BWD) {
        float m = -INFINITY, l = 0.f;
        for (int c0 = 0; c0 < T; c0 += LCH) {
            const int len = min(LCH, T - c0), ntc = ((len + 31) >> 5) << 1;
            __syncthreads();
            load_image(kimg, qbase + D + (long)c0 * ld, ld, len, ntc * 16);
            images_ready();
#pragma unroll 2
            for (int t = 0; t < ntc; ++t) {
                float4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < KS; ++s) acc = mfma16<F16>(frag_k(kimg, t * 16, s, lane), qf[s], acc);
                float sv[4], tm = -INFINITY;
                const int key0 = c0 + 16 * t + 4 * lg;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    sv[r] = (key0 + r < T && (!mk || mk[key0 + r])) ? acc[r] * sc : -INFINITY;
                    tm = fmaxf(tm, sv[r]);
                }
                if (tm > m) { l *= __builtin_amdgcn_exp2f(m - tm); m = tm; }
                if (m > -INFINITY) l += (__builtin_amdgcn_exp2f(sv[0] - m) + __builtin_amdgcn_exp2f(sv[1] - m)) + (__builtin_amdgcn_exp2f(sv[2] - m) + __builtin_amdgcn_exp2f(sv[3] - m));
            }
        }
        const float M = group_max(m);
        const float L = group_sum(m > -INFINITY ? l * __builtin_amdgcn_exp2f(m - M) : 0.f);
        lse = (qok && L > 0.f) ? M + __builtin_amdgcn_logf(L) : INFINITY;
        if (a.lse && lg == 0 && qin) a.lse[row_idx0 + q] = lse;
    } else {
        lse = qin ? a.lse[row_idx0 + q] : INFINITY;
    }
    float dl = 0.f;
    short8_t dof[KS];
    if (BWD) {
        const bf16_t* dobase = a.dout + row0 * D + hh * HD;
        const bf16_t* obase = a.out_fwd + row0 * D + hh * HD;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            dof[s] = frag_own(dobase, D, q0, T, s, lane);
            const short8_t of = frag_own(obase, D, q0, T, s, lane);
#pragma unroll
            for (int e = 0; e < 8; ++e) dl += H16<F16>::to_f32((uint16_t)dof[s][e]) * H16<F16>::to_f32((uint16_t)of[e]);
        }
        dl = group_sum(dl);
        if (lg == 0 && qin) a.delta[row_idx0 + q] = dl;
    }
    float4_t o[ND];
#pragma unroll
    for (int dt = 0; dt < ND; ++dt) o[dt] = float4_t{0.f, 0.f, 0.f, 0.f};
    for (int c0 = 0; c0 < T; c0 += LCH) {
        const int len = min(LCH, T - c0), ntc = ((len + 31) >> 5) << 1;
        __syncthreads();
        load_image(kimg, qbase + D + (long)c0 * ld, ld, len, ntc * 16);
        load_image(vimg, qbase + 2 * D + (long)c0 * ld, ld, len, ntc * 16);
        images_ready();
#pragma unroll 1
        for (int s2 = 0; s2 < ntc / 2; ++s2) {
            uint2 pk[2];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int t = 2 * s2 + half;
                float4_t acc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    acc = mfma16<F16>(frag_k(kimg, t * 16, s, lane), qf[s], acc);
                    if (BWD) dp = mfma16<F16>(frag_k(vimg, t * 16, s, lane), dof[s], dp);
                }
                float pv[4];
                const int key0 = c0 + 16 * t + 4 * lg;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    pv[r] = __builtin_amdgcn_exp2f((key0 + r < T && (!mk || mk[key0 + r])) ? acc[r] * sc - lse : -INFINITY);
                if (!BWD) pk[half] = pack4<F16>(pv[0], pv[1], pv[2], pv[3]);
                else pk[half] = pack4<F16>(pv[0] * (dp[0] - dl) * a.scale, pv[1] * (dp[1] - dl) * a.scale,
                                           pv[2] * (dp[2] - dl) * a.scale, pv[3] * (dp[3] - dl) * a.scale);
            }
            const short8_t pf = join(pk[0], pk[1]);
            const char* timg = BWD ? kimg : vimg;
#pragma unroll
            for (int dt = 0; dt < ND; ++dt) o[dt] = mfma16<F16>(frag_t(timg, s2, dt, lane), pf, o[dt]);
        }
    }
    if (qin) {
        bf16_t* orow = BWD ? a.dqkv + (row0 + q) * ld + hh * HD + 4 * lg : a.out + (row0 + q) * D + hh * HD + 4 * lg;
#pragma unroll
        for (int dt = 0; dt < ND; ++dt)
            *reinterpret_cast<uint2*>(orow + dt * 16) = pack4<F16>(o[dt][0], o[dt][1], o[dt][2], o[dt][3]);
    }
}

template <bool F16>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 8))) void attn_kv_long_kernel(AttnArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* qimg = smem;
    char* doimg = smem + LCH * ROWB;
    float* lse_s = reinterpret_cast<float*>(smem + 2 * LCH * ROWB);
    float* dl_s = lse_s + LCH;
    const int D = a.heads * HD;
    const int b = blockIdx.x / a.heads, hh = blockIdx.x % a.heads;
    const long ld = 3L * D;
    const long row0 = a.cu ? (long)a.cu[b] : (long)b * a.T;
    const int T = a.cu ? a.cu[b + 1] - a.cu[b] : a.T;
    const int kb0 = blockIdx.y * 64;
    if (kb0 >= T) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const bf16_t* qbase = a.qkv + row0 * ld + hh * HD;
    const uint8_t* mk = a.mask ? a.mask + (long)b * T : nullptr;
    const int k0 = kb0 + w * 16, key = k0 + li;
    const bool kin = key < T;
    const bool kok = kin && (!mk || mk[key]);
    const float sc = a.scale * kLog2e;
    short8_t kf[KS], vf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        kf[s] = frag_own(qbase + D, ld, k0, T, s, lane);
        vf[s] = frag_own(qbase + 2 * D, ld, k0, T, s, lane);
    }
    float4_t dv[ND], dk[ND];
#pragma unroll
    for (int dt = 0; dt < ND; ++dt) { dv[dt] = float4_t{0.f, 0.f, 0.f, 0.f}; dk[dt] = dv[dt]; }
    for (int c0 = 0; c0 < T; c0 += LCH) {
        const int len = min(LCH, T - c0), ntc = ((len + 31) >> 5) << 1;
        __syncthreads();
        load_image(qimg, qbase + (long)c0 * ld, ld, len, ntc * 16);
        load_image(doimg, a.dout + (row0 + c0) * D + hh * HD, D, len, ntc * 16);
        for (int t = threadIdx.x; t < ntc * 16; t += blockDim.x) {
            const long idx = (long)hh * a.Mtot + row0 + c0 + t;
            const bool ok = t < len && (!mk || mk[c0 + t]);
            lse_s[t] = ok ? a.lse[idx] : INFINITY;
            dl_s[t] = ok ? a.delta[idx] : 0.f;
        }
        images_ready();
#pragma unroll 1
        for (int u2 = 0; u2 < ntc / 2; ++u2) {
            uint2 pk[2], dsk[2];
            short8_t fq[2][KS], fd[2][KS];
#pragma unroll
            for (int half = 0; half < 2; ++half)
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    fq[half][s] = frag_k(qimg, (2 * u2 + half) * 16, s, lane);
                    fd[half][s] = frag_k(doimg, (2 * u2 + half) * 16, s, lane);
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int u = 2 * u2 + half;
                float4_t s_ = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    s_ = mfma16<F16>(fq[half][s], kf[s], s_);
                    dp = mfma16<F16>(fd[half][s], vf[s], dp);
                }
                const float4 l4 = *reinterpret_cast<const float4*>(lse_s + 16 * u + 4 * lg);
                const float4 d4 = *reinterpret_cast<const float4*>(dl_s + 16 * u + 4 * lg);
                const float lq[4] = {l4.x, l4.y, l4.z, l4.w}, dq[4] = {d4.x, d4.y, d4.z, d4.w};
                float pv[4], dsv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pp = __builtin_amdgcn_exp2f(kok ? s_[r] * sc - lq[r] : -INFINITY);
                    pv[r] = pp;
                    dsv[r] = pp * (dp[r] - dq[r]) * a.scale;
                }
                pk[half] = pack4<F16>(pv[0], pv[1], pv[2], pv[3]);
                dsk[half] = pack4<F16>(dsv[0], dsv[1], dsv[2], dsv[3]);
            }
            const short8_t pf = join(pk[0], pk[1]), df = join(dsk[0], dsk[1]);
#pragma unroll
            for (int dt = 0; dt < ND; ++dt) {
                dv[dt] = mfma16<F16>(frag_t(doimg, u2, dt, lane), pf, dv[dt]);
                dk[dt] = mfma16<F16>(frag_t(qimg, u2, dt, lane), df, dk[dt]);
            }
        }
    }
    if (kin) {
        bf16_t* krow = a.dqkv + (row0 + key) * ld + D + hh * HD + 4 * lg;
#pragma unroll
        for (int dt = 0; dt < ND; ++dt) {
            *reinterpret_cast<uint2*>(krow + dt * 16) = pack4<F16>(dk[dt][0], dk[dt][1], dk[dt][2], dk[dt][3]);
            *reinterpret_cast<uint2*>(krow + D + dt * 16) = pack4<F16>(dv[dt][0], dv[dt][1], dv[dt][2], dv[dt][3]);
        }
    }
}

template <typename K>
int set_lds(K kern, size_t bytes)
{
    if (bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

// (two images of NT * 16 rows + row statistics must fit 160 KiB: 608 rows of 64-wide heads, 416 rows of 96-wide ones)
inline int pick_nt(int T) { return T <= 160 ? 10 : (T <= 224 ? 14 : (T <= 416 ? 26 : ((T <= 608 && HD <= 64) ? 38 : 0))); }
inline int pick_threads(int T) { const int tiles = (T + 15) / 16; return (tiles % 3 == 0) ? 192 : 256; }

template <int NT, bool F16>
int launch_all(const AttnArgs& a_in, int B, int mode, hipStream_t stream)
{
    AttnArgs a = a_in;
    if (a.colparts && HD > 64 && NT > 10) return (int)hipErrorInvalidValue;      // (not built there: see kCols)
    const int threads = pick_threads(a.T);
    // Per-wave output staging (store_tile) for the backward passes, when the LDS has room for it next to the images.
    // Measured (tools/attn_prof.sh, T = 129): dK/dV 136 -> 126 us, dQ 111 -> 109 us; the forward gets SLOWER with it
    // (86 -> 95 us: 47.7 instead of 40 KiB per workgroup drops a resident workgroup), so it keeps the direct stores.
    size_t stg = (size_t)(threads / 64) * STG_BYTES;
    if (mode == 0 || (size_t)2 * NT * 16 * ROWB + (size_t)2 * NT * 16 * sizeof(float) + stg > 160 * 1024) stg = 0;
    a.stage_out = stg ? 1 : 0;
    const size_t img = (size_t)2 * NT * 16 * ROWB + stg;
    const dim3 grid(B * a.heads);
    int rc;
    if (mode == 0) {
        const bool full = !a.cu && !a.mask && (((a.T + 31) >> 5) << 1) >= NT && NT <= 14;
        if (full) {
            auto k = attn_q_pass_kernel<NT, false, true, F16, false>;
            if ((rc = set_lds(k, img))) return rc;
            hipLaunchKernelGGL(k, grid, dim3(threads), img, stream, a);
        } else {
            if (a.mask) {
                auto k = attn_q_pass_kernel<NT, false, false, F16, true>;
                if ((rc = set_lds(k, img))) return rc;
                hipLaunchKernelGGL(k, grid, dim3(threads), img, stream, a);
            } else {
                auto k = attn_q_pass_kernel<NT, false, false, F16, false>;
                if ((rc = set_lds(k, img))) return rc;
                hipLaunchKernelGGL(k, grid, dim3(threads), img, stream, a);
            }
        }
        EDITOR_LAUNCH_CHECK();
    } else {
        const bool full = !a.cu && !a.mask && (((a.T + 31) >> 5) << 1) >= NT && NT <= 14;
        if (a.mask) {
            auto k1 = attn_q_pass_kernel<NT, true, false, F16, true>;
            if ((rc = set_lds(k1, img))) return rc;
            hipLaunchKernelGGL(k1, grid, dim3(threads), img, stream, a);
        } else if (full) {
            auto k1 = attn_q_pass_kernel<NT, true, true, F16, false>;
            if ((rc = set_lds(k1, img))) return rc;
            hipLaunchKernelGGL(k1, grid, dim3(threads), img, stream, a);
        } else {
            auto k1 = attn_q_pass_kernel<NT, true, false, F16, false>;
            if ((rc = set_lds(k1, img))) return rc;
            hipLaunchKernelGGL(k1, grid, dim3(threads), img, stream, a);
        }
        EDITOR_LAUNCH_CHECK();
        const size_t lds2 = img + (size_t)2 * NT * 16 * sizeof(float);          // images | lse, delta | staging
        if (full) {
            if constexpr (NT <= 14) {                    // (`full` implies it; the unrolled form of longer sequences is not built)
                auto k2 = attn_kv_pass_kernel<NT, F16, true>;
                if ((rc = set_lds(k2, lds2))) return rc;
                hipLaunchKernelGGL(k2, grid, dim3(threads), lds2, stream, a);
            }
        } else {
            auto k2 = attn_kv_pass_kernel<NT, F16, false>;
            if ((rc = set_lds(k2, lds2))) return rc;
            hipLaunchKernelGGL(k2, grid, dim3(threads), lds2, stream, a);
        }
        EDITOR_LAUNCH_CHECK();
    }
    return 0;
}

template <bool F16>
int launch_long(const AttnArgs& a, int B, int mode, hipStream_t stream)
{
    if (a.probs || a.colparts) return (int)hipErrorInvalidValue;   // (no probability output / column sums: the backbone's sequences are short)
    const dim3 grid(B * a.heads, (a.T + 63) / 64);
    const size_t img = (size_t)2 * LCH * ROWB;
    int rc;
    if (mode == 0) {
        auto k = attn_q_long_kernel<false, F16>;
        if ((rc = set_lds(k, img))) return rc;
        hipLaunchKernelGGL(k, grid, dim3(256), img, stream, a);
        EDITOR_LAUNCH_CHECK();
    } else {
        auto k1 = attn_q_long_kernel<true, F16>;
        if ((rc = set_lds(k1, img))) return rc;
        hipLaunchKernelGGL(k1, grid, dim3(256), img, stream, a);
        EDITOR_LAUNCH_CHECK();
        auto k2 = attn_kv_long_kernel<F16>;
        const size_t lds2 = img + (size_t)2 * LCH * sizeof(float);
        if ((rc = set_lds(k2, lds2))) return rc;
        hipLaunchKernelGGL(k2, grid, dim3(256), lds2, stream, a);
        EDITOR_LAUNCH_CHECK();
    }
    return 0;
}

template <bool F16>
int dispatch(const AttnArgs& a, int B, int mode, hipStream_t stream)
{
    switch (pick_nt(a.T)) {
        case 10: return launch_all<10, F16>(a, B, mode, stream);
        case 14: return launch_all<14, F16>(a, B, mode, stream);
        case 26: return launch_all<26, F16>(a, B, mode, stream);
        case 38: return launch_all<38, F16>(a, B, mode, stream);      // 3 x 193 tokens: joint HMA block of the 384x128 configs
        default: return launch_long<F16>(a, B, mode, stream);         // > 608 tokens: chunked form
    }
}

template <bool F16>
int attention_fwd_h16(const uint16_t* qkv, int B, int T, int heads, int hd, float scale, const uint8_t* mask, uint16_t* out,
                      float* probs, int ldp, float* lse, const int* cu, long Mtot, hipStream_t stream)
{
    if (hd != HD || T < 1 || B < 1) return (int)hipErrorInvalidValue;
    if (probs && (ldp < T || (ldp & 3) || (reinterpret_cast<uintptr_t>(probs) & 15) || cu)) return (int)hipErrorInvalidValue;
    if (cu && mask) return (int)hipErrorInvalidValue;             // packed sequences hold only live tokens
    if (!cu) Mtot = (long)B * T;
    AttnArgs a{qkv, nullptr, nullptr, out, nullptr, probs, lse, nullptr, mask, T, heads, scale, ldp, cu, Mtot, 0};
    return dispatch<F16>(a, B, 0, stream);
}

template <bool F16>
int attention_bwd_h16(const uint16_t* qkv, const uint16_t* dout, const uint16_t* out, const float* lse, int B, int T, int heads,
                      int hd, float scale, const uint8_t* mask, uint16_t* dqkv, float* workspace, const int* cu, long Mtot,
                      float* colparts, hipStream_t stream)
{
    if (hd != HD || T < 1 || B < 1 || !workspace || !lse || (cu && mask)) return (int)hipErrorInvalidValue;
    if (!cu) Mtot = (long)B * T;
    AttnArgs a{qkv, dout, out, nullptr, dqkv, nullptr, const_cast<float*>(lse), workspace, mask, T, heads, scale, 0, cu, Mtot, 0,
               colparts};
    return dispatch<F16>(a, B, 1, stream);
}

template <bool F16>
int rollout_step_h16(const uint16_t* qkv, const float* lse, const float* r_in, int B, int T, int heads, int hd, float scale,
                     float* r_out, int final_step, hipStream_t stream)
{
    if (hd != HD || T < 2 || B < 1 || !qkv || !lse || !r_out) return (int)hipErrorInvalidValue;
    const int threads = pick_threads(T);
    const dim3 grid(B * heads);
    const long Mtot = (long)B * T;
#define ROLL_CASE(NTV) case NTV: {                                                                                   \
        auto k = attn_rollout_step_kernel<NTV, F16>;                                                                     \
        const size_t lds = (size_t)NTV * 16 * ROWB + (size_t)2 * NTV * 16 * sizeof(float);                              \
        int rc = set_lds(k, lds); if (rc) return rc;                                                                     \
        hipLaunchKernelGGL(k, grid, dim3(threads), lds, stream, qkv, lse, r_in, T, heads, scale, Mtot, r_out, final_step); \
        break; }
    switch (pick_nt(T)) {
        ROLL_CASE(10) ROLL_CASE(14) ROLL_CASE(26) ROLL_CASE(38)
        default: return (int)hipErrorInvalidValue;                     // the backbone's sequences are <= 608 tokens
    }
#undef ROLL_CASE
    EDITOR_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// ---- entry points.  This file is compiled once per head width (editor_amd/build.py: -DATTN_HD=32 / 64 / 96); the 64-wide build
// carries the C-ABI names of include/editor_hip.h and forwards the other widths to the suffixed builds of the same source.
#if ATTN_HD == 64
#define ATTN_ENTRY(name) name
#define ATTN_DECL_WIDTHS(name, ...) extern "C" int name##_hd32(__VA_ARGS__); extern "C" int name##_hd96(__VA_ARGS__);
#define ATTN_OTHER_WIDTHS(name, ...) do { if (hd == 32) return name##_hd32(__VA_ARGS__); if (hd == 96) return name##_hd96(__VA_ARGS__); } while (0)
#elif ATTN_HD == 32
#define ATTN_ENTRY(name) name##_hd32
#define ATTN_DECL_WIDTHS(name, ...)
#define ATTN_OTHER_WIDTHS(name, ...) do { } while (0)
#else
#define ATTN_ENTRY(name) name##_hd96
#define ATTN_DECL_WIDTHS(name, ...)
#define ATTN_OTHER_WIDTHS(name, ...) do { } while (0)
#endif

#define FWD_ARGS const uint16_t* qkv, int B, int T, int heads, int hd, float scale, const uint8_t* mask, uint16_t* out, \
                 float* probs, int ldp, float* lse, const int* cu, long Mtot, hipStream_t stream
ATTN_DECL_WIDTHS(editor_attention_fwd_bf16, FWD_ARGS)
ATTN_DECL_WIDTHS(editor_attention_fwd_f16, FWD_ARGS)
extern "C" int ATTN_ENTRY(editor_attention_fwd_bf16)(FWD_ARGS)
{
    ATTN_OTHER_WIDTHS(editor_attention_fwd_bf16, qkv, B, T, heads, hd, scale, mask, out, probs, ldp, lse, cu, Mtot, stream);
    return attention_fwd_h16<false>(qkv, B, T, heads, hd, scale, mask, out, probs, ldp, lse, cu, Mtot, stream);
}
extern "C" int ATTN_ENTRY(editor_attention_fwd_f16)(FWD_ARGS)
{
    ATTN_OTHER_WIDTHS(editor_attention_fwd_f16, qkv, B, T, heads, hd, scale, mask, out, probs, ldp, lse, cu, Mtot, stream);
    return attention_fwd_h16<true>(qkv, B, T, heads, hd, scale, mask, out, probs, ldp, lse, cu, Mtot, stream);
}
#undef FWD_ARGS

#define BWD_ARGS const uint16_t* qkv, const uint16_t* dout, const uint16_t* out, const float* lse, int B, int T, int heads, int hd, \
                 float scale, const uint8_t* mask, uint16_t* dqkv, float* workspace, const int* cu, long Mtot, hipStream_t stream
ATTN_DECL_WIDTHS(editor_attention_bwd_bf16, BWD_ARGS)
ATTN_DECL_WIDTHS(editor_attention_bwd_f16, BWD_ARGS)
extern "C" int ATTN_ENTRY(editor_attention_bwd_bf16)(BWD_ARGS)
{
    ATTN_OTHER_WIDTHS(editor_attention_bwd_bf16, qkv, dout, out, lse, B, T, heads, hd, scale, mask, dqkv, workspace, cu, Mtot, stream);
    return attention_bwd_h16<false>(qkv, dout, out, lse, B, T, heads, hd, scale, mask, dqkv, workspace, cu, Mtot, nullptr, stream);
}
extern "C" int ATTN_ENTRY(editor_attention_bwd_f16)(BWD_ARGS)
{
    ATTN_OTHER_WIDTHS(editor_attention_bwd_f16, qkv, dout, out, lse, B, T, heads, hd, scale, mask, dqkv, workspace, cu, Mtot, stream);
    return attention_bwd_h16<true>(qkv, dout, out, lse, B, T, heads, hd, scale, mask, dqkv, workspace, cu, Mtot, nullptr, stream);
}
#undef BWD_ARGS

// the same backward that ALSO leaves the column sums of dqkv (the qkv bias gradient) as one partial row per sequence:
// colparts[B][3 * heads * hd] floats, to be folded by editor_reduce_rows(_multi) - instead of a colsum pass over the 16-bit dqkv.
// Sequences of <= 608 tokens (the two-pass kernels); longer ones: hipErrorInvalidValue (the caller sums dqkv itself).
#define BWDC_ARGS const uint16_t* qkv, const uint16_t* dout, const uint16_t* out, const float* lse, int B, int T, int heads, int hd, \
                  float scale, const uint8_t* mask, uint16_t* dqkv, float* workspace, const int* cu, long Mtot, float* colparts, \
                  hipStream_t stream
ATTN_DECL_WIDTHS(editor_attention_bwd_colsum_bf16, BWDC_ARGS)
ATTN_DECL_WIDTHS(editor_attention_bwd_colsum_f16, BWDC_ARGS)
extern "C" int ATTN_ENTRY(editor_attention_bwd_colsum_bf16)(BWDC_ARGS)
{
    ATTN_OTHER_WIDTHS(editor_attention_bwd_colsum_bf16, qkv, dout, out, lse, B, T, heads, hd, scale, mask, dqkv, workspace, cu, Mtot, colparts, stream);
    if (!colparts) return (int)hipErrorInvalidValue;
    return attention_bwd_h16<false>(qkv, dout, out, lse, B, T, heads, hd, scale, mask, dqkv, workspace, cu, Mtot, colparts, stream);
}
extern "C" int ATTN_ENTRY(editor_attention_bwd_colsum_f16)(BWDC_ARGS)
{
    ATTN_OTHER_WIDTHS(editor_attention_bwd_colsum_f16, qkv, dout, out, lse, B, T, heads, hd, scale, mask, dqkv, workspace, cu, Mtot, colparts, stream);
    if (!colparts) return (int)hipErrorInvalidValue;
    return attention_bwd_h16<true>(qkv, dout, out, lse, B, T, heads, hd, scale, mask, dqkv, workspace, cu, Mtot, colparts, stream);
}
#undef BWDC_ARGS

#define ROLL_ARGS const uint16_t* qkv, const float* lse, const float* r_in, int B, int T, int heads, int hd, float scale, \
                  float* r_out, int final_step, hipStream_t stream
ATTN_DECL_WIDTHS(editor_attn_rollout_step_bf16, ROLL_ARGS)
ATTN_DECL_WIDTHS(editor_attn_rollout_step_f16, ROLL_ARGS)
extern "C" int ATTN_ENTRY(editor_attn_rollout_step_bf16)(ROLL_ARGS)
{
    ATTN_OTHER_WIDTHS(editor_attn_rollout_step_bf16, qkv, lse, r_in, B, T, heads, hd, scale, r_out, final_step, stream);
    return rollout_step_h16<false>(qkv, lse, r_in, B, T, heads, hd, scale, r_out, final_step, stream);
}
extern "C" int ATTN_ENTRY(editor_attn_rollout_step_f16)(ROLL_ARGS)
{
    ATTN_OTHER_WIDTHS(editor_attn_rollout_step_f16, qkv, lse, r_in, B, T, heads, hd, scale, r_out, final_step, stream);
    return rollout_step_h16<true>(qkv, lse, r_in, B, T, heads, hd, scale, r_out, final_step, stream);
}
#undef ROLL_ARGS
