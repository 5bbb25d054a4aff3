// Split-precision attention forward (COMPUTE_DTYPE 'f16x2'), gfx950 / CDNA4.
// Restates Attention.forward (vit_pytorch.py:184-198) and AttentionMask.forward (:240-258) at fp32-class accuracy on the
// HALF matrix cores: q, k, v arrive as pairs x = hi + lo of IEEE-half matrices (the qkv product's split output) and every
// contraction is three v_mfma_f32_16x16x32_f16 into one fp32 accumulator,
//     S   = Q_lo K_hi^T + Q_hi K_lo^T + Q_hi K_hi^T                       (lo.lo, 2^-22 relative, dropped)
//     O   = (P_hi V_lo + P_lo V_hi + P_hi V_hi) 2^-12,   P' = 2^12 softmax(S scale) = P_hi + P_lo
// (the probabilities are scaled by 2^12 before they are split so that the low-order half of a 1/T-sized probability stays
// in half's normal range; the power of two is divided out of the finished fp32 tile).  The softmax itself is the
// reference's fp32 formula exp(s - max) / sum.  Outputs: the attention output as a half pair again (operand of the proj
// product), the row log-sum-exp for the 16-bit backward (editor_attention_bwd_f16 on the hi halves), and - for the backbone -
// the fp32 probabilities the rollout consumes (vit_pytorch.py:638-644, SFTS.py:145-153).
//
// One workgroup per (sample, head) with the K and V pairs of the whole sequence in LDS (4 images; T <= 288), each wave
// sweeping 16-query tiles; longer sequences (joint HMA block) run 64 queries per workgroup and stream the keys through LDS
// in 128-row chunks, twice (row statistics, then P V) - the structure of attn_q_long_kernel.
#include "common.h"
#include "../../include/editor_hip.h"
#include "attn_common.h"

namespace {

struct SplitAttnArgs {
    const bf16_t* qkv_hi; const bf16_t* qkv_lo;
    bf16_t* out_hi; bf16_t* out_lo; float* probs; float* lse;
    const uint8_t* mask;
    int T, heads; float scale;
    int ldp;
    const int* cu; long Mtot;
    int cap;                              // rows per LDS image (whole form: the padded sequence; chunked form: the chunk)
};

constexpr int SCH = 128;                  // key chunk of the streamed form
constexpr float kPScale = 4096.f;         // 2^12

template <bool MULTI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(HD <= 64 ? 4 : 3, 8))) void attn_fwd_split_kernel(SplitAttnArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int cap = a.cap;
    char* khi = smem;
    char* klo = smem + cap * ROWB;
    char* vhi = smem + 2 * cap * ROWB;
    char* vlo = smem + 3 * cap * ROWB;
    const int D = a.heads * HD;
    const int b = blockIdx.x / a.heads, hh = blockIdx.x % a.heads;
    const long ld = 3L * D;
    const long row0 = a.cu ? (long)a.cu[b] : (long)b * a.T;
    const int T = a.cu ? a.cu[b + 1] - a.cu[b] : a.T;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const bf16_t* qh = a.qkv_hi + row0 * ld + hh * HD;
    const bf16_t* ql = a.qkv_lo + row0 * ld + hh * HD;
    const uint8_t* mk = a.mask ? a.mask + (long)b * T : nullptr;     // dense-masked form only (packed rows are all live)
    const float sc = a.scale * kLog2e;
    const long row_idx0 = (long)hh * a.Mtot + row0;
    const long prow0 = ((long)b * a.heads + hh) * a.T;

    // S^T tile (keys 16t .. 16t+15 of the resident images x this wave's 16 queries): lane (i, g) gets keys 16t + 4g + r
    auto scores = [&](int t, const short8_t (&qfh)[KS], const short8_t (&qfl)[KS]) {
        float4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const short8_t kh = frag_k(khi, t * 16, s, lane), kl = frag_k(klo, t * 16, s, lane);
            acc = mfma16<true>(kl, qfh[s], acc);
            acc = mfma16<true>(kh, qfl[s], acc);
            acc = mfma16<true>(kh, qfh[s], acc);
        }
        return acc;
    };
    auto key_ok = [&](int key) { return key < T && (!mk || mk[key]); };

    // sweep 1 over the resident keys [c0, c0 + 16 ntc): running max / sum of this lane's keys
    auto sweep1 = [&](int c0, int ntc, const short8_t (&qfh)[KS], const short8_t (&qfl)[KS], float& m, float& l) {
#pragma unroll 2
        for (int t = 0; t < ntc; ++t) {
            const float4_t acc = scores(t, qfh, qfl);
            float sv[4], tm = -INFINITY;
            const int key0 = c0 + 16 * t + 4 * lg;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sv[r] = key_ok(key0 + r) ? acc[r] * sc : -INFINITY;
                tm = fmaxf(tm, sv[r]);
            }
            if (tm > m) { l *= __builtin_amdgcn_exp2f(m - tm); m = tm; }
            if (m > -INFINITY)
                l += (__builtin_amdgcn_exp2f(sv[0] - m) + __builtin_amdgcn_exp2f(sv[1] - m)) +
                     (__builtin_amdgcn_exp2f(sv[2] - m) + __builtin_amdgcn_exp2f(sv[3] - m));
        }
    };
    // sweep 2: P = exp2(s - M) / L for the resident keys; O'^T += V^T P'^T with P' = 2^12 P as a half pair
    auto sweep2 = [&](int c0, int ntc, const short8_t (&qfh)[KS], const short8_t (&qfl)[KS], float Ms, float inv, float* pr,
                      float4_t (&o)[ND]) {
        const float invs = inv * kPScale;
#pragma unroll 1
        for (int s2 = 0; s2 < ntc / 2; ++s2) {
            uint2 ph[2], pl[2];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int t = 2 * s2 + half;
                const float4_t acc = scores(t, qfh, qfl);
                const int key0 = c0 + 16 * t + 4 * lg;
                float e[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) e[r] = __builtin_amdgcn_exp2f(key_ok(key0 + r) ? acc[r] * sc - Ms : -INFINITY);
                if (pr && key0 < a.ldp) *reinterpret_cast<float4*>(pr + key0) = make_float4(e[0] * inv, e[1] * inv, e[2] * inv, e[3] * inv);
                const float p0 = e[0] * invs, p1 = e[1] * invs, p2 = e[2] * invs, p3 = e[3] * invs;
                ph[half] = pack4<true>(p0, p1, p2, p3);
                const float2_t_ h0 = H16<true>::unpack2(ph[half].x), h1 = H16<true>::unpack2(ph[half].y);
                pl[half] = pack4<true>(p0 - h0.x, p1 - h0.y, p2 - h1.x, p3 - h1.y);
            }
            const short8_t pfh = join(ph[0], ph[1]), pfl = join(pl[0], pl[1]);
#pragma unroll
            for (int dt = 0; dt < ND; ++dt) {
                const short8_t vh = frag_t(vhi, s2, dt, lane), vl = frag_t(vlo, s2, dt, lane);
                o[dt] = mfma16<true>(vl, pfh, o[dt]);
                o[dt] = mfma16<true>(vh, pfl, o[dt]);
                o[dt] = mfma16<true>(vh, pfh, o[dt]);
            }
        }
    };
    auto finish_stats = [&](float m, float l, bool qok, int q, float& Ms, float& inv) {
        const float M = group_max(m);
        const float L = group_sum(m > -INFINITY ? l * __builtin_amdgcn_exp2f(m - M) : 0.f);
        const bool live = qok && L > 0.f;
        Ms = M > -INFINITY ? M : 0.f;
        inv = live ? 1.f / L : 0.f;
        if (a.lse && lg == 0 && q < T) a.lse[row_idx0 + q] = live ? M + __builtin_amdgcn_logf(L) : INFINITY;
    };
    auto store_out = [&](int q0, float4_t (&o)[ND]) {
        if (q0 + li >= T) return;
        const long off = (row0 + q0 + li) * D + hh * HD + 4 * lg;
#pragma unroll
        for (int dt = 0; dt < ND; ++dt) {
            const float v0 = o[dt][0] * (1.f / kPScale), v1 = o[dt][1] * (1.f / kPScale), v2 = o[dt][2] * (1.f / kPScale),
                        v3 = o[dt][3] * (1.f / kPScale);
            const uint2 h = pack4<true>(v0, v1, v2, v3);
            const float2_t_ h0 = H16<true>::unpack2(h.x), h1 = H16<true>::unpack2(h.y);
            *reinterpret_cast<uint2*>(a.out_hi + off + dt * 16) = h;
            *reinterpret_cast<uint2*>(a.out_lo + off + dt * 16) = pack4<true>(v0 - h0.x, v1 - h0.y, v2 - h1.x, v3 - h1.y);
        }
    };

    if constexpr (!MULTI) {
        const int nt = ((T + 31) >> 5) << 1;                         // populated key tiles (even count), nt * 16 <= cap
        load_image(khi, qh + D, ld, T, nt * 16);
        load_image(klo, ql + D, ld, T, nt * 16);
        load_image(vhi, qh + 2 * D, ld, T, nt * 16);
        load_image(vlo, ql + 2 * D, ld, T, nt * 16);
        images_ready();
        for (int q0 = w * 16; q0 < T; q0 += nw * 16) {
            const int q = q0 + li;
            const bool qok = q < T && (!mk || mk[q]);
            short8_t qfh[KS], qfl[KS];
#pragma unroll
            for (int s = 0; s < KS; ++s) { qfh[s] = frag_own(qh, ld, q0, T, s, lane); qfl[s] = frag_own(ql, ld, q0, T, s, lane); }
            float m = -INFINITY, l = 0.f, Ms, inv;
            sweep1(0, nt, qfh, qfl, m, l);
            finish_stats(m, l, qok, q, Ms, inv);
            float* pr = (a.probs && q < T) ? a.probs + (prow0 + q) * a.ldp : nullptr;
            float4_t o[ND];
#pragma unroll
            for (int dt = 0; dt < ND; ++dt) o[dt] = float4_t{0.f, 0.f, 0.f, 0.f};
            sweep2(0, nt, qfh, qfl, Ms, inv, pr, o);
            store_out(q0, o);
        }
    } else {
        const int qb0 = blockIdx.y * 64;
        if (qb0 >= T) return;                                         // (whole workgroup: before any barrier)
        const int q0 = qb0 + w * 16, q = q0 + li;
        const bool active = q0 < T;                                   // (wave-uniform; idle waves still load and synchronise)
        const bool qok = q < T && (!mk || mk[q]);
        short8_t qfh[KS], qfl[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) { qfh[s] = frag_own(qh, ld, q0, T, s, lane); qfl[s] = frag_own(ql, ld, q0, T, s, lane); }
        float m = -INFINITY, l = 0.f, Ms = 0.f, inv = 0.f;
        for (int c0 = 0; c0 < T; c0 += SCH) {
            const int len = min(SCH, T - c0), ntc = ((len + 31) >> 5) << 1;
            __syncthreads();
            load_image(khi, qh + D + (long)c0 * ld, ld, len, ntc * 16);
            load_image(klo, ql + D + (long)c0 * ld, ld, len, ntc * 16);
            images_ready();
            if (active) sweep1(c0, ntc, qfh, qfl, m, l);
        }
        if (active) finish_stats(m, l, qok, q, Ms, inv);
        float* pr = (a.probs && q < T) ? a.probs + (prow0 + q) * a.ldp : nullptr;
        float4_t o[ND];
#pragma unroll
        for (int dt = 0; dt < ND; ++dt) o[dt] = float4_t{0.f, 0.f, 0.f, 0.f};
        for (int c0 = 0; c0 < T; c0 += SCH) {
            const int len = min(SCH, T - c0), ntc = ((len + 31) >> 5) << 1;
            __syncthreads();
            load_image(khi, qh + D + (long)c0 * ld, ld, len, ntc * 16);
            load_image(klo, ql + D + (long)c0 * ld, ld, len, ntc * 16);
            load_image(vhi, qh + 2 * D + (long)c0 * ld, ld, len, ntc * 16);
            load_image(vlo, ql + 2 * D + (long)c0 * ld, ld, len, ntc * 16);
            images_ready();
            if (active) sweep2(c0, ntc, qfh, qfl, Ms, inv, pr, o);
        }
        if (active) store_out(q0, o);
    }
}

// Short sequences (T <= 224: the backbone's 129 / 193 tokens and the per-modality HMA blocks): the lane's 4 * NT scores of a
// 16-query tile stay in REGISTERS, so S = Q K^T (three MFMA passes per k-step) is formed once and each key costs one
// exponential - the streamed form above recomputes the scores in its second sweep and pays the running-max rescales of the
// first (measured at T = 129, B = 384: 309 us per layer against 79 us for the 16-bit kernel; this form: see DESIGN.md).
template <int NT>
// (waves_per_eu >= 2: VGPR-form MFMAs, no v_accvgpr_read_b32 per score element - see attn_kv_pass_kernel)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((NT <= 10 && HD <= 64) ? 3 : 2, 8))) void attn_fwd_split_reg_kernel(SplitAttnArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int Tp = NT * 16;
    char* khi = smem;
    char* klo = smem + Tp * ROWB;
    char* vhi = smem + 2 * Tp * ROWB;
    char* vlo = smem + 3 * Tp * ROWB;
    const int D = a.heads * HD;
    const int b = blockIdx.x / a.heads, hh = blockIdx.x % a.heads;
    const long ld = 3L * D;
    const long row0 = a.cu ? (long)a.cu[b] : (long)b * a.T;
    const int T = a.cu ? a.cu[b + 1] - a.cu[b] : a.T;
    const int nt = min(NT, ((T + 31) >> 5) << 1);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const bf16_t* qh = a.qkv_hi + row0 * ld + hh * HD;
    const bf16_t* ql = a.qkv_lo + row0 * ld + hh * HD;
    const uint8_t* mk = a.mask ? a.mask + (long)b * T : nullptr;
    load_image(khi, qh + D, ld, T, nt * 16);
    load_image(klo, ql + D, ld, T, nt * 16);
    load_image(vhi, qh + 2 * D, ld, T, nt * 16);
    load_image(vlo, ql + 2 * D, ld, T, nt * 16);
    short8_t qnh[KS], qnl[KS];                                       // the wave's first query fragments travel with the images
#pragma unroll
    for (int s = 0; s < KS; ++s) { qnh[s] = frag_own(qh, ld, w * 16, T, s, lane); qnl[s] = frag_own(ql, ld, w * 16, T, s, lane); }
    images_ready();
    const float sc = a.scale * kLog2e;
    const long row_idx0 = (long)hh * a.Mtot + row0;
    const long prow0 = ((long)b * a.heads + hh) * a.T;
    const int klim = T - 4 * lg;                                    // unmasked: key 16 t + 4 g + r is valid  <=>  16 t + r < klim

    for (int q0 = w * 16; q0 < T; q0 += nw * 16) {
        const int q = q0 + li;
        const bool qok = q < T && (!mk || mk[q]);
        short8_t qfh[KS], qfl[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) { qfh[s] = qnh[s]; qfl[s] = qnl[s]; }
        if (q0 + nw * 16 < T) {
#pragma unroll
            for (int s = 0; s < KS; ++s) { qnh[s] = frag_own(qh, ld, q0 + nw * 16, T, s, lane); qnl[s] = frag_own(ql, ld, q0 + nw * 16, T, s, lane); }
        }
        float4_t sreg[NT];
        float m = -INFINITY;
#pragma unroll
        for (int t = 0; t < NT; t += 2) {
            if (t >= nt) { sreg[t] = sreg[t + 1] = float4_t{-INFINITY, -INFINITY, -INFINITY, -INFINITY}; continue; }
            // two key tiles at a time: eight fragment reads in flight, then twelve MFMAs on two independent accumulators
            short8_t kh0[KS], kl0[KS], kh1[KS], kl1[KS];
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                kh0[s] = frag_k(khi, t * 16, s, lane); kl0[s] = frag_k(klo, t * 16, s, lane);
                kh1[s] = frag_k(khi, t * 16 + 16, s, lane); kl1[s] = frag_k(klo, t * 16 + 16, s, lane);
            }
            float4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                a0 = mfma16<true>(kl0[s], qfh[s], a0); a1 = mfma16<true>(kl1[s], qfh[s], a1);
                a0 = mfma16<true>(kh0[s], qfl[s], a0); a1 = mfma16<true>(kh1[s], qfl[s], a1);
                a0 = mfma16<true>(kh0[s], qfh[s], a0); a1 = mfma16<true>(kh1[s], qfh[s], a1);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k0 = 16 * t + 4 * lg + r, k1 = k0 + 16;
                const bool v0 = mk ? (k0 < T && mk[k0]) : (16 * t + r < klim), v1 = mk ? (k1 < T && mk[k1]) : (16 * t + 16 + r < klim);
                sreg[t][r] = v0 ? a0[r] * sc : -INFINITY;
                sreg[t + 1][r] = v1 ? a1[r] * sc : -INFINITY;
                m = fmaxf(m, fmaxf(sreg[t][r], sreg[t + 1][r]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        const float M = group_max(m);
        const float Ms = M > -INFINITY ? M : 0.f;
        float l = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) { sreg[t][r] = __builtin_amdgcn_exp2f(sreg[t][r] - Ms); l += sreg[t][r]; }
        const float L = group_sum(l);
        const bool live = qok && L > 0.f;
        const float inv = live ? 1.f / L : 0.f, invs = inv * kPScale;
        if (a.lse && lg == 0 && q < T) a.lse[row_idx0 + q] = live ? M + __builtin_amdgcn_logf(L) : INFINITY;
        float* pr = (a.probs && q < T) ? a.probs + (prow0 + q) * a.ldp : nullptr;
        float4_t o[ND];
#pragma unroll
        for (int dt = 0; dt < ND; ++dt) o[dt] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s2 = 0; s2 < NT / 2; ++s2) {
            if (2 * s2 >= nt) continue;
            uint2 ph[2], pl[2];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int t = 2 * s2 + half;
                if (pr && 16 * t + 4 * lg < a.ldp)
                    *reinterpret_cast<float4*>(pr + 16 * t + 4 * lg) = make_float4(sreg[t][0] * inv, sreg[t][1] * inv, sreg[t][2] * inv, sreg[t][3] * inv);
                const float p0 = sreg[t][0] * invs, p1 = sreg[t][1] * invs, p2 = sreg[t][2] * invs, p3 = sreg[t][3] * invs;
                ph[half] = pack4<true>(p0, p1, p2, p3);
                const float2_t_ h0 = H16<true>::unpack2(ph[half].x), h1 = H16<true>::unpack2(ph[half].y);
                pl[half] = pack4<true>(p0 - h0.x, p1 - h0.y, p2 - h1.x, p3 - h1.y);
            }
            const short8_t pfh = join(ph[0], ph[1]), pfl = join(pl[0], pl[1]);
            short8_t vh[ND], vl[ND];
#pragma unroll
            for (int dt = 0; dt < ND; ++dt) { vh[dt] = frag_t(vhi, s2, dt, lane); vl[dt] = frag_t(vlo, s2, dt, lane); }
#pragma unroll
            for (int dt = 0; dt < ND; ++dt) {
                o[dt] = mfma16<true>(vl[dt], pfh, o[dt]);
                o[dt] = mfma16<true>(vh[dt], pfl, o[dt]);
                o[dt] = mfma16<true>(vh[dt], pfh, o[dt]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (q < T) {
            const long off = (row0 + q) * D + hh * HD + 4 * lg;
#pragma unroll
            for (int dt = 0; dt < ND; ++dt) {
                const float v0 = o[dt][0] * (1.f / kPScale), v1 = o[dt][1] * (1.f / kPScale), v2 = o[dt][2] * (1.f / kPScale),
                            v3 = o[dt][3] * (1.f / kPScale);
                const uint2 h = pack4<true>(v0, v1, v2, v3);
                const float2_t_ h0 = H16<true>::unpack2(h.x), h1 = H16<true>::unpack2(h.y);
                *reinterpret_cast<uint2*>(a.out_hi + off + dt * 16) = h;
                *reinterpret_cast<uint2*>(a.out_lo + off + dt * 16) = pack4<true>(v0 - h0.x, v1 - h0.y, v2 - h1.x, v3 - h1.y);
            }
        }
    }
}

// Rollout step of the split-precision mode (SFTS.py:150-153, row-vector form; round 4): r_out[k] = sum_q r_in[q] P_l[q,k] with
// P_l RECOMPUTED from layer l's q / k half pairs and the forward's row log-sum-exp, P[q,k] = exp2(s[q,k] - lse[q]), s the same
// three-pass fp32-class score the forward formed (Q_lo K_hi + Q_hi K_lo + Q_hi K_hi) - instead of read from the materialised
// (L,3B,h,T,T) fp32 probabilities (3.8 GB written by the forward and read back at B = 128).  Skeleton of the 16-bit
// attn_rollout_step_kernel (attention_bf16.hip): own = keys, the Q pair images + lse + r_in in LDS.
template <int NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void attn_rollout_step_split_kernel(const bf16_t* __restrict__ qkv_hi, const bf16_t* __restrict__ qkv_lo,
    const float* __restrict__ lse, const float* __restrict__ r_in, int T, int heads, float scale, long Mtot, float* __restrict__ r_out,
    int final_step)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int Tp = NT * 16;
    char* qhi = smem;
    char* qlo = smem + Tp * ROWB;
    float* lse_s = reinterpret_cast<float*>(smem + 2 * Tp * ROWB);
    float* w_s = lse_s + Tp;
    const int D = heads * HD;
    const int b = blockIdx.x / heads, hh = blockIdx.x % heads;
    const long ld = 3L * D;
    const long row0 = (long)b * T;
    const int nt = min(NT, ((T + 31) >> 5) << 1);
    const bf16_t* bh = qkv_hi + row0 * ld + hh * HD;
    const bf16_t* bl = qkv_lo + row0 * ld + hh * HD;
    load_image(qhi, bh, ld, T, nt * 16);
    load_image(qlo, bl, ld, T, nt * 16);
    for (int t = threadIdx.x; t < Tp; t += blockDim.x) {
        lse_s[t] = t < T ? lse[(long)hh * Mtot + row0 + t] : INFINITY;          // +inf (pad rows, dead queries) -> probability 0
        w_s[t] = t < T ? (r_in ? r_in[(long)blockIdx.x * T + t] : (t == 0 ? 1.f : 0.f)) : 0.f;
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    short8_t knh[KS], knl[KS];                                       // own key fragments: first tile's travel with the images
#pragma unroll
    for (int s = 0; s < KS; ++s) { knh[s] = frag_own(bh + D, ld, w * 16, T, s, lane); knl[s] = frag_own(bl + D, ld, w * 16, T, s, lane); }
    images_ready();
    const int li = lane & 15, lg = lane >> 4;
    const float sc = scale * kLog2e;
    for (int k0 = w * 16; k0 < T; k0 += nw * 16) {
        const int key = k0 + li;
        short8_t kh[KS], kl[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) { kh[s] = knh[s]; kl[s] = knl[s]; }
        if (k0 + nw * 16 < T) {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                knh[s] = frag_own(bh + D, ld, k0 + nw * 16, T, s, lane); knl[s] = frag_own(bl + D, ld, k0 + nw * 16, T, s, lane);
            }
        }
        float acc = 0.f;
#pragma unroll 2
        for (int u = 0; u < nt; ++u) {
            float4_t s_ = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS; ++s) {                          // lane (i, g): queries 16u + 4g + r  x  key k0 + i
                const short8_t fh = frag_k(qhi, u * 16, s, lane), fl = frag_k(qlo, u * 16, s, lane);
                s_ = mfma16<true>(fl, kh[s], s_);
                s_ = mfma16<true>(fh, kl[s], s_);
                s_ = mfma16<true>(fh, kh[s], s_);
            }
            const float4 l4 = *reinterpret_cast<const float4*>(lse_s + 16 * u + 4 * lg);
            const float4 w4 = *reinterpret_cast<const float4*>(w_s + 16 * u + 4 * lg);
            acc = fmaf(__builtin_amdgcn_exp2f(s_[0] * sc - l4.x), w4.x, acc);
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

template <auto KERN>
int set_lds_dev(size_t bytes)
{
    if (bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)KERN, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

}  // namespace

// ---- entry points (this file is compiled once per head width, as attention_bf16.hip: editor_amd/build.py -DATTN_HD=32 / 64 / 96) ----
#if ATTN_HD == 64
#define SPLIT_ENTRY(name) name
#define SPLIT_DECL_WIDTHS(name, ...) extern "C" int name##_hd32(__VA_ARGS__); extern "C" int name##_hd96(__VA_ARGS__);
#define SPLIT_OTHER_WIDTHS(name, ...) do { if (hd == 32) return name##_hd32(__VA_ARGS__); if (hd == 96) return name##_hd96(__VA_ARGS__); } while (0)
#elif ATTN_HD == 32
#define SPLIT_ENTRY(name) name##_hd32
#define SPLIT_DECL_WIDTHS(name, ...)
#define SPLIT_OTHER_WIDTHS(name, ...) do { } while (0)
#else
#define SPLIT_ENTRY(name) name##_hd96
#define SPLIT_DECL_WIDTHS(name, ...)
#define SPLIT_OTHER_WIDTHS(name, ...) do { } while (0)
#endif
constexpr size_t kLdsMax = 160 * 1024;

#define FWD2_ARGS const uint16_t* qkv_hi, const uint16_t* qkv_lo, int B, int T, int heads, int hd, float scale, const uint8_t* mask, \
                  uint16_t* out_hi, uint16_t* out_lo, float* probs, int ldp, float* lse, const int* cu, long Mtot, hipStream_t stream
SPLIT_DECL_WIDTHS(editor_attention_fwd_f16x2, FWD2_ARGS)
extern "C" int SPLIT_ENTRY(editor_attention_fwd_f16x2)(FWD2_ARGS)
{
    SPLIT_OTHER_WIDTHS(editor_attention_fwd_f16x2, qkv_hi, qkv_lo, B, T, heads, hd, scale, mask, out_hi, out_lo, probs, ldp, lse, cu, Mtot,
                       stream);
    if (hd != HD || T < 1 || B < 1 || !qkv_hi || !qkv_lo || !out_hi || !out_lo) return (int)hipErrorInvalidValue;
    if (probs && (ldp < T || (ldp & 3) || (reinterpret_cast<uintptr_t>(probs) & 15) || cu)) return (int)hipErrorInvalidValue;
    if (cu && mask) return (int)hipErrorInvalidValue;
    if (!cu) Mtot = (long)B * T;
    const int rows = (((T + 31) >> 5) << 1) * 16;
    SplitAttnArgs a{qkv_hi, qkv_lo, out_hi, out_lo, probs, lse, mask, T, heads, scale, ldp, cu, Mtot, rows};
    const int tiles = (T + 15) / 16;
    const int threads = (tiles % 3 == 0) ? 192 : 256;
    // (four images - K and V pairs - of the whole sequence must fit the CU's 160 KiB: 288 rows of 64-wide heads, 208 of 96-wide ones)
    if (rows <= 160 && (size_t)4 * 160 * ROWB <= kLdsMax) {
        // scores in registers (one pass): NT = 10 (T <= 160) or 14 (T <= 224) key tiles
        const size_t lds = (size_t)4 * 160 * ROWB;
        if (int rc = set_lds_dev<attn_fwd_split_reg_kernel<10>>(lds)) return rc;
        hipLaunchKernelGGL(attn_fwd_split_reg_kernel<10>, dim3(B * heads), dim3(threads), lds, stream, a);
    } else if (rows <= 224 && (size_t)4 * 224 * ROWB <= kLdsMax) {
        const size_t lds = (size_t)4 * 224 * ROWB;
        if (int rc = set_lds_dev<attn_fwd_split_reg_kernel<14>>(lds)) return rc;
        hipLaunchKernelGGL(attn_fwd_split_reg_kernel<14>, dim3(B * heads), dim3(threads), lds, stream, a);
    } else if (rows <= 288 && (size_t)4 * rows * ROWB <= kLdsMax) {
        const size_t lds = (size_t)4 * rows * ROWB;
        if (int rc = set_lds_dev<attn_fwd_split_kernel<false>>(lds)) return rc;
        hipLaunchKernelGGL(attn_fwd_split_kernel<false>, dim3(B * heads), dim3(threads), lds, stream, a);
    } else {
        a.cap = SCH;
        const size_t lds = (size_t)4 * SCH * ROWB;
        if (int rc = set_lds_dev<attn_fwd_split_kernel<true>>(lds)) return rc;
        hipLaunchKernelGGL(attn_fwd_split_kernel<true>, dim3(B * heads, (T + 63) / 64), dim3(256), lds, stream, a);
    }
    EDITOR_LAUNCH_CHECK();
    return 0;
}
#undef FWD2_ARGS

#define ROLL2_ARGS const uint16_t* qkv_hi, const uint16_t* qkv_lo, const float* lse, const float* r_in, int B, int T, int heads, int hd, \
                   float scale, float* r_out, int final_step, hipStream_t stream
SPLIT_DECL_WIDTHS(editor_attn_rollout_step_f16x2, ROLL2_ARGS)
extern "C" int SPLIT_ENTRY(editor_attn_rollout_step_f16x2)(ROLL2_ARGS)
{
    SPLIT_OTHER_WIDTHS(editor_attn_rollout_step_f16x2, qkv_hi, qkv_lo, lse, r_in, B, T, heads, hd, scale, r_out, final_step, stream);
    if (hd != HD || T < 2 || B < 1 || !qkv_hi || !qkv_lo || !lse || !r_out) return (int)hipErrorInvalidValue;
    const int tiles = (T + 15) / 16;
    const int threads = (tiles % 3 == 0) ? 192 : 256;
    const dim3 grid(B * heads);
    const long Mtot = (long)B * T;
#define ROLL_CASE(NTV) {                                                                                                \
        const size_t lds = (size_t)2 * NTV * 16 * ROWB + (size_t)2 * NTV * 16 * sizeof(float);                             \
        if (lds > kLdsMax) return (int)hipErrorInvalidValue;                                                               \
        if (int rc = set_lds_dev<attn_rollout_step_split_kernel<NTV>>(lds)) return rc;                                    \
        hipLaunchKernelGGL(attn_rollout_step_split_kernel<NTV>, grid, dim3(threads), lds, stream, qkv_hi, qkv_lo, lse, r_in, T, \
                           heads, scale, Mtot, r_out, final_step); }
    if (T <= 160) ROLL_CASE(10)
    else if (T <= 224) ROLL_CASE(14)
    else if (T <= 416) ROLL_CASE(26)
    else if (T <= 608) ROLL_CASE(38)
    else return (int)hipErrorInvalidValue;                            // the backbone's sequences are <= 608 tokens (416 at 96 columns)
#undef ROLL_CASE
    EDITOR_LAUNCH_CHECK();
    return 0;
}
#undef ROLL2_ARGS
