// Training-step kernels around the hot path (SURVEY.md 8(f) rows N4 and the drop-path RNG of K5), gfx950.
//   fused multi-tensor SGD   torch.optim.SGD(momentum, weight_decay) over the per-parameter groups of
//                            solver/make_optimizer.py:4-29 (bias lr x2) in ONE launch, HBM bound (5 x 4 B per weight)
//   drop-path row scales     per-sample Bernoulli keep / keep_prob of every block and branch (vit_pytorch.py:52-69,511)
//                            for the whole backbone in one launch (counter-based RNG)
#include "common.h"
#include "../../include/editor_hip.h"

namespace {

constexpr int kChunk = 16384;           // elements per workgroup

// chunk c covers elements [off, off+len) of tensor t;  g' = g + wd*p ; m = mu*m + g' (m starts at 0) ; p -= lr*m
template <bool F16>
__global__ __launch_bounds__(256) void sgd_multi_kernel(float* const* __restrict__ p_ptrs, const float* const* __restrict__ g_ptrs,
    float* const* __restrict__ m_ptrs, const int* __restrict__ chunk_tensor, const long* __restrict__ chunk_off,
    const long* __restrict__ numel, const float* __restrict__ lr, const float* __restrict__ wd, float momentum,
    uint16_t* const* __restrict__ h_ptrs, int* __restrict__ nonfinite, const float* __restrict__ inv_scale_dev,
    const int* __restrict__ skip)
{
    // amp.GradScaler.step semantics (engine/processor.py:94-96): when this step's gradients hold an inf / nan (flag written
    // by grad_check_multi_kernel BEFORE this launch) nothing is updated - parameters, momentum and the 16-bit shadows keep
    // their values; gradients that carry the (device-resident) loss scale are unscaled on the way in
    if (skip && *skip) return;
    const float gsc = inv_scale_dev ? *inv_scale_dev : 1.f;
    const int t = chunk_tensor[blockIdx.x];
    const long off = chunk_off[blockIdx.x];
    const long n = min((long)kChunk, numel[t] - off);
    float* __restrict__ p = p_ptrs[t] + off;
    const float* __restrict__ g = g_ptrs[t] + off;
    float* __restrict__ m = m_ptrs[t] + off;
    if (!g_ptrs[t]) return;                                   // parameter without a gradient this step
    const float l = lr[t], w = wd[t];
    // optional bf16 shadow of the updated parameter (the GEMM operand copy), written in the same pass
    uint16_t* __restrict__ h = (h_ptrs && h_ptrs[t]) ? h_ptrs[t] + off : nullptr;
    const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m)) & 15) == 0;
    bool bad = false;                                         // a non-finite gradient element (f16 loss-scale overflow)
    if (vec) {
        for (long i = threadIdx.x * 4L; i + 3 < n; i += 1024) {
            float4 pv = *reinterpret_cast<float4*>(p + i);
            float4 gv = *reinterpret_cast<const float4*>(g + i);
            bad |= !(isfinite(gv.x) && isfinite(gv.y) && isfinite(gv.z) && isfinite(gv.w));
            gv.x *= gsc; gv.y *= gsc; gv.z *= gsc; gv.w *= gsc;
            float4 mv = *reinterpret_cast<float4*>(m + i);
            const float4 d = make_float4(gv.x + w * pv.x, gv.y + w * pv.y, gv.z + w * pv.z, gv.w + w * pv.w);
            mv = make_float4(momentum * mv.x + d.x, momentum * mv.y + d.y, momentum * mv.z + d.z, momentum * mv.w + d.w);
            pv.x -= l * mv.x; pv.y -= l * mv.y; pv.z -= l * mv.z; pv.w -= l * mv.w;
            *reinterpret_cast<float4*>(m + i) = mv;
            *reinterpret_cast<float4*>(p + i) = pv;
            if (h) { uint2 o; o.x = H16<F16>::pack2(pv.x, pv.y); o.y = H16<F16>::pack2(pv.z, pv.w); *reinterpret_cast<uint2*>(h + i) = o; }
        }
        for (long i = (n & ~3L) + threadIdx.x; i < n; i += 256) {
            bad |= !isfinite(g[i]);
            const float d = g[i] * gsc + w * p[i];
            const float mv = momentum * m[i] + d;
            m[i] = mv; p[i] -= l * mv;
            if (h) h[i] = H16<F16>::from_f32(p[i]);
        }
    } else {
        for (long i = threadIdx.x; i < n; i += 256) {
            bad |= !isfinite(g[i]);
            const float d = g[i] * gsc + w * p[i];
            const float mv = momentum * m[i] + d;
            m[i] = mv; p[i] -= l * mv;
            if (h) h[i] = H16<F16>::from_f32(p[i]);
        }
    }
    if (nonfinite && bad) atomicOr(nonfinite, 1);
}

// found[0] |= 1 when any gradient element of the step is inf / nan (torch._amp_foreach_non_finite_check_and_unscale_'s
// check, engine/processor.py:94-95 via GradScaler.step): one pass over the gradients BEFORE the update, eight 16-byte loads
// in flight per thread.  sticky (optional) accumulates across steps for FusedSGD.found_inf().
__global__ __launch_bounds__(256) void grad_check_multi_kernel(const float* const* __restrict__ g_ptrs, const int* __restrict__ chunk_tensor,
    const long* __restrict__ chunk_off, const long* __restrict__ numel, int* __restrict__ found, int* __restrict__ sticky)
{
    const int t = chunk_tensor[blockIdx.x];
    if (!g_ptrs[t]) return;
    const long off = chunk_off[blockIdx.x];
    const long n = min((long)kChunk, numel[t] - off);
    const float* __restrict__ g = g_ptrs[t] + off;
    // |x| as integer >= 0x7f800000  <=>  inf or nan
    uint32_t acc = 0u;
    if ((reinterpret_cast<uintptr_t>(g) & 15) == 0) {
        const long n4 = n >> 2;
        for (long i0 = threadIdx.x; i0 < n4; i0 += 256 * 8) {
            uint4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const long i = i0 + j * 256;
                v[j] = i < n4 ? *reinterpret_cast<const uint4*>(g + i * 4) : make_uint4(0u, 0u, 0u, 0u);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                acc |= (uint32_t)((v[j].x & 0x7fffffffu) >= 0x7f800000u) | (uint32_t)((v[j].y & 0x7fffffffu) >= 0x7f800000u) |
                       (uint32_t)((v[j].z & 0x7fffffffu) >= 0x7f800000u) | (uint32_t)((v[j].w & 0x7fffffffu) >= 0x7f800000u);
        }
        for (long i = (n & ~3L) + threadIdx.x; i < n; i += 256) acc |= (uint32_t)((__float_as_uint(g[i]) & 0x7fffffffu) >= 0x7f800000u);
    } else {
        for (long i = threadIdx.x; i < n; i += 256) acc |= (uint32_t)((__float_as_uint(g[i]) & 0x7fffffffu) >= 0x7f800000u);
    }
    if (__builtin_amdgcn_ballot_w64(acc != 0u) != 0ull && (threadIdx.x & 63) == 0) {
        atomicOr(found, 1);
        if (sticky) atomicOr(sticky, 1);
    }
}

// amp.GradScaler.update (torch/amp/grad_scaler.py `_amp_update_scale_`) on device-resident state, one thread:
// found -> scale *= backoff, tracker = 0; else tracker += 1 and at growth_interval scale *= growth, tracker = 0.
// Also refreshes inv_scale = 1 / scale and clears `found` for the next step.
__global__ void scaler_update_kernel(float* scale, float* inv_scale, int* tracker, int* found, float growth, float backoff, int interval)
{
    float s = scale[0];
    if (found[0]) { s *= backoff; tracker[0] = 0; }
    else if (++tracker[0] >= interval) { const float g = s * growth; if (isfinite(g)) s = g; tracker[0] = 0; }
    scale[0] = s;
    inv_scale[0] = 1.f / s;
    found[0] = 0;
}

// split-precision pairs (COMPUTE_DTYPE 'f16x2') of many fp32 tensors in one launch: hi = half(p * scale), lo = half(p * scale - hi),
// over the same chunk tables as sgd_multi_kernel (the forward's weight operands, refreshed after the update)
__global__ __launch_bounds__(256) void split_multi_kernel(const float* const* __restrict__ p_ptrs, uint16_t* const* __restrict__ hi_ptrs,
    uint16_t* const* __restrict__ lo_ptrs, const int* __restrict__ chunk_tensor, const long* __restrict__ chunk_off,
    const long* __restrict__ numel, float scale)
{
    const int t = chunk_tensor[blockIdx.x];
    if (!hi_ptrs[t]) return;
    const long off = chunk_off[blockIdx.x];
    const long n = min((long)kChunk, numel[t] - off);          // (shadowed tensors are >= 2-D GEMM weights: n % 4 == 0, 16-byte aligned)
    const float* __restrict__ p = p_ptrs[t] + off;
    uint16_t* __restrict__ hi = hi_ptrs[t] + off;
    uint16_t* __restrict__ lo = lo_ptrs[t] + off;
    for (long i = threadIdx.x * 4L; i + 3 < n; i += 1024) {
        float4 v = *reinterpret_cast<const float4*>(p + i);
        v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
        uint2 h; h.x = pack_f16x2(v.x, v.y); h.y = pack_f16x2(v.z, v.w);
        const float2_t_ a = H16<true>::unpack2(h.x), b = H16<true>::unpack2(h.y);
        uint2 l; l.x = pack_f16x2(v.x - a.x, v.y - a.y); l.y = pack_f16x2(v.z - b.x, v.w - b.y);
        *reinterpret_cast<uint2*>(hi + i) = h;
        *reinterpret_cast<uint2*>(lo + i) = l;
    }
    for (long i = (n & ~3L) + threadIdx.x; i < n; i += 256) {
        const float v = p[i] * scale;
        const uint16_t h = f32_to_f16(v);
        hi[i] = h; lo[i] = f32_to_f16(v - f16_to_f32(h));
    }
}

// Transposed 16-bit copies of the GEMM weights for the dgrad products: dx = dy W reduces over the ROWS of the nn.Linear
// weight W (N_out, K_in); with W^T (K_in, N_out) stored as well both dgrad operands are k-major and the product runs on
// the plain ds_read_b128 fragment path (2 400 instead of 2 840 cycles per K-tile with ds_read_b64_tr_b16).  One launch
// for all weights: tile table (tensor, tile row, tile col), 64x64 tiles through LDS, 16-byte accesses on both sides.
__global__ __launch_bounds__(256) void transpose_multi_kernel(const uint16_t* const* __restrict__ src, uint16_t* const* __restrict__ dst,
    const int* __restrict__ rows, const int* __restrict__ cols, const int* __restrict__ tile_tensor, const int* __restrict__ tile_r,
    const int* __restrict__ tile_c)
{
    __shared__ uint16_t tile[64][72];                        // padded rows: conflict-free column reads
    const int t = tile_tensor[blockIdx.x];
    const int R = rows[t], C = cols[t];
    const int r0 = tile_r[blockIdx.x] * 64, c0 = tile_c[blockIdx.x] * 64;
    const uint16_t* __restrict__ s = src[t];
    uint16_t* __restrict__ d = dst[t];
    // read 64 rows x 64 cols: thread -> (row = tid/8 + 32*j, 8-element chunk = tid%8)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = (threadIdx.x >> 3) + 32 * j, ch = threadIdx.x & 7;
        const uint4 v = *reinterpret_cast<const uint4*>(s + (long)(r0 + r) * C + c0 + ch * 8);
        *reinterpret_cast<uint4*>(&tile[r][ch * 8]) = v;
    }
    __syncthreads();
    // write 64 rows of the transpose (= columns of the tile), 8 consecutive source rows per thread
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = (threadIdx.x >> 3) + 32 * j, ch = threadIdx.x & 7;
        uint16_t v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = tile[ch * 8 + e][c];
        uint4 o;
        o.x = v[0] | ((uint32_t)v[1] << 16); o.y = v[2] | ((uint32_t)v[3] << 16);
        o.z = v[4] | ((uint32_t)v[5] << 16); o.w = v[6] | ((uint32_t)v[7] << 16);
        *reinterpret_cast<uint4*>(d + (long)(c0 + c) * R + r0 + ch * 8) = o;
    }
}

__device__ __forceinline__ uint64_t mix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
// scales[(l*2 + branch)*B*T + b*T + t] = keep(l,branch,b) / keep_prob_l ; keep = floor(keep_prob + U[0,1))
__global__ void droppath_kernel(const float* __restrict__ rates, int L, long B, int T, uint64_t seed, float* __restrict__ scales)
{
    const long total = (long)L * 2 * B * T;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long s = (uint32_t)e / (uint32_t)T;             // (layer, branch, sample); total < 2^32: host check, 32-bit division
        const int l = (int)((uint32_t)s / (uint32_t)(2 * B));
        const float keep_prob = 1.f - rates[l];
        const uint64_t r = mix64(seed * 0x100000001B3ull + (uint64_t)s);
        const float u = (float)(r >> 40) * (1.f / 16777216.f);
        scales[e] = floorf(keep_prob + u) / keep_prob;
    }
}

// seed from device memory, so that a captured hipGraph draws new masks on every replay: state[0] is read by the
// kernel above (through `seed_dev`) and advanced by this one afterwards
__global__ void droppath_dev_kernel(const float* __restrict__ rates, int L, long B, int T, const long* __restrict__ state,
                                    float* __restrict__ scales)
{
    const uint64_t seed = (uint64_t)state[0];
    const long total = (long)L * 2 * B * T;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long s = (uint32_t)e / (uint32_t)T;
        const int l = (int)((uint32_t)s / (uint32_t)(2 * B));
        const float keep_prob = 1.f - rates[l];
        const uint64_t r = mix64(seed * 0x100000001B3ull + (uint64_t)s);
        const float u = (float)(r >> 40) * (1.f / 16777216.f);
        scales[e] = floorf(keep_prob + u) / keep_prob;
    }
}
__global__ void advance_state_kernel(long* state) { state[0] += 1; }

// torch.optim.AdamW (solver/make_optimizer.py:23-24: OPTIMIZER_NAME 'AdamW', per-parameter lr / weight decay groups, betas (0.9, 0.999),
// eps 1e-8, no amsgrad) over the same chunk tables as sgd_multi_kernel, the operations of torch's single-tensor path in its order:
//   p *= 1 - lr * wd ; m = lerp(m, g, 1 - b1) ; v = b2 * v + (1 - b2) g g ; p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// t lives in device memory (step[0], advanced by adam_advance_kernel before the update: hipGraph-replay safe).  Overflow skip,
// loss-scale and 16-bit shadow as sgd_multi_kernel; a skipped step does not advance t (adam_advance_kernel reads the flag too).
template <bool F16>
__global__ __launch_bounds__(256) void adamw_multi_kernel(float* const* __restrict__ p_ptrs, const float* const* __restrict__ g_ptrs,
    float* const* __restrict__ m_ptrs, float* const* __restrict__ v_ptrs, const int* __restrict__ chunk_tensor,
    const long* __restrict__ chunk_off, const long* __restrict__ numel, const float* __restrict__ lr, const float* __restrict__ wd,
    double beta1d, double beta2d, float eps, const float* __restrict__ step, uint16_t* const* __restrict__ h_ptrs,
    int* __restrict__ nonfinite, const float* __restrict__ inv_scale_dev, const int* __restrict__ skip)
{
    if (skip && *skip) return;
    const float gsc = inv_scale_dev ? *inv_scale_dev : 1.f;
    const int t = chunk_tensor[blockIdx.x];
    if (!g_ptrs[t]) return;
    const long off = chunk_off[blockIdx.x];
    const long n = min((long)kChunk, numel[t] - off);
    float* __restrict__ p = p_ptrs[t] + off;
    const float* __restrict__ g = g_ptrs[t] + off;
    float* __restrict__ m = m_ptrs[t] + off;
    float* __restrict__ v = v_ptrs[t] + off;
    uint16_t* __restrict__ h = (h_ptrs && h_ptrs[t]) ? h_ptrs[t] + off : nullptr;
    const float l = lr[t], decay = 1.f - l * wd[t];
    // the scalar factors in double, as torch computes them on the host (1 - 0.999f differs from 0.001 by 1.3e-5: too coarse in float)
    const double tt = (double)step[0];
    const float beta1 = (float)beta1d, beta2 = (float)beta2d;
    const float omb1 = (float)(1.0 - beta1d), omb2 = (float)(1.0 - beta2d);
    const float bc2s = (float)sqrt(1.0 - pow(beta2d, tt));
    const float step_size = (float)((double)l / (1.0 - pow(beta1d, tt)));
    bool bad = false;
    for (long i = threadIdx.x; i < n; i += 256) {
        const float gr = g[i];
        bad |= !isfinite(gr);
        const float gv = gr * gsc;
        float pv = p[i] * decay;
        float mv = m[i];
        mv = mv + omb1 * (gv - mv);                            // lerp_(grad, 1 - beta1)
        const float vv = beta2 * v[i] + omb2 * (gv * gv);       // mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
        const float denom = sqrtf(vv) / bc2s + eps;
        pv -= step_size * (mv / denom);
        m[i] = mv; v[i] = vv; p[i] = pv;
        if (h) h[i] = H16<F16>::from_f32(pv);
    }
    if (nonfinite && __builtin_amdgcn_ballot_w64(bad) != 0ull && (threadIdx.x & 63) == 0) atomicOr(nonfinite, 1);
}
__global__ void adam_advance_kernel(float* step, const int* skip) { if (!(skip && *skip)) step[0] += 1.f; }

// Stochastic-depth COMPACTION plan (round 6): for every (block, branch) the token rows ordered live samples first - a sample
// whose draw is 0 contributes nothing to the branch (vit_pytorch.py:66-68: x.div(keep_prob) * 0) and gets no gradient through
// it, so the branch's LayerNorm / products / their backward run on the live prefix only.  One workgroup per (block, branch):
//   perm[r]  = slot of token row r in the compacted order (live samples keep their relative order, dropped samples follow)
//   inv[c]   = token row of slot c;   live = number of live ROWS (live samples x T).
// Derived from the scales tensor itself (scale == 0 <=> dropped), so teacher-forced masks take the same path.
__global__ __launch_bounds__(256) void droppath_plan_kernel(const float* __restrict__ scales, long B, int T, int* __restrict__ perm,
                                                            int* __restrict__ inv, int* __restrict__ live)
{
    extern __shared__ int slot[];                              // [B] compacted sample position
    __shared__ int part[256];                                  // live samples in each thread's chunk -> exclusive prefix
    const long u = blockIdx.x;                                 // (block, branch) unit
    const float* sc = scales + u * B * T;
    const int per = (int)((B + 255) / 256);                    // consecutive samples per thread (order-preserving scan)
    const long s0 = (long)threadIdx.x * per, s1 = min(B, s0 + per);
    int mine = 0;
    for (long s = s0; s < s1; ++s) { const int k = sc[s * T] != 0.f ? 1 : 0; slot[s] = k; mine += k; }
    part[threadIdx.x] = mine;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {                  // Hillis-Steele inclusive scan over the 256 chunk counts
        const int v = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    const int nl = part[255];
    int pl = part[threadIdx.x] - mine;                         // live samples before this chunk
    int pd = nl + (int)s0 - pl;                                // dropped samples before it, behind all live ones
    for (long s = s0; s < s1; ++s) { const int k = slot[s]; slot[s] = k ? pl++ : pd++; }
    if (threadIdx.x == 0 && blockIdx.y == 0) live[u] = nl * T;
    __syncthreads();
    int* pu = perm + u * B * T;
    int* iu = inv + u * B * T;
    // (every workgroup of a unit repeats the scan - a few hundred samples - and fills its share of the rows)
    const uint32_t rows = (uint32_t)(B * T);                   // (< 2^31: checked by the host; 32-bit division)
    for (uint32_t e = blockIdx.y * blockDim.x + threadIdx.x; e < rows; e += gridDim.y * blockDim.x) {
        const uint32_t s = e / (uint32_t)T;
        const int tok = (int)(e - s * (uint32_t)T);
        const int c = slot[s] * T + tok;
        pu[e] = c;
        iu[c] = (int)e;
    }
}

}  // namespace

extern "C" int editor_sgd_multi(float* const* p_ptrs, const float* const* g_ptrs, float* const* m_ptrs,
    const int* chunk_tensor, const long* chunk_off, const long* numel, const float* lr, const float* wd, float momentum,
    long nchunks, uint16_t* const* h_ptrs, int shadow_dtype, int* nonfinite, const float* inv_scale, const int* skip,
    hipStream_t stream)
{
    if (nchunks < 1) return 0;
    if (h_ptrs && shadow_dtype != 1 && shadow_dtype != 2) return (int)hipErrorInvalidValue;
    if (shadow_dtype == 2)
        hipLaunchKernelGGL(sgd_multi_kernel<true>, dim3((unsigned)nchunks), dim3(256), 0, stream, p_ptrs, g_ptrs, m_ptrs,
                           chunk_tensor, chunk_off, numel, lr, wd, momentum, h_ptrs, nonfinite, inv_scale, skip);
    else
        hipLaunchKernelGGL(sgd_multi_kernel<false>, dim3((unsigned)nchunks), dim3(256), 0, stream, p_ptrs, g_ptrs, m_ptrs,
                           chunk_tensor, chunk_off, numel, lr, wd, momentum, h_ptrs, nonfinite, inv_scale, skip);
    EDITOR_LAUNCH_CHECK();
    return 0;
}

extern "C" int editor_grad_check_multi(const float* const* g_ptrs, const int* chunk_tensor, const long* chunk_off,
    const long* numel, long nchunks, int* found, int* sticky, hipStream_t stream)
{
    if (nchunks < 1) return 0;
    if (!found) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(grad_check_multi_kernel, dim3((unsigned)nchunks), dim3(256), 0, stream, g_ptrs, chunk_tensor, chunk_off,
                       numel, found, sticky);
    EDITOR_LAUNCH_CHECK();
    return 0;
}

extern "C" int editor_scaler_update(float* scale, float* inv_scale, int* tracker, int* found, float growth, float backoff,
                                    int interval, hipStream_t stream)
{
    if (!scale || !inv_scale || !tracker || !found || interval < 1) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(scaler_update_kernel, dim3(1), dim3(1), 0, stream, scale, inv_scale, tracker, found, growth, backoff, interval);
    EDITOR_LAUNCH_CHECK();
    return 0;
}

extern "C" int editor_split_multi(const float* const* p_ptrs, uint16_t* const* hi_ptrs, uint16_t* const* lo_ptrs,
    const int* chunk_tensor, const long* chunk_off, const long* numel, long nchunks, float scale, hipStream_t stream)
{
    if (nchunks < 1) return 0;
    hipLaunchKernelGGL(split_multi_kernel, dim3((unsigned)nchunks), dim3(256), 0, stream, p_ptrs, hi_ptrs, lo_ptrs, chunk_tensor,
                       chunk_off, numel, scale);
    EDITOR_LAUNCH_CHECK();
    return 0;
}
extern "C" int editor_sgd_chunk_elems(void) { return kChunk; }

extern "C" int editor_adamw_multi(float* const* p_ptrs, const float* const* g_ptrs, float* const* m_ptrs, float* const* v_ptrs,
    const int* chunk_tensor, const long* chunk_off, const long* numel, const float* lr, const float* wd, double beta1, double beta2,
    float eps, float* step, long nchunks, uint16_t* const* h_ptrs, int shadow_dtype, int* nonfinite, const float* inv_scale,
    const int* skip, hipStream_t stream)
{
    if (nchunks < 1) return 0;
    if (!step || !v_ptrs || (h_ptrs && shadow_dtype != 1 && shadow_dtype != 2)) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(1), 0, stream, step, skip);
    if (shadow_dtype == 2)
        hipLaunchKernelGGL(adamw_multi_kernel<true>, dim3((unsigned)nchunks), dim3(256), 0, stream, p_ptrs, g_ptrs, m_ptrs, v_ptrs,
                           chunk_tensor, chunk_off, numel, lr, wd, beta1, beta2, eps, step, h_ptrs, nonfinite, inv_scale, skip);
    else
        hipLaunchKernelGGL(adamw_multi_kernel<false>, dim3((unsigned)nchunks), dim3(256), 0, stream, p_ptrs, g_ptrs, m_ptrs, v_ptrs,
                           chunk_tensor, chunk_off, numel, lr, wd, beta1, beta2, eps, step, h_ptrs, nonfinite, inv_scale, skip);
    EDITOR_LAUNCH_CHECK();
    return 0;
}

extern "C" int editor_transpose_multi(const uint16_t* const* src, uint16_t* const* dst, const int* rows, const int* cols,
    const int* tile_tensor, const int* tile_r, const int* tile_c, long ntiles, hipStream_t stream)
{
    if (ntiles < 1) return 0;
    hipLaunchKernelGGL(transpose_multi_kernel, dim3((unsigned)ntiles), dim3(256), 0, stream, src, dst, rows, cols, tile_tensor,
                       tile_r, tile_c);
    EDITOR_LAUNCH_CHECK();
    return 0;
}

extern "C" int editor_droppath_scales(const float* rates, int L, long B, int T, long seed, float* scales, hipStream_t stream)
{
    const long total = (long)L * 2 * B * T;
    if (total >= (1L << 32)) return (int)hipErrorInvalidValue;
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(droppath_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, rates, L, B, T, (uint64_t)seed, scales);
    EDITOR_LAUNCH_CHECK();
    return 0;
}

extern "C" int editor_droppath_plan(const float* scales, int L, long B, int T, int* perm, int* inv, int* live, hipStream_t stream)
{
    if (L < 1 || B < 1 || T < 1 || B > 12288 || B * T > 0x7FFFFFFFL || !scales || !perm || !inv || !live) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(droppath_plan_kernel, dim3((unsigned)(L * 2), 16), dim3(256), (size_t)B * sizeof(int), stream, scales, B, T,
                       perm, inv, live);
    EDITOR_LAUNCH_CHECK();
    return 0;
}

extern "C" int editor_droppath_scales_dev(const float* rates, int L, long B, int T, long* state, float* scales, hipStream_t stream)
{
    const long total = (long)L * 2 * B * T;
    if (total >= (1L << 32)) return (int)hipErrorInvalidValue;
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(droppath_dev_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, rates, L, B, T, state, scales);
    hipLaunchKernelGGL(advance_state_kernel, dim3(1), dim3(1), 0, stream, state);
    EDITOR_LAUNCH_CHECK();
    return 0;
}
