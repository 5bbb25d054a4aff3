// Device-side train-time input transform, gfx950: everything of the reference's per-image
// transform chain that follows the decode + resize (data/datasets/make_dataloader.py:245-253) in ONE pass over the batch:
//   RandomHorizontalFlip -> Pad(p, fill 0) -> RandomCrop(H,W) -> ToTensor (/255) -> Normalize(mean,std)
//   -> RandomErasing(mode='pixel', max_count=1)      (make_dataloader.py:55-146)
// The random draws are made on the host in the reference's order (editor_amd/data.py) and arrive as a per-image
// parameter table; the per-pixel N(0,1) fill of the erased rectangle is either supplied (parity tests) or generated on
// the device from a counter-based generator.  HBM-bound byte work: 1 byte read and 4 bytes written per output element.
#include "common.h"
#include "../../include/editor_hip.h"

namespace {

__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

// params per image: {flip, crop_top, crop_left, erase (0/1), e_top, e_left, e_h, e_w}
// in: uint8 (B,H,W,3) interleaved (the decoder's layout); out: fp32 (B,3,H,W)
__global__ void augment_kernel(const uint8_t* __restrict__ in, const int* __restrict__ params, int B, int H, int W, int pad,
    float m0, float m1, float m2, float s0, float s1, float s2, const float* __restrict__ noise, unsigned long long seed,
    float* __restrict__ out)
{
    const long n = (long)B * H * W;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const int x = (int)(e % W), y = (int)((e / W) % H), b = (int)(e / ((long)W * H));
        const int* p = params + b * 8;
        // output pixel (y,x) <- padded-and-flipped image at (crop_top + y, crop_left + x)
        const int py = p[1] + y - pad, px0 = p[2] + x - pad;
        const bool inside = py >= 0 && py < H && px0 >= 0 && px0 < W;
        const int px = p[0] ? W - 1 - px0 : px0;                       // the flip precedes the padding
        const bool erase = p[3] && y >= p[4] && y < p[4] + p[6] && x >= p[5] && x < p[5] + p[7];
        const uint8_t* src = in + (((long)b * H + (inside ? py : 0)) * W + (inside ? px : 0)) * 3;
        const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v;
            if (erase) {
                if (noise) {
                    v = noise[(((long)b * 3 + c) * H + y) * W + x];
                } else {                                               // Box-Muller on two counter-based uniforms
                    const unsigned long long h = mix64(seed ^ (unsigned long long)((((long)b * 3 + c) * H + y) * W + x));
                    const float u1 = ((unsigned)(h >> 40) + 1u) * (1.f / 16777216.f), u2 = (unsigned)((h >> 8) & 0xffffffu) * (1.f / 16777216.f);
                    v = sqrtf(-2.f * logf(u1)) * cosf(6.283185307179586f * u2);
                }
            } else {
                const float t = inside ? __fdiv_rn((float)src[c], 255.f) : 0.f;   // ToTensor; padding is fill 0 BEFORE normalisation
                v = __fdiv_rn(t - mean[c], sd[c]);
            }
            out[(((long)b * 3 + c) * H + y) * W + x] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// T.Resize on decoded uint8 images = Pillow's ImagingResample, 8-bit path (src/libImaging/Resample.c; torchvision 0.14.1
// calls PIL.Image.resize for PIL inputs): separable, horizontal pass then vertical pass, 22-bit fixed-point weights from
// a host-built table (window start, tap count, taps per output coordinate), accumulator seeded with 1 << 21, result
// clip8(acc >> 22); the intermediate image is uint8, as in Pillow.  Integer arithmetic: bit-exact by construction.
// One thread per output pixel (3 channels); HBM-bound byte work - taps of neighbouring outputs overlap in L1/L2.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint8_t clip8(int v) { v >>= 22; return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// out[b, y, xx, c] = sum_k in[b, y, x0[xx] + k, c] * w[xx][k]          (in: (B,H,Win,3), out: (B,H,Wout,3))
__global__ void resize_h_kernel(const uint8_t* __restrict__ in, int B, int H, int Win, int Wout, const int* __restrict__ bounds,
                                const int* __restrict__ kk, int ksize, uint8_t* __restrict__ out)
{
    const long n = (long)B * H * Wout;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const int xx = (int)(e % Wout);
        const long row = e / Wout;                                   // b * H + y
        const int x0 = bounds[2 * xx], cnt = bounds[2 * xx + 1];
        const uint8_t* src = in + (row * Win + x0) * 3;
        const int* w = kk + (long)xx * ksize;
        int a0 = 1 << 21, a1 = 1 << 21, a2 = 1 << 21;
        for (int k = 0; k < cnt; ++k) {
            const int wk = w[k];
            a0 += src[3 * k] * wk; a1 += src[3 * k + 1] * wk; a2 += src[3 * k + 2] * wk;
        }
        uint8_t* o = out + e * 3;
        o[0] = clip8(a0); o[1] = clip8(a1); o[2] = clip8(a2);
    }
}
// out[b, yy, x, c] = sum_k in[b, y0[yy] + k, x, c] * w[yy][k]          (in: (B,Hin,W,3), out: (B,Hout,W,3))
__global__ void resize_v_kernel(const uint8_t* __restrict__ in, int B, int Hin, int Hout, int W, const int* __restrict__ bounds,
                                const int* __restrict__ kk, int ksize, uint8_t* __restrict__ out)
{
    const long n = (long)B * Hout * W;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const int x = (int)(e % W);
        const int yy = (int)((e / W) % Hout);
        const long b = e / ((long)W * Hout);
        const int y0 = bounds[2 * yy], cnt = bounds[2 * yy + 1];
        const uint8_t* src = in + ((b * Hin + y0) * W + x) * 3;
        const int* w = kk + (long)yy * ksize;
        int a0 = 1 << 21, a1 = 1 << 21, a2 = 1 << 21;
        for (int k = 0; k < cnt; ++k) {
            const int wk = w[k];
            const uint8_t* s = src + (long)k * W * 3;
            a0 += s[0] * wk; a1 += s[1] * wk; a2 += s[2] * wk;
        }
        uint8_t* o = out + e * 3;
        o[0] = clip8(a0); o[1] = clip8(a1); o[2] = clip8(a2);
    }
}

}  // namespace

extern "C" int editor_resize_u8(const uint8_t* in, int B, int Hin, int Win, int Hout, int Wout, const int* xbounds,
                                const int* xk, int xksize, const int* ybounds, const int* yk, int yksize, uint8_t* tmp,
                                uint8_t* out, editor_stream_t stream)
{
    if (B <= 0 || Hin <= 0 || Win <= 0 || Hout <= 0 || Wout <= 0 || !in || !out) return (int)hipErrorInvalidValue;
    const bool need_h = Wout != Win, need_v = Hout != Hin;
    if ((need_h && (!xbounds || !xk || xksize < 1)) || (need_v && (!ybounds || !yk || yksize < 1)) || (need_h && need_v && !tmp))
        return (int)hipErrorInvalidValue;
    auto blocks = [](long n) { long b = (n + 255) / 256; return (unsigned)(b > 65536 ? 65536 : b); };
    if (!need_h && !need_v) {
        hipError_t e = hipMemcpyAsync(out, in, (size_t)B * Hin * Win * 3, hipMemcpyDeviceToDevice, (hipStream_t)stream);
        return (int)e;
    }
    const uint8_t* vsrc = in;
    if (need_h) {
        uint8_t* hdst = need_v ? tmp : out;                           // (B, Hin, Wout, 3)
        resize_h_kernel<<<blocks((long)B * Hin * Wout), 256, 0, (hipStream_t)stream>>>(in, B, Hin, Win, Wout, xbounds, xk, xksize, hdst);
        EDITOR_LAUNCH_CHECK();
        vsrc = hdst;
    }
    if (need_v) {
        resize_v_kernel<<<blocks((long)B * Hout * Wout), 256, 0, (hipStream_t)stream>>>(vsrc, B, Hin, Hout, Wout, ybounds, yk, yksize, out);
        EDITOR_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int editor_augment_u8(const uint8_t* in, const int* params, int B, int H, int W, int pad, const float* mean,
                                 const float* stdv, const float* noise, unsigned long long seed, float* out,
                                 editor_stream_t stream)
{
    if (B <= 0 || H <= 0 || W <= 0 || pad < 0 || !mean || !stdv) return 1;
    const long n = (long)B * H * W;
    long blocks = (n + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    augment_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(in, params, B, H, W, pad, mean[0], mean[1], mean[2],
                                                                      stdv[0], stdv[1], stdv[2], noise, seed, out);
    EDITOR_LAUNCH_CHECK();
    return 0;
}
