// Training input pipeline on the device: the reference's albumentations Compose
// (notebook/notebook.ipynb:387-394) + ToTensor/Normalize (:412-414) as ONE kernel over the uint8
// batch.  The reference runs these on two CPU DataLoader workers per frame; here the per-sample
// random parameters are drawn on the host (cilrs_mi355/data.py) and every pixel is one thread.
//
//   RandomBrightnessContrast -> HueSaturationValue -> GaussianBlur -> GaussNoise -> CoarseDropout
//
// albumentations and cv2 are absent from the build image, so each step restates the libraries'
// published behaviour (PARITY UNPINNED against them); the arithmetic below is mirrored operation
// by operation by oracle/augment_oracle.py, which the tests compare against.
#include "common.h"

namespace cilrs {
namespace {

#pragma clang fp contract(off)      // keep a*b+c as two roundings: numpy-identical arithmetic

__device__ __forceinline__ float clip255(float v) { return fminf(fmaxf(v, 0.f), 255.f); }

// stages 1-2 for one pixel: brightness/contrast LUT, then the HSV shifts.  Returns uint8-valued
// floats in rgb[3].
__device__ __forceinline__ void colour_stages(const unsigned char* __restrict__ p,
                                              const cilrs_aug_params& a, float rgb[3]) {
    float c[3] = {(float)p[0], (float)p[1], (float)p[2]};
    if (a.rbc_on) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float t = c[k] * a.alpha;
            t = t + a.beta255;
            c[k] = truncf(clip255(t));           // LUT built in float32, clip, astype(uint8)
        }
    }
    if (a.hsv_on) {
        const float r = c[0], g = c[1], b = c[2];
        const float v = fmaxf(r, fmaxf(g, b)), mn = fminf(r, fminf(g, b));
        const float d = v - mn;
        float s = 0.f, h = 0.f;
        if (v > 0.f) s = d * 255.f / v;
        if (d > 0.f) {
            if (v == r) h = 60.f * (g - b) / d;
            else if (v == g) h = 120.f + 60.f * (b - r) / d;
            else h = 240.f + 60.f * (r - g) / d;
            if (h < 0.f) h = h + 360.f;
        }
        // 8-bit HSV as OpenCV stores it: H in half-degrees [0,180), S, V in [0,255]
        float H = rintf(h * 0.5f);
        if (H >= 180.f) H = H - 180.f;
        const float S = rintf(s), V = v;
        // the three LUTs (hue wraps mod 180, sat / val clip), truncated to uint8
        float H2 = fmodf(H + a.hue, 180.f);
        if (H2 < 0.f) H2 = H2 + 180.f;
        H2 = truncf(H2);
        const float S2 = truncf(clip255(S + a.sat));
        const float V2 = truncf(clip255(V + a.val));
        // HSV -> RGB
        const float hh = H2 * 2.f / 60.f;        // sector coordinate in [0,6)
        const float sec = floorf(hh);
        const float f = hh - sec;
        const float sn = S2 / 255.f;
        const float pp = V2 * (1.f - sn);
        const float qq = V2 * (1.f - sn * f);
        const float tt = V2 * (1.f - sn * (1.f - f));
        const int isec = (int)sec;
        float R, G, B;
        switch (isec) {
            case 0: R = V2; G = tt; B = pp; break;
            case 1: R = qq; G = V2; B = pp; break;
            case 2: R = pp; G = V2; B = tt; break;
            case 3: R = pp; G = qq; B = V2; break;
            case 4: R = tt; G = pp; B = V2; break;
            default: R = V2; G = pp; B = qq; break;
        }
        c[0] = rintf(clip255(R));
        c[1] = rintf(clip255(G));
        c[2] = rintf(clip255(B));
    }
    rgb[0] = c[0]; rgb[1] = c[1]; rgb[2] = c[2];
}

__device__ __forceinline__ int reflect101(int i, const int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return i;
}

__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ __launch_bounds__(256) void augment_u8_kernel(
    const unsigned char* __restrict__ frames, const cilrs_aug_params* __restrict__ params,
    const int B, const int H, const int W, float* __restrict__ out_f32,
    unsigned char* __restrict__ out_u8, const float m0, const float m1, const float m2,
    const float d0, const float d1, const float d2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * H * W) return;
    const int x = i % W, y = (i / W) % H, b = i / (W * H);
    const cilrs_aug_params a = params[b];
    const unsigned char* img = frames + (size_t)b * H * W * 3;
    float c[3];
    if (a.blur_k > 1) {
        // separable Gaussian written as one fixed-order 2-D sum over the colour-adjusted taps
        // (cv2.GaussianBlur semantics: reflect-101 border)
        const int r = min(a.blur_k >> 1, 2);
        float acc[3] = {0.f, 0.f, 0.f};
        for (int dy = -r; dy <= r; ++dy) {
            const int yy = reflect101(y + dy, H);
            float row[3] = {0.f, 0.f, 0.f};
            for (int dx = -r; dx <= r; ++dx) {
                const int xx = reflect101(x + dx, W);
                float t[3];
                colour_stages(img + ((size_t)yy * W + xx) * 3, a, t);
                const float w = a.blur_w[dx < 0 ? -dx : dx];
#pragma unroll
                for (int k = 0; k < 3; ++k) row[k] = row[k] + w * t[k];
            }
            const float w = a.blur_w[dy < 0 ? -dy : dy];
#pragma unroll
            for (int k = 0; k < 3; ++k) acc[k] = acc[k] + w * row[k];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) c[k] = rintf(clip255(acc[k]));
    } else {
        colour_stages(img + ((size_t)y * W + x) * 3, a, c);
    }
    if (a.noise_std255 > 0.f) {
        // per-channel Gaussian noise: Box-Muller on a counter-based hash of (seed, element)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const unsigned long long e = ((unsigned long long)(y * W + x)) * 3ull + k;
            const unsigned long long hsh = splitmix64(a.noise_seed + e * 0xD1B54A32D192ED03ull);
            const float u1 = ((float)(unsigned int)(hsh >> 40) + 1.0f) * (1.0f / 16777216.0f);
            const float u2 = (float)(unsigned int)((hsh >> 8) & 0xFFFFFFu) * (1.0f / 16777216.0f);
            const float rad = sqrtf(-2.0f * logf(u1));
            const float n = rad * cosf(6.2831853071795864f * u2);
            c[k] = rintf(clip255(c[k] + n * a.noise_std255));
        }
    }
    for (int hI = 0; hI < min(a.nholes, 3); ++hI)
        if (y >= a.hole_y0[hI] && y < a.hole_y1[hI] && x >= a.hole_x0[hI] && x < a.hole_x1[hI])
            c[0] = c[1] = c[2] = 0.f;            // CoarseDropout fill = 0
    if (out_u8) {
        unsigned char* o = out_u8 + (size_t)i * 3;
        o[0] = (unsigned char)c[0]; o[1] = (unsigned char)c[1]; o[2] = (unsigned char)c[2];
    }
    if (out_f32) {
        float* o = out_f32 + (size_t)i * 3;      // /255, Normalize (notebook.ipynb:412-414)
        o[0] = (c[0] / 255.0f - m0) / d0;
        o[1] = (c[1] / 255.0f - m1) / d1;
        o[2] = (c[2] / 255.0f - m2) / d2;
    }
}

}  // namespace

int launch_augment_u8(const unsigned char* frames, const cilrs_aug_params* params, int B, int H,
                      int W, float* out_f32, unsigned char* out_u8, hipStream_t s) {
    CILRS_CHECK(B >= 1 && H >= 2 && W >= 2 && (size_t)B * H * W < (1u << 31),
                "augment: bad batch geometry");
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    augment_u8_kernel<<<cdiv(B * H * W, 256), 256, 0, s>>>(frames, params, B, H, W, out_f32, out_u8,
                                                           mean[0], mean[1], mean[2], stdv[0],
                                                           stdv[1], stdv[2]);
    CILRS_LAUNCH_CHECK();
    return 0;
}

}  // namespace cilrs
