// The small pieces around the trunk: layout transforms, the narrow linear layers of the heads,
// command-conditioned branch selection, the CILRS losses (forward + gradient in one launch), the
// fused multi-tensor Adam step and the gradient-norm reduction.
//
// Reference semantics:
//   heads / gather      model/autonomous_drive.py:371-399
//   loss (Config B)     notebook/notebook.ipynb:514-527   (5*L1 steer + L1 throttle + L1 brake +
//                                                          0.5*MSE speed)
//   loss (Config A)     configs/train_config.json:30-32   (MSE controls + 0.05*MSE speed)
//   clip + Adam         notebook/notebook.ipynb:533-534, 553-555 (torch.optim.Adam, coupled L2)
//   preprocessing       model/autonomous_drive.py:481-485, 897-902
#include "common.h"

namespace cilrs {

namespace {

int grid1d(size_t total, int cap = 2048) {
    size_t b = (total + 255) / 256;
    if (b > (size_t)cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

// ---- layout transforms -------------------------------------------------------------------------
// f32 NCHW (arbitrary element strides, C = 3) -> dense NHWC with the channel padded to 4 (= 0)
__global__ __launch_bounds__(256) void nchw3_to_nhwc4_kernel(const float* __restrict__ x,
                                                             float* __restrict__ out, const int N,
                                                             const int H, const int W,
                                                             const long sn, const long sc,
                                                             const long sh, const long sw) {
    const size_t total = (size_t)N * H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        size_t p = i;
        const int w = (int)(p % W); p /= W;
        const int h = (int)(p % H);
        const int n = (int)(p / H);
        const float* src = x + n * sn + h * sh + w * sw;
        f32x4 v = {src[0], src[sc], src[2 * sc], 0.f};
        *reinterpret_cast<f32x4*>(out + i * 4) = v;
    }
}

// uint8 RGB HWC frames -> normalised NHWC4: (v/255 - mean)/std  (autonomous_drive.py:898-901)
__global__ __launch_bounds__(256) void u8hwc_to_nhwc4_kernel(const unsigned char* __restrict__ x,
                                                             float* __restrict__ out,
                                                             const size_t npix, const float m0,
                                                             const float m1, const float m2,
                                                             const float s0, const float s1,
                                                             const float s2) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix;
         i += (size_t)gridDim.x * blockDim.x) {
        const unsigned char* p = x + i * 3;
        f32x4 v;
        v[0] = ((float)p[0] / 255.0f - m0) / s0;
        v[1] = ((float)p[1] / 255.0f - m1) / s1;
        v[2] = ((float)p[2] / 255.0f - m2) / s2;
        v[3] = 0.f;
        *reinterpret_cast<f32x4*>(out + i * 4) = v;
    }
}

// Camera frames -> network input in one pass: cv2.resize(frame, (W, H)) [bilinear, uint8 result],
// /255, HWC->CHW, Normalize (autonomous_drive.py:897-902; camera layout :868-872 = 600x800 BGRA
// with the first three bytes of each pixel kept, in that order).  The bilinear step restates
// OpenCV's 8-bit INTER_LINEAR: 11-bit fixed-point coefficients rounded to nearest-even,
// horizontal pass in int, vertical pass ((b*(v>>4))>>16 summed, +2, >>2).
__device__ __forceinline__ void resize_coef(const int d, const double scale, const int ssize,
                                            const bool zero_frac_at_edge, int& s0, int& c0,
                                            int& c1) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int si = (int)floorf(f);
    f -= (float)si;
    if (zero_frac_at_edge) {          // horizontal: out-of-range taps lose their fraction
        if (si < 0) { f = 0.f; si = 0; }
        if (si >= ssize - 1) { f = 0.f; si = ssize - 1; }
    }
    s0 = si;
    c0 = __float2int_rn((1.f - f) * 2048.f);
    c1 = __float2int_rn(f * 2048.f);
}

__global__ __launch_bounds__(256) void camera_to_nhwc4_kernel(
    const unsigned char* __restrict__ src, float* __restrict__ out, const int B, const int sh,
    const int sw, const int pix_stride, const long row_stride, const long frame_stride,
    const int H, const int W, const double scale_y, const double scale_x, const float m0,
    const float m1, const float m2, const float d0, const float d1, const float d2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * H * W) return;
    const int dx = i % W, dy = (i / W) % H, b = i / (W * H);
    int sx, a0, a1, sy, b0, b1;
    resize_coef(dx, scale_x, sw, true, sx, a0, a1);
    resize_coef(dy, scale_y, sh, false, sy, b0, b1);
    const int sx1 = min(sx + 1, sw - 1);
    const int y0 = min(max(sy, 0), sh - 1), y1 = min(max(sy + 1, 0), sh - 1);
    const unsigned char* f = src + (size_t)b * frame_stride;
    const unsigned char* r0 = f + (size_t)y0 * row_stride;
    const unsigned char* r1 = f + (size_t)y1 * row_stride;
    const float mean[3] = {m0, m1, m2}, stdv[3] = {d0, d1, d2};
    f32x4 v;
    v[3] = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int h0 = (int)r0[(size_t)sx * pix_stride + c] * a0 +
                       (int)r0[(size_t)sx1 * pix_stride + c] * a1;
        const int h1 = (int)r1[(size_t)sx * pix_stride + c] * a0 +
                       (int)r1[(size_t)sx1 * pix_stride + c] * a1;
        int u = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
        u = min(max(u, 0), 255);
        v[c] = ((float)u / 255.0f - mean[c]) / stdv[c];
    }
    *reinterpret_cast<f32x4*>(out + (size_t)i * 4) = v;
}

// stem weights: OHWI with I = 3 -> I padded to 4
__global__ void pad_cin3_to_4_kernel(const float* __restrict__ w3, float* __restrict__ w4,
                                     const int n_taps_total) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_taps_total) return;
    f32x4 v = {w3[i * 3], w3[i * 3 + 1], w3[i * 3 + 2], 0.f};
    *reinterpret_cast<f32x4*>(w4 + i * 4) = v;
}

// ---- narrow linear layers (in = 1, out = 3, out = 1) ---------------------------------------------
// in >= 64: one wave per sample row, lanes split K (coalesced), shuffle tree per output (<= 4);
// otherwise one thread per output element.
__global__ __launch_bounds__(256) void linear_small_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ y, const int B, const int in, const int out, const int x_ld,
    const int y_ld, const int relu) {
    if (in >= 64 && out <= 4) {
        const int lane = threadIdx.x & 63;
        const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
        if (b >= B) return;
        const float* xr = x + (size_t)b * x_ld;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int k = lane; k < in; k += 64) {
            const float xv = xr[k];
#pragma unroll
            for (int o = 0; o < 4; ++o)
                if (o < out) acc[o] = fmaf(xv, w[(size_t)o * in + k], acc[o]);
        }
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            float v = acc[o];
#pragma unroll
            for (int sft = 32; sft > 0; sft >>= 1) v += __shfl_xor(v, sft);
            if (lane == 0 && o < out) {
                v += bias[o];
                if (relu) v = fmaxf(v, 0.f);
                y[(size_t)b * y_ld + o] = v;
            }
        }
        return;
    }
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * out) return;
    const int b = i / out, o = i - b * out;
    const float* xr = x + (size_t)b * x_ld;
    const float* wr = w + (size_t)o * in;
    float acc = 0.f;
    for (int k = 0; k < in; ++k) acc = fmaf(xr[k], wr[k], acc);
    acc += bias[o];
    if (relu) acc = fmaxf(acc, 0.f);
    y[(size_t)b * y_ld + o] = acc;
}

// dx[b][i] = sum_o dy[b][o]*W[o][i], optionally masked by (act > 0) * scale
__global__ __launch_bounds__(256) void linear_small_bwd_dx_kernel(
    const float* __restrict__ dy, const float* __restrict__ w, const float* __restrict__ act,
    float* __restrict__ dx, const int B, const int in, const int out, const int dy_ld,
    const int dx_ld, const int act_ld, const float scale) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * in) return;
    const int b = i / in, k = i - b * in;
    float acc = 0.f;
    for (int o = 0; o < out; ++o) acc = fmaf(dy[(size_t)b * dy_ld + o], w[(size_t)o * in + k], acc);
    if (act) acc = act[(size_t)b * act_ld + k] > 0.f ? acc * scale : 0.f;
    dx[(size_t)b * dx_ld + k] = acc;
}

// dW[o][i] = sum_b dy[b][o]*x[b][i]; db[o] = sum_b dy[b][o]   (thread per dW element, + out for db)
__global__ __launch_bounds__(256) void linear_small_bwd_dw_kernel(
    const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dw,
    float* __restrict__ db, const int B, const int in, const int out, const int dy_ld,
    const int x_ld, const int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < out * in) {
        const int o = i / in, k = i - o * in;
        float a4[4] = {0.f, 0.f, 0.f, 0.f};
        int b = 0;
        for (; b + 3 < B; b += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                a4[u] = fmaf(dy[(size_t)(b + u) * dy_ld + o], x[(size_t)(b + u) * x_ld + k], a4[u]);
        }
        for (; b < B; ++b) a4[0] = fmaf(dy[(size_t)b * dy_ld + o], x[(size_t)b * x_ld + k], a4[0]);
        const float acc = (a4[0] + a4[1]) + (a4[2] + a4[3]);
        dw[i] = accumulate ? dw[i] + acc : acc;
    } else if (i < out * in + out) {
        const int o = i - out * in;
        float a4[4] = {0.f, 0.f, 0.f, 0.f};
        int b = 0;
        for (; b + 3 < B; b += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) a4[u] += dy[(size_t)(b + u) * dy_ld + o];
        }
        for (; b < B; ++b) a4[0] += dy[(size_t)b * dy_ld + o];
        const float acc = (a4[0] + a4[1]) + (a4[2] + a4[3]);
        db[o] = accumulate ? db[o] + acc : acc;
    }
}

// db[o] = sum_b dy[b][o] for the wide layers: 64 columns x 4 row groups per block
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ dy,
                                                     float* __restrict__ db, const int B,
                                                     const int out, const int dy_ld,
                                                     const int accumulate) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int o = blockIdx.x * 64 + cl;
    float a4[4] = {0.f, 0.f, 0.f, 0.f};
    if (o < out) {
        int b = g;
        for (; b + 12 < B; b += 16) {
#pragma unroll
            for (int u = 0; u < 4; ++u) a4[u] += dy[(size_t)(b + 4 * u) * dy_ld + o];
        }
        for (; b < B; b += 4) a4[0] += dy[(size_t)b * dy_ld + o];
    }
    red[g][cl] = (a4[0] + a4[1]) + (a4[2] + a4[3]);
    __syncthreads();
    if (g == 0 && o < out) {
        const float acc = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
        db[o] = accumulate ? db[o] + acc : acc;
    }
}

// d[b][c] = act[b][c] > 0 ? d[b][c]*scale : 0
__global__ __launch_bounds__(256) void relu_mask_kernel(float* __restrict__ d,
                                                        const float* __restrict__ act, const int B,
                                                        const int cols, const int d_ld,
                                                        const int act_ld, const float scale) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * cols) return;
    const int b = i / cols, c = i - b * cols;
    const size_t o = (size_t)b * d_ld + c;
    d[o] = act[(size_t)b * act_ld + c] > 0.f ? d[o] * scale : 0.f;
}

// out[b][c] = ((p0+p1)+p2)+... (+ tail for c < cols_tail) -- fixed order, deterministic
__global__ __launch_bounds__(256) void sum_parts_kernel(const SumParts parts, float* __restrict__ out,
                                                        const int B, const int cols,
                                                        const int cols_tail) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * cols) return;
    const int c = i % cols;
    float v = parts.p[0][i];
    for (int k = 1; k < parts.n; ++k) v += parts.p[k][i];
    if (c < cols_tail) v += parts.tail[i];
    out[i] = v;
}

// inverted dropout in place; keep = hash(seed, stream, idx) >= p
__device__ __forceinline__ unsigned int mix32(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    x ^= x >> 31;
    return (unsigned int)(x >> 32);
}
__global__ __launch_bounds__(256) void dropout_kernel(float* __restrict__ a, const int B,
                                                      const int cols, const int ld, const float p,
                                                      const unsigned long long seed,
                                                      const unsigned long long stream) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * cols) return;
    const int b = i / cols, c = i - b * cols;
    const unsigned int r = mix32(seed * 0x2545F4914F6CDD1Dull + (stream << 40) + (unsigned)i);
    const float u = (float)(r >> 8) * (1.0f / 16777216.0f);
    const size_t o = (size_t)b * ld + c;
    a[o] = (u >= p) ? a[o] / (1.0f - p) : 0.f;
}

// ---- small-batch inference heads ---------------------------------------------------------------
// blockIdx.y == 0: AdaptiveAvgPool2d + Flatten of sample blockIdx.x -> combined[b][0:512]
// blockIdx.y == 1: speed encoder Linear(1,128)+ReLU, Linear(128,128)+ReLU -> combined[b][512:640]
// (autonomous_drive.py:366-374, 390-392)
__global__ __launch_bounds__(256) void heads_small_pre_kernel(
    const float* __restrict__ feat, const int HW, const float* __restrict__ speed,
    const float* __restrict__ w0, const float* __restrict__ b0, const float* __restrict__ w1,
    const float* __restrict__ b1, float* __restrict__ combined) {
    const int b = blockIdx.x;
    const int t = threadIdx.x;
    float* out = combined + (size_t)b * 640;
    if (blockIdx.y == 0) {
        if (t >= 128 || feat == nullptr) return;     // feat == NULL: already pooled (fp16 trunk)
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int p = 0; p < HW; ++p)
            s += *reinterpret_cast<const f32x4*>(feat + ((size_t)b * HW + p) * 512 + t * 4);
        *reinterpret_cast<f32x4*>(out + t * 4) = s / (float)HW;
        return;
    }
    __shared__ float s1[128];
    if (t < 128) s1[t] = fmaxf(fmaf(speed[b], w0[t], 0.f) + b0[t], 0.f);
    __syncthreads();
    const int lane = t & 63, wave = t >> 6;
    const float xa = s1[lane], xb = s1[lane + 64];
    for (int o = wave * 32; o < wave * 32 + 32; ++o) {
        const float* wr = w1 + (size_t)o * 128;
        float v = fmaf(xb, wr[lane + 64], xa * wr[lane]);
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) v += __shfl_xor(v, sft);
        if (lane == 0) out[512 + o] = fmaxf(v + b1[o], 0.f);
    }
}

// One wave per output feature: the weight row stays in registers and is applied to every sample
// row that selects it (chain 0: rows whose command picks branch k; chain 1: all rows).
__global__ __launch_bounds__(256) void heads_small_layer_kernel(const HeadsSmallArgs a) {
    const int lane = threadIdx.x & 63;
    int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int chain = o >= a.out[0] ? 1 : 0;
    if (chain) o -= a.out[0];
    if (o >= a.out[chain]) return;
    const int in = a.in[chain];
    const int nq = in >> 2;
    const float* xbase = a.x[chain];
    float* ybase = a.y[chain];
    const int y_ld = a.y_ld[chain];
    const int nk = chain ? 1 : a.ncmd;
    for (int k = 0; k < nk; ++k) {
        const int widx = chain ? a.ncmd : k;
        if (!chain) {
            bool any = false;
            for (int b = 0; b < a.B; ++b) {
                const long long c = a.cmd[b];
                any |= ((c < 0 || c >= a.ncmd) ? 0 : (int)c) == k;
            }
            if (!any) continue;
        }
        const float* wr = a.w[widx] + (size_t)o * in;
        f32x4 wv[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int q = lane + 64 * j;
            wv[j] = q < nq ? *reinterpret_cast<const f32x4*>(wr + q * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const float bias = a.b[widx][o];
        for (int b = 0; b < a.B; ++b) {
            if (!chain) {
                const long long c = a.cmd[b];
                if (((c < 0 || c >= a.ncmd) ? 0 : (int)c) != k) continue;
            }
            const float* xr = xbase + (size_t)b * a.x_ld;
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int q = lane + 64 * j;
                if (q < nq) {
                    const f32x4 xv = *reinterpret_cast<const f32x4*>(xr + q * 4);
                    acc = fmaf(xv[0], wv[j][0], acc);
                    acc = fmaf(xv[1], wv[j][1], acc);
                    acc = fmaf(xv[2], wv[j][2], acc);
                    acc = fmaf(xv[3], wv[j][3], acc);
                }
            }
#pragma unroll
            for (int sft = 32; sft > 0; sft >>= 1) acc += __shfl_xor(acc, sft);
            if (lane == 0) {
                float v = acc + bias;
                if (a.relu) v = fmaxf(v, 0.f);
                ybase[(size_t)b * y_ld + o] = v;
            }
        }
    }
    if (a.status && !chain && o == 0 && lane == 0) {   // torch.gather would raise
        for (int b = 0; b < a.B; ++b)
            if (a.cmd[b] < 0 || a.cmd[b] >= a.ncmd) *a.status = 1;
    }
}

// controls[b][j] = all_out[cmd[b]][b][j]   (torch.stack + gather, autonomous_drive.py:395-398)
__global__ void branch_gather_kernel(const float* __restrict__ all_out,
                                     const long long* __restrict__ cmd,
                                     float* __restrict__ controls, const int B, const int nbranch,
                                     int* __restrict__ status) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * 3) return;
    const int b = i / 3, j = i - b * 3;
    long long k = cmd[b];
    if (k < 0 || k >= nbranch) {            // torch.gather raises; flag it for the host
        if (status) *status = 1;
        k = 0;
    }
    controls[i] = all_out[((size_t)k * B + b) * 4 + j];
}

// d_all[k][b][j] = (cmd[b] == k) ? dcontrols[b][j] : 0    (padded to 4 columns)
__global__ void branch_scatter_kernel(const float* __restrict__ dcontrols,
                                      const long long* __restrict__ cmd,
                                      float* __restrict__ d_all, const int B, const int nbranch) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nbranch * B * 4) return;
    const int j = i & 3;
    const int b = (i >> 2) % B;
    const int k = (i >> 2) / B;
    d_all[i] = (j < 3 && cmd[b] == k) ? dcontrols[b * 3 + j] : 0.f;
}

// ---- evaluation-report accumulators (schema: evaluation_report.json:1-73) ----------------------
// acc[0..31]   channel c (steer, throttle, brake, speed) x {n, Sp, St, Spt, Spp, Stt, S|d|, Sdd}
// acc[32..67]  command k x {n, S|d_steer|, S|d_throttle|, S|d_brake|, Sp, St, Spt, Spp, Stt (steer)}
// acc[68..71]  rows with |d_steer| <= 0.01, 0.02, 0.05, 0.1
// Deterministic: slot j is summed by one wave (lanes stride the rows, shuffle tree) in double.
__device__ __forceinline__ double eval_term(const int j, const float* __restrict__ pc,
                                            const float* __restrict__ tc,
                                            const float* __restrict__ ps,
                                            const float* __restrict__ ts,
                                            const long long* __restrict__ cmd, const int b) {
    if (j < 32) {
        const int c = j >> 3, q = j & 7;
        const double p = c < 3 ? (double)pc[b * 3 + c] : (double)ps[b];
        const double t = c < 3 ? (double)tc[b * 3 + c] : (double)ts[b];
        const double d = p - t;
        switch (q) {
            case 0: return 1.0;
            case 1: return p;
            case 2: return t;
            case 3: return p * t;
            case 4: return p * p;
            case 5: return t * t;
            case 6: return fabs(d);
            default: return d * d;
        }
    }
    if (j < 68) {
        const int k = (j - 32) / 9, q = (j - 32) % 9;
        if (cmd[b] != k) return 0.0;
        const double p = (double)pc[b * 3], t = (double)tc[b * 3];
        switch (q) {
            case 0: return 1.0;
            case 1: return fabs(p - t);
            case 2: return fabs((double)pc[b * 3 + 1] - (double)tc[b * 3 + 1]);
            case 3: return fabs((double)pc[b * 3 + 2] - (double)tc[b * 3 + 2]);
            case 4: return p;
            case 5: return t;
            case 6: return p * t;
            case 7: return p * p;
            default: return t * t;
        }
    }
    const double thr[4] = {0.01, 0.02, 0.05, 0.1};
    return fabs((double)pc[b * 3] - (double)tc[b * 3]) <= thr[j - 68] ? 1.0 : 0.0;
}

__global__ __launch_bounds__(256) void eval_accumulate_kernel(
    const float* __restrict__ pc, const float* __restrict__ tc, const float* __restrict__ ps,
    const float* __restrict__ ts, const long long* __restrict__ cmd, const int B,
    double* __restrict__ acc, float* __restrict__ steer_err) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int j = wave; j < kEvalAccDoubles; j += 4) {
        double s = 0.0;
        for (int b = lane; b < B; b += 64) s += eval_term(j, pc, tc, ps, ts, cmd, b);
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) s += __shfl_xor(s, sft);
        if (lane == 0) acc[j] += s;
    }
    if (steer_err)
        for (int b = threadIdx.x; b < B; b += 256)
            steer_err[b] = (float)fabs((double)pc[b * 3] - (double)tc[b * 3]);
}

// ---- loss forward + gradient, one block --------------------------------------------------------
// kind 0: MSE (Config A)  total = mean_{b,c}(d^2) + w3 * mean_b(ds^2); per-channel = mean_b(d_c^2)
// kind 1: L1  (Config B)  total = sum_c w_c * mean_b|d_c| + w3 * mean_b(ds^2)
// out[6] = total, control, steer, throttle, brake, speed
__global__ __launch_bounds__(256) void loss_kernel(
    const float* __restrict__ pc, const float* __restrict__ tc, const float* __restrict__ ps,
    const float* __restrict__ ts, const int B, const int kind, const float w0, const float w1,
    const float w2, const float w3, const float grad_scale, float* __restrict__ dpc,
    float* __restrict__ dps, float* __restrict__ out) {
    __shared__ double red[4][256];
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    const float invB = 1.0f / (float)B;
    const float wc[3] = {w0, w1, w2};
    for (int b = threadIdx.x; b < B; b += 256) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float d = pc[b * 3 + c] - tc[b * 3 + c];
            float g;
            if (kind == 0) {
                a[c] += (double)d * (double)d;
                g = 2.0f * d / (3.0f * (float)B);
            } else {
                a[c] += (double)fabsf(d);
                g = wc[c] * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * invB;
            }
            if (dpc) dpc[b * 3 + c] = g * grad_scale;
        }
        const float d = ps[b] - ts[b];
        a[3] += (double)d * (double)d;
        if (dps) dps[b] = w3 * 2.0f * d * invB * grad_scale;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) red[c][threadIdx.x] = a[c];
    __syncthreads();
    if (threadIdx.x == 0) {
        double s[4] = {0.0, 0.0, 0.0, 0.0};
        for (int t = 0; t < 256; ++t)
            for (int c = 0; c < 4; ++c) s[c] += red[c][t];
        const float steer = (float)(s[0] / B), thr = (float)(s[1] / B), brk = (float)(s[2] / B);
        const float spd = (float)(s[3] / B);
        float control;
        if (kind == 0) control = (float)((s[0] + s[1] + s[2]) / (3.0 * B));
        else control = w0 * steer + w1 * thr + w2 * brk;
        out[0] = control + w3 * spd;
        out[1] = control;
        out[2] = steer;
        out[3] = thr;
        out[4] = brk;
        out[5] = spd;
    }
}

// ---- gradient norm + Adam ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sqnorm_partial_kernel(const float* __restrict__ g,
                                                             const size_t n4,
                                                             double* __restrict__ partial) {
    __shared__ double red[256];
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(g + i * 4);
        acc += (double)v[0] * v[0] + (double)v[1] * v[1] + (double)v[2] * v[2] +
               (double)v[3] * v[3];
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// out[0] = ||g||_2, out[1] = clip coefficient min(1, max_norm/(norm+1e-6)) (1 if max_norm <= 0)
__global__ void sqnorm_finalize_kernel(const double* __restrict__ partial, const int nblk,
                                       const float max_norm, float* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s = 0.0;
    for (int i = 0; i < nblk; ++i) s += partial[i];
    const float norm = (float)sqrt(s);
    out[0] = norm;
    float coef = 1.0f;
    if (max_norm > 0.f) {
        coef = max_norm / (norm + 1e-6f);
        if (coef > 1.0f) coef = 1.0f;
    }
    out[1] = coef;
}

// torch.optim.Adam (coupled L2), one launch over the flat arena.
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p,
                                                   const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   const size_t n4, const float omb1,
                                                   const float beta2, const float omb2,
                                                   const float eps, const float wd,
                                                   const float neg_step_size,
                                                   const float bc2_sqrt,
                                                   const float* __restrict__ gscale_ptr,
                                                   const float gscale_const) {
    const float gs = gscale_ptr ? gscale_ptr[1] * gscale_const : gscale_const;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (size_t)gridDim.x * blockDim.x) {
        f32x4 pv = *reinterpret_cast<const f32x4*>(p + i * 4);
        f32x4 gv = *reinterpret_cast<const f32x4*>(g + i * 4);
        f32x4 mv = *reinterpret_cast<const f32x4*>(m + i * 4);
        f32x4 vv = *reinterpret_cast<const f32x4*>(v + i * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float gg = gv[e] * gs;                             // clip_grad_norm_ scaling
            gg = gg + wd * pv[e];                              // grad.add(param, alpha=wd)
            mv[e] = mv[e] + omb1 * (gg - mv[e]);               // exp_avg.lerp_(grad, 1-beta1)
            vv[e] = vv[e] * beta2 + (omb2 * gg) * gg;          // mul_(beta2).addcmul_(g,g,1-beta2)
            const float denom = sqrtf(vv[e]) / bc2_sqrt + eps;
            pv[e] = pv[e] + (neg_step_size * mv[e]) / denom;   // addcdiv_(exp_avg, denom, -step)
        }
        *reinterpret_cast<f32x4*>(p + i * 4) = pv;
        *reinterpret_cast<f32x4*>(m + i * 4) = mv;
        *reinterpret_cast<f32x4*>(v + i * 4) = vv;
    }
}

__global__ __launch_bounds__(256) void scale_kernel(float* __restrict__ g, const size_t n4,
                                                    const float* __restrict__ coef_ptr,
                                                    const float c) {
    const float s = coef_ptr ? coef_ptr[1] * c : c;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (size_t)gridDim.x * blockDim.x) {
        f32x4 v = *reinterpret_cast<f32x4*>(g + i * 4);
        *reinterpret_cast<f32x4*>(g + i * 4) = v * s;
    }
}

}  // namespace

int launch_nchw3_to_nhwc4(const float* x, float* out, int N, int H, int W, long sn, long sc,
                          long sh, long sw, hipStream_t s) {
    const size_t total = (size_t)N * H * W;
    nchw3_to_nhwc4_kernel<<<grid1d(total), 256, 0, s>>>(x, out, N, H, W, sn, sc, sh, sw);
    CILRS_LAUNCH_CHECK();
    return 0;
}

int launch_u8hwc_to_nhwc4(const unsigned char* x, float* out, size_t npix, const float* mean,
                          const float* stdv, hipStream_t s) {
    u8hwc_to_nhwc4_kernel<<<grid1d(npix), 256, 0, s>>>(x, out, npix, mean[0], mean[1], mean[2],
                                                       stdv[0], stdv[1], stdv[2]);
    CILRS_LAUNCH_CHECK();
    return 0;
}

int launch_camera_to_nhwc4(const unsigned char* src, float* out, int B, int sh, int sw,
                           int pix_stride, long row_stride, long frame_stride, int H, int W,
                           const float* mean, const float* stdv, hipStream_t s) {
    CILRS_CHECK(sh >= 1 && sw >= 1 && (pix_stride == 3 || pix_stride == 4) &&
                    row_stride >= (long)sw * pix_stride && frame_stride >= (long)sh * row_stride,
                "camera frame: bad geometry / strides");
    // cv::resize: scale = 1 / (dsize / ssize), in double
    const double scale_x = 1.0 / ((double)W / (double)sw);
    const double scale_y = 1.0 / ((double)H / (double)sh);
    camera_to_nhwc4_kernel<<<cdiv(B * H * W, 256), 256, 0, s>>>(
        src, out, B, sh, sw, pix_stride, row_stride, frame_stride, H, W, scale_y, scale_x, mean[0],
        mean[1], mean[2], stdv[0], stdv[1], stdv[2]);
    CILRS_LAUNCH_CHECK();
    return 0;
}

int launch_pad_cin3_to_4(const float* w3, float* w4, int n_taps_total, hipStream_t s) {
    pad_cin3_to_4_kernel<<<cdiv(n_taps_total, 256), 256, 0, s>>>(w3, w4, n_taps_total);
    CILRS_LAUNCH_CHECK();
    return 0;
}

int launch_linear_small_fwd(const float* x, const float* w, const float* bias, float* y, int B,
                            int in, int out, int x_ld, int y_ld, int relu, hipStream_t s) {
    const int blocks = (in >= 64 && out <= 4) ? cdiv(B, 4) : cdiv(B * out, 256);
    linear_small_fwd_kernel<<<blocks, 256, 0, s>>>(x, w, bias, y, B, in, out, x_ld, y_ld, relu);
    CILRS_LAUNCH_CHECK();
    return 0;
}

int launch_linear_small_bwd(const float* dy, const float* x, const float* w, const float* act,
                            float act_scale, float* dx, float* dw, float* db, int B, int in,
                            int out, int dy_ld, int x_ld, int dx_ld, int act_ld, int accumulate,
                            hipStream_t s) {
    if (dx) {
        linear_small_bwd_dx_kernel<<<cdiv(B * in, 256), 256, 0, s>>>(dy, w, act, dx, B, in, out,
                                                                     dy_ld, dx_ld, act_ld,
                                                                     act_scale);
        CILRS_LAUNCH_CHECK();
    }
    if (dw) {
        linear_small_bwd_dw_kernel<<<cdiv(out * in + out, 256), 256, 0, s>>>(
            dy, x, dw, db, B, in, out, dy_ld, x_ld, accumulate);
        CILRS_LAUNCH_CHECK();
    }
    return 0;
}

int launch_colsum(const float* dy, float* db, int B, int out, int dy_ld, int accumulate,
                  hipStream_t s) {
    colsum_kernel<<<cdiv(out, 64), 256, 0, s>>>(dy, db, B, out, dy_ld, accumulate);
    CILRS_LAUNCH_CHECK();
    return 0;
}

int launch_relu_mask(float* d, const float* act, int B, int cols, int d_ld, int act_ld,
                     float scale, hipStream_t s) {
    relu_mask_kernel<<<cdiv(B * cols, 256), 256, 0, s>>>(d, act, B, cols, d_ld, act_ld, scale);
    CILRS_LAUNCH_CHECK();
    return 0;
}

int launch_sum_parts(const SumParts& parts, float* out, int B, int cols, int cols_tail, hipStream_t s) {
    CILRS_CHECK(parts.n >= 1 && parts.n <= kMaxCmd && parts.tail, "sum_parts: bad part list");
    sum_parts_kernel<<<cdiv(B * cols, 256), 256, 0, s>>>(parts, out, B, cols, cols_tail);
    CILRS_LAUNCH_CHECK();
    return 0;
}

int launch_dropout(float* a, int B, int cols, int ld, float p, unsigned long long seed,
                   unsigned long long stream, hipStream_t s) {
    CILRS_CHECK(p >= 0.f && p < 1.f, "dropout: p=%f out of range", (double)p);
    dropout_kernel<<<cdiv(B * cols, 256), 256, 0, s>>>(a, B, cols, ld, p, seed, stream);
    CILRS_LAUNCH_CHECK();
    return 0;
}

int launch_heads_small_pre(const float* feat, int HW, const float* speed, const float* w0,
                           const float* b0, const float* w1, const float* b1, float* combined,
                           int B, hipStream_t s) {
    heads_small_pre_kernel<<<dim3(B, 2), 256, 0, s>>>(feat, HW, speed, w0, b0, w1, b1, combined);
    CILRS_LAUNCH_CHECK();
    return 0;
}

int launch_heads_small_layer(const HeadsSmallArgs& a, hipStream_t s) {
    CILRS_CHECK(a.B >= 1 && a.B <= kHeadsSmallMaxB, "heads_small: batch out of range");
    CILRS_CHECK(a.ncmd >= 1 && a.ncmd <= kMaxCmd, "heads_small: number of commands out of range");
    CILRS_CHECK(a.in[0] % 4 == 0 && a.in[0] <= 768 && a.in[1] % 4 == 0 && a.in[1] <= 768 &&
                    a.x_ld % 4 == 0, "heads_small: bad width");
    heads_small_layer_kernel<<<cdiv(a.out[0] + a.out[1], 4), 256, 0, s>>>(a);
    CILRS_LAUNCH_CHECK();
    return 0;
}

int launch_eval_accumulate(const float* pc, const float* tc, const float* ps, const float* ts,
                           const long long* cmd, int B, double* acc, float* steer_err,
                           hipStream_t s) {
    eval_accumulate_kernel<<<1, 256, 0, s>>>(pc, tc, ps, ts, cmd, B, acc, steer_err);
    CILRS_LAUNCH_CHECK();
    return 0;
}

// the backward pass needs the inputs of the heads: one launch instead of two device-to-device
// hipMemcpyAsync calls (blit kernels with their own fences in the middle of the step)
namespace {
__global__ void keep_head_inputs_kernel(const float* __restrict__ speed, const long long* __restrict__ cmd,
                                        float* __restrict__ speed_dst, long long* __restrict__ cmd_dst,
                                        const int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) {
        speed_dst[i] = speed[i];
        cmd_dst[i] = cmd[i];
    }
}
}  // namespace
int launch_keep_head_inputs(const float* speed, const long long* cmd, float* speed_dst,
                            long long* cmd_dst, int B, hipStream_t s) {
    keep_head_inputs_kernel<<<cdiv(B, 256), 256, 0, s>>>(speed, cmd, speed_dst, cmd_dst, B);
    CILRS_LAUNCH_CHECK();
    return 0;
}

int launch_branch_gather(const float* all_out, const long long* cmd, float* controls, int B,
                         int nbranch, int* status, hipStream_t s) {
    branch_gather_kernel<<<cdiv(B * 3, 256), 256, 0, s>>>(all_out, cmd, controls, B, nbranch,
                                                          status);
    CILRS_LAUNCH_CHECK();
    return 0;
}

int launch_branch_scatter(const float* dcontrols, const long long* cmd, float* d_all, int B,
                          int nbranch, hipStream_t s) {
    branch_scatter_kernel<<<cdiv(nbranch * B * 4, 256), 256, 0, s>>>(dcontrols, cmd, d_all, B,
                                                                    nbranch);
    CILRS_LAUNCH_CHECK();
    return 0;
}

int launch_loss(const float* pc, const float* tc, const float* ps, const float* ts, int B,
                int kind, const float* w, float grad_scale, float* dpc, float* dps, float* out,
                hipStream_t s) {
    CILRS_CHECK(kind == 0 || kind == 1, "loss: kind must be 0 (MSE) or 1 (L1)");
    loss_kernel<<<1, 256, 0, s>>>(pc, tc, ps, ts, B, kind, w[0], w[1], w[2], w[3], grad_scale,
                                  dpc, dps, out);
    CILRS_LAUNCH_CHECK();
    return 0;
}

constexpr int kNormBlocks = 1024;
size_t sqnorm_scratch_bytes() { return kNormBlocks * sizeof(double); }

int launch_grad_sqnorm(const float* g, size_t n, float max_norm, double* partial, float* out,
                       hipStream_t s) {
    CILRS_CHECK(n % 4 == 0, "grad_sqnorm: n must be a multiple of 4");
    const size_t n4 = n / 4;
    const int blocks = grid1d(n4, kNormBlocks);
    sqnorm_partial_kernel<<<blocks, 256, 0, s>>>(g, n4, partial);
    CILRS_LAUNCH_CHECK();
    sqnorm_finalize_kernel<<<1, 64, 0, s>>>(partial, blocks, max_norm, out);
    CILRS_LAUNCH_CHECK();
    return 0;
}

int launch_adam(float* p, const float* g, float* m, float* v, size_t n, double lr, double beta1,
                double beta2, double eps, double wd, long long step, const float* clip_out,
                float gscale, hipStream_t s) {
    CILRS_CHECK(n % 4 == 0, "adam: n must be a multiple of 4");
    CILRS_CHECK(step >= 1, "adam: step must be >= 1");
    // scalar prep in double exactly as torch.optim.adam._single_tensor_adam does in Python
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    const float neg_step_size = (float)(-(lr / bc1));
    const float bc2_sqrt = (float)sqrt(bc2);
    const size_t n4 = n / 4;
    adam_kernel<<<grid1d(n4, 4096), 256, 0, s>>>(p, g, m, v, n4, (float)(1.0 - beta1),
                                                 (float)beta2, (float)(1.0 - beta2), (float)eps,
                                                 (float)wd, neg_step_size, bc2_sqrt, clip_out,
                                                 gscale);
    CILRS_LAUNCH_CHECK();
    return 0;
}

int launch_scale(float* g, size_t n, const float* coef_ptr, float c, hipStream_t s) {
    CILRS_CHECK(n % 4 == 0, "scale: n must be a multiple of 4");
    scale_kernel<<<grid1d(n / 4, 4096), 256, 0, s>>>(g, n / 4, coef_ptr, c);
    CILRS_LAUNCH_CHECK();
    return 0;
}

}  // namespace cilrs
