// 16-bit stem for batched serving (BASELINE.json configs[3] / configs[4]): conv 7x7 / stride 2 / pad 3
// with 3 input channels + folded eval-mode BatchNorm + ReLU, then max-pool 3x3 / stride 2 / pad 1,
// on 16-bit activations (reference: visual_encoder.0-3, model/autonomous_drive.py:366-370 under
// model.eval()).  The fp32 path spends 93 us (B=64, 88x200) / 336 us (B=64, 176x400) on the stem
// convolution alone: 3 channels padded to 4 on the generic-tap fp32 implicit GEMM.
//
// Here the reduction index is laid out k = kh*32 + kw*4 + c with kw padded 7 -> 8 and c 3 -> 4
// (224 = 14 steps of v_mfma_f32_32x32x16_*): one MFMA k-group of a lane (8 values) is TWO adjacent
// input pixels of the channel-padded fp32 image (32 contiguous bytes), and consecutive output
// pixels of a row start 32 bytes apart -- the A operand is read straight from global memory,
// coalesced, converted to 16 bits in registers (no LDS staging, no im2col).  The folded weights
// (64 x 224 halfs = 28 KB) sit in LDS for the life of the block, which walks 128-pixel chunks.
#include "common.h"

namespace cilrs {
namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
template <typename T> struct SVec8;
template <> struct SVec8<_Float16> { typedef h8 type; };
template <> struct SVec8<__bf16> { typedef b8 type; };
__device__ __forceinline__ f32x16 smfma(const h8 a, const h8 b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 smfma(const b8 a, const b8 b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

constexpr int SK = 224;            // 7 x 8 x 4
constexpr int SKP = 232;           // LDS row pitch (halfs): 464 bytes, conflict-free b128 reads

// w16[co][kh][kw8][c4] = (T)(w[co][kh][kw][c] * scale[co]) (zero for kw = 7, c = 3), bias = shift
template <typename T>
__global__ __launch_bounds__(256) void fold_stem_kernel(const float* __restrict__ w,
                                                        const float* __restrict__ stats,
                                                        T* __restrict__ w16,
                                                        float* __restrict__ bias) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 64 * SK) {
        const int co = i / SK, k = i - co * SK;
        const int kh = k >> 5, kw = (k >> 2) & 7, c = k & 3;
        float v = 0.f;
        if (kw < 7 && c < 3) v = w[((co * 7 + kh) * 7 + kw) * 3 + c] * stats[2 * 64 + co];
        w16[i] = (T)v;
    }
    if (i < 64) bias[i] = stats[3 * 64 + i];
}

template <typename T>
__global__ __launch_bounds__(256) void stem_f16_kernel(const float* __restrict__ x4,
                                                       const T* __restrict__ w16,
                                                       const float* __restrict__ bias,
                                                       T* __restrict__ z, const int N, const int H,
                                                       const int W, const int Ho, const int Wo) {
    typedef typename SVec8<T>::type v8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* Ws = reinterpret_cast<T*>(smem_raw);                       // [64][SKP]
    T* Os = Ws + 64 * SKP;                                        // [4 waves][32][64 + 8] staging
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    for (int i = tid; i < 64 * SK / 8; i += 256) {               // 16-byte chunks
        const int co = i / (SK / 8), ch = i - co * (SK / 8);
        *reinterpret_cast<f32x4*>(&Ws[co * SKP + ch * 8]) =
            *reinterpret_cast<const f32x4*>(&w16[co * SK + ch * 8]);
    }
    __syncthreads();
    const int M = N * Ho * Wo;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)x4, 0, (int)(unsigned)((size_t)N * H * W * 16), 0x00020000);
    const float b0 = bias[l31], b1 = bias[32 + l31];
    T* Ow = Os + wave * 32 * 72;
    for (int m0 = blockIdx.x * 128 + wave * 32; m0 < M; m0 += gridDim.x * 128) {
        const int m = min(m0 + l31, M - 1);
        const int n = m / (Ho * Wo), rem = m - n * (Ho * Wo);
        const int oh = rem / Wo, ow = rem - oh * Wo;
        const int hb = oh * 2 - 3, wb = ow * 2 - 3 + 2 * lh;      // this lane's pixel pair base
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
        for (int kh = 0; kh < 7; ++kh) {
            const int ih = hb + kh;
            const bool rok = ih >= 0 && ih < H;
            f32x4 p[4];                                           // pixels of the two steps
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int iw = wb + 4 * (q >> 1) + (q & 1);
                const bool ok = rok && iw >= 0 && iw < W;
                const unsigned off = ok ? (unsigned)((((size_t)(n * H + ih) * W + iw)) * 16)
                                        : 0xFFFFFFFFu;
                p[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
            }
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const f32x4 pa = p[2 * half], pb = p[2 * half + 1];
                const v8 av = {(T)pa[0], (T)pa[1], (T)pa[2], (T)pa[3],
                               (T)pb[0], (T)pb[1], (T)pb[2], (T)pb[3]};
                const int ks = (kh * 2 + half) * 16 + lh * 8;
                const v8 bv0 = *reinterpret_cast<const v8*>(&Ws[l31 * SKP + ks]);
                const v8 bv1 = *reinterpret_cast<const v8*>(&Ws[(32 + l31) * SKP + ks]);
                acc0 = smfma(av, bv0, acc0);
                acc1 = smfma(av, bv1, acc1);
            }
        }
        // + folded shift, ReLU, 16-bit; the 32 x 64 tile goes through this wave's LDS patch so that
        // every lane stores 16 bytes of one output row
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
            Ow[row * 72 + l31] = (T)fmaxf(acc0[r] + b0, 0.f);
            Ow[row * 72 + 32 + l31] = (T)fmaxf(acc1[r] + b1, 0.f);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int row = pass * 8 + (lane >> 3), c8 = (lane & 7) * 8;
            const f32x4 v = *reinterpret_cast<const f32x4*>(&Ow[row * 72 + c8]);
            if (m0 + row < M)
                *reinterpret_cast<f32x4*>(&z[(size_t)(m0 + row) * 64 + c8]) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // before the patch is reused
    }
}

// MaxPool2d(3, 2, 1) on 16-bit NHWC, 8 channels (16 bytes) per thread
template <typename T>
__global__ __launch_bounds__(256) void maxpool_f16_kernel(const T* __restrict__ x,
                                                          T* __restrict__ out, const int N,
                                                          const int H, const int W, const int C,
                                                          const int Ho, const int Wo) {
    typedef typename SVec8<T>::type v8;
    const int c8n = C >> 3;
    const size_t total = (size_t)N * Ho * Wo * c8n;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % c8n);
        size_t p = i / c8n;
        const int ow = (int)(p % Wo); p /= Wo;
        const int oh = (int)(p % Ho);
        const int n = (int)(p / Ho);
        float best[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) best[e] = -INFINITY;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int h = oh * 2 - 1 + kh;
            if (h < 0 || h >= H) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int w = ow * 2 - 1 + kw;
                if (w < 0 || w >= W) continue;
                const v8 v = *reinterpret_cast<const v8*>(x + ((size_t)(n * H + h) * W + w) * C + q * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) best[e] = fmaxf(best[e], (float)v[e]);
            }
        }
        v8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (T)best[e];
        *reinterpret_cast<v8*>(out + i * 8) = o;
    }
}

}  // namespace

int launch_fold_stem_f16(const float* w, const float* stats, void* w16, float* bias, int bf16,
                         hipStream_t s) {
    if (bf16)
        fold_stem_kernel<__bf16><<<cdiv(64 * SK, 256), 256, 0, s>>>(
            w, stats, reinterpret_cast<__bf16*>(w16), bias);
    else
        fold_stem_kernel<_Float16><<<cdiv(64 * SK, 256), 256, 0, s>>>(
            w, stats, reinterpret_cast<_Float16*>(w16), bias);
    CILRS_LAUNCH_CHECK();
    return 0;
}

template <typename T>
static int launch_stem_t(const float* x4, const void* w16, const float* bias, void* z, int N, int H,
                         int W, int Ho, int Wo, hipStream_t s) {
    constexpr size_t lds = (size_t)(64 * SKP + 4 * 32 * 72) * 2;
    const int M = N * Ho * Wo;
    const int blocks = cdiv(M, 128) < 2048 ? cdiv(M, 128) : 2048;
    stem_f16_kernel<T><<<blocks, 256, lds, s>>>(x4, reinterpret_cast<const T*>(w16), bias,
                                                reinterpret_cast<T*>(z), N, H, W, Ho, Wo);
    CILRS_LAUNCH_CHECK();
    return 0;
}

int launch_stem_f16(const float* x4, const void* w16, const float* bias, void* z, int N, int H,
                    int W, int bf16, hipStream_t s) {
    CILRS_CHECK((size_t)N * H * W * 16 < (1ull << 32), "stem_f16: input larger than 4 GB");
    const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
    return bf16 ? launch_stem_t<__bf16>(x4, w16, bias, z, N, H, W, Ho, Wo, s)
                : launch_stem_t<_Float16>(x4, w16, bias, z, N, H, W, Ho, Wo, s);
}

int launch_maxpool_f16(const void* x, void* out, int N, int H, int W, int C, int bf16,
                       hipStream_t s) {
    CILRS_CHECK(C % 8 == 0, "maxpool_f16: C %% 8");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const size_t total = (size_t)N * Ho * Wo * (C / 8);
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    if (bf16)
        maxpool_f16_kernel<__bf16><<<blocks, 256, 0, s>>>(reinterpret_cast<const __bf16*>(x),
                                                         reinterpret_cast<__bf16*>(out), N, H, W,
                                                         C, Ho, Wo);
    else
        maxpool_f16_kernel<_Float16><<<blocks, 256, 0, s>>>(reinterpret_cast<const _Float16*>(x),
                                                           reinterpret_cast<_Float16*>(out), N, H,
                                                           W, C, Ho, Wo);
    CILRS_LAUNCH_CHECK();
    return 0;
}

}  // namespace cilrs
