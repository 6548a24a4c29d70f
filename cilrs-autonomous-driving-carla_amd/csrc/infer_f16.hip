// 16-bit inference trunk: fp16 (BASELINE config 5: batched serving) and bf16 (BASELINE config 3:
// the ResNet-50 variant's "bf16 MFMA path"), BatchNorm folded into the convolutions; reference
// path model/autonomous_drive.py:389-399 under model.eval().
//
// Every trunk convolution after the stem (3x3 and 1x1, stride 1 and 2: BasicBlock and Bottleneck
// alike) runs as an implicit GEMM on v_mfma_f32_32x32x16_f16 / v_mfma_f32_32x32x16_bf16 (16-bit
// operands, fp32 accumulation): activations are 16-bit NHWC, weights
// are re-folded each forward (w * gamma * rstd -> fp16, beta - mean * gamma * rstd -> fp32 bias) by
// one table-driven kernel, the epilogue adds bias (+ the fp16 residual), applies ReLU and writes
// fp16.  The stem (3 input channels) and the heads stay on the fp32 kernels.
//
// Kernel shape: 64x64 output tile per 256-thread block (4 waves x one 32x32 MFMA tile), K-tile 64
// halfs inside one filter tap, global loads two K-tiles ahead through buffer loads (out-of-image
// taps read zeros), LDS double buffer with a 72-half pitch (conflict-free ds_read_b128).
#include "common.h"

namespace cilrs {
namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef _Float16 half_t;      // storage type of the shared 16-bit buffers (bit container)

constexpr int HBM = 64, HBN = 64, HBK = 64, HPITCH = 72;     // halfs

template <typename T> struct Vec8;
template <> struct Vec8<_Float16> { typedef h8 type; };
template <> struct Vec8<__bf16> { typedef b8 type; };
__device__ __forceinline__ f32x16 mfma16(const h8 a, const h8 b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma16(const b8 a, const b8 b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

template <typename T>
__global__ __launch_bounds__(256, 2) void conv_f16_kernel(const ConvF16Args a) {
    typedef typename Vec8<T>::type v8;
    __shared__ __attribute__((aligned(16))) T As[2][HBM * HPITCH];
    __shared__ __attribute__((aligned(16))) T Bs[2][HBN * HPITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int tilesN = a.Cout / HBN;
    const int m0 = (blockIdx.x / tilesN) * HBM, n0 = (blockIdx.x % tilesN) * HBN;
    const int M = a.N * a.Ho * a.Wo, HoWo = a.Ho * a.Wo;
    const int ntaps = a.K * a.K, cin_tiles = a.Cin / HBK;
    const int nt = ntaps * cin_tiles;
    const long Krow = (long)ntaps * a.Cin;

    // per-thread gather rows: rows r0 and r0 + 32, 16-byte chunk kq of the 128-byte K-tile row
    const int kq = tid & 7, r0 = tid >> 3;
    unsigned rowOff[2], rowMask[2], wOff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + r0 + 32 * i;
        rowMask[i] = 0u;
        rowOff[i] = 0u;
        if (m < M) {
            const int n = m / HoWo, rem = m - n * HoWo;
            const int oh = rem / a.Wo, ow = rem - oh * a.Wo;
            const int hb = oh * a.stride - a.pad, wb = ow * a.stride - a.pad;
            rowOff[i] = (unsigned)((((long)(n * a.H + hb) * a.W + wb) * a.Cin + kq * 8) * 2);
            for (int t = 0; t < ntaps; ++t) {
                const int h = hb + t / a.K, w = wb + t % a.K;
                if (h >= 0 && w >= 0 && h < a.H && w < a.W) rowMask[i] |= 1u << t;
            }
        }
        wOff[i] = (unsigned)(((long)(n0 + r0 + 32 * i) * Krow + kq * 8) * 2);
    }
    int tapA_v = 0, tapB_v = 0;              // per-tap byte offsets, one tap per lane
    if (lane < ntaps) {
        tapA_v = ((lane / a.K) * a.W + lane % a.K) * a.Cin * 2;
        tapB_v = lane * a.Cin * 2;
    }
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.x, 0, (int)(unsigned)((size_t)a.N * a.H * a.W * a.Cin * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, -1, 0x00020000);

    int ld_tap = 0, ld_c = 0;
    auto load_tile = [&](f32x4(&ra)[2], f32x4(&rb)[2]) {
        const unsigned toff = (unsigned)__builtin_amdgcn_readlane(tapA_v, ld_tap) +
                              (unsigned)(ld_c * HBK * 2);
        const unsigned koff = (unsigned)__builtin_amdgcn_readlane(tapB_v, ld_tap) +
                              (unsigned)(ld_c * HBK * 2);
        const unsigned bit = 1u << ld_tap;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const unsigned off = (rowMask[i] & bit) ? rowOff[i] + toff : 0xFFFFFFFFu;
            ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)off, 0, 0));
            rb[i] = __builtin_bit_cast(
                f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, (int)(wOff[i] + koff), 0, 0));
        }
        if (++ld_c == cin_tiles) { ld_c = 0; ++ld_tap; }
    };
    auto store_tile = [&](int buf, const f32x4(&ra)[2], const f32x4(&rb)[2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<f32x4*>(&As[buf][(r0 + 32 * i) * HPITCH + kq * 8]) = ra[i];
            *reinterpret_cast<f32x4*>(&Bs[buf][(r0 + 32 * i) * HPITCH + kq * 8]) = rb[i];
        }
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    auto compute = [&](int buf) {
        const T* Ab = &As[buf][(wm * 32 + l31) * HPITCH + lh * 8];
        const T* Bb = &Bs[buf][(wn * 32 + l31) * HPITCH + lh * 8];
#pragma unroll
        for (int q = 0; q < HBK / 16; ++q) {
            const v8 av = *reinterpret_cast<const v8*>(Ab + q * 16);
            const v8 bv = *reinterpret_cast<const v8*>(Bb + q * 16);
            acc = mfma16(av, bv, acc);
        }
    };

    f32x4 ra0[2], rb0[2], ra1[2], rb1[2];
    int it = 0;
    if (nt >= 4) {
        load_tile(ra0, rb0);
        load_tile(ra1, rb1);
        store_tile(0, ra0, rb0);
        __syncthreads();
        for (; it + 3 < nt; it += 2) {
            load_tile(ra0, rb0);
            __builtin_amdgcn_sched_barrier(0);
            compute(0);
            store_tile(1, ra1, rb1);
            __syncthreads();
            load_tile(ra1, rb1);
            __builtin_amdgcn_sched_barrier(0);
            compute(1);
            store_tile(0, ra0, rb0);
            __syncthreads();
        }
    } else {
        if (nt > 0) load_tile(ra0, rb0);
        if (nt > 1) load_tile(ra1, rb1);
        if (nt > 0) store_tile(0, ra0, rb0);
        __syncthreads();
    }
    for (; it < nt; it += 2) {
        if (it + 2 < nt) load_tile(ra0, rb0);
        compute(0);
        if (it + 1 < nt) store_tile(1, ra1, rb1);
        __syncthreads();
        if (it + 1 >= nt) break;
        if (it + 3 < nt) load_tile(ra1, rb1);
        compute(1);
        if (it + 2 < nt) store_tile(0, ra0, rb0);
        __syncthreads();
    }

    // ---- epilogue: + folded-BN bias (+ 16-bit residual), ReLU, 16-bit store --------------------
    const int co = n0 + wn * 32 + l31;
    const float bias = a.bias[co];
    const T* res = reinterpret_cast<const T*>(a.residual);
    T* yout = reinterpret_cast<T*>(a.y);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m >= M) continue;
        float v = acc[r] + bias;
        if (res) v += (float)res[(size_t)m * a.Cout + co];
        if (a.relu) v = fmaxf(v, 0.f);
        yout[(size_t)m * a.Cout + co] = (T)v;
    }
}

// w16[conv][o][k] = half(w[o][k] * scale[o]),  bias[o] = shift[o]   (blockIdx.y = table entry)
template <typename T>
__global__ __launch_bounds__(256) void fold_bn_f16_kernel(const FoldF16Table t,
                                                          const float* __restrict__ params,
                                                          const float* __restrict__ ws,
                                                          T* __restrict__ w16,
                                                          float* __restrict__ bias) {
    const int l = blockIdx.y;
    const float* w = params + t.w[l];
    const float* stats = ws + t.stats[l];         // [mean | rstd | w = gamma*rstd | b]
    const int C = t.cout[l];
    const size_t krow = t.krow[l], n = (size_t)C * krow;
    T* dst = w16 + t.w16[l];
    typedef T h4 __attribute__((ext_vector_type(4)));
    for (size_t i4 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i4 < n / 4;     // krow % 4 == 0
         i4 += (size_t)gridDim.x * blockDim.x) {
        const size_t i = i4 * 4;
        const float sc = stats[2 * C + (int)(i / krow)];
        const f32x4 v = *reinterpret_cast<const f32x4*>(w + i);
        h4 o = {(T)(v[0] * sc), (T)(v[1] * sc), (T)(v[2] * sc), (T)(v[3] * sc)};
        *reinterpret_cast<h4*>(dst + i) = o;
    }
    if (blockIdx.x == 0)
        for (int o = threadIdx.x; o < C; o += blockDim.x) bias[t.bias[l] + o] = stats[3 * C + o];
}

template <typename T>
__global__ __launch_bounds__(256) void f32_to_f16_kernel(const float* __restrict__ x,
                                                         T* __restrict__ y, const size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (size_t)gridDim.x * blockDim.x) {
        typedef T h4 __attribute__((ext_vector_type(4)));
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4);
        h4 o = {(T)v[0], (T)v[1], (T)v[2], (T)v[3]};
        *reinterpret_cast<h4*>(y + i * 4) = o;
    }
}

// AdaptiveAvgPool2d((1,1)) + Flatten of the fp16 feature map -> fp32 combined[:, 0:C]
template <typename T>
__global__ __launch_bounds__(256) void avgpool_f16_kernel(const T* __restrict__ x,
                                                          float* __restrict__ out, const int N,
                                                          const int HW, const int C,
                                                          const int out_ld) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * C) return;
    const int c = i % C, n = i / C;
    float s = 0.f;
    for (int p = 0; p < HW; ++p) s += (float)x[((size_t)n * HW + p) * C + c];
    out[(size_t)n * out_ld + c] = s / (float)HW;
}

}  // namespace

int launch_conv_f16(const ConvF16Args& a, hipStream_t s) {
    CILRS_CHECK(a.Cin % HBK == 0 && a.Cout % HBN == 0 && a.K * a.K <= 16,
                "conv_f16: Cin %% 64, Cout %% 64, <= 16 taps");
    CILRS_CHECK((size_t)a.N * a.H * a.W * a.Cin * 2 < (1ull << 32), "conv_f16: input too large");
    const int M = a.N * a.Ho * a.Wo;
    const int grid = cdiv(M, HBM) * (a.Cout / HBN);
    if (a.bf16) conv_f16_kernel<__bf16><<<grid, 256, 0, s>>>(a);
    else conv_f16_kernel<_Float16><<<grid, 256, 0, s>>>(a);
    CILRS_LAUNCH_CHECK();
    return 0;
}

int launch_fold_bn_f16(const FoldF16Table& t, const float* params, const float* ws, void* w16,
                       float* bias, int bf16, hipStream_t s) {
    if (bf16)
        fold_bn_f16_kernel<__bf16><<<dim3(128, t.n), 256, 0, s>>>(
            t, params, ws, reinterpret_cast<__bf16*>(w16), bias);
    else
        fold_bn_f16_kernel<_Float16><<<dim3(128, t.n), 256, 0, s>>>(
            t, params, ws, reinterpret_cast<_Float16*>(w16), bias);
    CILRS_LAUNCH_CHECK();
    return 0;
}

int launch_f32_to_f16(const float* x, void* y, size_t n, int bf16, hipStream_t s) {
    CILRS_CHECK(n % 4 == 0, "f32_to_f16: n %% 4");
    const size_t n4 = n / 4;
    const int blocks = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
    if (bf16) f32_to_f16_kernel<__bf16><<<blocks, 256, 0, s>>>(x, reinterpret_cast<__bf16*>(y), n4);
    else f32_to_f16_kernel<_Float16><<<blocks, 256, 0, s>>>(x, reinterpret_cast<_Float16*>(y), n4);
    CILRS_LAUNCH_CHECK();
    return 0;
}

int launch_avgpool_f16(const void* x, float* out, int N, int HW, int C, int out_ld, int bf16,
                       hipStream_t s) {
    if (bf16)
        avgpool_f16_kernel<__bf16><<<cdiv(N * C, 256), 256, 0, s>>>(
            reinterpret_cast<const __bf16*>(x), out, N, HW, C, out_ld);
    else
        avgpool_f16_kernel<_Float16><<<cdiv(N * C, 256), 256, 0, s>>>(
            reinterpret_cast<const _Float16*>(x), out, N, HW, C, out_ld);
    CILRS_LAUNCH_CHECK();
    return 0;
}

}  // namespace cilrs
