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

#include <stdlib.h>

namespace cilrs {
namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef _Float16 half_t;      // storage type of the shared 16-bit buffers (bit container)

constexpr int HBK = 64, HPITCH = 72;     // halfs: K-tile, LDS row pitch

template <typename T> struct Vec8;
template <> struct Vec8<_Float16> { typedef h8 type; };
template <> struct Vec8<__bf16> { typedef b8 type; };
__device__ __forceinline__ f32x16 mfma16(const h8 a, const h8 b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma16(const b8 a, const b8 b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// BMxBN output tile per 256-thread block, 4 waves as 2x2, each wave (BM/64) x (BN/64) MFMA tiles:
//   64x64   one 32x32 accumulator per wave, 36 KB LDS, four blocks per CU -- layers with few output
//           pixels (many small blocks beat a handful of big ones);
//   128x128 four accumulators per wave, 72 KB LDS, two blocks per CU -- a quarter of the global
//           loads, LDS writes and address work per MFMA.  Measured on MI355X it LOSES on every
//           layer of both networks at B=64 (ResNet-50 variant 4.74 vs 3.68 ms per batch, ResNet-34
//           fp16 1.48 vs 0.92 ms): these convolutions have 1-18 K-tiles per block and are bound
//           by memory-level parallelism, which two fat blocks per CU halve.  Opt-in only
//           (CILRS_F16_TILE=128).
// TRAIN: fp32 raw result (+ fp32 addend, + BatchNorm column partials) instead of the folded
// 16-bit epilogue, and the stride-2 data-gradient gather (ConvF16Args::up2).
template <typename T, int BM, int BN, bool TRAIN = false>
__global__ __launch_bounds__(256, (BM == 64 ? 2 : 2)) void conv_f16_kernel(const ConvF16Args a) {
    typedef typename Vec8<T>::type v8;
    constexpr int TM = BM / 64, TN = BN / 64;          // MFMA tiles per wave
    constexpr int AP = BM / 32, BP = BN / 32;          // 32-row load passes per K-tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* As = reinterpret_cast<T*>(smem_raw);                        // [2][BM][HPITCH]
    T* Bs = As + 2 * BM * HPITCH;                                  // [2][BN][HPITCH]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int tilesN = a.Cout / BN;
    // XCD-aware block order: consecutive ids go round-robin over the 8 XCDs; give each XCD a
    // contiguous run of tiles so neighbouring tiles (shared rows / weights) share an L2
    int logical;
    {
        const int nwg = (int)gridDim.x, bid = (int)blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    // stride-2 data gradient by output-parity class (up2 == 2): this block's class, its own
    // enumerated grid [N][cHo][cWo] (pixel (2 i + ph, 2 j + pw) of the gradient) and tap subset
    int cls = 0, cHo = a.Ho, cWo = a.Wo, mt_base = 0;
    if (TRAIN && a.up2 == 2) {
        while (cls < 3 && logical >= a.cls_tile_begin[cls + 1]) ++cls;
        logical -= a.cls_tile_begin[cls];
        cHo = a.cls_Ho[cls]; cWo = a.cls_Wo[cls];
        mt_base = a.cls_tile_begin[cls] / tilesN;
    }
    const int m0 = (logical / tilesN) * BM, n0 = (logical % tilesN) * BN;
    const int M = a.N * cHo * cWo, HoWo = cHo * cWo;
    const int ntaps_all = a.K * a.K, cin_tiles = a.Cin / HBK;
    const int ntaps = (TRAIN && a.up2 == 2) ? a.cls_ntaps[cls] : ntaps_all;
    const int nt = ntaps * cin_tiles;
    const long Krow = (long)ntaps_all * a.Cin;

    // per-thread gather rows: rows r0 + 32 i, 16-byte chunk kq of the 128-byte K-tile row
    const int kq = tid & 7, r0 = tid >> 3;
    unsigned rowOff[AP], rowMask[AP], wOff[BP];
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        const int m = m0 + r0 + 32 * i;
        rowMask[i] = 0u;
        rowOff[i] = 0u;
        if (m < M) {
            const int n = m / HoWo, rem = m - n * HoWo;
            const int oh = rem / cWo, ow = rem - oh * cWo;
            if (TRAIN && a.up2 == 2) {
                // base = gathered pixel (oh, ow) of dy; tap t sits (cls_dh, cls_dw) away from it
                rowOff[i] = (unsigned)((((long)(n * a.H + oh) * a.W + ow) * a.Cin + kq * 8) * 2);
                for (int t = 0; t < ntaps; ++t) {
                    const int h2 = oh + a.cls_dh[cls][t], w2 = ow + a.cls_dw[cls][t];
                    if (h2 >= 0 && w2 >= 0 && h2 < a.H && w2 < a.W) rowMask[i] |= 1u << t;
                }
            } else if (TRAIN && a.up2) {
                // parity-matching taps of row (oh + pad) sit at (oh + pad)/2 - kh/2 (see up2)
                const int bh = oh + a.pad, bw = ow + a.pad;
                rowOff[i] = (unsigned)((((long)(n * a.H + (bh >> 1)) * a.W + (bw >> 1)) * a.Cin +
                                        kq * 8) * 2);
                for (int t = 0; t < ntaps; ++t) {
                    const int kh = a.K - 1 - t / a.K, kw = a.K - 1 - t % a.K;
                    const int h2 = bh - kh, w2 = bw - kw;
                    if (h2 >= 0 && w2 >= 0 && !(h2 & 1) && !(w2 & 1) && (h2 >> 1) < a.H &&
                        (w2 >> 1) < a.W)
                        rowMask[i] |= 1u << t;
                }
            } else {
                const int hb = oh * a.stride - a.pad, wb = ow * a.stride - a.pad;
                rowOff[i] = (unsigned)((((long)(n * a.H + hb) * a.W + wb) * a.Cin + kq * 8) * 2);
                for (int t = 0; t < ntaps; ++t) {
                    const int h = hb + t / a.K, w = wb + t % a.K;
                    if (h >= 0 && w >= 0 && h < a.H && w < a.W) rowMask[i] |= 1u << t;
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < BP; ++i)
        wOff[i] = (unsigned)(((long)(n0 + r0 + 32 * i) * Krow + kq * 8) * 2);
    int tapA_v = 0, tapB_v = 0;              // per-tap byte offsets, one tap per lane
    if (lane < ntaps) {
        if (TRAIN && a.up2 == 2) {
            tapA_v = (a.cls_dh[cls][lane] * a.W + a.cls_dw[cls][lane]) * a.Cin * 2;
        } else if (TRAIN && a.up2) {
            const int kh = a.K - 1 - lane / a.K, kw = a.K - 1 - lane % a.K;
            tapA_v = -(((kh >> 1) * a.W + (kw >> 1)) * a.Cin * 2);
        } else {
            tapA_v = ((lane / a.K) * a.W + lane % a.K) * a.Cin * 2;
        }
        tapB_v = ((TRAIN && a.up2 == 2) ? a.cls_tap[cls][lane] : lane) * a.Cin * 2;
    }
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.x, 0, (int)(unsigned)((size_t)a.N * a.H * a.W * a.Cin * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, -1, 0x00020000);

    int ld_tap = 0, ld_c = 0;
    auto load_tile = [&](f32x4(&ra)[AP], f32x4(&rb)[BP]) {
        const unsigned toff = (unsigned)__builtin_amdgcn_readlane(tapA_v, ld_tap) +
                              (unsigned)(ld_c * HBK * 2);
        const unsigned koff = (unsigned)__builtin_amdgcn_readlane(tapB_v, ld_tap) +
                              (unsigned)(ld_c * HBK * 2);
        const unsigned bit = 1u << ld_tap;
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const unsigned off = (rowMask[i] & bit) ? rowOff[i] + toff : 0xFFFFFFFFu;
            ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)off, 0, 0));
        }
#pragma unroll
        for (int i = 0; i < BP; ++i)
            rb[i] = __builtin_bit_cast(
                f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, (int)(wOff[i] + koff), 0, 0));
        if (++ld_c == cin_tiles) { ld_c = 0; ++ld_tap; }
    };
    auto store_tile = [&](int buf, const f32x4(&ra)[AP], const f32x4(&rb)[BP]) {
#pragma unroll
        for (int i = 0; i < AP; ++i)
            *reinterpret_cast<f32x4*>(&As[(buf * BM + r0 + 32 * i) * HPITCH + kq * 8]) = ra[i];
#pragma unroll
        for (int i = 0; i < BP; ++i)
            *reinterpret_cast<f32x4*>(&Bs[(buf * BN + r0 + 32 * i) * HPITCH + kq * 8]) = rb[i];
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto compute = [&](int buf) {
        const T* Ab = &As[(buf * BM + wm * (BM / 2) + l31) * HPITCH + lh * 8];
        const T* Bb = &Bs[(buf * BN + wn * (BN / 2) + l31) * HPITCH + lh * 8];
#pragma unroll
        for (int q = 0; q < HBK / 16; ++q) {
            v8 av[TM], bv[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                av[i] = *reinterpret_cast<const v8*>(Ab + i * 32 * HPITCH + q * 16);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bv[j] = *reinterpret_cast<const v8*>(Bb + j * 32 * HPITCH + q * 16);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = mfma16(av[i], bv[j], acc[i][j]);
        }
    };

    f32x4 ra0[AP], rb0[BP], ra1[AP], rb1[BP];
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

    if constexpr (TRAIN) {
        // ---- training epilogue: BatchNorm column partials of the raw tile (rows >= M are exact
        // zeros; fixed order => deterministic), then the tile through LDS to 16-byte stores:
        // fp32 (y32) or ROUNDED to 16 bits (y16), + an fp32 or 16-bit addend.  With a 16-bit
        // result every derived reduction is taken from the rounded values ----
        float* stage = reinterpret_cast<float*>(smem_raw);
        constexpr int SP = BN + 4;
        const bool out16 = a.y16 != nullptr;
        // rows of the column partials: the launch's M-tiles (all parity classes of a stride-2 gradient)
        const size_t nmt_all = a.up2 == 2 ? (size_t)(a.cls_tile_begin[4] / tilesN)
                                          : (size_t)((M + BM - 1) / BM);
        if (a.bn_partial != nullptr) {
            float* red = stage;                           // [2][BN][2]
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = out16 ? (float)(T)acc[i][j][r] : acc[i][j][r];
                        s1 += v;
                        s2 = fmaf(v, v, s2);
                    }
                s1 += __shfl_xor(s1, 32);
                s2 += __shfl_xor(s2, 32);
                if (lh == 0) {
                    red[(wm * BN + wn * (BN / 2) + j * 32 + l31) * 2] = s1;
                    red[(wm * BN + wn * (BN / 2) + j * 32 + l31) * 2 + 1] = s2;
                }
            }
            __syncthreads();
            if (tid < BN) {
                const float t1 = red[tid * 2] + red[(BN + tid) * 2];
                const float t2 = red[tid * 2 + 1] + red[(BN + tid) * 2 + 1];
                const size_t mt = (size_t)(mt_base + logical / tilesN), nmt = nmt_all;
                a.bn_partial[(size_t)(n0 + tid) * nmt + mt] = t1;
                a.bn_partial[(size_t)(a.Cout + n0 + tid) * nmt + mt] = t2;
            }
            __syncthreads();
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = wn * (BN / 2) + j * 32 + l31;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    stage[row * SP + col] = acc[i][j][r];
                }
        }
        __syncthreads();
        constexpr int TPR = BN / 8, RPP = 256 / TPR;
        const int c8 = (tid % TPR) * 8, rsub = tid / TPR;
        const bool bwd = a.bwd_partial != nullptr;
        f32x4 mlo = {0.f, 0.f, 0.f, 0.f}, mhi = mlo, rlo = mlo, rhi = mlo;
        f32x4 s1lo = mlo, s1hi = mlo, s2lo = mlo, s2hi = mlo;
        if (bwd) {
            mlo = *reinterpret_cast<const f32x4*>(a.bwd_stats + n0 + c8);
            mhi = *reinterpret_cast<const f32x4*>(a.bwd_stats + n0 + c8 + 4);
            rlo = *reinterpret_cast<const f32x4*>(a.bwd_stats + a.Cout + n0 + c8);
            rhi = *reinterpret_cast<const f32x4*>(a.bwd_stats + a.Cout + n0 + c8 + 4);
        }
        const T* add16 = reinterpret_cast<const T*>(a.addend16);
        const T* bz16 = reinterpret_cast<const T*>(a.bwd_z16);
        const T* by16 = reinterpret_cast<const T*>(a.bwd_y16);
        T* y16 = reinterpret_cast<T*>(a.y16);
#pragma unroll
        for (int pass = 0; pass < BM / RPP; ++pass) {
            const int row = pass * RPP + rsub;
            const int m = m0 + row;
            if (m >= M) continue;
            f32x4 lo = *reinterpret_cast<const f32x4*>(&stage[row * SP + c8]);
            f32x4 hi = *reinterpret_cast<const f32x4*>(&stage[row * SP + c8 + 4]);
            size_t pix = (size_t)m;
            if (a.up2 == 2) {           // row m of the class grid -> pixel (2 i + ph, 2 j + pw) of dx
                const int n = m / HoWo, rem = m - n * HoWo;
                const int ci = rem / cWo, cj = rem - ci * cWo;
                pix = ((size_t)n * a.Ho + 2 * ci + a.cls_ph[cls]) * a.Wo + 2 * cj + a.cls_pw[cls];
            }
            const size_t o = pix * a.Cout + n0 + c8;
            if (a.addend32) {
                lo += *reinterpret_cast<const f32x4*>(a.addend32 + o);
                hi += *reinterpret_cast<const f32x4*>(a.addend32 + o + 4);
            }
            if (add16) {
                const v8 av = *reinterpret_cast<const v8*>(add16 + o);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    lo[e] += (float)av[e];
                    hi[e] += (float)av[4 + e];
                }
            }
            if (out16) {
                v8 ov;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ov[e] = (T)lo[e];
                    ov[4 + e] = (T)hi[e];
                }
                *reinterpret_cast<v8*>(y16 + o) = ov;
#pragma unroll
                for (int e = 0; e < 4; ++e) {        // what was stored is what everything downstream sees
                    lo[e] = (float)ov[e];
                    hi[e] = (float)ov[4 + e];
                }
            } else {
                *reinterpret_cast<f32x4*>(a.y32 + o) = lo;
                *reinterpret_cast<f32x4*>(a.y32 + o + 4) = hi;
            }
            if (bwd) {
                f32x4 ylo, yhi, zlo = {1.f, 1.f, 1.f, 1.f}, zhi = zlo;
                if (by16) {
                    const v8 yv = *reinterpret_cast<const v8*>(by16 + o);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        ylo[e] = (float)yv[e];
                        yhi[e] = (float)yv[4 + e];
                    }
                    if (a.bwd_relu) {
                        const v8 zv = *reinterpret_cast<const v8*>(bz16 + o);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            zlo[e] = (float)zv[e];
                            zhi[e] = (float)zv[4 + e];
                        }
                    }
                } else {
                    ylo = *reinterpret_cast<const f32x4*>(a.bwd_y + o);
                    yhi = *reinterpret_cast<const f32x4*>(a.bwd_y + o + 4);
                    if (a.bwd_relu) {
                        zlo = *reinterpret_cast<const f32x4*>(a.bwd_z + o);
                        zhi = *reinterpret_cast<const f32x4*>(a.bwd_z + o + 4);
                    }
                }
                f32x4 glo = lo, ghi = hi;
                if (a.bwd_relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        glo[e] = zlo[e] > 0.f ? glo[e] : 0.f;
                        ghi[e] = zhi[e] > 0.f ? ghi[e] : 0.f;
                    }
                }
                s1lo += glo; s1hi += ghi;
                s2lo += glo * ((ylo - mlo) * rlo);
                s2hi += ghi * ((yhi - mhi) * rhi);
            }
        }
        if (bwd) {
            // column sums over the tile's rows: [RPP row-threads][BN] through LDS, fixed order
            float* red1 = stage + BM * SP;                       // [RPP][BN]
            float* red2 = red1 + RPP * BN;
            *reinterpret_cast<f32x4*>(&red1[rsub * BN + c8]) = s1lo;
            *reinterpret_cast<f32x4*>(&red1[rsub * BN + c8 + 4]) = s1hi;
            *reinterpret_cast<f32x4*>(&red2[rsub * BN + c8]) = s2lo;
            *reinterpret_cast<f32x4*>(&red2[rsub * BN + c8 + 4]) = s2hi;
            __syncthreads();
            if (tid < BN) {
                float t1 = 0.f, t2 = 0.f;
#pragma unroll 8
                for (int r = 0; r < RPP; ++r) {
                    t1 += red1[r * BN + tid];
                    t2 += red2[r * BN + tid];
                }
                const size_t mt = (size_t)(mt_base + logical / tilesN), nmt = nmt_all;
                a.bwd_partial[(size_t)(n0 + tid) * nmt + mt] = t1;
                a.bwd_partial[(size_t)(a.Cout + n0 + tid) * nmt + mt] = t2;
            }
        }
        return;
    }
    // ---- epilogue: + folded-BN bias (+ 16-bit residual), ReLU, 16-bit store --------------------
    // The accumulator layout gives a lane ONE column and 16 rows: stored directly that is 2-byte
    // stores in 64-byte runs (and 2-byte residual loads) -- the wide 1x1 convolutions of the
    // Bottleneck trunk, which write 4x what they read, were bound by exactly that.  So the tile
    // goes through LDS once (fp32, acc + bias): every thread then owns 8 consecutive columns of
    // a row -- one 16-byte residual load, one 16-byte store, whole 128-byte lines per 8 lanes.
    // Same fp32 operation order per element as before: (acc + bias) + residual, ReLU, round.
    constexpr int SP = BN + 4;                               // staging pitch (floats)
    float* stage = reinterpret_cast<float*>(smem_raw);      // [BM][SP]; the K loop ended on a barrier
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = wn * (BN / 2) + j * 32 + l31;
        const float bias = a.bias[n0 + col];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                stage[row * SP + col] = acc[i][j][r] + bias;
            }
        }
    }
    __syncthreads();
    const T* res = reinterpret_cast<const T*>(a.residual);
    T* yout = reinterpret_cast<T*>(a.y);
    constexpr int TPR = BN / 8;                              // threads per output row
    constexpr int RPP = 256 / TPR;                           // rows per pass
    const int c8 = (tid % TPR) * 8, rsub = tid / TPR;
#pragma unroll
    for (int pass = 0; pass < BM / RPP; ++pass) {
        const int row = pass * RPP + rsub;
        const int m = m0 + row;
        if (m >= M) continue;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(&stage[row * SP + c8]);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(&stage[row * SP + c8 + 4]);
        float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        const size_t o = (size_t)m * a.Cout + n0 + c8;
        if (res) {
            const v8 rv = *reinterpret_cast<const v8*>(res + o);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += (float)rv[e];
        }
        v8 ov;
#pragma unroll
        for (int e = 0; e < 8; ++e) ov[e] = (T)(a.relu ? fmaxf(v[e], 0.f) : v[e]);
        *reinterpret_cast<v8*>(yout + o) = ov;
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

template <typename T, int BM, int BN>
static int launch_conv_f16_cfg(const ConvF16Args& a, int M, hipStream_t s) {
    constexpr size_t lds = (size_t)2 * (BM + BN) * HPITCH * 2;
    static bool attr_set = false;
    if (!attr_set) {
        CILRS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16_kernel<T, BM, BN>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    conv_f16_kernel<T, BM, BN><<<cdiv(M, BM) * (a.Cout / BN), 256, lds, s>>>(a);
    CILRS_LAUNCH_CHECK();
    return 0;
}

int launch_conv_f16(const ConvF16Args& a, hipStream_t s) {
    CILRS_CHECK(a.Cin % HBK == 0 && a.Cout % 64 == 0 && a.K * a.K <= 16,
                "conv_f16: Cin %% 64, Cout %% 64, <= 16 taps");
    CILRS_CHECK((size_t)a.N * a.H * a.W * a.Cin * 2 < (1ull << 32), "conv_f16: input too large");
    const int M = a.N * a.Ho * a.Wo;
    // 64x64 tiles everywhere (see the kernel's header); CILRS_F16_TILE=128 forces the big tile
    static const int force = experiment_env("CILRS_F16_TILE", 0);
    const bool big = a.Cout % 128 == 0 && force == 128;
    if (big)
        return a.bf16 ? launch_conv_f16_cfg<__bf16, 128, 128>(a, M, s)
                      : launch_conv_f16_cfg<_Float16, 128, 128>(a, M, s);
    return a.bf16 ? launch_conv_f16_cfg<__bf16, 64, 64>(a, M, s)
                  : launch_conv_f16_cfg<_Float16, 64, 64>(a, M, s);
}

// Parity classes of a stride-2 data gradient.  Forward: y[oh][ow] = sum x[2 oh - p + kh][..] w[kh][..],
// so dx[h][w] only receives taps with kh = (h + p) mod 2 (mod 2), kw likewise: 1 + 2 + 2 + 4 of the 9
// taps of a 3x3 filter over the four classes (h + p, w + p) mod 2 -- 2.25 taps per pixel instead of
// the 9 (seven of them masked to zero) the up-sampled form multiplies; a 1x1 / stride-2 filter has
// one class with one tap and three classes that only copy the addend.  Class (ph, pw) enumerates
// dx pixels (2 i + h0, 2 j + w0), h0 = (ph - p) mod 2; tap kh reads dy row i + (h0 + p - kh) / 2.
// Weights: the transposed, tap-FLIPPED copy wT[ci][K-1-kh][K-1-kw][co] (flipped index into cls_tap).
void conv_f16_up2_classes(ConvF16Args& c) {
    const int K = c.K, p = c.pad, tilesN = c.Cout / 64;
    c.up2 = 2;
    c.cls_tile_begin[0] = 0;
    for (int cls = 0; cls < 4; ++cls) {
        const int ph = cls >> 1, pw = cls & 1;                 // (h + p) mod 2, (w + p) mod 2
        const int h0 = ((ph - p) % 2 + 2) % 2, w0 = ((pw - p) % 2 + 2) % 2;
        c.cls_ph[cls] = h0; c.cls_pw[cls] = w0;
        c.cls_Ho[cls] = c.Ho > h0 ? (c.Ho - h0 + 1) / 2 : 0;
        c.cls_Wo[cls] = c.Wo > w0 ? (c.Wo - w0 + 1) / 2 : 0;
        int nt = 0;
        for (int kh = ph; kh < K; kh += 2)
            for (int kw = pw; kw < K; kw += 2) {
                c.cls_tap[cls][nt] = (K - 1 - kh) * K + (K - 1 - kw);
                c.cls_dh[cls][nt] = (h0 + p - kh) / 2;          // exact: h0 + p - kh is even
                c.cls_dw[cls][nt] = (w0 + p - kw) / 2;
                ++nt;
            }
        c.cls_ntaps[cls] = nt;
        // (a class no tap reaches only copies the addend: nothing to do when that is in place)
        const bool in_place = c.y16 != nullptr && c.addend16 == c.y16 && !c.bwd_partial;
        const int Mc = (nt == 0 && in_place) ? 0 : c.N * c.cls_Ho[cls] * c.cls_Wo[cls];
        c.cls_tile_begin[cls + 1] = c.cls_tile_begin[cls] + cdiv(Mc, 64) * tilesN;
    }
}

template <typename T>
static int launch_conv_f16_train_blocks(const ConvF16Args& a, int blocks, hipStream_t s) {
    constexpr size_t lds = (size_t)2 * (64 + 64) * HPITCH * 2;
    if (once_per_device(reinterpret_cast<const void*>(&conv_f16_kernel<T, 64, 64, true>)))
        CILRS_HIP(hipFuncSetAttribute(
            reinterpret_cast<const void*>(&conv_f16_kernel<T, 64, 64, true>),
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (blocks <= 0) return 0;
    conv_f16_kernel<T, 64, 64, true><<<blocks, 256, lds, s>>>(a);
    CILRS_LAUNCH_CHECK();
    return 0;
}

template <typename T>
static int launch_conv_f16_train_t(const ConvF16Args& a, int M, hipStream_t s) {
    constexpr size_t lds = (size_t)2 * (64 + 64) * HPITCH * 2;
    static bool attr_set = false;
    if (!attr_set) {
        CILRS_HIP(hipFuncSetAttribute(
            reinterpret_cast<const void*>(&conv_f16_kernel<T, 64, 64, true>),
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    conv_f16_kernel<T, 64, 64, true><<<cdiv(M, 64) * (a.Cout / 64), 256, lds, s>>>(a);
    CILRS_LAUNCH_CHECK();
    return 0;
}

// (conv_f16_train_mtiles -- rows of the column partials a launch writes -- lives with the tile plan
//  in conv16.hip)
// may the BatchNorm-backward reductions ride on this launch's epilogue?
bool conv_f16_train_can_fuse_bwd(const ConvF16Args& a) { (void)a; return true; }   // (parity classes included)

int launch_conv_f16_train(const ConvF16Args& a, hipStream_t s) {
    CILRS_CHECK(a.Cin % HBK == 0 && a.Cout % 64 == 0 && a.K * a.K <= 16,
                "conv_f16_train: Cin %% 64, Cout %% 64, <= 16 taps");
    CILRS_CHECK((a.y32 != nullptr) != (a.y16 != nullptr) && ((uintptr_t)a.y32 & 15) == 0 &&
                    ((uintptr_t)a.addend32 & 15) == 0 && ((uintptr_t)a.y16 & 15) == 0 &&
                    ((uintptr_t)a.addend16 & 15) == 0 && !(a.addend32 && a.addend16),
                "conv_f16_train: exactly one of the fp32 / 16-bit outputs, 16-byte aligned");
    CILRS_CHECK((size_t)a.N * a.H * a.W * a.Cin * 2 < (1ull << 32), "conv_f16_train: input too large");
    CILRS_CHECK(!a.up2 || a.stride == 2, "conv_f16_train: up2 is the stride-2 data gradient");
    CILRS_CHECK(!a.bwd_partial ||
                    (a.bwd_stats &&
                     ((a.bwd_y && (a.bwd_z || !a.bwd_relu)) || (a.bwd_y16 && (a.bwd_z16 || !a.bwd_relu)))),
                "conv_f16_train: BatchNorm-backward partials need y / stats (/ z)");
    const int M = a.N * a.Ho * a.Wo;
    if (a.up2 == 1) {           // stride-2 data gradient: ONE launch over the four parity classes
        ConvF16Args c = a;
        conv_f16_up2_classes(c);
        const int blocks = c.cls_tile_begin[4];
        return a.bf16 ? launch_conv_f16_train_blocks<__bf16>(c, blocks, s)
                      : launch_conv_f16_train_blocks<_Float16>(c, blocks, s);
    }
    // large layers: persistent 128-row tiles (conv16.hip); -1 = this launch stays on 64x64
    const int rc = launch_conv16_large(a, s);
    if (rc >= 0) return rc;
    return a.bf16 ? launch_conv_f16_train_t<__bf16>(a, M, s)
                  : launch_conv_f16_train_t<_Float16>(a, M, s);
}

template <typename T>
__global__ __launch_bounds__(256) void transpose_flip_f16_kernel(const float* __restrict__ w,
                                                                 T* __restrict__ wT, const int Cout,
                                                                 const int K, const int Cin) {
    // one thread per (ci, tap', co): reads are strided, the tensors are small (<= 9.4 MB)
    const size_t total = (size_t)Cin * K * K * Cout;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int co = (int)(i % Cout);
        size_t r = i / Cout;
        const int tp = (int)(r % (K * K));
        const int ci = (int)(r / (K * K));
        const int kh = K - 1 - tp / K, kw = K - 1 - tp % K;
        wT[i] = (T)w[(((size_t)co * K + kh) * K + kw) * Cin + ci];
    }
}

int launch_transpose_flip_f16(const float* w, void* wT, int Cout, int K, int Cin, int bf16,
                              hipStream_t s) {
    const size_t total = (size_t)Cin * K * K * Cout;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    if (bf16)
        transpose_flip_f16_kernel<__bf16><<<blocks, 256, 0, s>>>(w, reinterpret_cast<__bf16*>(wT),
                                                                Cout, K, Cin);
    else
        transpose_flip_f16_kernel<_Float16><<<blocks, 256, 0, s>>>(
            w, reinterpret_cast<_Float16*>(wT), Cout, K, Cin);
    CILRS_LAUNCH_CHECK();
    return 0;
}

// every convolution's transposed, tap-flipped 16-bit weights in one launch.  Per filter tap this is
// a [Cout][Cin] -> [Cin][Cout] transpose: 32x32 tiles through LDS, so the fp32 reads run along Cin
// (128-byte rows) and the 16-bit writes along Cout (64-byte rows).  blockIdx.x walks the tiles of
// all convolutions (tile_begin = prefix sums; Cout and Cin are multiples of 32 here).
template <typename T>
__global__ __launch_bounds__(256) void transpose_flip_all_kernel(const TransposeF16Table t,
                                                                 const float* __restrict__ params,
                                                                 T* __restrict__ wT16) {
    __shared__ float tile[32][33];
    int l = 0;
    while (l + 1 < t.n && (int)blockIdx.x >= t.tile_begin[l + 1]) ++l;
    int b = (int)blockIdx.x - t.tile_begin[l];
    const int Cout = t.cout[l], K = t.k[l], Cin = t.cin[l];
    const int tci = Cin / 32, tco = Cout / 32;
    const int ci0 = (b % tci) * 32; b /= tci;
    const int co0 = (b % tco) * 32;
    const int tp = b / tco;                                  // flipped tap index (kh', kw')
    const int kh = K - 1 - tp / K, kw = K - 1 - tp % K;
    const float* w = params + t.w[l];
    T* wT = wT16 + t.wT[l];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
    for (int r = ty; r < 32; r += 8)
        tile[r][tx] = w[(((size_t)(co0 + r) * K + kh) * K + kw) * Cin + ci0 + tx];
    __syncthreads();
#pragma unroll
    for (int r = ty; r < 32; r += 8)
        wT[((size_t)(ci0 + r) * K * K + tp) * Cout + co0 + tx] = (T)tile[tx][r];
}

int launch_transpose_flip_f16_all(const TransposeF16Table& t, const float* params, void* wT16,
                                  int bf16, hipStream_t s) {
    if (t.n <= 0) return 0;
    const int total = t.tile_begin[t.n];
    if (bf16)
        transpose_flip_all_kernel<__bf16><<<total, 256, 0, s>>>(t, params,
                                                               reinterpret_cast<__bf16*>(wT16));
    else
        transpose_flip_all_kernel<_Float16><<<total, 256, 0, s>>>(
            t, params, reinterpret_cast<_Float16*>(wT16));
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
