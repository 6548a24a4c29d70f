// Implicit-GEMM convolution on the gfx950 exact-f32 matrix pipe (v_mfma_f32_32x32x2_f32).
//
// Replaces the cuDNN/ATen convolutions that torchvision's ResNet-34 dispatches inside
// CILRS.visual_encoder (reference model/autonomous_drive.py:365-370; shapes SURVEY.md 2b/8a) and
// the 128..640-wide nn.Linear layers of the heads (autonomous_drive.py:371-387).
//
// GEMM view (no im2col buffer is ever materialised):
//     M = N*Ho*Wo output pixels, N = Cout, K = KH*KW*Cin, A[m][k] gathered on the fly from the
//     NHWC activation, B = OHWI weights.  A K-tile (32 floats) lies inside ONE filter tap, so the
//     gather is a 128-byte contiguous read per output pixel (one full cache line, NHWC).
//
// Block = 256 threads = 4 waves; block tile BM x BN, K-tile 32; LDS double-buffered, register
// staged (global loads for tile t+1 are in flight while tile t is multiplied).  LDS rows are
// K-contiguous with a 4-float pad (pitch 36 floats = 144 B): a ds_read_b128 gives one lane four
// k-values, and r -> 9r mod 16 being a bijection makes every 16-lane read group conflict-free.
// Lane half h of the wave owns k = 8q+4h+e (e = 0..3) of each 8-k group for BOTH operands, so the
// fmaf chain order is fixed and results are run-to-run deterministic.
#include "common.h"

namespace cilrs {

namespace {

constexpr int BK = 32;
constexpr int APITCH = BK + 4;

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    // blocks b and b+8 share an XCD (round-robin dispatch): give each XCD a contiguous chunk of
    // logical tiles so neighbouring tiles (shared input halo / shared weights) hit one L2.
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

template <int BM, int BN, int WM, int WN, bool TAP_UNIFORM, int W_MODE>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvArgs a, const int M,
                                                         const int Ktot, const int KT) {
    static_assert(WM * WN == 4, "4 waves");
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int A_PASSES = BM / 32, B_PASSES = BN / 32;
    constexpr int B_FLOATS = (W_MODE == 0) ? BN * APITCH : BK * (BN + 4);
    constexpr int BPITCH1 = BN + 4;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                       // [2][BM][APITCH]
    float* Bs = smem + 2 * BM * APITCH;     // [2][B_FLOATS]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;

    const int tilesN = a.Cout / BN;
    const int nwg = gridDim.x;
    const int logical = xcd_remap(blockIdx.x, nwg);
    const int m0 = (logical / tilesN) * BM;
    const int n0 = (logical % tilesN) * BN;

    // K-tile range of this split
    int kt_begin = 0, kt_end = KT;
    if (a.splitk > 1) {
        const int per = (KT + a.splitk - 1) / a.splitk;
        kt_begin = blockIdx.z * per;
        kt_end = min(KT, kt_begin + per);
    }

    // ---- per-thread gather rows (fixed for the whole K loop) --------------------------------
    const int kq = tid & 7, r0 = tid >> 3;
    int rowN[A_PASSES], rowH[A_PASSES], rowW[A_PASSES];
    const int HoWo = a.Ho * a.Wo;
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
        const int m = m0 + r0 + 32 * i;
        if (m < M) {
            const int n = m / HoWo, rem = m - n * HoWo;
            const int oh = rem / a.Wo, ow = rem - oh * a.Wo;
            rowN[i] = n;
            rowH[i] = oh * a.stride - a.pad;
            rowW[i] = ow * a.stride - a.pad;
        } else {
            rowN[i] = -1; rowH[i] = 0; rowW[i] = 0;
        }
    }
    const int cin_tiles = TAP_UNIFORM ? (a.Cin / BK) : 1;

    f32x4 ra[A_PASSES], rb[B_PASSES];

    auto load_tile = [&](int kt) {
        // ---- A: gathered activation rows ----
        if constexpr (TAP_UNIFORM) {
            const int tap = kt / cin_tiles;
            const int c0 = (kt - tap * cin_tiles) * BK;
            const int kh = tap / a.KW, kw = tap - kh * a.KW;
#pragma unroll
            for (int i = 0; i < A_PASSES; ++i) {
                const int hup = rowH[i] + kh, wup = rowW[i] + kw;
                bool ok = (rowN[i] >= 0) && (hup >= 0) && (wup >= 0);
                int h = hup, w = wup;
                if (a.dil == 2) {
                    ok = ok && (((hup | wup) & 1) == 0);
                    h = hup >> 1; w = wup >> 1;
                }
                ok = ok && (h < a.H) && (w < a.W);
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (ok) {
                    const float* p = a.x + (size_t)((rowN[i] * a.H + h) * a.W + w) * a.x_ld +
                                     c0 + kq * 4;
                    v = *reinterpret_cast<const f32x4*>(p);
                }
                ra[i] = v;
            }
        } else {
            // generic: each k-quad may sit in a different tap (Cin % 4 == 0, e.g. the stem's
            // channel-padded Cin = 4)
            const int k = kt * BK + kq * 4;
            const int tap = k / a.Cin, ci = k - tap * a.Cin;
            const int kh = tap / a.KW, kw = tap - kh * a.KW;
            const bool kok = k < Ktot;
#pragma unroll
            for (int i = 0; i < A_PASSES; ++i) {
                const int h = rowH[i] + kh, w = rowW[i] + kw;
                const bool ok = kok && (rowN[i] >= 0) && (h >= 0) && (w >= 0) && (h < a.H) &&
                                (w < a.W);
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (ok) {
                    const float* p =
                        a.x + (size_t)((rowN[i] * a.H + h) * a.W + w) * a.x_ld + ci;
                    v = *reinterpret_cast<const f32x4*>(p);
                }
                ra[i] = v;
            }
        }
        // ---- B: weights ----
        if constexpr (W_MODE == 0) {
            const int k = kt * BK + kq * 4;
            const bool kok = TAP_UNIFORM || (k < Ktot);
#pragma unroll
            for (int i = 0; i < B_PASSES; ++i) {
                const int co = n0 + r0 + 32 * i;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (kok) v = *reinterpret_cast<const f32x4*>(a.w + (size_t)co * Ktot + k);
                rb[i] = v;
            }
        } else {
            // dgrad: B[k = forward co][j = forward ci] at the FLIPPED tap
            constexpr int JQ = BN / 4;
            constexpr int ROWS = 256 / JQ;
            const int tap = kt / cin_tiles;
            const int c0 = (kt - tap * cin_tiles) * BK;
            const int ftap = a.KH * a.KW - 1 - tap;
            const int jq = tid % JQ, kr0 = tid / JQ;
            const size_t wrow = (size_t)a.KH * a.KW * a.w_cin;
#pragma unroll
            for (int i = 0; i < B_PASSES; ++i) {
                const int kr = kr0 + ROWS * i;
                rb[i] = *reinterpret_cast<const f32x4*>(
                    a.w + (size_t)(c0 + kr) * wrow + (size_t)ftap * a.w_cin + n0 + jq * 4);
            }
        }
    };

    auto store_tile = [&](int buf) {
        float* Ab = As + buf * BM * APITCH;
        float* Bb = Bs + buf * B_FLOATS;
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i)
            *reinterpret_cast<f32x4*>(Ab + (r0 + 32 * i) * APITCH + kq * 4) = ra[i];
        if constexpr (W_MODE == 0) {
#pragma unroll
            for (int i = 0; i < B_PASSES; ++i)
                *reinterpret_cast<f32x4*>(Bb + (r0 + 32 * i) * APITCH + kq * 4) = rb[i];
        } else {
            constexpr int JQ = BN / 4;
            constexpr int ROWS = 256 / JQ;
            const int jq = tid % JQ, kr0 = tid / JQ;
#pragma unroll
            for (int i = 0; i < B_PASSES; ++i)
                *reinterpret_cast<f32x4*>(Bb + (kr0 + ROWS * i) * BPITCH1 + jq * 4) = rb[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (kt_begin < kt_end) {
        load_tile(kt_begin);
        store_tile(0);
    }
    __syncthreads();

    int buf = 0;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const bool more = (kt + 1 < kt_end);
        if (more) load_tile(kt + 1);

        const float* Ab = As + buf * BM * APITCH + (wm * WTM + l31) * APITCH + lh * 4;
        const float* Bb = Bs + buf * B_FLOATS;
#pragma unroll
        for (int q = 0; q < BK / 8; ++q) {
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * APITCH + q * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (W_MODE == 0) {
                    bf[j] = *reinterpret_cast<const f32x4*>(
                        Bb + (wn * WTN + j * 32 + l31) * APITCH + q * 8 + lh * 4);
                } else {
                    const float* p = Bb + (q * 8 + lh * 4) * BPITCH1 + wn * WTN + j * 32 + l31;
                    bf[j][0] = p[0];
                    bf[j][1] = p[BPITCH1];
                    bf[j][2] = p[2 * BPITCH1];
                    bf[j][3] = p[3 * BPITCH1];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e],
                                                                         acc[i][j], 0, 0, 0);
        }
        if (more) store_tile(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }

    // ---- epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const bool partial = a.splitk > 1;
    float* yout = partial ? a.y + (size_t)blockIdx.z * M * a.y_ld : a.y;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int co = n0 + wn * WTN + j * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int m = m0 + wm * WTM + i * 32 + row;
                if (m < M) {
                    float v = acc[i][j][r];
                    const size_t o = (size_t)m * a.y_ld + co;
                    if (!partial) {
                        if (a.bias) v += a.bias[co];
                        if (a.relu) v = fmaxf(v, 0.f);
                        if (a.mask)
                            v = (a.mask[(size_t)m * a.mask_ld + co] > 0.f) ? v * a.mask_scale
                                                                             : 0.f;
                        if (a.addend) v += a.addend[o];
                    }
                    yout[o] = v;
                }
            }
        }
    }
}

// Sum split-K partials (fixed order => deterministic) and apply the epilogue.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const ConvArgs a, const float* part,
                                                            const int M, const int splits) {
    const int cq = a.Cout >> 2;
    const size_t total = (size_t)M * cq;
    const size_t slab = (size_t)M * a.y_ld;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(idx / cq);
        const int co = (int)(idx - (size_t)m * cq) * 4;
        const size_t o = (size_t)m * a.y_ld + co;
        f32x4 v = *reinterpret_cast<const f32x4*>(part + o);
        for (int s = 1; s < splits; ++s) v += *reinterpret_cast<const f32x4*>(part + s * slab + o);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float t = v[e];
            if (a.bias) t += a.bias[co + e];
            if (a.relu) t = fmaxf(t, 0.f);
            if (a.mask)
                t = (a.mask[(size_t)m * a.mask_ld + co + e] > 0.f) ? t * a.mask_scale : 0.f;
            if (a.addend) t += a.addend[o + e];
            v[e] = t;
        }
        *reinterpret_cast<f32x4*>(a.y + o) = v;
    }
}

template <int BM, int BN, int WM, int WN, bool TU, int MODE>
int launch_cfg(const ConvArgs& a, int M, int Ktot, int KT, hipStream_t s) {
    constexpr int B_FLOATS = (MODE == 0) ? BN * APITCH : BK * (BN + 4);
    constexpr size_t lds = (size_t)(2 * BM * APITCH + 2 * B_FLOATS) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        CILRS_HIP(hipFuncSetAttribute(
            reinterpret_cast<const void*>(&conv_igemm_kernel<BM, BN, WM, WN, TU, MODE>),
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    dim3 grid(cdiv(M, BM) * (a.Cout / BN), 1, a.splitk > 1 ? a.splitk : 1);
    conv_igemm_kernel<BM, BN, WM, WN, TU, MODE><<<grid, 256, lds, s>>>(a, M, Ktot, KT);
    CILRS_LAUNCH_CHECK();
    return 0;
}

// cost model: rounds of blocks over 256 CUs x per-block MFMA work, with a mild penalty for the
// smaller tiles' extra L2 traffic.  Returns the estimated cost (arbitrary units).
double cfg_cost(int M, int Cout, int KT, int BM, int BN, int splitk, double penalty) {
    if (Cout % BN) return 1e30;
    const double blocks = (double)cdiv(M, BM) * (Cout / BN) * splitk;
    const double rounds = (double)((long)((blocks + 255) / 256));
    const double ktiles = (double)cdiv(KT, splitk);
    double cost = rounds * ktiles * BM * BN * penalty;
    // fixed per-block prologue/epilogue ~ 1.5 K-tiles of work
    cost += rounds * 1.5 * BM * BN;
    if (splitk > 1) cost += (double)M * Cout * (splitk + 1) * 0.02 + 30000.0;
    return cost;
}

struct Choice { int cfg; int splitk; };

Choice choose(int M, int Cout, int KT, bool allow_split) {
    static const int bm[3] = {128, 128, 64};
    static const int bn[3] = {128, 64, 64};
    static const double pen[3] = {1.0, 1.06, 1.18};
    static const int splits[6] = {1, 2, 3, 4, 6, 8};
    Choice best{1, 1};
    double bc = 1e30;
    for (int c = 0; c < 3; ++c)
        for (int si = 0; si < 6; ++si) {
            const int sk = splits[si];
            if (sk > 1 && (!allow_split || KT < 4 * sk)) continue;
            const double cost = cfg_cost(M, Cout, KT, bm[c], bn[c], sk, pen[c]);
            if (cost < bc) { bc = cost; best = {c, sk}; }
        }
    return best;
}

}  // namespace

int launch_conv_igemm(const ConvArgs& a_in, hipStream_t s) {
    ConvArgs a = a_in;
    const int M = a.N * a.Ho * a.Wo;
    const int Ktot = a.KH * a.KW * a.Cin;
    const bool uniform = (a.Cin % BK) == 0;
    const int KT = uniform ? a.KH * a.KW * (a.Cin / BK) : cdiv(Ktot, BK);
    CILRS_CHECK(a.Cout % 64 == 0, "conv_igemm: Cout=%d must be a multiple of 64", a.Cout);
    CILRS_CHECK(a.Cin % 4 == 0 && a.x_ld % 4 == 0 && a.y_ld % 4 == 0,
                "conv_igemm: Cin/x_ld/y_ld must be multiples of 4 (%d,%d,%d)", a.Cin, a.x_ld,
                a.y_ld);
    CILRS_CHECK(a.dil == 1 || a.dil == 2, "conv_igemm: dil must be 1 or 2");
    CILRS_CHECK(uniform || (a.w_mode == 0 && a.dil == 1),
                "conv_igemm: the generic-tap path is forward-only, undilated");
    CILRS_CHECK(a.w_mode == 0 || a.w_cin % 4 == 0, "conv_igemm: w_cin must be a multiple of 4");
    CILRS_CHECK(((uintptr_t)a.x & 15) == 0 && ((uintptr_t)a.w & 15) == 0 &&
                    ((uintptr_t)a.y & 15) == 0,
                "conv_igemm: pointers must be 16-byte aligned");
    CILRS_CHECK(M > 0 && (size_t)a.N * a.H * a.W < (1u << 31), "conv_igemm: bad M");

    // ---- tile / split-K choice ----
    const size_t slab = (size_t)M * a.y_ld;
    const bool can_split = a.scratch != nullptr && a.scratch_floats >= 2 * slab && uniform;
    Choice ch = choose(M, a.Cout, KT, can_split);
    if (a.force_splitk > 0) ch.splitk = a.force_splitk;
    if (a.force_cfg >= 0) ch.cfg = a.force_cfg;
    while (ch.splitk > 1 && (size_t)ch.splitk * slab > a.scratch_floats) --ch.splitk;
    CILRS_CHECK(ch.splitk == 1 || (a.scratch && uniform), "conv_igemm: split-K needs scratch");
    CILRS_CHECK(ch.cfg >= 0 && ch.cfg < 3 && a.Cout % (ch.cfg == 0 ? 128 : 64) == 0,
                "conv_igemm: tile config %d does not fit Cout=%d", ch.cfg, a.Cout);
    a.splitk = ch.splitk;
    float* final_y = a.y;
    if (a.splitk > 1) a.y = a.scratch;

    int rc = 1;
#define CILRS_DISPATCH(BM_, BN_)                                                         \
    do {                                                                                 \
        if (a.w_mode == 1) rc = launch_cfg<BM_, BN_, 2, 2, true, 1>(a, M, Ktot, KT, s);  \
        else if (uniform)  rc = launch_cfg<BM_, BN_, 2, 2, true, 0>(a, M, Ktot, KT, s);  \
        else               rc = launch_cfg<BM_, BN_, 2, 2, false, 0>(a, M, Ktot, KT, s); \
    } while (0)
    if (ch.cfg == 0) CILRS_DISPATCH(128, 128);
    else if (ch.cfg == 1) CILRS_DISPATCH(128, 64);
    else CILRS_DISPATCH(64, 64);
#undef CILRS_DISPATCH
    if (rc) return rc;

    if (a.splitk > 1) {
        const float* part = a.scratch;
        a.y = final_y;
        const size_t total = (size_t)M * (a.Cout / 4);
        const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
        splitk_reduce_kernel<<<blocks, 256, 0, s>>>(a, part, M, a.splitk);
        CILRS_LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace cilrs
