// Implicit-GEMM convolution on the gfx950 exact-f32 matrix pipe (v_mfma_f32_32x32x2_f32).
//
// Replaces the cuDNN/ATen convolutions that torchvision's ResNet-34 dispatches inside
// CILRS.visual_encoder (reference model/autonomous_drive.py:365-370; shapes SURVEY.md 2b/8a) and
// the 128..640-wide nn.Linear layers of the heads (autonomous_drive.py:371-387), forward and
// data-gradient.
//
// GEMM view (no im2col buffer is ever materialised):
//     M = N*Ho*Wo output pixels, N = Cout, K = taps*Cin, A[m][k] gathered on the fly from the
//     NHWC activation, B = OHWI weights.  A K-tile (32 floats) lies inside ONE filter tap, so the
//     gather is a 128-byte contiguous read per output pixel (one full cache line, NHWC).
//
// Block = 256 threads = 4 waves; block tile BM x BN, K-tile 32.  Global loads run TWO K-tiles
// ahead of the MFMAs (two register sets, LDS double-buffered): the measured load-to-use latency
// under load (~4-6k cycles) is several K-tiles of MFMA work, one tile of prefetch left the matrix
// pipe ~45 % idle.  Operands come through buffer loads: a wave-uniform base plus a 32-bit byte
// offset per row; per-row valid-tap masks are computed once, and a tap that falls outside the image
// is given the offset ~0, which is out of the buffer's range, so the hardware returns zeros -- no
// zero page, no pointer select, no divergent branch in the K loop (the generic-tap path of the
// 3-channel stem still gathers through pointers and a zero page).  LDS rows are K-contiguous; the
// 64x64 tile uses an unpadded XOR-swizzled image (32 KB per block, five blocks per CU), the larger
// tiles a 4-float pad (pitch 36 floats): a ds_read_b128 gives one lane four k-values and every
// 16-lane read group is conflict-free.  Lane half h owns k = 8q+4h+e (e = 0..3) of each 8-k group
// for BOTH operands, so the fmaf chain order is fixed: results are run-to-run deterministic.
//
// What a launch of this kernel costs at the benchmark sizes (tools/fill_probe.py, MI355X): with
// the grid an exact multiple of the 1,280 resident blocks the 64x64 tile sustains 126-134 TFLOP/s
// (0.80-0.85 of the fp32 matrix peak) from the second round on; every launch adds a fixed ~12 us
// (T = 11.9 us + 2.5 us per K-tile for one round of five blocks per CU: dependent-launch boundary,
// write-back of the ~20 MB of dirty output, cold first tiles, index set-up), and a layer whose tile
// count is not a multiple of the slots idles the short CUs (layer2: 1,100 tiles on 1,280 slots).
#include "common.h"

#include <stdlib.h>
#include <string.h>

namespace cilrs {

namespace {

constexpr int BK = 32;
constexpr int APITCH = BK + 4;
constexpr int kSplitKInKernelDefault = 0;      // see launch_conv_igemm

__device__ float g_zero_page[64];   // zero-initialised; target of out-of-image gathers

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    // blocks b and b+8 share an XCD (round-robin dispatch): give each XCD a contiguous chunk of
    // logical tiles so neighbouring tiles (shared input halo / shared weights) hit one L2.
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

// DMA = true: tiles go global -> LDS directly (LDS-DMA, global_load_lds 16 B/lane), three LDS
// stages, loads two K-tiles ahead with a COUNTED vmcnt and raw s_barrier: no VGPR round trip, no
// ds_write, a quarter of the registers.  The LDS image is then lane-linear (128-B rows, no pad);
// bank conflicts are avoided by XOR-swizzling the 16-B chunk index with (row>>1)&7 on the SOURCE
// address and on the ds_read side (synthetic ladder tools/mfma_probe.hip: 104 vs 70 TF at one
// block/CU, 123 vs 110 at three).
// The register-staged 64x64 tile also uses the unpadded, XOR-swizzled LDS image: 32 KB per block
// instead of 36.9 KB lets FIVE blocks share a CU (160 KB LDS, <= 96 VGPRs), i.e. 1,280 block slots
// instead of 1,024 -- layer2's 1,100 tiles then fit in one round.
// Timing experiments only (tools/igemm_dbg.sh builds side libraries with -DCILRS_IGEMM_DBG=mask; results
// are numerically meaningless): 1 = no global loads in the K loop, 2 = no LDS stores, 4 = no
// barriers, 8 = no LDS reads (operands stay in registers), 16 = ADD the work of a BatchNorm apply
// fused into the A-operand path (two 16-byte table loads, scale/shift, ReLU, padding mask per
// element between the global load and the LDS store): what VERDICT-style "apply folded into the
// next convolution's operand load" would cost this K loop
#ifndef CILRS_IGEMM_DBG
#define CILRS_IGEMM_DBG 0
#endif
template <int BM, int BN, bool TAP_UNIFORM, bool DMA>
constexpr bool igemm_swz() { return DMA || (BM == 64 && BN == 64 && TAP_UNIFORM); }
template <int BM, int BN, bool TAP_UNIFORM, bool DMA>
constexpr int igemm_min_blocks() { return (!DMA && BM == 64 && BN == 64 && TAP_UNIFORM) ? 5 : 2; }

// MULTI: the launch covers a.multi.n sub-problems (parity classes of a stride-2 data gradient);
// a block finds its class from its logical id and takes the class's output grid, taps and K
// length from a.multi -- four short launches of 0.4 rounds each become one launch of whole rounds.
template <int BM, int BN, int WM, int WN, bool TAP_UNIFORM, int W_MODE, bool DMA, bool MULTI = false>
__global__ __launch_bounds__(256, (igemm_min_blocks<BM, BN, TAP_UNIFORM, DMA>()))
void conv_igemm_kernel(const ConvArgs a, const int M_, const int Krow, const int KT_) {
    static_assert(WM * WN == 4, "4 waves");
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int A_PASSES = BM / 32, B_PASSES = BN / 32;
    constexpr bool SWZ = igemm_swz<BM, BN, TAP_UNIFORM, DMA>();
    constexpr int APIT = SWZ ? BK : APITCH;                       // A (and k-contiguous B) pitch
    constexpr int BPITCH1 = SWZ ? BN : BN + 4;                    // W_MODE 1 (k-major B) pitch
    constexpr int B_FLOATS = (W_MODE == 0) ? BN * APIT : BK * BPITCH1;
    constexpr int NSTAGE = DMA ? 3 : 2;
    constexpr int JQ = BN / 4, BROWS = 256 / JQ;     // W_MODE 1 loader shape

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                           // [NSTAGE][BM][APIT]
    float* Bs = smem + NSTAGE * BM * APIT;      // [NSTAGE][B_FLOATS]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;

    wave_priority(a.prio_mode < 16 ? a.prio_mode : 0);
    wave_stagger(a.prio_mode);
    const int tilesN = a.Cout / BN;
    int logical = xcd_remap(blockIdx.x, gridDim.x);
    // geometry of this block's (sub-)problem
    int cls = 0;
    int P_Ho = a.Ho, P_Wo = a.Wo, P_h0 = a.out_h0, P_w0 = a.out_w0, P_ntaps = a.ntaps;
    int M = M_, KT = KT_;
    if constexpr (MULTI) {
        while (cls + 1 < a.multi.n && logical >= a.multi.tile_begin[cls + 1]) ++cls;
        logical -= a.multi.tile_begin[cls];
        P_Ho = a.multi.Ho[cls]; P_Wo = a.multi.Wo[cls];
        P_h0 = a.multi.out_h0[cls]; P_w0 = a.multi.out_w0[cls];
        P_ntaps = a.multi.ntaps[cls];
        M = a.N * P_Ho * P_Wo;
        KT = P_ntaps * (a.Cin / BK);
    }
    auto tap_dh = [&](int t) { if constexpr (MULTI) return a.multi.tap_dh[cls][t]; else return a.tap_dh[t]; };
    auto tap_dw = [&](int t) { if constexpr (MULTI) return a.multi.tap_dw[cls][t]; else return a.tap_dw[t]; };
    auto tap_w = [&](int t) { if constexpr (MULTI) return a.multi.tap_w[cls][t]; else return a.tap_w[t]; };
    const int m0 = (logical / tilesN) * BM;
    const int n0 = (logical % tilesN) * BN;

    // K-tile range of this split
    int kt_begin = 0, kt_end = KT;
    if (a.splitk > 1) {
        const int per = (KT + a.splitk - 1) / a.splitk;
        kt_begin = blockIdx.z * per;
        kt_end = min(KT, kt_begin + per);
    }
    const int nt = max(kt_end - kt_begin, 0);

    // ---- per-thread gather rows (fixed for the whole K loop) --------------------------------
    const int kq = tid & 7, r0 = tid >> 3;
    // logical 16-B chunk this thread fetches: LDS position kq of row r holds chunk kq^((r>>1)&7)
    // when DMA (rows r0 + 32 i all share (r>>1)&7)
    const int kql = DMA ? (kq ^ ((r0 >> 1) & 7)) : kq;
    long rowBase[A_PASSES];        // element offset of the row's base pixel (+ kql*4)
    unsigned rowMask[A_PASSES];    // uniform path: bit t = tap t inside the image
    int rowH[A_PASSES], rowW[A_PASSES];   // generic path only
    const int HoWo = P_Ho * P_Wo;
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
        const int m = m0 + r0 + 32 * i;
        rowMask[i] = 0u; rowBase[i] = 0; rowH[i] = -(1 << 20); rowW[i] = 0;
        if (m < M) {
            const int n = m / HoWo, rem = m - n * HoWo;
            const int oh = rem / P_Wo, ow = rem - oh * P_Wo;
            const int hb = oh * a.stride - a.pad, wb = ow * a.stride - a.pad;
            rowBase[i] = ((long)(n * a.H + hb) * a.W + wb) * a.x_ld + kql * 4;
            if constexpr (TAP_UNIFORM) {
                unsigned msk = 0u;
                for (int t = 0; t < P_ntaps; ++t) {
                    const int h = hb + tap_dh(t), w = wb + tap_dw(t);
                    if (h >= 0 && w >= 0 && h < a.H && w < a.W) msk |= 1u << t;
                }
                rowMask[i] = msk;
            } else {
                rowH[i] = hb; rowW[i] = wb;
            }
        }
    }
    // weight row bases
    long wBase[B_PASSES];
    const long wrow = (long)a.KH * a.KW * a.w_cin;
    const int jq = tid % JQ, kr0 = tid / JQ;
#pragma unroll
    for (int i = 0; i < B_PASSES; ++i) {
        if constexpr (W_MODE == 0) wBase[i] = (long)(n0 + r0 + 32 * i) * Krow + kql * 4;
        else wBase[i] = (long)(kr0 + BROWS * i) * wrow + n0 + jq * 4;
    }
    const int cin_tiles = TAP_UNIFORM ? (a.Cin / BK) : 1;
    const float* zero = g_zero_page + kq * 4;

    // Register-staged uniform path: buffer loads (wave-uniform base + 32-bit byte offset; a row
    // whose tap falls outside the image gets offset ~0 = out of range = hardware returns zeros),
    // and the per-tap offsets live one per lane and are fetched with v_readlane -- no scalar
    // memory access and no pointer select inside the K loop.
    constexpr bool BUF = TAP_UNIFORM && !DMA;
    __amdgpu_buffer_rsrc_t rsA, rsB;
    unsigned rowOff[A_PASSES], wOff[B_PASSES];
    int tapA_v = 0, tapB_v = 0;
    if constexpr (BUF) {
        const size_t xbytes = (size_t)a.N * a.H * a.W * a.x_ld * sizeof(float);
        rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)(unsigned)xbytes, 0x00020000);
        rsB = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, -1, 0x00020000);
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) rowOff[i] = (unsigned)(rowBase[i] * 4);
#pragma unroll
        for (int i = 0; i < B_PASSES; ++i) wOff[i] = (unsigned)(wBase[i] * 4);
        if (lane < P_ntaps) {
            tapA_v = (tap_dh(lane) * a.W + tap_dw(lane)) * a.x_ld * 4;
            tapB_v = tap_w(lane) * (W_MODE == 0 ? a.Cin : a.w_cin) * 4;
        }
    }

    // scalar cursor (tap, cin chunk) of the NEXT tile to load
    int ld_tap = kt_begin / cin_tiles;
    int ld_c = kt_begin - ld_tap * cin_tiles;
    int ld_kt = kt_begin;

    // DMA destinations of this wave (wave-uniform bases; lane l lands at base + 16*l bytes)
    const int a_dst0 = (wave * 8) * BK;                              // + 32*i*BK per pass
    const int b_dst0 = (W_MODE == 0) ? (wave * 8) * BK : (wave * (64 / JQ)) * BN;
    int dma_stage = 0;
    auto put_a = [&](f32x4(&ra)[A_PASSES], int i, const float* p) {
        if constexpr (DMA) {
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)p,
                (__attribute__((address_space(3))) void*)(As + dma_stage * BM * APIT + a_dst0 +
                                                           32 * i * BK),
                16, 0, 0);
        } else {
            ra[i] = *reinterpret_cast<const f32x4*>(p);
        }
    };
    auto put_b = [&](f32x4(&rb)[B_PASSES], int i, const float* p) {
        if constexpr (DMA) {
            constexpr int STEP = (W_MODE == 0) ? 32 * BK : BROWS * BN;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)p,
                (__attribute__((address_space(3))) void*)(Bs + dma_stage * B_FLOATS + b_dst0 +
                                                           STEP * i),
                16, 0, 0);
        } else {
            rb[i] = *reinterpret_cast<const f32x4*>(p);
        }
    };
    auto load_tile = [&](f32x4(&ra)[A_PASSES], f32x4(&rb)[B_PASSES]) {
        if constexpr (BUF) {
            const unsigned toff = (unsigned)__builtin_amdgcn_readlane(tapA_v, ld_tap) +
                                  (unsigned)(ld_c * BK * 4);
            const unsigned bit = 1u << ld_tap;
#pragma unroll
            for (int i = 0; i < A_PASSES; ++i) {
                const unsigned off = (rowMask[i] & bit) ? rowOff[i] + toff : 0xFFFFFFFFu;
                if constexpr (CILRS_IGEMM_DBG & 1) ra[i] = f32x4{(float)off, 1.f, 1.f, 1.f};
                else
                ra[i] = __builtin_bit_cast(
                    f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)off, 0, 0));
            }
            unsigned koff = (unsigned)__builtin_amdgcn_readlane(tapB_v, ld_tap);
            if constexpr (W_MODE == 0) koff += (unsigned)(ld_c * BK * 4);
            else koff += (unsigned)(ld_c * BK) * (unsigned)(wrow * 4);
#pragma unroll
            for (int i = 0; i < B_PASSES; ++i) {
                if constexpr (CILRS_IGEMM_DBG & 1) rb[i] = f32x4{(float)(wOff[i] + koff), 1.f, 1.f, 1.f};
                else
                rb[i] = __builtin_bit_cast(
                    f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, (int)(wOff[i] + koff), 0, 0));
            }
        } else if constexpr (TAP_UNIFORM) {
            const long toff = ((long)tap_dh(ld_tap) * a.W + tap_dw(ld_tap)) * a.x_ld +
                              ld_c * BK;
#pragma unroll
            for (int i = 0; i < A_PASSES; ++i) {
                const bool ok = (rowMask[i] >> ld_tap) & 1u;
                const float* p = ok ? (a.x + (rowBase[i] + toff)) : zero;
                put_a(ra, i, p);
            }
            long koff;
            if constexpr (W_MODE == 0) koff = (long)tap_w(ld_tap) * a.Cin + ld_c * BK;
            else koff = (long)(ld_c * BK) * wrow + (long)tap_w(ld_tap) * a.w_cin;
#pragma unroll
            for (int i = 0; i < B_PASSES; ++i) put_b(rb, i, a.w + (wBase[i] + koff));
        } else {
            // generic: each k-quad may sit in a different tap (Cin % 4 == 0, e.g. the stem's
            // channel-padded Cin = 4); forward only
            const int k = ld_kt * BK + kql * 4;
            const int tap = k / a.Cin, ci = k - tap * a.Cin;
            const int kh = tap / a.KW, kw = tap - kh * a.KW;
            const bool kok = k < Krow;
#pragma unroll
            for (int i = 0; i < A_PASSES; ++i) {
                const int h = rowH[i] + kh, w = rowW[i] + kw;
                const bool ok = kok && (h >= 0) && (w >= 0) && (h < a.H) && (w < a.W);
                const float* p =
                    ok ? (a.x + (rowBase[i] - kql * 4 + ((long)kh * a.W + kw) * a.x_ld + ci))
                       : zero;
                put_a(ra, i, p);
            }
#pragma unroll
            for (int i = 0; i < B_PASSES; ++i) {
                const float* p = kok ? (a.w + (wBase[i] + (long)ld_kt * BK)) : zero;
                put_b(rb, i, p);
            }
        }
        if constexpr (DMA) dma_stage = (dma_stage + 1 == NSTAGE) ? 0 : dma_stage + 1;
        ++ld_kt;
        if (++ld_c == cin_tiles) { ld_c = 0; ++ld_tap; }
    };

    int st_tap = kt_begin / cin_tiles, st_c = kt_begin - (kt_begin / cin_tiles) * cin_tiles;
    auto store_tile = [&](int buf, const f32x4(&ra_in)[A_PASSES], const f32x4(&rb)[B_PASSES]) {
        if constexpr (CILRS_IGEMM_DBG & 2) {
            asm volatile("" ::"v"(ra_in[0]), "v"(rb[0]), "v"(ra_in[A_PASSES - 1]), "v"(rb[B_PASSES - 1]));
            return;
        }
        f32x4 ra[A_PASSES];
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) ra[i] = ra_in[i];
        if constexpr ((CILRS_IGEMM_DBG & 16) && TAP_UNIFORM) {
            const int ch = (st_c * BK + kq * 4) & 1023;          // stand-in scale / shift tables
            const f32x4 sc = *reinterpret_cast<const f32x4*>(a.w + ch);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(a.w + 1024 + ch);
            const unsigned bit = 1u << st_tap;
#pragma unroll
            for (int i = 0; i < A_PASSES; ++i) {
                const bool in_image = (rowMask[i] & bit) != 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = fmaxf(fmaf(ra[i][e], sc[e], sh[e]), 0.f);
                    ra[i][e] = in_image ? v : 0.f;
                }
            }
            if (++st_c == cin_tiles) { st_c = 0; ++st_tap; }
        }
        float* Ab = As + buf * BM * APIT;
        float* Bb = Bs + buf * B_FLOATS;
        // swizzled image: logical 16-B chunk kq of row r sits at position kq ^ ((r >> 1) & 7)
        const int kpos = (SWZ ? (kq ^ ((r0 >> 1) & 7)) : kq) * 4;
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i)
            *reinterpret_cast<f32x4*>(Ab + (r0 + 32 * i) * APIT + kpos) = ra[i];
        if constexpr (W_MODE == 0) {
#pragma unroll
            for (int i = 0; i < B_PASSES; ++i)
                *reinterpret_cast<f32x4*>(Bb + (r0 + 32 * i) * APIT + kpos) = rb[i];
        } else {
#pragma unroll
            for (int i = 0; i < B_PASSES; ++i)
                *reinterpret_cast<f32x4*>(Bb + (kr0 + BROWS * i) * BPITCH1 + jq * 4) = rb[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int swz = (l31 >> 1) & 7;           // DMA: chunk swizzle of this lane's rows
    auto compute = [&](int buf) {
        const float* Ab = As + buf * BM * APIT + (wm * WTM + l31) * APIT;
        const float* Bb = Bs + buf * B_FLOATS;
        // operand fragments of k-group q+1 are read from LDS while group q is multiplied
        f32x4 af[2][TM], bf[2][TN];
        auto frags = [&](int q, f32x4(&fa)[TM], f32x4(&fb)[TN]) {
            if constexpr (CILRS_IGEMM_DBG & 8) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[i] = f32x4{(float)q, 1.f, 2.f, 3.f};
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[j] = f32x4{(float)buf, 1.f, 2.f, 3.f};
                return;
            }
            const int chunk = SWZ ? (((2 * q + lh) ^ swz) * 4) : (q * 8 + lh * 4);
#pragma unroll
            for (int i = 0; i < TM; ++i)
                fa[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * APIT + chunk);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (W_MODE == 0) {
                    fb[j] = *reinterpret_cast<const f32x4*>(
                        Bb + (wn * WTN + j * 32 + l31) * APIT + chunk);
                } else {
                    const float* p = Bb + (q * 8 + lh * 4) * BPITCH1 + wn * WTN + j * 32 + l31;
                    fb[j][0] = p[0];
                    fb[j][1] = p[BPITCH1];
                    fb[j][2] = p[2 * BPITCH1];
                    fb[j][3] = p[3 * BPITCH1];
                }
            }
        };
        frags(0, af[0], bf[0]);
#pragma unroll
        for (int q = 0; q < BK / 8; ++q) {
            if (q + 1 < BK / 8) frags(q + 1, af[(q + 1) & 1], bf[(q + 1) & 1]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                            af[q & 1][i][e], bf[q & 1][j][e], acc[i][j], 0, 0, 0);
        }
        // pin the interleave (the scheduler otherwise sinks every read to just before its use
        // and the wave eats one LDS round trip per k-group): reads of groups 0 and 1, then
        // [multiply group q | read group q+2] ...
        constexpr int NR = TM + (W_MODE == 0 ? TN : 4 * TN);      // LDS reads per k-group
        constexpr int NM = 4 * TM * TN;                           // MFMAs per k-group
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * NR, 0);
#pragma unroll
        for (int q = 0; q < BK / 8; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
            if (q + 2 < BK / 8) __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
        }
    };

    if constexpr (DMA) {
        // ---- LDS-DMA pipeline: tile t+2 is issued while tile t is multiplied; per iteration one
        // counted wait (this wave's loads of tile t landed) + one raw barrier (everyone's did, and
        // everyone finished reading the stage about to be refilled)
        constexpr int LPT = A_PASSES + B_PASSES;          // DMA instructions per wave per tile
        f32x4 dummy_a[A_PASSES], dummy_b[B_PASSES];
        int issued = 0;
        for (; issued < nt && issued < 2; ++issued) load_tile(dummy_a, dummy_b);
        int stage = 0;
        for (int t = 0; t < nt; ++t) {
            if (issued > t + 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (issued < nt) { load_tile(dummy_a, dummy_b); ++issued; }
            compute(stage);
            stage = (stage + 1 == NSTAGE) ? 0 : stage + 1;
        }
        __syncthreads();      // all reads done before the epilogue reuses LDS
    } else {
    // ---- main loop: loads run two tiles ahead (register sets 0/1), LDS double-buffered ----
    f32x4 ra0[A_PASSES], rb0[B_PASSES], ra1[A_PASSES], rb1[B_PASSES];
    int it = 0;
    if (nt >= 4) {
        // steady state: tiles it+2 and it+3 exist, so every load / LDS store -- here and in this
        // prologue -- is unconditional and the compiler's wait counts are exact (one guarded
        // store anywhere on the path makes it re-wait for the in-flight register set at the top
        // of every iteration, which halves the prefetch distance)
        load_tile(ra0, rb0);
        load_tile(ra1, rb1);
        store_tile(0, ra0, rb0);
        __syncthreads();
        for (; it + 3 < nt; it += 2) {
            load_tile(ra0, rb0);
            __builtin_amdgcn_sched_barrier(0);   // issue the loads BEFORE the multiplies: two
            compute(0);                          // compute phases of latency cover, not one
            store_tile(1, ra1, rb1);
            if constexpr (!(CILRS_IGEMM_DBG & 4)) __syncthreads();
            load_tile(ra1, rb1);
            __builtin_amdgcn_sched_barrier(0);
            compute(1);
            store_tile(0, ra0, rb0);
            if constexpr (!(CILRS_IGEMM_DBG & 4)) __syncthreads();
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
    }

    // ---- split-K: every K-slice block parks its partial tile in a slab; the block that draws the
    // last ticket of the tile sums the slabs in slice order (deterministic) and runs the epilogue.
    // Hand-off WITHOUT release / acquire fences (cdna guide G16, "every load sc1" form; the same
    // protocol as csrc/infer_b1.hip): slab stores are write-through (`sc1`: a whole 128-B line per
    // wave instruction), every storing wave drains them, a workgroup barrier, ONE relaxed
    // agent-scope ticket add whose returned value tells the last arriver, which then reads every
    // slab with `sc1` loads.  (Round 1's version fenced -- an agent release per block writes back
    // the XCD's whole L2, ~20 MB of dirty output here -- and lost to a separate reduce launch.)
    bool inkernel_reduce = false;
    if (a.splitk > 1 && a.tile_counters != nullptr) {
        inkernel_reduce = true;
        float* slab = a.scratch + (size_t)blockIdx.z * M * a.y_ld;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int co = n0 + wn * WTN + j * 32 + l31;
                const int mbase = m0 + wm * WTM + i * 32 + 4 * lh;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mbase + (r & 3) + 8 * (r >> 2);
                    if (m < M)
                        __hip_atomic_store(&slab[(size_t)m * a.y_ld + co], acc[i][j][r],
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        volatile int* flag = reinterpret_cast<volatile int*>(smem);
        if (tid == 0) {
            const int prev = __hip_atomic_fetch_add(&a.tile_counters[logical], 1, __ATOMIC_RELAXED,
                                                    __HIP_MEMORY_SCOPE_AGENT);
            const int last = (prev == a.splitk - 1) ? 1 : 0;
            if (last)       // leave the counter ready for the next launch; nobody else touches it
                __hip_atomic_store(&a.tile_counters[logical], 0, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            flag[0] = last;
        }
        __syncthreads();
        const int is_last = flag[0];
        __syncthreads();                         // flag word is reused as scratch below
        if (!is_last) return;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int co = n0 + wn * WTN + j * 32 + l31;
                const int mbase = m0 + wm * WTM + i * 32 + 4 * lh;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
                for (int z = 0; z < a.splitk; ++z) {
                    const float* sl = a.scratch + (size_t)z * M * a.y_ld;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = mbase + (r & 3) + 8 * (r >> 2);
                        if (m < M)
                            acc[i][j][r] += __hip_atomic_load(&sl[(size_t)m * a.y_ld + co],
                                                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
    }

    // ---- fused BatchNorm statistics: per-channel sum / sum of squares of this M-tile's rows
    // (rows >= M are exact zeros).  Fixed summation order => deterministic.
    if (a.bn_partial != nullptr && (a.splitk <= 1 || inkernel_reduce)) {
        float* red = smem;                       // [WM][BN][2]; the K loop ended on a barrier
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = acc[i][j][r];
                    s1 += v;
                    s2 = fmaf(v, v, s2);
                }
            s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 32);
            if (lh == 0) {
                red[(wm * BN + wn * WTN + j * 32 + l31) * 2] = s1;
                red[(wm * BN + wn * WTN + j * 32 + l31) * 2 + 1] = s2;
            }
        }
        __syncthreads();
        if (tid < BN) {
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) {
                t1 += red[(w * BN + tid) * 2];
                t2 += red[(w * BN + tid) * 2 + 1];
            }
            // channel-major partials [2][Cout][M-tiles]: the finalize reads each channel's row
            const size_t mt = (size_t)(logical / tilesN), nmt = (size_t)((M + BM - 1) / BM);
            a.bn_partial[(size_t)(n0 + tid) * nmt + mt] = t1;
            a.bn_partial[(size_t)(a.Cout + n0 + tid) * nmt + mt] = t2;
        }
    }

    // ---- epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    // Optional operands are fetched as 16 independent loads per tile (clamped row, no per-element
    // branch) so their latency overlaps instead of serialising.
    const bool partial = a.splitk > 1 && !inkernel_reduce;
    const bool dense = (a.out_sh == 1) && (a.out_sw == 1) && (a.out_H == P_Ho) &&
                       (a.out_W == P_Wo);
    float* yout = partial ? a.y + (size_t)blockIdx.z * M * a.y_ld : a.y;
    // fused BatchNorm-backward reductions of the layer this data gradient feeds: per channel
    // sum(g) and sum(g * xhat), g = dz * (z > 0), xhat = (y - mean) * rstd
    const bool bwd_red = a.bwd_partial != nullptr && !partial;
    float bs1[TN], bs2[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) { bs1[j] = 0.f; bs2[j] = 0.f; }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int co = n0 + wn * WTN + j * 32 + l31;
            const int mbase = m0 + wm * WTM + i * 32 + 4 * lh;
            int pix[16];     // output pixel index of each of the 16 rows (clamped for loads)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = min(mbase + (r & 3) + 8 * (r >> 2), M - 1);
                if (dense || partial) {
                    pix[r] = m;
                } else {
                    const int n = m / HoWo, rem = m - n * HoWo;
                    const int oh = rem / P_Wo, ow = rem - oh * P_Wo;
                    pix[r] = (n * a.out_H + oh * a.out_sh + P_h0) * a.out_W +
                             ow * a.out_sw + P_w0;
                }
            }
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = acc[i][j][r];
            if (!partial) {
                if (a.ch_scale) {
                    const float sc = a.ch_scale[co], sh = a.ch_shift[co];
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = v[r] * sc + sh;
                }
                if (a.bias) {
                    const float b = a.bias[co];
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] += b;
                }
                if (a.relu) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.f);
                }
                if (a.mask) {
                    float mk[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) mk[r] = a.mask[(size_t)pix[r] * a.mask_ld + co];
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = mk[r] > 0.f ? v[r] * a.mask_scale : 0.f;
                }
                if (a.addend) {
                    float ad[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) ad[r] = a.addend[(size_t)pix[r] * a.y_ld + co];
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] += ad[r];
                }
                if (a.relu_post) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.f);
                }
            }
            if (bwd_red) {
                float zz[16], yy[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    zz[r] = a.bwd_relu ? a.bwd_z[(size_t)pix[r] * a.y_ld + co] : 1.f;
                    yy[r] = a.bwd_y[(size_t)pix[r] * a.y_ld + co];
                }
                const float mean = a.bwd_stats[co], rstd = a.bwd_stats[a.Cout + co];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mbase + (r & 3) + 8 * (r >> 2);
                    const float g = (m < M && zz[r] > 0.f) ? v[r] : 0.f;
                    bs1[j] += g;
                    bs2[j] = fmaf(g, (yy[r] - mean) * rstd, bs2[j]);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mbase + (r & 3) + 8 * (r >> 2);
                if (m < M) yout[(size_t)pix[r] * a.y_ld + co] = v[r];
            }
        }
    }
    if (bwd_red) {
        __syncthreads();                          // LDS may still hold the forward-stats scratch
        float* red = smem;                        // [WM][BN][2]
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float s1 = bs1[j] + __shfl_xor(bs1[j], 32);
            float s2 = bs2[j] + __shfl_xor(bs2[j], 32);
            if (lh == 0) {
                red[(wm * BN + wn * WTN + j * 32 + l31) * 2] = s1;
                red[(wm * BN + wn * WTN + j * 32 + l31) * 2 + 1] = s2;
            }
        }
        __syncthreads();
        if (tid < BN) {
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) {
                t1 += red[(w * BN + tid) * 2];
                t2 += red[(w * BN + tid) * 2 + 1];
            }
            const size_t mt = (size_t)(logical / tilesN), nmt = (size_t)((M + BM - 1) / BM);
            a.bwd_partial[(size_t)(n0 + tid) * nmt + mt] = t1;
            a.bwd_partial[(size_t)(a.Cout + n0 + tid) * nmt + mt] = t2;
        }
    }
}

__device__ __forceinline__ int out_pixel(const ConvArgs& a, int m, int HoWo) {
    const int n = m / HoWo, rem = m - n * HoWo;
    const int oh = rem / a.Wo, ow = rem - oh * a.Wo;
    return (n * a.out_H + oh * a.out_sh + a.out_h0) * a.out_W + ow * a.out_sw + a.out_w0;
}

// Sum split-K partials (fixed order => deterministic) and apply the epilogue.  VEC = 4: one float4
// per thread; VEC = 1: one float per thread (small problems: 4x the threads).  Four independent
// load streams over the splits.
// Split-K reduce that also emits the BatchNorm column partials of the finished tensor (dense
// output only), so a split-K convolution needs no separate column-reduce pass:
//   MODE 0 (forward):  partial = sum(v), sum(v*v) over the block's rows
//   MODE 1 (data gradient): partial = sum(g), sum(g * xhat), g = v * (z > 0), xhat = (y-mean)*rstd
// Layout and determinism as bn_colreduce_kernel (channel-major partial[2][C][gridDim.x]).
template <int MODE>
__global__ __launch_bounds__(256) void splitk_reduce_cols_kernel(const ConvArgs a,
                                                                 const float* __restrict__ part,
                                                                 const int M, const int splits,
                                                                 const int rows_per_block,
                                                                 float* __restrict__ partial) {
    __shared__ float red[2][256 * 4];
    const int C = a.Cout;
    const int tpr = C >> 2;                // threads per row (one float4 each)
    const int rpi = 256 / tpr;             // rows per iteration
    const int q = threadIdx.x % tpr, rsub = threadIdx.x / tpr;
    const int co = q * 4;
    const size_t slab = (size_t)M * a.y_ld;
    const int row_end = min(M, (int)(blockIdx.x + 1) * rows_per_block);
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    f32x4 mean = {0.f, 0.f, 0.f, 0.f}, rstd = {0.f, 0.f, 0.f, 0.f};
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f}, bias = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 1) {
        mean = *reinterpret_cast<const f32x4*>(a.bwd_stats + co);
        rstd = *reinterpret_cast<const f32x4*>(a.bwd_stats + C + co);
    }
    if (a.ch_scale) {
        sc = *reinterpret_cast<const f32x4*>(a.ch_scale + co);
        sh = *reinterpret_cast<const f32x4*>(a.ch_shift + co);
    }
    if (a.bias) bias = *reinterpret_cast<const f32x4*>(a.bias + co);
    // one row: everything after its operands are in registers (same arithmetic, same order, whatever
    // the loop shape below)
    auto finish_row = [&](const int m, const size_t o, f32x4 v, const f32x4 mk, const f32x4 ad,
                          const f32x4 zz, const f32x4 yy) {
        if (a.ch_scale) v = v * sc + sh;
        if (a.bias) v += bias;
        if (a.relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (a.mask) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = mk[e] > 0.f ? v[e] * a.mask_scale : 0.f;
        }
        if (a.addend) v += ad;
        if (a.relu_post) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        *reinterpret_cast<f32x4*>(a.y + o) = v;
        if (MODE == 0) {
            s1 += v;
#pragma unroll
            for (int e = 0; e < 4; ++e) s2[e] = fmaf(v[e], v[e], s2[e]);
        } else {
            f32x4 g = v;
            if (a.bwd_relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) g[e] = zz[e] > 0.f ? g[e] : 0.f;
            }
            s1 += g;
#pragma unroll
            for (int e = 0; e < 4; ++e) s2[e] = fmaf(g[e], (yy[e] - mean[e]) * rstd[e], s2[e]);
        }
        (void)m;
    };
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    int m = blockIdx.x * rows_per_block + rsub;
    if (splits <= 4) {
        // up to four slabs (the Winograd channel split: three): a thread's rows are few (three at
        // layer3) and each is a chain of dependent round trips -- THREE rows per trip, every load
        // of the trip in flight before the first use
        // (two for the data-gradient form: with three it needs 136 registers, and a launch on the
        //  critical path of the backward pass must fit NEXT to the weight-gradient workgroups that
        //  occupy every CU -- ~110 registers and 13 KB of LDS are what they leave.  Measured: the
        //  136-register version, equally fast alone, made the overlapped step 0.56 ms SLOWER.)
        constexpr int U = MODE == 1 ? 2 : 3;
        for (; m < row_end; m += U * rpi) {
            f32x4 sl[U][4], mk[U], ad[U], zz[U], yy[U];
            size_t o[U];
            bool live[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int mu = m + u * rpi;
                live[u] = mu < row_end;
                o[u] = (size_t)(live[u] ? mu : m) * a.y_ld + co;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    sl[u][k] = k < splits ? *reinterpret_cast<const f32x4*>(part + (size_t)k * slab + o[u]) : zero4;
                mk[u] = a.mask ? *reinterpret_cast<const f32x4*>(a.mask + (size_t)(live[u] ? mu : m) * a.mask_ld + co) : zero4;
                ad[u] = a.addend ? *reinterpret_cast<const f32x4*>(a.addend + o[u]) : zero4;
                zz[u] = (MODE == 1 && a.bwd_relu) ? *reinterpret_cast<const f32x4*>(a.bwd_z + o[u]) : zero4;
                yy[u] = MODE == 1 ? *reinterpret_cast<const f32x4*>(a.bwd_y + o[u]) : zero4;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (!live[u]) continue;
                const f32x4 v = (sl[u][0] + sl[u][1]) + (sl[u][2] + sl[u][3]);
                finish_row(m + u * rpi, o[u], v, mk[u], ad[u], zz[u], yy[u]);
            }
        }
    } else {
        for (; m < row_end; m += rpi) {
            const size_t o = (size_t)m * a.y_ld + co;
            f32x4 acc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = zero4;
            int sidx = 0;
            for (; sidx + 3 < splits; sidx += 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    acc[u] += *reinterpret_cast<const f32x4*>(part + (size_t)(sidx + u) * slab + o);
            }
            for (; sidx < splits; ++sidx)
                acc[0] += *reinterpret_cast<const f32x4*>(part + (size_t)sidx * slab + o);
            const f32x4 v = (acc[0] + acc[1]) + (acc[2] + acc[3]);
            const f32x4 mk = a.mask ? *reinterpret_cast<const f32x4*>(a.mask + (size_t)m * a.mask_ld + co) : zero4;
            const f32x4 ad = a.addend ? *reinterpret_cast<const f32x4*>(a.addend + o) : zero4;
            const f32x4 zz = (MODE == 1 && a.bwd_relu) ? *reinterpret_cast<const f32x4*>(a.bwd_z + o) : zero4;
            const f32x4 yy = MODE == 1 ? *reinterpret_cast<const f32x4*>(a.bwd_y + o) : zero4;
            finish_row(m, o, v, mk, ad, zz, yy);
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        red[0][threadIdx.x * 4 + e] = s1[e];
        red[1][threadIdx.x * 4 + e] = s2[e];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float a1 = 0.f, a2 = 0.f;
        const int qq = c >> 2, e = c & 3;
        for (int rs = 0; rs < rpi; ++rs) {
            a1 += red[0][(rs * tpr + qq) * 4 + e];
            a2 += red[1][(rs * tpr + qq) * 4 + e];
        }
        partial[(size_t)c * gridDim.x + blockIdx.x] = a1;
        partial[(size_t)(C + c) * gridDim.x + blockIdx.x] = a2;
    }
}

template <int VEC>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const ConvArgs a, const float* part,
                                                            const int M, const int splits) {
    const int cq = a.Cout / VEC;
    const size_t total = (size_t)M * cq;
    const size_t slab = (size_t)M * a.y_ld;
    const int HoWo = a.Ho * a.Wo;
    const bool dense = (a.out_sh == 1) && (a.out_sw == 1) && (a.out_H == a.Ho) &&
                       (a.out_W == a.Wo);
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(idx / cq);
        const int co = (int)(idx - (size_t)m * cq) * VEC;
        const size_t o = (size_t)m * a.y_ld + co;
        float v[VEC];
        {
            float acc[4][VEC];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[u][e] = 0.f;
            int sidx = 0;
            for (; sidx + 3 < splits; sidx += 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if constexpr (VEC == 4) {
                        const f32x4 t =
                            *reinterpret_cast<const f32x4*>(part + (size_t)(sidx + u) * slab + o);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[u][e] += t[e];
                    } else {
                        acc[u][0] += part[(size_t)(sidx + u) * slab + o];
                    }
                }
            }
            for (; sidx < splits; ++sidx) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[0][e] += part[(size_t)sidx * slab + o + e];
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[e] = (acc[0][e] + acc[1][e]) + (acc[2][e] + acc[3][e]);
        }
        const int pix = dense ? m : out_pixel(a, m, HoWo);
        const size_t po = (size_t)pix * a.y_ld + co;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            float t = v[e];
            if (a.ch_scale) t = t * a.ch_scale[co + e] + a.ch_shift[co + e];
            if (a.bias) t += a.bias[co + e];
            if (a.relu) t = fmaxf(t, 0.f);
            if (a.mask)
                t = (a.mask[(size_t)pix * a.mask_ld + co + e] > 0.f) ? t * a.mask_scale : 0.f;
            if (a.addend) t += a.addend[po + e];
            if (a.relu_post) t = fmaxf(t, 0.f);
            v[e] = t;
        }
        if constexpr (VEC == 4) {
            f32x4 ov = {v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(a.y + po) = ov;
        } else {
            a.y[po] = v[0];
        }
    }
}

// y_sub = addend_sub (or 0) for an output-parity class that no filter tap reaches
__global__ __launch_bounds__(256) void subgrid_fill_kernel(const ConvArgs a, const int M) {
    const int cq = a.Cout >> 2;
    const size_t total = (size_t)M * cq;
    const int HoWo = a.Ho * a.Wo;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(idx / cq);
        const int co = (int)(idx - (size_t)m * cq) * 4;
        const size_t po = (size_t)out_pixel(a, m, HoWo) * a.y_ld + co;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (a.addend) v = *reinterpret_cast<const f32x4*>(a.addend + po);
        *reinterpret_cast<f32x4*>(a.y + po) = v;
    }
}

template <int BM, int BN, int WM, int WN, bool TU, int MODE, bool DMA>
int launch_cfg(const ConvArgs& a, int M, int Krow, int KT, hipStream_t s) {
    constexpr bool SWZ = igemm_swz<BM, BN, TU, DMA>();
    constexpr int APIT = SWZ ? BK : APITCH;
    constexpr int B_FLOATS = (MODE == 0) ? BN * APIT : BK * (SWZ ? BN : BN + 4);
    constexpr int NSTAGE = DMA ? 3 : 2;
    constexpr size_t lds = (size_t)(NSTAGE * BM * APIT + NSTAGE * B_FLOATS) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        CILRS_HIP(hipFuncSetAttribute(
            reinterpret_cast<const void*>(&conv_igemm_kernel<BM, BN, WM, WN, TU, MODE, DMA>),
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    dim3 grid(cdiv(M, BM) * (a.Cout / BN), 1, a.splitk > 1 ? a.splitk : 1);
    conv_igemm_kernel<BM, BN, WM, WN, TU, MODE, DMA><<<grid, 256, lds, s>>>(a, M, Krow, KT);
    CILRS_LAUNCH_CHECK();
    return 0;
}

// Cost model (microseconds).  A CU finishes b co-resident blocks of a config at the chip-wide rate
// rate[b] (TFLOP/s, from the synthetic ladder tools/mfma_probe.hip, derated by what the real
// kernels reach); the busiest CU runs ceil(blocks/256) blocks in groups of `occ`.
struct Cfg { int bm, bn, occ; bool dma; float rate[5]; };
const Cfg kCfg[6] = {
    {128, 128, 2, false, {100.f, 113.f, 0.f, 0.f}},     // 0: register-staged 128x128
    {128, 64, 2, false, {81.f, 100.f, 0.f, 0.f}},       // 1: register-staged 128x64
    {64, 64, 5, false, {64.f, 90.f, 101.f, 108.f, 112.f}},   // 2: register-staged 64x64 (32 KB LDS)
    {128, 128, 1, true, {125.f, 0.f, 0.f, 0.f}},        // 3: LDS-DMA 128x128 (96 KB LDS)
    {128, 64, 2, true, {110.f, 121.f, 0.f, 0.f}},       // 4: LDS-DMA 128x64  (72 KB)
    {64, 64, 3, true, {94.f, 106.f, 111.f, 0.f}},       // 5: LDS-DMA 64x64   (48 KB)
};
constexpr int kNumCfg = 6;

double cfg_cost(int M, int Cout, int KT, const Cfg& c, int splitk) {
    if (Cout % c.bn) return 1e30;
    const double blocks = (double)cdiv(M, c.bm) * (Cout / c.bn) * splitk;
    const long per_cu = (long)((blocks + 255) / 256);
    const double ktiles = (double)cdiv(KT, splitk);
    const double unit = 2.0 * c.bm * c.bn * BK * ktiles;               // flops per block
    const long full = per_cu / c.occ, rem = per_cu % c.occ;
    double t = 0.0;
    if (full) t += full * unit * 256.0 * c.occ / (c.rate[c.occ - 1] * 1e6);
    if (rem) t += unit * 256.0 * rem / (c.rate[rem - 1] * 1e6);
    t += (full + (rem ? 1 : 0)) * 2.5;                                   // fill / drain per group
    if (splitk > 1) t += 3.0 + 0.12 * splitk + (double)M * Cout * 4.0 * (splitk + 1) / 3.0e6;  // slabs + reduce
    return t;
}

struct Choice { int cfg; int splitk; };

Choice choose(int M, int Cout, int KT, bool allow_split, int force_cfg, int force_splitk) {
    static const int splits[11] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48};
    static const int use_dma = experiment_env("CILRS_IGEMM_DMA", 0);
    Choice best{2, 1};
    double bc = 1e30;
    for (int c = 0; c < kNumCfg; ++c) {
        if (force_cfg >= 0 && c != force_cfg) continue;
        if (force_cfg < 0 && use_dma == 0 && kCfg[c].dma) continue;
        if (force_cfg < 0 && use_dma == 2 && !kCfg[c].dma) continue;
        for (int si = 0; si < 11; ++si) {
            int sk = splits[si];
            if (force_splitk > 0) {
                if (si > 0) break;
                sk = force_splitk;
            } else if (sk > 1 && (!allow_split || KT < 2 * sk)) {
                continue;
            }
            const double cost = cfg_cost(M, Cout, KT, kCfg[c], sk);
            if (cost < bc) { bc = cost; best = {c, sk}; }
        }
    }
    return best;
}

}  // namespace

int launch_conv_igemm(const ConvArgs& a_in, hipStream_t s) {
    ConvArgs a = a_in;
    a.prio_mode = wave_priority_mode();
    const int M = a.N * a.Ho * a.Wo;
    const bool uniform = (a.Cin % BK) == 0;
    // the uniform path addresses both operands with 32-bit byte offsets (buffer loads)
    CILRS_CHECK(!uniform || ((size_t)a.N * a.H * a.W * a.x_ld * sizeof(float) < (1ull << 32) &&
                             (size_t)(a.w_mode == 0 ? a.Cout : a.Cin) * a.KH * a.KW *
                                     (a.w_mode == 0 ? a.Cin : a.w_cin) * sizeof(float) < (1ull << 32)),
                "conv_igemm: tensor larger than 4 GB");
    if (a.out_H == 0) {          // dense output
        a.out_H = a.Ho; a.out_W = a.Wo; a.out_sh = a.out_sw = 1; a.out_h0 = a.out_w0 = 0;
    }
    if (uniform && a.ntaps == 0) {   // dense KH x KW table, forward order
        CILRS_CHECK(a.KH * a.KW <= 16, "conv_igemm: more than 16 taps on the uniform path");
        a.ntaps = a.KH * a.KW;
        for (int t = 0; t < a.ntaps; ++t) {
            a.tap_dh[t] = t / a.KW;
            a.tap_dw[t] = t % a.KW;
            a.tap_w[t] = (a.w_mode == 0) ? t : a.ntaps - 1 - t;      // dgrad: flipped filter
        }
    }
    const int KT = uniform ? a.ntaps * (a.Cin / BK) : cdiv(a.KH * a.KW * a.Cin, BK);
    // W_MODE 0 weight rows are addressed with the forward row pitch KH*KW*Cin
    const int Krow = a.KH * a.KW * a.Cin;
    CILRS_CHECK(a.Cout % 64 == 0, "conv_igemm: Cout=%d must be a multiple of 64", a.Cout);
    CILRS_CHECK(a.Cin % 4 == 0 && a.x_ld % 4 == 0 && a.y_ld % 4 == 0,
                "conv_igemm: Cin/x_ld/y_ld must be multiples of 4 (%d,%d,%d)", a.Cin, a.x_ld,
                a.y_ld);
    CILRS_CHECK(uniform || a.w_mode == 0, "conv_igemm: the generic-tap path is forward-only");
    CILRS_CHECK(a.w_mode == 0 || a.w_cin % 4 == 0, "conv_igemm: w_cin must be a multiple of 4");
    CILRS_CHECK(((uintptr_t)a.x & 15) == 0 && ((uintptr_t)a.w & 15) == 0 &&
                    ((uintptr_t)a.y & 15) == 0,
                "conv_igemm: pointers must be 16-byte aligned");
    CILRS_CHECK(M > 0 && (size_t)a.N * a.H * a.W < (1u << 31), "conv_igemm: bad M");
    CILRS_CHECK(a.ntaps <= 16, "conv_igemm: too many taps");

    // ---- tile / split-K choice ----
    const size_t slab = (size_t)M * a.y_ld;
    const bool can_split = a.scratch != nullptr && a.scratch_floats >= 2 * slab && uniform;
    Choice ch = choose(M, a.Cout, KT, can_split, a.force_cfg, a.force_splitk);
    while (ch.splitk > 1 && (size_t)ch.splitk * slab > a.scratch_floats) --ch.splitk;
    CILRS_CHECK(ch.splitk == 1 || (a.scratch && uniform), "conv_igemm: split-K needs scratch");
    CILRS_CHECK(ch.cfg >= 0 && ch.cfg < kNumCfg && a.Cout % kCfg[ch.cfg].bn == 0,
                "conv_igemm: tile config %d does not fit Cout=%d", ch.cfg, a.Cout);
    a.splitk = ch.splitk;
    // in-kernel reduction needs one ticket counter per output tile (zero before the launch; the
    // last arriver re-zeroes it).  Counters live at the head of the scratch unless given.
    const int n_tiles = cdiv(M, kCfg[ch.cfg].bm) * (a.Cout / kCfg[ch.cfg].bn);
    // Round 1 measured the FENCED in-kernel combine slower than a separate reduce launch (train
    // step +9 %); the fence-free form (sc1 slabs + ticket) is A/B'd with CILRS_SPLITK_INKERNEL=0|1
    // (profiles/r03_splitk_ab.log).
    static const int inkernel_on =
        getenv("CILRS_SPLITK_INKERNEL") ? atoi(getenv("CILRS_SPLITK_INKERNEL")) : kSplitKInKernelDefault;
    bool inkernel = false;
    if (!inkernel_on) a.tile_counters = nullptr;
    if (a.splitk > 1 && inkernel_on) {
        if (a.tile_counters == nullptr) {
            // carve [counters | slabs] out of the caller's scratch and zero the counters
            const size_t cnt_floats = ((size_t)n_tiles + 63) / 64 * 64;
            if (a.scratch_floats >= cnt_floats + (size_t)a.splitk * slab) {
                a.tile_counters = reinterpret_cast<int*>(a.scratch);
                CILRS_HIP(hipMemsetAsync(a.tile_counters, 0, cnt_floats * sizeof(float), s));
                a.scratch += cnt_floats;
                a.scratch_floats -= cnt_floats;
                inkernel = true;
            }
        } else {
            inkernel = n_tiles <= a.tile_counters_cap;
            if (!inkernel) a.tile_counters = nullptr;
        }
    }
    const bool dense_out = a.out_sh == 1 && a.out_sw == 1 && a.out_H == a.Ho && a.out_W == a.Wo;
    // split-K + separate reduce: the reduce kernel can emit the BatchNorm column partials itself
    // when the output is dense and the channel count tiles a 256-thread block
    const int tpr = a.Cout >> 2;
    const bool cols_ok = a.splitk > 1 && !inkernel && dense_out && a.Cout % 4 == 0 && tpr <= 256 &&
                         256 % tpr == 0 && a.y_ld % 4 == 0 && (!a.mask || a.mask_ld % 4 == 0);
    int cols_rows = 0, cols_nblk = 0;
    if (cols_ok && (a.bn_partial || a.bwd_partial)) {
        const int rpi = 256 / tpr;
        cols_rows = cdiv(cdiv(M, 1024), rpi) * rpi;       // <= 1024 partial rows per channel
        cols_nblk = cdiv(M, cols_rows);
    }
    float* const fwd_partial = a.bn_partial;
    float* const bwd_partial = a.bwd_partial;
    if (a_in.bn_nblk)
        *a_in.bn_nblk = !a.bn_partial ? 0
                        : (a.splitk == 1 || inkernel) ? cdiv(M, kCfg[ch.cfg].bm) : cols_nblk;
    if (!(a.splitk == 1 && dense_out)) a.bwd_partial = nullptr;
    if (a_in.bwd_nblk)
        *a_in.bwd_nblk = a.bwd_partial ? cdiv(M, kCfg[ch.cfg].bm) : (bwd_partial ? cols_nblk : 0);
    float* final_y = a.y;
    if (a.splitk > 1 && !inkernel) a.y = a.scratch;      // legacy: partials + separate reduce

    int rc = 1;
#define CILRS_DISPATCH(BM_, BN_, DMA_)                                                          \
    do {                                                                                        \
        if (a.w_mode == 1) rc = launch_cfg<BM_, BN_, 2, 2, true, 1, DMA_>(a, M, Krow, KT, s);   \
        else if (uniform)  rc = launch_cfg<BM_, BN_, 2, 2, true, 0, DMA_>(a, M, Krow, KT, s);   \
        else               rc = launch_cfg<BM_, BN_, 2, 2, false, 0, DMA_>(a, M, Krow, KT, s);  \
    } while (0)
    switch (ch.cfg) {
        case 0: CILRS_DISPATCH(128, 128, false); break;
        case 1: CILRS_DISPATCH(128, 64, false); break;
        case 2: CILRS_DISPATCH(64, 64, false); break;
        case 3: CILRS_DISPATCH(128, 128, true); break;
        case 4: CILRS_DISPATCH(128, 64, true); break;
        default: CILRS_DISPATCH(64, 64, true); break;
    }
#undef CILRS_DISPATCH
    if (rc) return rc;

    if (a.splitk > 1 && !inkernel) {
        const float* part = a.scratch;
        a.y = final_y;
        const size_t total4 = (size_t)M * (a.Cout / 4);
        if (cols_nblk > 0 && fwd_partial) {
            splitk_reduce_cols_kernel<0><<<cols_nblk, 256, 0, s>>>(a, part, M, a.splitk, cols_rows,
                                                                   fwd_partial);
        } else if (cols_nblk > 0 && bwd_partial) {
            splitk_reduce_cols_kernel<1><<<cols_nblk, 256, 0, s>>>(a, part, M, a.splitk, cols_rows,
                                                                   bwd_partial);
        } else if (total4 < 65536) {  // small problem: one float per thread, 4x the parallelism
            const size_t total = total4 * 4;
            splitk_reduce_kernel<1><<<(int)((total + 255) / 256), 256, 0, s>>>(a, part, M, a.splitk);
        } else {
            const int blocks = (int)((total4 + 255) / 256 < 2048 ? (total4 + 255) / 256 : 2048);
            splitk_reduce_kernel<4><<<blocks, 256, 0, s>>>(a, part, M, a.splitk);
        }
        CILRS_LAUNCH_CHECK();
    }
    return 0;
}

// Sum `splits` dense [M][C] fp32 slabs in slab order into y (+ addend) and emit the BatchNorm column
// partials of the result ([2][C][rows], rows = slab_reduce_rows): mode 0 forward statistics,
// mode 1 BatchNorm-backward reductions.  Used by the channel-split Winograd launches
// (conv_wino.hip); the same kernel as the implicit GEMM's split-K reduce.
int slab_reduce_rows(int M, int C, int* rows_per_block) {
    const int tpr = C >> 2, rpi = 256 / tpr;
    const int rows = cdiv(cdiv(M, 1024), rpi) * rpi;
    if (rows_per_block) *rows_per_block = rows;
    return cdiv(M, rows);
}
int launch_slab_reduce_cols(int mode, const float* slabs, int splits, float* y, const float* addend,
                            int M, int C, const float* bwd_z, const float* bwd_y,
                            const float* bwd_stats, int bwd_relu, float* partial, hipStream_t s) {
    CILRS_CHECK(slabs && y && partial && splits >= 1, "slab_reduce: NULL argument");
    CILRS_CHECK(C % 4 == 0 && (C >> 2) <= 256 && 256 % (C >> 2) == 0, "slab_reduce: channel count %d", C);
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.y = y; a.addend = addend; a.Cout = C; a.y_ld = C;
    a.bwd_z = bwd_z; a.bwd_y = bwd_y; a.bwd_stats = bwd_stats; a.bwd_relu = bwd_relu;
    int rows = 0;
    const int nblk = slab_reduce_rows(M, C, &rows);
    if (mode == 0) splitk_reduce_cols_kernel<0><<<nblk, 256, 0, s>>>(a, slabs, M, splits, rows, partial);
    else splitk_reduce_cols_kernel<1><<<nblk, 256, 0, s>>>(a, slabs, M, splits, rows, partial);
    CILRS_LAUNCH_CHECK();
    return 0;
}

int launch_conv_dgrad(const DgradArgs& d, hipStream_t s) {
    CILRS_CHECK(d.stride == 1 || d.stride == 2, "dgrad: stride must be 1 or 2");
    CILRS_CHECK(d.K * d.K <= 16, "dgrad: filter too large");
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x = d.dy; a.w = d.w; a.y = d.dx; a.addend = d.addend;
    a.mask = d.mask; a.mask_ld = d.mask_ld; a.mask_scale = d.mask_scale;
    a.N = d.N; a.H = d.Ho; a.W = d.Wo; a.Cin = d.Cout;     // gathered tensor = dy
    a.Cout = d.Cin;                                        // columns written = forward Cin
    a.KH = a.KW = d.K;
    a.x_ld = d.dy_ld; a.y_ld = d.dx_ld; a.w_mode = 1; a.w_cin = d.Cin;
    a.scratch = d.scratch; a.scratch_floats = d.scratch_floats;
    a.force_cfg = d.force_cfg; a.force_splitk = d.force_splitk;
    a.tile_counters = d.tile_counters; a.tile_counters_cap = d.tile_counters_cap;
    if (d.bwd_nblk) *d.bwd_nblk = 0;
    if (d.stride == 1) {
        a.bwd_z = d.bwd_z; a.bwd_y = d.bwd_y; a.bwd_stats = d.bwd_stats; a.bwd_relu = d.bwd_relu;
        a.bwd_partial = d.bwd_partial; a.bwd_nblk = d.bwd_nblk;
        a.Ho = d.H; a.Wo = d.W; a.stride = 1; a.pad = d.K - 1 - d.pad;
        return launch_conv_igemm(a, s);           // dense table with flipped taps
    }
    // stride 2: dx pixel (hi, wi) = (2*oh + ph, 2*ow + pw) only sees taps kh with
    // (ph + pad - kh) even, reading dy row oh + (ph + pad - kh)/2
    //
    // Default: ALL FOUR parity classes in one launch (a class alone is 0.4 rounds of blocks with a
    // quarter of the K length; launched separately they ran at 52-61 TF).  Classes are ordered by
    // K length, longest first; a class no tap reaches (1x1 / s2) still gets blocks: with zero
    // K-tiles they write the addend (or zeros) through the ordinary epilogue.
    // ... when a class alone does not fill the chip.  A class that brings a full round of blocks
    // of its own (the ResNet-50 variant at 176x400: 2,200 tiles per class) runs 36 % FASTER as four
    // launches with the cost model's own tile per class (tools/conv_bench.py --r50: 257 vs 349 us).
    const long cls_tiles = (long)cdiv(d.N * ((d.H + 1) / 2) * ((d.W + 1) / 2), 64) * (d.Cin / 64);
    const bool fuse = d.force_cfg < 0 && d.force_splitk <= 0 && (d.Cout % BK) == 0 &&
                      (d.Cin % 64) == 0 && cls_tiles < 1280;
    if (fuse) {
        ConvArgs c = a;
        c.stride = 1; c.pad = 0;
        c.out_H = d.H; c.out_W = d.W; c.out_sh = 2; c.out_sw = 2; c.out_h0 = c.out_w0 = 0;
        c.Ho = (d.H + 1) / 2; c.Wo = (d.W + 1) / 2;          // class (0,0): never empty
        c.splitk = 1; c.scratch = nullptr; c.tile_counters = nullptr;
        c.prio_mode = wave_priority_mode();
        struct Cls { int ph, pw, nt, Ho, Wo; int dh[4], dw[4], tw[4]; } cl[4];
        int ncl = 0;
        for (int ph = 0; ph < 2; ++ph)
            for (int pw = 0; pw < 2; ++pw) {
                Cls k;
                k.ph = ph; k.pw = pw; k.nt = 0;
                k.Ho = (d.H - ph + 1) / 2; k.Wo = (d.W - pw + 1) / 2;
                if (k.Ho <= 0 || k.Wo <= 0) continue;
                for (int kh = 0; kh < d.K; ++kh) {
                    if ((ph + d.pad - kh) & 1) continue;
                    for (int kw = 0; kw < d.K; ++kw) {
                        if ((pw + d.pad - kw) & 1) continue;
                        CILRS_CHECK(k.nt < 4, "dgrad: more than 4 taps in a parity class");
                        k.dh[k.nt] = (ph + d.pad - kh) / 2;
                        k.dw[k.nt] = (pw + d.pad - kw) / 2;
                        k.tw[k.nt] = kh * d.K + kw;
                        ++k.nt;
                    }
                }
                // a class no tap reaches (three of the four of a 1x1 / stride-2 filter) has nothing
                // to add: when the result accumulates IN PLACE (addend == dx, no mask) its pixels
                // already hold their final value -- no blocks for it (the down-sample branch's
                // data gradient then touches a quarter of dx instead of reading and rewriting all
                // of it: 70 -> ~25 us at B=128)
                if (k.nt == 0 && d.addend == d.dx && d.addend != nullptr && d.mask == nullptr) continue;
                cl[ncl++] = k;
            }
        for (int i = 1; i < ncl; ++i)                        // longest K first (stable)
            for (int j = i; j > 0 && cl[j].nt > cl[j - 1].nt; --j) {
                const Cls t = cl[j]; cl[j] = cl[j - 1]; cl[j - 1] = t;
            }
        ConvMulti& mu = c.multi;
        memset(&mu, 0, sizeof(mu));
        mu.n = ncl;
        const int tilesN = c.Cout / 64;
        int total = 0;
        for (int i = 0; i < ncl; ++i) {
            mu.tile_begin[i] = total;
            total += cdiv(c.N * cl[i].Ho * cl[i].Wo, 64) * tilesN;
            mu.Ho[i] = cl[i].Ho; mu.Wo[i] = cl[i].Wo;
            mu.out_h0[i] = cl[i].ph; mu.out_w0[i] = cl[i].pw; mu.ntaps[i] = cl[i].nt;
            for (int t = 0; t < cl[i].nt; ++t) {
                mu.tap_dh[i][t] = cl[i].dh[t]; mu.tap_dw[i][t] = cl[i].dw[t];
                mu.tap_w[i][t] = cl[i].tw[t];
            }
        }
        for (int i = ncl; i < 5; ++i) mu.tile_begin[i] = total;
        CILRS_CHECK((size_t)c.N * c.H * c.W * c.x_ld * sizeof(float) < (1ull << 32) &&
                        (size_t)c.Cin * c.KH * c.KW * c.w_cin * sizeof(float) < (1ull << 32),
                    "conv_dgrad: tensor larger than 4 GB");
        CILRS_CHECK(((uintptr_t)c.x & 15) == 0 && ((uintptr_t)c.w & 15) == 0 &&
                        ((uintptr_t)c.y & 15) == 0 && c.x_ld % 4 == 0 && c.y_ld % 4 == 0 &&
                        c.w_cin % 4 == 0,
                    "conv_dgrad: operands must be 16-byte aligned");
        constexpr size_t lds = (size_t)(2 * 64 * BK + 2 * BK * 64) * sizeof(float);
        static bool attr_set = false;
        if (!attr_set) {
            CILRS_HIP(hipFuncSetAttribute(
                reinterpret_cast<const void*>(
                    &conv_igemm_kernel<64, 64, 2, 2, true, 1, false, true>),
                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_set = true;
        }
        conv_igemm_kernel<64, 64, 2, 2, true, 1, false, true>
            <<<dim3(total, 1, 1), 256, lds, s>>>(c, 0, c.KH * c.KW * c.Cin, 0);
        CILRS_LAUNCH_CHECK();
        return 0;
    }
    for (int ph = 0; ph < 2; ++ph)
        for (int pw = 0; pw < 2; ++pw) {
            ConvArgs c = a;
            c.Ho = (d.H - ph + 1) / 2; c.Wo = (d.W - pw + 1) / 2;
            if (c.Ho <= 0 || c.Wo <= 0) continue;
            c.stride = 1; c.pad = 0;
            c.out_H = d.H; c.out_W = d.W; c.out_sh = 2; c.out_sw = 2; c.out_h0 = ph; c.out_w0 = pw;
            int nt = 0;
            for (int kh = 0; kh < d.K; ++kh) {
                if ((ph + d.pad - kh) & 1) continue;
                for (int kw = 0; kw < d.K; ++kw) {
                    if ((pw + d.pad - kw) & 1) continue;
                    c.tap_dh[nt] = (ph + d.pad - kh) / 2;     // exact: numerator is even
                    c.tap_dw[nt] = (pw + d.pad - kw) / 2;
                    c.tap_w[nt] = kh * d.K + kw;
                    ++nt;
                }
            }
            c.ntaps = nt;
            if (nt == 0) {
                const int M = c.N * c.Ho * c.Wo;
                const size_t total = (size_t)M * (c.Cout / 4);
                const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
                subgrid_fill_kernel<<<blocks, 256, 0, s>>>(c, M);
                CILRS_LAUNCH_CHECK();
                continue;
            }
            if (launch_conv_igemm(c, s)) return 1;
        }
    return 0;
}

}  // namespace cilrs
