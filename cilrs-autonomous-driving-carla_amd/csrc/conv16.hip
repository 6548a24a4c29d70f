// Training convolutions of the bf16 mode on large tiles: forward and data gradient of the trunk
// (BASELINE.json configs[3], "bf16 MFMA path"; no counterpart in the reference, which trains in
// fp32 -- notebook/notebook.ipynb:549-555; arithmetic defined by oracle/bf16_emulation.py).
//
// Why a second kernel next to conv_f16_kernel<64, 64, TRAIN> (infer_f16.hip): measured in round 4
// (profiles/r04_prof_bf16_v1/), a 64x64 tile moves 16 KB from L2 into LDS for 0.5 MFLOP -- at the
// bf16 pipe's rate that is 12 TB/s of L2 -> CU traffic for the ResNet-34 layers (the limit that
// run hit), and the 1x1 convolutions of the Bottleneck variant have ONE to four K-tiles per block:
// their blocks live 5-6 us of which the matrix pipe works 0.1.  So:
//   * 128x128 (128x64) output tile per 256-thread block, each wave 64x64 (64x32) = 2x2 (2x1)
//     accumulators of v_mfma_f32_32x32x16: half (two thirds) of the L2 and LDS bytes per flop;
//   * PERSISTENT blocks: a block walks tiles (adjacent tiles = the channel tiles of the same rows,
//     on the same XCD), and the global loads run ahead ACROSS tile boundaries -- the next tile's
//     first K-tile is in flight while this tile's epilogue runs;
//   * same operand path as the 64x64 kernel: buffer loads (an out-of-image tap gets offset ~0 and
//     reads zeros), 16-byte chunks, LDS double buffer with a 72-half pitch (conflict-free
//     ds_read_b128), one barrier per K-tile;
//   * same epilogue contract (ConvF16Args): result rounded to 16 bits (or fp32), optional 16-bit
//     addend, BatchNorm batch statistics of the ROUNDED result as per-tile column partials,
//     BatchNorm-backward reductions of the produced gradient, all in fixed order.
// Small layers (too few 128-row tiles to fill the chip) stay on the 64x64 kernel: see
// conv16_plan().
#include "common.h"

#include <stdlib.h>

namespace cilrs {
namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));

#ifndef CILRS_CONV16_DBG
#define CILRS_CONV16_DBG 0      // timing experiments (tools/conv16_dbg.sh): 1 no multiplies / LDS reads,
#endif                          // 2 no operand loads, 4 no epilogue -- results are then meaningless
constexpr int PBK = 64;                  // halfs per K-tile row: 128 bytes, UNPADDED (LDS-DMA images
                                         // are lane-linear), bank conflicts removed by a chunk swizzle

template <typename T> struct PVec8;
template <> struct PVec8<_Float16> { typedef h8 type; };
template <> struct PVec8<__bf16> { typedef b8 type; };
__device__ __forceinline__ f32x16 pmfma(const h8 a, const h8 b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 pmfma(const b8 a, const b8 b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// LDS: NST stages of one K-tile each: A [128 rows][128 B] then B [BN rows][128 B]
template <int BN> constexpr int stage_bytes() { return (128 + BN) * 128; }
template <int BN, int NST> constexpr size_t conv16p_lds() { return (size_t)NST * stage_bytes<BN>(); }

// tile `t` of the launch -> block-local order: XCD x (= blockIdx & 7) owns a contiguous range of
// tiles, its resident blocks (slot = blockIdx >> 3) walk that range side by side
struct TileWalk { int begin, end, step; };
__device__ __forceinline__ TileWalk tile_walk(const int ntiles) {
    const int G = (int)gridDim.x, bid = (int)blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int slots = (G >> 3) + ((G & 7) > xcd ? 1 : 0);        // blocks of this launch on the XCD
    const int q = ntiles >> 3, r = ntiles & 7;
    const int lo = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const int n = q + (xcd < r ? 1 : 0);
    TileWalk w{lo + slot, lo + n, slots};
    return w;
}

// Operands go global -> LDS directly (`buffer_load_dwordx4 ... lds`: no register destination, so
// nothing for the compiler to copy or to wait for), NST stages, the loads of K-tile g + NST - 1 are
// issued while K-tile g is multiplied, ONE counted s_waitcnt (this wave's loads of tile g have
// landed: buffer operations complete in issue order, so "at most the (NST-2) newer tiles' loads
// outstanding" is exact; anything an epilogue issued in between only makes the wait stricter)
// and ONE raw s_barrier per K-tile (everyone's have, and everyone finished reading the stage that
// is refilled next).  Out-of-image taps / rows past M / tiles past the walk get offset ~0: the
// range check writes zeros.  (A register pipeline was built first and dropped: hipcc drains to
// vmcnt(0) before every barrier of this control flow, and inline-asm loads had their destination
// registers copied before the data landed -- profiles/r04_conv16_notes.log.)
// The LDS-DMA loads are issued through inline asm: with the builtin the compiler knows they write
// LDS and orders EVERY later LDS access of the kernel behind them -- the epilogue's staging accesses
// then drain the whole look-ahead (vmcnt(0..2) in front of every ds_write / ds_read; measured: 10 us
// per 128x64 tile of a one-K-tile convolution).  No register destination, so nothing the compiler
// could copy early; the hand-counted wait + barrier of the K loop orders the data.
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 make_rsrc_words(const void* p, const unsigned bytes) {
    const unsigned long long v = (unsigned long long)p;
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)v);
    r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((v >> 32) & 0xffffu));
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}
// lane l: 16 bytes from buffer offset `voff` -> LDS byte address lds_base + 16 l (lds_base uniform)
__device__ __forceinline__ void dma16(const unsigned lds_base, const i32x4 rsrc, const unsigned voff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                 :: "s"(lds_base), "v"(voff), "s"(rsrc) : "memory", "m0");
}

// workgroup barrier for LDS traffic only: the wave's own LDS operations have completed, then
// s_barrier.  (__syncthreads() also drains vmcnt -- i.e. the whole look-ahead -- at every use.)
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// EPI (compile-time, so that no load of the epilogue is issued or consumed under a run-time
// condition: a load the compiler cannot PROVE consumed stays "pending" at the loop's back edge and
// is waited for -- together with the whole look-ahead -- inside the K loop): bit 0 = 16-bit
// addend, bit 1 = BatchNorm-backward reductions (with the ReLU mask).
template <typename T, int BN, int NST, int EPI>
__global__ __launch_bounds__(256, 2) void conv16p_kernel(const ConvF16Args a) {
    typedef typename PVec8<T>::type v8;
    constexpr bool ADD = (EPI & 1) != 0, BWD = (EPI & 2) != 0;
    constexpr int BM = 128;
    constexpr int TM = 2, TN = BN / 64;                // 32x32 accumulators per wave (2x2 waves)
    constexpr int AP = BM / 32, BP = BN / 32;          // DMA instructions per wave per K-tile
    constexpr int LPT = AP + BP;
    constexpr int STB = stage_bytes<BN>();
    constexpr int RR = BN == 64 ? 64 : 32;             // rows staged per epilogue round
    constexpr int SP = BN + 4;                         // staging pitch (floats)
    static_assert(RR * SP * 4 <= STB, "a staging round must fit in one free stage");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int M = a.N * a.Ho * a.Wo, HoWo = a.Ho * a.Wo;
    const int tilesN = a.Cout / BN, tilesM = (M + BM - 1) / BM;
    const int ntaps = a.K * a.K, cin_tiles = a.Cin / PBK;
    const int nt = ntaps * cin_tiles;
    const long Krow = (long)ntaps * a.Cin;
    const TileWalk walk = tile_walk(tilesM * tilesN);

    // ---- load side: its own tile cursor, NST - 1 K-tiles ahead of the compute side.  Thread ->
    // row r0 + 32 i of the tile and LDS position kq of that row, which holds logical 16-byte chunk
    // kq ^ ((row >> 1) & 7): the 16 lanes of a ds_read_b128 group then fall on 16 different slots
    const int kq = lane & 7, r0 = tid >> 3;
    const int kql = kq ^ ((r0 >> 1) & 7);
    unsigned rowOff[AP], rowMask[AP], wOff[BP];
    int tapA_v = 0, tapB_v = 0;              // per-tap byte offsets, one tap per lane
    if (lane < ntaps) {
        tapA_v = ((lane / a.K) * a.W + lane % a.K) * a.Cin * 2;
        tapB_v = lane * a.Cin * 2;
    }
    const i32x4 rsA = make_rsrc_words(a.x, (unsigned)((size_t)a.N * a.H * a.W * a.Cin * 2));
    const i32x4 rsB = make_rsrc_words(a.w, (unsigned)((size_t)a.Cout * Krow * 2));
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane(
        (int)(unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_raw);
    int ld_tile = walk.begin, ld_tap = 0, ld_c = 0, ld_stage = 0;
    auto load_setup = [&]() {
        const bool live = ld_tile < walk.end;          // past the walk: every load reads zeros
        const int m0 = (ld_tile / tilesN) * BM, n0 = (ld_tile % tilesN) * BN;
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const int m = m0 + r0 + 32 * i;
            rowMask[i] = 0u;
            rowOff[i] = 0u;
            if (live && m < M) {
                const int n = m / HoWo, rem = m - n * HoWo;
                const int oh = rem / a.Wo, ow = rem - oh * a.Wo;
                const int hb = oh * a.stride - a.pad, wb = ow * a.stride - a.pad;
                rowOff[i] = (unsigned)((((long)(n * a.H + hb) * a.W + wb) * a.Cin + kql * 8) * 2);
                for (int t = 0; t < ntaps; ++t) {
                    const int h = hb + t / a.K, w = wb + t % a.K;
                    if (h >= 0 && w >= 0 && h < a.H && w < a.W) rowMask[i] |= 1u << t;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < BP; ++i)
            wOff[i] = live ? (unsigned)(((long)(n0 + r0 + 32 * i) * Krow + kql * 8) * 2) : 0xFFFFFF00u;
        ld_tap = 0;
        ld_c = 0;
    };
    auto issue = [&]() {                 // the next K-tile of the walk -> stage ld_stage
        const unsigned toff = (unsigned)__builtin_amdgcn_readlane(tapA_v, ld_tap) +
                              (unsigned)(ld_c * PBK * 2);
        const unsigned koff = (unsigned)__builtin_amdgcn_readlane(tapB_v, ld_tap) +
                              (unsigned)(ld_c * PBK * 2);
        const unsigned bit = 1u << ld_tap;
        const unsigned dstA = lds0 + (unsigned)(ld_stage * STB + wave * (8 * 128));
        const unsigned dstB = dstA + BM * 128;
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const unsigned off = (rowMask[i] & bit) ? rowOff[i] + toff : 0xFFFFFFFFu;
            dma16(dstA + i * (32 * 128), rsA, off);
        }
#pragma unroll
        for (int i = 0; i < BP; ++i) dma16(dstB + i * (32 * 128), rsB, wOff[i] + koff);
        ld_stage = ld_stage + 1 == NST ? 0 : ld_stage + 1;
        if (++ld_c == cin_tiles) {
            ld_c = 0;
            if (++ld_tap == ntaps) {
                ld_tile += walk.step;
                load_setup();
            }
        }
    };

    f32x16 acc[TM][TN];
    const int swz = (l31 >> 1) & 7;
    auto compute = [&](const int st) {
        const unsigned char* Ab = smem_raw + st * STB + (wm * 64 + l31) * 128;
        const unsigned char* Bb = smem_raw + st * STB + BM * 128 + (wn * (BN / 2) + l31) * 128;
#pragma unroll
        for (int q = 0; q < PBK / 16; ++q) {
            const int ch = ((2 * q + lh) ^ swz) * 16;
            v8 av[TM], bv[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                av[i] = *reinterpret_cast<const v8*>(Ab + i * (32 * 128) + ch);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bv[j] = *reinterpret_cast<const v8*>(Bb + j * (32 * 128) + ch);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = pmfma(av[i], bv[j], acc[i][j]);
        }
    };

    const bool out16 = a.y16 != nullptr;
    const T* add16 = reinterpret_cast<const T*>(a.addend16);
    const T* bz16 = reinterpret_cast<const T*>(a.bwd_z16);
    const T* by16 = reinterpret_cast<const T*>(a.bwd_y16);
    T* y16 = reinterpret_cast<T*>(a.y16);

    load_setup();
#pragma unroll
    for (int p = 0; p < NST - 1; ++p) issue();
    int cs = 0;                                        // stage of the K-tile multiplied next
    for (int tile = walk.begin; tile < walk.end; tile += walk.step) {
        const int mt = tile / tilesN;
        const int m0 = mt * BM, n0 = (tile - mt * tilesN) * BN;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int kt = 0; kt < nt; ++kt) {
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NST - 2) * LPT) : "memory");
            if constexpr (!(CILRS_CONV16_DBG & 2)) issue();   // K-tile g + NST - 1 -> the stage read at g - 1
            if constexpr (!(CILRS_CONV16_DBG & 1)) compute(cs);
            cs = cs + 1 == NST ? 0 : cs + 1;
        }
        // ---- epilogue; staging lives in the stage just multiplied (the other NST - 1 are being
        //      filled), RR rows of the tile per round ----
        if constexpr ((CILRS_CONV16_DBG & 4) != 0) {
            if (acc[0][0][0] == 123.456f) y16[tid] = (T)1.f;       // (keeps the accumulators alive)
            continue;
        }
        const int fs = cs == 0 ? NST - 1 : cs - 1;
        float* stage = reinterpret_cast<float*>(smem_raw + fs * STB);
        lds_barrier();                               // every wave is done reading that stage
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
            lds_barrier();
            if (tid < BN) {
                const float t1 = red[tid * 2] + red[(BN + tid) * 2];
                const float t2 = red[tid * 2 + 1] + red[(BN + tid) * 2 + 1];
                a.bn_partial[(size_t)(n0 + tid) * tilesM + mt] = t1;
                a.bn_partial[(size_t)(a.Cout + n0 + tid) * tilesM + mt] = t2;
            }
            lds_barrier();
        }
        constexpr int TPR = BN / 8, RPP = 256 / TPR;
        const int c8 = (tid % TPR) * 8, rsub = tid / TPR;
        f32x4 mlo = {0.f, 0.f, 0.f, 0.f}, mhi = mlo, rlo = mlo, rhi = mlo;
        f32x4 s1lo = mlo, s1hi = mlo, s2lo = mlo, s2hi = mlo;
        if constexpr (BWD) {
            mlo = *reinterpret_cast<const f32x4*>(a.bwd_stats + n0 + c8);
            mhi = *reinterpret_cast<const f32x4*>(a.bwd_stats + n0 + c8 + 4);
            rlo = *reinterpret_cast<const f32x4*>(a.bwd_stats + a.Cout + n0 + c8);
            rhi = *reinterpret_cast<const f32x4*>(a.bwd_stats + a.Cout + n0 + c8 + 4);
        }
#pragma unroll
        for (int rd = 0; rd < BM / RR; ++rd) {
            // this round's epilogue operands (addend, and y / z of the BatchNorm whose reductions
            // ride along) are requested first: their latency runs under the staging + barrier
            constexpr int NP = RR / RPP;
            v8 addv[NP], yv[NP], zv[NP];
#pragma unroll
            for (int pass = 0; pass < NP; ++pass) {
                const int m = m0 + rd * RR + pass * RPP + rsub;
                const size_t o = (size_t)(m < M ? m : 0) * a.Cout + n0 + c8;
                if constexpr (ADD) addv[pass] = *reinterpret_cast<const v8*>(add16 + o);
                if constexpr (BWD) {
                    yv[pass] = *reinterpret_cast<const v8*>(by16 + o);
                    zv[pass] = *reinterpret_cast<const v8*>(bz16 + o);
                }
            }
            // the waves that own rows rd * RR .. + RR of the tile park them in the staging area
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int trow = wm * 64 + i * 32 - rd * RR;       // wave-uniform
                if (trow >= 0 && trow < RR) {
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int col = wn * (BN / 2) + j * 32 + l31;
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            stage[(trow + (r & 3) + 8 * (r >> 2) + 4 * lh) * SP + col] = acc[i][j][r];
                    }
                }
            }
            lds_barrier();
            // (every requested operand is consumed on every path -- a row past M only skips its
            //  stores: a load left pending at the loop's back edge would make the compiler wait
            //  for it, and with it for the whole look-ahead, inside the K loop)
#pragma unroll
            for (int pass = 0; pass < NP; ++pass) {
                const int row = pass * RPP + rsub;
                const int m = m0 + rd * RR + row;
                const bool live = m < M;
                f32x4 lo = *reinterpret_cast<const f32x4*>(&stage[row * SP + c8]);
                f32x4 hi = *reinterpret_cast<const f32x4*>(&stage[row * SP + c8 + 4]);
                const size_t o = (size_t)(live ? m : 0) * a.Cout + n0 + c8;
                if constexpr (ADD) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        lo[e] += (float)addv[pass][e];
                        hi[e] += (float)addv[pass][4 + e];
                    }
                }
                if (out16) {
                    v8 ov;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        ov[e] = (T)lo[e];
                        ov[4 + e] = (T)hi[e];
                    }
                    if (live) *reinterpret_cast<v8*>(y16 + o) = ov;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        lo[e] = (float)ov[e];
                        hi[e] = (float)ov[4 + e];
                    }
                } else if (live) {
                    *reinterpret_cast<f32x4*>(a.y32 + o) = lo;
                    *reinterpret_cast<f32x4*>(a.y32 + o + 4) = hi;
                }
                if constexpr (BWD) {
                    f32x4 ylo, yhi;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        ylo[e] = (float)yv[pass][e];
                        yhi[e] = (float)yv[pass][4 + e];
                    }
                    f32x4 glo = lo, ghi = hi;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        glo[e] = (float)zv[pass][e] > 0.f ? glo[e] : 0.f;
                        ghi[e] = (float)zv[pass][4 + e] > 0.f ? ghi[e] : 0.f;
                    }
                    if (!live) { glo = f32x4{0.f, 0.f, 0.f, 0.f}; ghi = glo; }
                    s1lo += glo; s1hi += ghi;
                    s2lo += glo * ((ylo - mlo) * rlo);
                    s2hi += ghi * ((yhi - mhi) * rhi);
                }
            }
            lds_barrier();                             // the staged rows are consumed
        }
        if constexpr (BWD) {
            // column sums over the tile's rows: [RPP row-threads][BN] through LDS, fixed order
            float* red1 = stage;                                 // [RPP][BN]
            float* red2 = red1 + RPP * BN;
            *reinterpret_cast<f32x4*>(&red1[rsub * BN + c8]) = s1lo;
            *reinterpret_cast<f32x4*>(&red1[rsub * BN + c8 + 4]) = s1hi;
            *reinterpret_cast<f32x4*>(&red2[rsub * BN + c8]) = s2lo;
            *reinterpret_cast<f32x4*>(&red2[rsub * BN + c8 + 4]) = s2hi;
            lds_barrier();
            if (tid < BN) {
                float t1 = 0.f, t2 = 0.f;
#pragma unroll 8
                for (int r = 0; r < RPP; ++r) {
                    t1 += red1[r * BN + tid];
                    t2 += red2[r * BN + tid];
                }
                a.bwd_partial[(size_t)(n0 + tid) * tilesM + mt] = t1;
                a.bwd_partial[(size_t)(a.Cout + n0 + tid) * tilesM + mt] = t2;
            }
            lds_barrier();
        }
    }
    // the look-ahead loads past the walk (zeros) must land before the workgroup's LDS is released
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// 0: the 64x64 kernel of infer_f16.hip; 1: 128x64; 2: 128x128
struct Conv16Plan { int cfg, bm, tiles, grid; };
// CILRS_CONV16_BIG=1: 128x128 tiles where the layer has enough of them (one block per CU)
bool big_ok() {
    static const int on = experiment_env("CILRS_CONV16_BIG", 0);
    return on != 0;
}
Conv16Plan conv16_plan(const ConvF16Args& a) {
    const int cus = device_cus();
    // CILRS_CONV16_TILE: 0 keeps every launch on the 64x64 kernel, 1 / 2 force a large tile (A/B)
    static const int force = getenv("CILRS_CONV16_TILE") ? atoi(getenv("CILRS_CONV16_TILE")) : -1;
    const int M = a.N * a.Ho * a.Wo;
    Conv16Plan p{0, 64, 0, 0};
    if (a.up2 || force == 0) return p;
    if (a.bwd_partial && (!a.bwd_y16 || !a.bwd_relu)) return p;   // (fp32 / mask-free reductions: old kernel)
    if (a.addend32) return p;
    const int t128 = cdiv(M, 128) * (a.Cout / 128), t64 = cdiv(M, 128) * (a.Cout / 64);
    // Measured per shape (tools/conv16_bench.py, profiles/r04_conv16_bench.log): the persistent
    // 128x64 kernel wins 4-14 % on the 1x1 convolutions of the Bottleneck trunk and on reductions
    // over >= 512 channels, the 64x64 kernel (four blocks per CU) wins 5-20 % on the 3x3
    // convolutions with up to 256 channels; 128x128 (one block per CU) loses everywhere.  Both sit
    // at the L2 -> LDS rate (~21 TB/s with nothing else running: tools/conv16_dbg.sh), which a
    // larger tile lowers per flop but pays for in resident waves.
    if (a.Cout % 128 == 0 && (force == 2 || (force < 0 && t128 >= 2 * cus && big_ok()))) {
        p.cfg = 2; p.tiles = t128;
    } else if (force == 1 || (force < 0 && t64 >= 2 * cus && (a.K == 1 || a.Cin >= 512))) {
        p.cfg = 1; p.tiles = t64;
    } else {
        return p;
    }
    p.bm = 128;
    // 128x64: 72 KB of LDS, two resident blocks per CU; 128x128: 96 KB, one
    static const int nst = experiment_env("CILRS_CONV16_NST", 3);
    const int resident = (p.cfg == 2 || nst > 3 ? 1 : 2) * cus;
    p.grid = p.tiles < resident ? p.tiles : resident;
    return p;
}

template <typename T, int BN, int EPI, int NST = 3>
int launch_conv16p_epi(const ConvF16Args& a, const Conv16Plan& p, hipStream_t s) {
    constexpr size_t lds = conv16p_lds<BN, NST>();
    if (once_per_device(reinterpret_cast<const void*>(&conv16p_kernel<T, BN, NST, EPI>))) {
        CILRS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv16p_kernel<T, BN, NST, EPI>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    conv16p_kernel<T, BN, NST, EPI><<<p.grid, 256, lds, s>>>(a);
    CILRS_LAUNCH_CHECK();
    return 0;
}
template <typename T, int BN>
int launch_conv16p(const ConvF16Args& a, const Conv16Plan& p, hipStream_t s) {
    const int epi = (a.addend16 ? 1 : 0) | (a.bwd_partial ? 2 : 0);
    // experiment (CILRS_CONV16_NST=4|5, forward form, 128x64): deeper rings, one block per CU
    static const int nst = experiment_env("CILRS_CONV16_NST", 3);
    if constexpr (BN == 64) {
        if (epi == 0 && nst == 4) return launch_conv16p_epi<T, BN, 0, 4>(a, p, s);
        if (epi == 0 && nst == 5) return launch_conv16p_epi<T, BN, 0, 5>(a, p, s);
    }
    switch (epi) {
        case 0: return launch_conv16p_epi<T, BN, 0>(a, p, s);
        case 1: return launch_conv16p_epi<T, BN, 1>(a, p, s);
        case 2: return launch_conv16p_epi<T, BN, 2>(a, p, s);
        default: return launch_conv16p_epi<T, BN, 3>(a, p, s);
    }
}

}  // namespace

// rows of the [2][Cout][M-tiles] column partials (bn_partial / bwd_partial) a launch writes
int conv_f16_train_mtiles(const ConvF16Args& a) {
    if (a.up2) {                 // four parity classes, 64-row tiles each (infer_f16.hip)
        ConvF16Args c = a;
        conv_f16_up2_classes(c);
        return c.cls_tile_begin[4] / (c.Cout / 64);
    }
    return cdiv(a.N * a.Ho * a.Wo, conv16_plan(a).bm);
}

// the large-tile path of launch_conv_f16_train: returns -1 when the launch belongs on the 64x64
// kernel, otherwise the launch status
int launch_conv16_large(const ConvF16Args& a, hipStream_t s) {
    const Conv16Plan p = conv16_plan(a);
    if (p.cfg == 0) return -1;
    if (p.cfg == 2)
        return a.bf16 ? launch_conv16p<__bf16, 128>(a, p, s) : launch_conv16p<_Float16, 128>(a, p, s);
    return a.bf16 ? launch_conv16p<__bf16, 64>(a, p, s) : launch_conv16p<_Float16, 64>(a, p, s);
}

}  // namespace cilrs
