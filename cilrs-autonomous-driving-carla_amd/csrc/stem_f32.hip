// The stem convolution of the TRAINING step in the reference's arithmetic (fp32): conv 7x7 / stride 2 /
// pad 3 on 3 input channels, 64 output channels (visual_encoder.0, model/autonomous_drive.py:366 --
// torchvision resnet34.conv1; trained by notebook/notebook.ipynb:549-555), with the batch statistics
// of the following BatchNorm as per-tile column partials.
//
// Why not the implicit GEMM of conv_igemm.hip: there the image is channel-padded 3 -> 4 and the
// reduction index runs over 49 taps x 4 channels = 196 (13 K-tiles of 16 = 208): a quarter of the
// MFMAs multiply the zero channel, the 13-iteration K loop is mostly prologue and epilogue, and every
// 64x64 tile re-reads the 50 KB of weights: 175 us at B = 128 (60 TFLOP/s of useful work).  Here:
//   * the reduction index is cut into 84 MFMA steps (kh, tap pair q, channel c) of
//     v_mfma_f32_32x32x2_f32: the lower half-wave multiplies tap kw = 2 q, the upper one tap
//     2 q + 1 (tap 7 = zero weights): 168 k values instead of 208, and ONE ds_read_b128 per lane --
//     its pixel of the channel-padded row image, the 64 lanes together one contiguous kilobyte --
//     feeds three MFMAs;
//   * the weights live in REGISTERS for the life of the workgroup (lane = output channel, 84 values:
//     its column of B for every MFMA step) -- no weight traffic at all after the prologue;
//   * a persistent workgroup walks tiles of 128 / 256 consecutive output pixels of one image; the
//     input rows the tile needs (2 R + 5 rows of the channel-padded image) are copied verbatim into
//     LDS by `buffer_load_dwordx4 ... lds` (rows outside the image and pixels past the row's end come
//     from out-of-range offsets: zeros, which is the padding), double-buffered: the next tile's rows
//     land while this tile is multiplied; the wait at the top of a tile counts the stores the
//     epilogue issued since (every wave issues the same number: masked rows go to an out-of-range
//     offset), so a tile never waits for its predecessor's stores to drain;
//   * the LDS address of a lane's operand is a per-lane base plus a COMPILE-TIME offset per step
//     (the row pitch is a template parameter): no address arithmetic in the loop.
// 8 waves = 4 pixel groups x 2 channel halves; a wave holds TM 32x32 accumulators.
#include "common.h"

namespace cilrs {
namespace {

typedef int s_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ s_i32x4 stem_rsrc(const void* p, const unsigned bytes) {
    const unsigned long long v = (unsigned long long)p;
    s_i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)v);
    r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((v >> 32) & 0xffffu));
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}
// lane l: 16 bytes from buffer offset `voff` -> LDS byte address lds_base + 16 l (lds_base uniform)
__device__ __forceinline__ void stem_dma16(const unsigned lds_base, const s_i32x4 rsrc, const unsigned voff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                 :: "s"(lds_base), "v"(voff), "s"(rsrc) : "memory", "m0");
}

constexpr int SSTEPS = 7 * 4 * 3;        // MFMA steps: filter row x tap pair x channel
#ifndef CILRS_STEM_DBG
#define CILRS_STEM_DBG 0      // timing experiments (tools/stem_dbg.sh): 1 no LDS operand reads, 2 no row
#endif                        // copies, 4 no epilogue stores -- results are then meaningless
constexpr int SPREFIX = 3;               // zero pixels in front of row 0 (the left padding of row 0)

// LDS floats of one row buffer
__host__ __device__ constexpr int stem_buf_floats(int rows, int pitch_px) { return (SPREFIX + rows * pitch_px) * 4; }

struct StemF32Args {
    const float* x4;        // [N][H][W][4]
    const float* w;         // OHWI [64][7][7][3]
    float* y;               // [N][Ho][Wo][64]
    float* bn_partial;      // [2][64][ntiles] or NULL
    int N, H, W, Ho, Wo;
    int tiles_per_img, ntiles, rows_max;
};

// DB: two row buffers, the next tile's rows land under this tile's MFMAs (one workgroup per CU);
// !DB: one buffer, load -> multiply -> store in turn, TWO workgroups per CU cover for each other
// (TM = 1: 128 registers, <= 80 KB of LDS).
template <int TM, int PITCH, bool DB>
__global__ __launch_bounds__(512) void stem_f32_kernel(const StemF32Args a) {
    constexpr int M = 128 * TM;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int mg = wave >> 1, nt = wave & 1;
    const int HoWo = a.Ho * a.Wo;
    const int buf_floats = stem_buf_floats(a.rows_max, PITCH);
    float* red = smem + (DB ? 2 : 1) * buf_floats;              // [4 pixel groups][64][2]
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane(
        (int)(unsigned)(size_t)(__attribute__((address_space(3))) float*)smem);
    const s_i32x4 rs = stem_rsrc(a.x4, (unsigned)((size_t)a.N * a.H * a.W * 16));

    // this lane's column of B: output channel nt * 32 + l31, tap 2 q + lh of step (kh, q, c)
    float bw[SSTEPS];
    {
        const float* wr = a.w + (size_t)(nt * 32 + l31) * 147;
#pragma unroll
        for (int s = 0; s < SSTEPS; ++s) {
            const int kh = s / 12, q = (s / 3) & 3, c = s % 3;            // compile time
            const int kw = 2 * q + lh;
            bw[s] = kw < 7 ? wr[(kh * 7 + kw) * 3 + c] : 0.f;
        }
        // (consumed here: otherwise the compiler's own counted waits for these loads sit in front
        //  of the MFMAs of EVERY tile, where they wait for the previous tile's stores instead)
#pragma unroll
        for (int s = 0; s < SSTEPS; ++s) asm volatile("" ::"v"(bw[s]));
    }
    if (tid < (DB ? 2 : 1) * SPREFIX * 4) smem[(tid / (SPREFIX * 4)) * buf_floats + tid % (SPREFIX * 4)] = 0.f;

    constexpr int SEGS = PITCH / 64;
    auto issue = [&](const int tile, const int buf) {            // the tile's input rows -> buffer `buf`
        if (tile >= a.ntiles) return;
        const int n = tile / a.tiles_per_img, p0 = (tile - n * a.tiles_per_img) * M;
        const int p1 = min(p0 + M, HoWo) - 1;
        const int oh0 = p0 / a.Wo, oh1 = p1 / a.Wo;
        const int rows = 2 * (oh1 - oh0) + 7, ih0 = 2 * oh0 - 3;
        const unsigned base = lds0 + (unsigned)(buf * buf_floats + SPREFIX * 4) * 4u;
        for (int u = wave; u < rows * SEGS; u += 8) {            // (wave-uniform)
            const int row = u / SEGS, seg = u - row * SEGS;
            const int ih = ih0 + row, px = seg * 64 + lane;
            const bool ok = ih >= 0 && ih < a.H && px < a.W;
            const unsigned off = ok ? (unsigned)(((n * a.H + ih) * a.W + px) * 16) : 0xFFFFFFFFu;
            const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane(
                (int)(base + (unsigned)((row * PITCH + seg * 64) * 16)));      // (uniform; M0 is scalar)
            if constexpr ((CILRS_STEM_DBG & 2) == 0) stem_dma16(dst, rs, off);
        }
    };

    // (stores through the builtin, not inline asm: the compiler must see that they read MFMA
    //  results -- it inserts the wait states between the last v_mfma and the first store)
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.y, 0, (int)(unsigned)((size_t)a.N * HoWo * 64 * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.bn_partial ? a.bn_partial : a.y), 0,
        a.bn_partial ? (int)(unsigned)((size_t)2 * 64 * a.ntiles * 4) : 0, 0x00020000);
    auto store32 = [&](const float v, const __amdgpu_buffer_rsrc_t rsrc, const unsigned off) {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc, (int)off, 0, 0);
    };
    constexpr int kStores = 16 * TM + 2;          // vector-memory operations a wave issues after its row copies

    int buf = 0;
    if constexpr (DB) {
        issue((int)blockIdx.x, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    for (int tile = (int)blockIdx.x; tile < a.ntiles; tile += (int)gridDim.x, buf ^= DB ? 1 : 0) {
        if constexpr (DB) {
            // this tile's rows have landed (everything older than the previous tile's stores); every
            // wave is done with the other buffer and with the reduction scratch
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(kStores) : "memory");
            issue(tile + (int)gridDim.x, buf ^ 1);
        } else {
            // every wave is done with the buffer; copy this tile's rows, wait for them
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            issue(tile, 0);
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        }

        const int n = tile / a.tiles_per_img, p0 = (tile - n * a.tiles_per_img) * M;
        const int oh0 = p0 / a.Wo;
        // A: this lane's pixel of each 32-pixel group, slot 2 ow + lh of row 2 (oh - oh0)
        const float* ab[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int p = p0 + (mg * TM + i) * 32 + l31;
            const int pc = p < HoWo ? p : HoWo - 1;
            const int oh = pc / a.Wo, ow = pc - oh * a.Wo;
            ab[i] = smem + buf * buf_floats + (2 * (oh - oh0) * PITCH + 2 * ow + lh) * 4;
        }
        f32x16 acc[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        // operands of pair g + 1 are requested before the MFMAs of pair g (an LDS read that an MFMA
        // waits on exposes its whole latency: measured 57 of 132 us with the reads issued at use)
        f32x4 av[2][TM];
        auto fetch = [&](const int g, f32x4* dst) {
            const int kh = g >> 2, q = g & 3;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if constexpr ((CILRS_STEM_DBG & 1) != 0) dst[i] = f32x4{(float)g, 1.f, (float)lane, 2.f};
                else if constexpr ((CILRS_STEM_DBG & 8) != 0)       // lane-linear: conflict-free by construction
                    dst[i] = *reinterpret_cast<const f32x4*>(smem + buf * buf_floats + lane * 4 + (g * 2 + i) * 256);
                else if constexpr ((CILRS_STEM_DBG & 16) != 0)      // one address for all lanes: broadcast
                    dst[i] = *reinterpret_cast<const f32x4*>(smem + buf * buf_floats + (g * 2 + i) * 256);
                else dst[i] = *reinterpret_cast<const f32x4*>(ab[i] + (kh * PITCH + 2 * q) * 4);
            }
        };
        fetch(0, av[0]);
#if (CILRS_STEM_DBG & 32)
        for (int rep = 0; rep < 2; ++rep)                  // (timing experiment: the MFMA loop twice)
#endif
#pragma unroll
        for (int g = 0; g < 28; ++g) {                    // (filter row, tap pair)
            // (scheduling fences: left alone, the machine scheduler sinks every read to its first use)
            if (g + 1 < 28) fetch(g + 1, av[(g + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g & 1][i][c], bw[g * 3 + c], acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- epilogue: y rows (32 consecutive channels = 128 bytes per lane group) + column partials ----
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int pb = p0 + (mg * TM + i) * 32 + 4 * lh;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int p = pb + 8 * (r >> 2) + (r & 3);
                const bool ok = p < HoWo;
                const float v = ok ? acc[i][r] : 0.f;
                store32(acc[i][r], rsY, (ok && !(CILRS_STEM_DBG & 4)) ? (unsigned)(((n * HoWo + p) * 64 + nt * 32 + l31) * 4) : 0xFFFFFFFFu);
                s1 += v;
                s2 = fmaf(v, v, s2);
            }
        }
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        if (lh == 0) {
            red[(mg * 64 + nt * 32 + l31) * 2] = s1;
            red[(mg * 64 + nt * 32 + l31) * 2 + 1] = s2;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        {
            float t1 = 0.f, t2 = 0.f;
            const int ch = tid & 63;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                t1 += red[(g * 64 + ch) * 2];
                t2 += red[(g * 64 + ch) * 2 + 1];
            }
            const bool w0 = tid < 64;              // (the other waves store to an out-of-range offset)
            store32(t1, rsP, w0 ? (unsigned)((ch * a.ntiles + tile) * 4) : 0xFFFFFFFFu);
            store32(t2, rsP, w0 ? (unsigned)(((64 + ch) * a.ntiles + tile) * 4) : 0xFFFFFFFFu);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (look-ahead rows must land before the LDS is released)
}

struct StemPlan { int tm, pitch, tiles_per_img, ntiles, rows_max, db; size_t lds; };
bool stem_f32_plan(int N, int H, int W, int Ho, int Wo, StemPlan* p) {
    const int pitch = W + SPREFIX <= 256 ? 256 : W + SPREFIX <= 448 ? 448 : 0;
    if (!pitch) return false;
    // CILRS_STEM_PLAN (experiments): 1 = TM 2 double-buffered, 2 = TM 1 double-buffered,
    // 3 = TM 1 single buffer, two workgroups per CU
    static const int force = experiment_env("CILRS_STEM_PLAN", 0);
    for (int cand = 0; cand < 3; ++cand) {
        const int tm = cand == 0 ? 2 : 1, db = cand < 2 ? 1 : 0;
        if (force && force != cand + 1) continue;
        const int M = 128 * tm;
        const int rmax = (Wo - 1 + M - 1) / Wo + 1;
        const int rows = 2 * (rmax - 1) + 7;
        const size_t lds = (size_t)((db ? 2 : 1) * stem_buf_floats(rows, pitch) + 4 * 64 * 2) * sizeof(float);
        if (lds <= (size_t)(db ? 160 : 80) * 1024) {
            p->tm = tm; p->pitch = pitch; p->rows_max = rows; p->lds = lds; p->db = db;
            p->tiles_per_img = cdiv(Ho * Wo, M);
            p->ntiles = N * p->tiles_per_img;
            return true;
        }
    }
    return false;
}

template <int TM, int PITCH, bool DB>
int launch_stem_f32_t(const StemF32Args& a, const StemPlan& p, hipStream_t s) {
    if (once_per_device(reinterpret_cast<const void*>(&stem_f32_kernel<TM, PITCH, DB>))) {
        // (the row count, hence the LDS size, of one instantiation varies with the image width: set
        //  the limit to the CU's 160 KB once, not to the first caller's size)
        CILRS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_f32_kernel<TM, PITCH, DB>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    const int resident = device_cus() * (DB ? 1 : 2);
    stem_f32_kernel<TM, PITCH, DB><<<p.ntiles < resident ? p.ntiles : resident, 512, p.lds, s>>>(a);
    CILRS_LAUNCH_CHECK();
    return 0;
}
template <int PITCH>
int launch_stem_f32_p(const StemF32Args& a, const StemPlan& p, hipStream_t s) {
    if (!p.db) return launch_stem_f32_t<1, PITCH, false>(a, p, s);
    return p.tm == 2 ? launch_stem_f32_t<2, PITCH, true>(a, p, s) : launch_stem_f32_t<1, PITCH, true>(a, p, s);
}


// ---- weight gradient of the stem ---------------------------------------------------------------------
// dW[co][kh][kw][c] = sum over output pixels of dy[pixel][co] * x[2 oh - 3 + kh][2 ow - 3 + kw][c]
// (loss.backward() through visual_encoder.0, notebook/notebook.ipynb:552).  The implicit-GEMM weight
// gradient pays for the channel pad like the forward pass (K = 196 columns, 64 x 13 tiles of short
// reductions cut into 1,024 slabs): 187 us at B = 128.  Here the product is D[co][tap] with the
// REDUCTION over pixels on v_mfma_f32_32x32x2_f32 (two pixels per step):
//   * A = dy: a lane's operand is (channel l & 31, pixel 2 s + (l >> 5)) -- 128 contiguous bytes per
//     half-wave, read straight from global memory one tile ahead (the register of step s is
//     re-requested for the next tile as soon as step s has used it);
//   * B = the input rows of the tile in LDS (the forward kernel's row image and LDS-DMA copy,
//     double-buffered): a lane's operand is tap j = 32 nt + (l & 31) of that pixel, i.e. a per-lane
//     constant (filter row, column, channel of ITS tap) plus a COMPILE-TIME offset per step (two
//     pixels = 64 bytes further along the row);
//   * every wave keeps the whole 64 x 160 product (2 x 5 accumulators, 147 real columns) for its
//     share of the pixels -- a tile is R output rows, a wave owns 1 / (8 / R) of one row -- through
//     ALL tiles of the workgroup; one cross-wave reduction through LDS at the very end, one
//     64 x 147 slab per workgroup, summed in slab order by a second launch (deterministic).
constexpr int WG_SLACK_PX = 64;          // zeroed pixels behind the last row (operands of masked pixels)
__host__ __device__ constexpr int wg_buf_floats(int rows, int pitch_px) {
    return (SPREFIX + rows * pitch_px + WG_SLACK_PX) * 4;
}

struct StemWgradArgs {
    const float* x4;        // [N][H][W][4]
    const float* dy;        // [N][Ho][Wo][64]
    float* slabs;           // [blocks][64 * 147]
    int N, H, W, Ho, Wo;
    int tiles_per_img, ntiles;
};

template <int PITCH, int R, int NS>
__global__ __launch_bounds__(512) void stem_wgrad_f32_kernel(const StemWgradArgs a) {
    constexpr int WPR = 8 / R;                       // waves per output row
    constexpr int ROWS = 2 * (R - 1) + 7;            // input rows of a tile
    constexpr int BUF = wg_buf_floats(ROWS, PITCH);
    constexpr int SEGS = PITCH / 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int wrow = wave / WPR, part = wave % WPR;
    const int ow_base = part * NS * 2;
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane(
        (int)(unsigned)(size_t)(__attribute__((address_space(3))) float*)smem);
    const s_i32x4 rs = stem_rsrc(a.x4, (unsigned)((size_t)a.N * a.H * a.W * 16));
    const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.dy, 0, (int)(unsigned)((size_t)a.N * a.Ho * a.Wo * 64 * 4), 0x00020000);

    // everything a masked pixel's operand can touch must be finite: prefix and slack of both buffers
    for (int i = tid; i < 2 * (SPREFIX + WG_SLACK_PX) * 4; i += 512) {
        const int b = i / ((SPREFIX + WG_SLACK_PX) * 4), k = i % ((SPREFIX + WG_SLACK_PX) * 4);
        smem[b * BUF + (k < SPREFIX * 4 ? k : (ROWS * PITCH) * 4 + k)] = 0.f;
    }

    auto issue = [&](const int tile, const int buf) {            // the tile's input rows -> buffer `buf`
        if (tile >= a.ntiles) return;
        const int n = tile / a.tiles_per_img, oh0 = (tile - n * a.tiles_per_img) * R;
        const int ih0 = 2 * oh0 - 3;
        const unsigned base = lds0 + (unsigned)(buf * BUF + SPREFIX * 4) * 4u;
        for (int u = wave; u < ROWS * SEGS; u += 8) {            // (wave-uniform)
            const int row = u / SEGS, seg = u - row * SEGS;
            const int ih = ih0 + row, px = seg * 64 + lane;
            const bool ok = ih >= 0 && ih < a.H && px < a.W;
            const unsigned off = ok ? (unsigned)(((n * a.H + ih) * a.W + px) * 16) : 0xFFFFFFFFu;
            const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane(
                (int)(base + (unsigned)((row * PITCH + seg * 64) * 16)));
            stem_dma16(dst, rs, off);
        }
    };
    // dy of (tile, step s, channel half mt) for this lane: pixel ow_base + 2 s + lh of row oh0 + wrow
    const unsigned lane_off = (unsigned)(((ow_base + lh) * 64 + l31) * 4);
    auto dy_load = [&](const int tile, const int s, const int mt) -> float {
        const int n = tile / a.tiles_per_img, oh = (tile - n * a.tiles_per_img) * R + wrow;
        const bool row_ok = tile < a.ntiles && oh < a.Ho;                       // (wave-uniform)
        const bool ok = row_ok && ow_base + 2 * s + lh < a.Wo;
        const unsigned off = ok ? lane_off + (unsigned)(s * 512 + mt * 128) : 0xFFFFFFFFu;
        // (the scalar offset must BE scalar: a per-lane value makes the compiler loop over lanes)
        const int soff = __builtin_amdgcn_readfirstlane(row_ok ? ((n * a.Ho + oh) * a.Wo) * 256 : 0);
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsD, (int)off, soff, 0));
    };
    // this lane's tap of each 32-column group -> float index of its operand of step 0 in buffer 0
    // (filter row, column, channel of the tap; the wave's row of the tile and first pixel); the
    // index moves to the other buffer and back with the tiles
    int bidx[5];
#pragma unroll
    for (int nt = 0; nt < 5; ++nt) {
        const int j = nt * 32 + l31;
        const int kh = j / 21, r = j - kh * 21;
        bidx[nt] = (j < 147 ? (kh * PITCH + r / 3) * 4 + r % 3 : 0) +
                   (2 * wrow * PITCH + 2 * (ow_base + lh)) * 4;
    }

    f32x16 acc[2][5];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 5; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    float av[2][NS];
    int buf = 0;
    issue((int)blockIdx.x, 0);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        av[0][s] = dy_load((int)blockIdx.x, s, 0);
        av[1][s] = dy_load((int)blockIdx.x, s, 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int tile = (int)blockIdx.x; tile < a.ntiles; tile += (int)gridDim.x, buf ^= 1) {
        // this tile's rows have landed: its copies are older than the 2 NS dy loads issued since
        // (a stricter count than needed is safe); every wave is done with the other buffer
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NS) : "memory");
        issue(tile + (int)gridDim.x, buf ^ 1);
        const int next = tile + (int)gridDim.x;
        float bv[2][5];
#pragma unroll
        for (int nt = 0; nt < 5; ++nt) bv[0][nt] = smem[bidx[nt]];
#if (CILRS_STEM_DBG & 64)
        for (int rep = 0; rep < 2; ++rep)                  // (timing experiment: the MFMA loop twice)
#endif
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (s + 1 < NS) {
#pragma unroll
                for (int nt = 0; nt < 5; ++nt) bv[(s + 1) & 1][nt] = smem[bidx[nt] + (s + 1) * 16];
            }
            __builtin_amdgcn_sched_barrier(0);
            const float a0 = av[0][s], a1 = av[1][s];
#pragma unroll
            for (int nt = 0; nt < 5; ++nt) {
                acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv[s & 1][nt], acc[0][nt], 0, 0, 0);
                acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv[s & 1][nt], acc[1][nt], 0, 0, 0);
            }
            av[0][s] = dy_load(next, s, 0);          // the next tile's operand of this step
            av[1][s] = dy_load(next, s, 1);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int nt = 0; nt < 5; ++nt) bidx[nt] += buf ? -BUF : BUF;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // ---- cross-wave reduction, 32 columns per round: red[wave][mt][r][lane] ----
    float* slab = a.slabs + (size_t)blockIdx.x * (64 * 147);
#pragma unroll
    for (int nt = 0; nt < 5; ++nt) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) smem[((wave * 2 + mt) * 16 + r) * 64 + lane] = acc[mt][nt][r];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = tid + 512 * q;                    // (mt, r, lane)
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) t += smem[w * 2048 + e];
            const int mt = e >> 10, r = (e >> 6) & 15, l = e & 63;
            const int co = mt * 32 + 8 * (r >> 2) + 4 * (l >> 5) + (r & 3);
            const int j = nt * 32 + (l & 31);
            if (j < 147) slab[co * 147 + j] = t;
        }
        __syncthreads();
    }
}

// dw[i] = sum over slabs in a fixed order: 32 outputs x 8 slab groups per block, each thread adds the
// slabs k = g, g + 8, ... of its output, thread g = 0 the eight partial sums
__global__ __launch_bounds__(256) void stem_wgrad_reduce_kernel(const float* __restrict__ slabs,
                                                                float* __restrict__ dw, const int nslabs) {
    __shared__ float part[8][32];
    const int o = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + o;
    float t = 0.f;
    if (i < 64 * 147) {
#pragma unroll 8
        for (int k = g; k < nslabs; k += 8) t += slabs[(size_t)k * (64 * 147) + i];
    }
    part[g][o] = t;
    __syncthreads();
    if (g == 0 && i < 64 * 147) {
        float r = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) r += part[q][o];
        dw[i] = r;
    }
}

struct StemWgradPlan { int pitch, R, tiles_per_img, ntiles, grid; size_t lds; };
bool stem_wgrad_plan(int N, int H, int W, int Ho, int Wo, StemWgradPlan* p) {
    const int pitch = W + SPREFIX <= 256 ? 256 : W + SPREFIX <= 448 ? 448 : 0;
    if (!pitch || (size_t)N * Ho * Wo * 256 >= (1ull << 32) || (size_t)N * H * W * 16 >= (1ull << 32))
        return false;
    // 25 steps (50 pixels) per wave: rows of 100 pixels on two waves, of 200 on four
    const int R = pitch == 256 ? 4 : 2;
    if (cdiv(cdiv(Wo, 2), 8 / R) != 25 || 2 * (Wo - 1) + 6 >= pitch + WG_SLACK_PX) return false;
    p->pitch = pitch; p->R = R;
    p->tiles_per_img = cdiv(Ho, R);
    p->ntiles = N * p->tiles_per_img;
    const int cus = device_cus();
    p->grid = p->ntiles < cus ? p->ntiles : cus;
    p->lds = (size_t)2 * wg_buf_floats(2 * (R - 1) + 7, pitch) * sizeof(float);
    if (p->lds < 8 * 2048 * sizeof(float)) p->lds = 8 * 2048 * sizeof(float);
    return p->lds <= 160 * 1024;
}

template <int PITCH, int R>
int launch_stem_wgrad_t(const StemWgradArgs& a, const StemWgradPlan& p, float* dw, hipStream_t s) {
    if (once_per_device(reinterpret_cast<const void*>(&stem_wgrad_f32_kernel<PITCH, R, 25>))) {
        CILRS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_wgrad_f32_kernel<PITCH, R, 25>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds));
    }
    stem_wgrad_f32_kernel<PITCH, R, 25><<<p.grid, 512, p.lds, s>>>(a);
    CILRS_LAUNCH_CHECK();
    stem_wgrad_reduce_kernel<<<cdiv(64 * 147, 32), 256, 0, s>>>(a.slabs, dw, p.grid);
    CILRS_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// rows of the [2][64][rows] column partials the launch writes; 0 = this geometry is not served here
// (the caller keeps the implicit GEMM)
int stem_f32_rows(int N, int H, int W) {
    StemPlan p;
    const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
    if ((size_t)N * H * W * 16 >= (1ull << 32)) return 0;
    return stem_f32_plan(N, H, W, Ho, Wo, &p) ? p.ntiles : 0;
}

int launch_stem_f32(const float* x4, const float* w, float* y, float* bn_partial, int N, int H, int W,
                    hipStream_t s) {
    StemPlan p;
    const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
    CILRS_CHECK(x4 && w && y, "stem_f32: NULL tensor");
    CILRS_CHECK((size_t)N * H * W * 16 < (1ull << 32) && stem_f32_plan(N, H, W, Ho, Wo, &p),
                "stem_f32: geometry %dx%dx%d not served", N, H, W);
    StemF32Args a;
    a.x4 = x4; a.w = w; a.y = y; a.bn_partial = bn_partial;
    a.N = N; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo;
    a.tiles_per_img = p.tiles_per_img; a.ntiles = p.ntiles; a.rows_max = p.rows_max;
    return p.pitch == 256 ? launch_stem_f32_p<256>(a, p, s) : launch_stem_f32_p<448>(a, p, s);
}

// scratch floats of launch_stem_wgrad_f32; 0 = this geometry is not served here
size_t stem_wgrad_f32_scratch_floats(int N, int H, int W) {
    StemWgradPlan p;
    const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
    return stem_wgrad_plan(N, H, W, Ho, Wo, &p) ? (size_t)p.grid * 64 * 147 : 0;
}

int launch_stem_wgrad_f32(const float* x4, const float* dy, float* dw, float* scratch, int N, int H,
                          int W, hipStream_t s) {
    StemWgradPlan p;
    const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
    CILRS_CHECK(x4 && dy && dw && scratch, "stem_wgrad_f32: NULL tensor");
    CILRS_CHECK(stem_wgrad_plan(N, H, W, Ho, Wo, &p), "stem_wgrad_f32: geometry %dx%dx%d not served", N, H, W);
    StemWgradArgs a;
    a.x4 = x4; a.dy = dy; a.slabs = scratch;
    a.N = N; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo;
    a.tiles_per_img = p.tiles_per_img; a.ntiles = p.ntiles;
    return p.pitch == 256 ? launch_stem_wgrad_t<256, 4>(a, p, dw, s) : launch_stem_wgrad_t<448, 2>(a, p, dw, s);
}

}  // namespace cilrs
