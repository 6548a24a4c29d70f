// Winograd F(2x2, 3x3) convolution on the gfx950 exact-f32 matrix pipe.
//
// The stride-1 3x3 convolutions of the BasicBlock stacks (30 of the 36 forward convolutions of the
// reference's ResNet-34 trunk, model/autonomous_drive.py:365-370, and their data gradients) are
// 90 % of the matrix work of a train step; F(2x2, 3x3) does them with 16 multiplications per 2x2
// output tile and input channel instead of 36 (2.25x fewer), at fp32 throughout:
//
//     Y = A^T [ sum_c (G g G^T) .* (B^T d B) ] A          d: 4x4 input patch, g: 3x3 filter
//
// i.e. 16 independent GEMMs, one per position xi of the 4x4 transform domain:
//     M_xi[tile][k] = sum_c V_xi[tile][c] * U_xi[c][k],   V = B^T d B,  U = G g G^T.
// FUSED: V is built in registers from the gathered patch and goes straight to LDS, M stays in the
// accumulators, the inverse transform runs in the epilogue -- neither V nor M ever touches HBM
// (unfused they cost more time than the convolution: 16 planes = ~650 MB per layer1 convolution).
// U is precomputed once per step by wino_weights_kernel (weights change every step), laid out
// [xi][C/8][K][8] so a block's slice of a reduction chunk is contiguous.
//
// Block = 512 threads (8 waves, one block per CU); tile = 64 output tiles (2x2 pixels each) x 64
// output channels x all 16 xi; wave w owns 16 tiles x 32 channels FOR ALL 16 xi (128 accumulator
// registers: 16 xi x 2 blocks of v_mfma_f32_16x16x4_f32 with the CHANNELS as the rows), so a lane
// ends up with the 16 transform-domain values of four consecutive channels of one tile and the
// inverse transform is register-local (the first version gave each wave 2 xi of the whole tile on
// 32x32x2 MFMAs and exchanged the accumulators through LDS: 17.5k cycles of epilogue per block
// against 4.5k here).  Reduction in chunks of 8 input channels:
//   * gather + transform: four waves (one per SIMD), thread = (tile, channel pair): 16 8-byte
//     buffer loads (a patch pixel outside the image gets the offset ~0 = out of range = zeros),
//     32 packed additions, 16 8-byte LDS stores;
//   * U chunk: 32 KB, contiguous 2 KB per xi;  both are fetched one chunk ahead into registers;
//   * multiply: per xi one V and two U fragments (8-byte LDS reads, 512 contiguous bytes per
//     wave, read two xi ahead), four MFMAs; two LDS stages;
//   * epilogue: inverse transform in registers, then the same fused work as the implicit-GEMM
//     kernel: residual / addend, BatchNorm batch statistics (forward) or BatchNorm-backward
//     reductions (data gradient) as per-block column partials, 16-byte stores (64 contiguous
//     bytes per tile pixel and wave).
// fp32 Winograd is not bit-identical to the direct sum (other rounding points): outputs differ
// by a few 1e-6 relative, inside the 1e-4 contract (tests/test_ops_gpu.py).
#include "common.h"

#include <type_traits>

namespace cilrs {
namespace {

constexpr int WT = 64;             // output tiles per block
constexpr int WK = 64;             // output channels per block
constexpr int WC = 8;              // input channels per reduction chunk
constexpr int WP = 8;              // LDS row pitch (floats) of the V / U images: no pad -- TWO stages fit
                                   // (2 x 64 KB) and a fragment read covers 512 contiguous bytes
constexpr int WTHREADS = 512;
// Position (floats) of (row r, channel pair p) inside one transform position's plane of ROWS rows x
// 8 channels: TWO sub-planes of ROWS x 16 bytes, pair p in sub-plane p >> 1.  An operand fragment
// read is one ds_read_b64 per lane, lane = (row l & 15, pair l >> 4); ds_read_b64 resolves bank
// conflicts per 32-lane half over 64 banks (= 256 bytes), and a half -- 16 rows x pairs {0, 1} or
// {2, 3} -- then reads 256 CONTIGUOUS bytes.  With plain 32-byte rows (round 3's layout) a half
// read 16-byte pieces 32 bytes apart, lanes l and l + 8 on the same banks: every fragment read
// took two LDS passes (rocprofv3: SQ_LDS_BANK_CONFLICT 80 per GPU cycle on this kernel, 0 on every
// other one -- profiles/r04_pmc/).  Measured alternatives: 32-byte rows with the pair slot XOR-ed
// by row bit 3 (also 64 distinct banks per half, but two 128-byte pieces instead of one 256-byte
// run): fewer counted conflicts (30 vs 45) and a SLOWER step (10.00 vs 9.75 ms) -- contiguity of
// the read matters more than the counter.  The 16-byte U stores are assigned so that a wave writes
// 1 KB of ONE sub-plane (row = lane): with consecutive lanes alternating between the sub-planes
// (the global order) even and odd lanes of an 8-lane group landed on the same banks.
template <int ROWS> __device__ __forceinline__ int plane_pos(const int r, const int p) {
    return (p >> 1) * (ROWS * 4) + r * 4 + (p & 1) * 2;
}
typedef unsigned int u32x4w __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// timing experiments (tools/wino_dbg.sh; results meaningless): 1 no global loads, 2 no transform +
// LDS stores of a chunk, 4 no MFMAs, 8 no LDS fragment reads
#ifndef CILRS_WINO_DBG
#define CILRS_WINO_DBG 0
#endif
// K loop structure (compile-time: a run-time choice between the two bodies spills ~400 VGPRs):
// 0 = two phases per chunk (half the waves multiply while the other half refills, then swap; two
// barriers), 1 = one barrier per chunk (every wave multiplies chunk ch, then stores its part of
// chunk ch + 1 into the other stage; measured 6 % slower on the step, profiles/r04_negative_results.log)
#ifndef CILRS_WINO_ONE_PHASE
#define CILRS_WINO_ONE_PHASE 0
#endif

// u = G g G^T;  G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
__device__ __forceinline__ void wino_filter(const float (&g)[3][3], float (&u)[16]) {
    float t[4][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        t[0][b] = g[0][b];
        t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
        t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
        t[3][b] = g[2][b];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        u[a * 4 + 0] = t[a][0];
        u[a * 4 + 1] = 0.5f * (t[a][0] + t[a][1] + t[a][2]);
        u[a * 4 + 2] = 0.5f * (t[a][0] - t[a][1] + t[a][2]);
        u[a * 4 + 3] = t[a][2];
    }
}

// every layer, both forms (WinoWeightTable): thread = (k, c) of an 8 x 32 block of one layer
__global__ __launch_bounds__(256) void wino_weights_all_kernel(const WinoWeightTable t,
                                                               const float* __restrict__ params,
                                                               float* __restrict__ ubase) {
    __shared__ float sf[16][8][33], sb[16][8][33];
    int e = 0;
    while (e + 1 < t.n && (int)blockIdx.x >= t.blk_begin[e + 1]) ++e;      // block-uniform
    const int K = t.K[e], C = t.C[e];
    const int b = blockIdx.x - t.blk_begin[e], ncb = C >> 5;
    const int kb = b / ncb, cb = b - kb * ncb;
    const int tid = threadIdx.x, kk = tid >> 5, cc = tid & 31;
    const float* w = params + t.w[e] + ((size_t)(kb * 8 + kk) * 9) * C + cb * 32 + cc;
    float g[3][3], gf[3][3], uf[16], ub[16];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) g[i][j] = w[(size_t)(i * 3 + j) * C];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) gf[i][j] = g[2 - i][2 - j];
    wino_filter(g, uf);
    wino_filter(gf, ub);
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) { sf[xi][kk][cc] = uf[xi]; sb[xi][kk][cc] = ub[xi]; }
    __syncthreads();
    float* U = ubase + t.u[e];
    float* Ud = ubase + t.ud[e];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int f = it * 256 + tid;
        {   // forward form [xi][C/8][K][8]: float4 index = [xi 16][c8 4][k 8][2]
            const int q = f & 1, k = (f >> 1) & 7, c8 = (f >> 4) & 3, xi = f >> 6;
            const float* src = &sf[xi][k][c8 * 8 + q * 4];
            *reinterpret_cast<f32x4*>(U + (((size_t)xi * (C >> 3) + cb * 4 + c8) * K + kb * 8 + k) * 8 + q * 4) =
                f32x4{src[0], src[1], src[2], src[3]};
        }
        {   // data-gradient form [xi][K/8][C][8]: float4 index = [xi 16][c 32][2]
            const int q = f & 1, c = (f >> 1) & 31, xi = f >> 6;
            *reinterpret_cast<f32x4*>(Ud + (((size_t)xi * (K >> 3) + kb) * C + cb * 32 + c) * 8 + q * 4) =
                f32x4{sb[xi][q * 4 + 0][c], sb[xi][q * 4 + 1][c], sb[xi][q * 4 + 2][c], sb[xi][q * 4 + 3][c]};
        }
    }
}

// U[xi][c/8][k][c%8] = (G g G^T)[xi] of g = w[k][.][.][c]  (OHWI weights)
// dgrad = 1: the data-gradient filter instead: g' = 180-degree flip of w[c_out..] with the channel
// roles swapped, i.e. U[xi][k/8][c][k%8] from w[k][2-kh][2-kw][c]
__global__ __launch_bounds__(256) void wino_weights_kernel(const float* __restrict__ w,
                                                           float* __restrict__ U, const int K,
                                                           const int C, const int dgrad) {
    // one thread per (k, c); c fastest: coalesced reads of w[k][t][c]
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= K * C) return;
    const int k = i / C, c = i - k * C;
    float g[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
            g[a][b] = w[((size_t)k * 9 + (dgrad ? (2 - a) * 3 + (2 - b) : a * 3 + b)) * C + c];
    // t = G g (4x3), u = t G^T (4x4); G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
    float t[4][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        t[0][b] = g[0][b];
        t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
        t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
        t[3][b] = g[2][b];
    }
    // reduction channel ci, output channel co of the GEMM this U feeds
    const int ci = dgrad ? k : c, co = dgrad ? c : k;
    const int Cin = dgrad ? K : C, Cout = dgrad ? C : K;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const float u0 = t[a][0];
        const float u1 = 0.5f * (t[a][0] + t[a][1] + t[a][2]);
        const float u2 = 0.5f * (t[a][0] - t[a][1] + t[a][2]);
        const float u3 = t[a][2];
        const float uu[4] = {u0, u1, u2, u3};
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int xi = a * 4 + b;
            U[(((size_t)xi * (Cin >> 3) + (ci >> 3)) * Cout + co) * 8 + (ci & 7)] = uu[b];
        }
    }
}

// PRE (launches with an addend and / or the BatchNorm-backward reductions, no channel split): the
// epilogue's operands do not wait for the K loop to end.  The addend is loaded in the prologue and
// planted in the accumulators -- A^T M A is linear and M[0][0], -M[0][3], -M[3][0], M[3][3] reach
// exactly one output pixel each -- and the y / z tensors of the reductions are requested when the
// last chunk's global loads have been consumed (their registers are free from then on; half of them:
// the register file holds 32 more, the other half is requested right after the loop), so the
// burst of 2 x 64 KB per block that all CUs used to issue TOGETHER after their last MFMA runs
// under the last one and a half chunks instead.
template <bool PRE>
__device__ __forceinline__ void wino_full_body(const WinoArgs& a, float* smem, const int bid) {
    constexpr int STAGE = 16 * WT * WP + 16 * WK * WP;      // floats per stage (64 KB)
    float* Vs = smem;                          // [2 stages]: [16][WT][WP] | [16][WK][WP]
    float* Us = smem + 16 * WT * WP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int tg = wave & 3, kg = wave >> 2;   // this wave's 16 tiles / 32 output channels
    const int TH = (a.H + 1) >> 1, TW = (a.W + 1) >> 1;
    const int tiles_img = TH * TW, total_tiles = a.N * tiles_img;
    const int nkt = a.K / WK;
    // channel split: consecutive blocks = the parts of one tile's reduction
    const int csplit = a.csplit > 1 ? a.csplit : 1;
    const int part = bid % csplit;
    const int tile_id = bid / csplit;
    const int blk_n = tile_id % nkt, blk_m = tile_id / nkt;
    const int k0 = blk_n * WK;
    const int nchunks_all = a.C / WC;
    const int per_part = (nchunks_all + csplit - 1) / csplit;
    const int ch0 = part * per_part;                                   // first chunk of this block
    const int nchunks = max(0, min(nchunks_all, ch0 + per_part) - ch0); // chunks of this block

    // ---- gather role (waves 0-3, one per SIMD): this thread's (tile, channel PAIR) and the 16
    //      patch pixels; 8-byte loads, packed additions, 8-byte LDS stores ----
    const bool gatherer = tid < 256;
    // (lane = (tile l >> 2, channel pair l & 3): four adjacent lanes fetch one pixel's 32 bytes.  With
    //  lane = (tile l & 15, pair l >> 4) the V stores of a half-wave become one 256-byte run -- fewer
    //  LDS conflicts, 38 -> 32 per cycle -- but the patch loads coalesce worse: step 9.75 -> 10.26 ms)
    const int g_tile = (tid >> 2) & 63, g_p = tid & 3;
    unsigned voff[16];
    {
        const int t = a.tile_begin + blk_m * WT + g_tile;
        const bool tv = t < total_tiles && gatherer;
        const int n = tv ? t / tiles_img : 0, rem = t - n * tiles_img;
        const int ty = rem / TW, tx = rem - ty * TW;
        const int h0 = 2 * ty - 1, w0 = 2 * tx - 1;
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const int h = h0 + (p >> 2), w = w0 + (p & 3);
            const bool ok = tv && h >= 0 && w >= 0 && h < a.H && w < a.W;
            voff[p] = ok ? (unsigned)((((n * a.H + h) * a.W + w) * a.C + g_p * 2) * 4) : 0xFFFFFFFFu;
        }
    }
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.x, 0, (int)(unsigned)((size_t)a.N * a.H * a.W * a.C * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsU = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.U, 0, (int)(unsigned)((size_t)16 * a.C * a.K * 4), 0x00020000);
    // U chunk: thread handles 4 float4: xi = 4 pass + (tid >> 7), float4 index tid & 127 of the
    // xi's contiguous [64 k][8 c] slice
    const int u_row = tid & 63, u_half = (tid >> 6) & 1, u_xi0 = tid >> 7;   // 16 bytes: row, channels 4 u_half ..

    f32x2 d[16];
    f32x4 ur[4];
    auto load_chunk = [&](const int ch_local) {
        const int ch = ch0 + ch_local;                  // absolute chunk of the reduction
        if (CILRS_WINO_DBG & 1) {
#pragma unroll
            for (int p = 0; p < 16; ++p) d[p] = f32x2{(float)(p + ch), 1.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) ur[q] = f32x4{1.f, 2.f, 3.f, (float)ch};
            return;
        }
        const int soff = ch * WC * 4;
        if (gatherer) {
#pragma unroll
            for (int p = 0; p < 16; ++p)
                d[p] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsX, (int)voff[p], soff, 0));
        }
        const int usoff = (ch * a.K + k0) * 32;           // bytes: [ch][k0] of an xi plane
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int xi = q * 4 + u_xi0;
            const unsigned off = (unsigned)(xi * nchunks_all * a.K * 8 + (u_row * 2 + u_half) * 4) * 4u;
            ur[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsU, (int)off, usoff, 0));
        }
    };
    // V = B^T d B in two steps, in place;  B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]]
    auto transform_col = [&](const int j) {         // r = B^T d, column j of the patch
        const f32x2 r0 = d[0 * 4 + j] - d[2 * 4 + j];
        const f32x2 r1 = d[1 * 4 + j] + d[2 * 4 + j];
        const f32x2 r2 = d[2 * 4 + j] - d[1 * 4 + j];
        const f32x2 r3 = d[1 * 4 + j] - d[3 * 4 + j];
        d[0 * 4 + j] = r0; d[1 * 4 + j] = r1; d[2 * 4 + j] = r2; d[3 * 4 + j] = r3;
    };
    auto transform_row = [&](const int i, const int stage) {     // V row i = r B, stored
        const f32x2 v0 = d[i * 4 + 0] - d[i * 4 + 2];
        const f32x2 v1 = d[i * 4 + 1] + d[i * 4 + 2];
        const f32x2 v2 = d[i * 4 + 2] - d[i * 4 + 1];
        const f32x2 v3 = d[i * 4 + 1] - d[i * 4 + 3];
        float* dst = Vs + stage * STAGE + (i * 4) * (WT * WP) + plane_pos<WT>(g_tile, g_p);
        *reinterpret_cast<f32x2*>(dst + 0 * WT * WP) = v0;
        *reinterpret_cast<f32x2*>(dst + 1 * WT * WP) = v1;
        *reinterpret_cast<f32x2*>(dst + 2 * WT * WP) = v2;
        *reinterpret_cast<f32x2*>(dst + 3 * WT * WP) = v3;
    };
    auto store_u = [&](const int stage) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<f32x4*>(Us + stage * STAGE + (q * 4 + u_xi0) * (WK * WP) + u_half * (WK * 4) + u_row * 4) = ur[q];
    };
    auto store_chunk = [&](const int stage) {
        if (CILRS_WINO_DBG & 2) {
            asm volatile("" ::"v"(d[0]), "v"(d[15]), "v"(ur[0]), "v"(ur[3]));
            return;
        }
        if (gatherer) {
#pragma unroll
            for (int j = 0; j < 4; ++j) transform_col(j);
#pragma unroll
            for (int i = 0; i < 4; ++i) transform_row(i, stage);
        }
        store_u(stage);
    };

    // M^T[xi] (channels x tiles) of this wave: rows = 2 x 16 output channels, columns = 16 tiles.
    // v_mfma_f32_16x16x4_f32: lane l holds rows 4 (l >> 4) .. + 3 of column l & 15, i.e. FOUR
    // CONSECUTIVE CHANNELS OF ONE TILE for all 16 xi: the inverse transform is register-local.
    f32x4 acc[16][2];
#pragma unroll
    for (int x = 0; x < 16; ++x)
#pragma unroll
        for (int m = 0; m < 2; ++m) acc[x][m] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- epilogue coordinates: this lane's tile (one per lane) and its 2 x 4 consecutive channels ----
    const int e_t = a.tile_begin + blk_m * WT + tg * 16 + l15;
    const bool e_tv = e_t < total_tiles;
    const int e_n = e_tv ? e_t / tiles_img : 0, e_rem = e_t - e_n * tiles_img;
    const int e_ty = e_rem / TW, e_tx = e_rem - e_ty * TW;
    const int co = k0 + kg * 32 + lq * 4;                   // + 16 mb
    const size_t ybytes = (size_t)a.N * a.H * a.W * a.K * 4;
    const int yrec = (int)(unsigned)ybytes;
    unsigned yoff[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int oy = 2 * e_ty + (q >> 1), ox = 2 * e_tx + (q & 1);
        const bool ok = e_tv && oy < a.H && ox < a.W;
        yoff[q] = ok ? (unsigned)((((e_n * a.H + oy) * a.W + ox) * a.K + co) * 4) : 0xFFFFFFFFu;
    }
    // (a tensor the launch does not have gets a zero-length buffer: its loads return zeros and
    //  cost no traffic, so every load below is issued and consumed unconditionally)
    f32x4 zz[2][4], yy[2][4];
    auto prefetch_epilogue = [&](const int mb) {     // (one 16-channel half: 32 registers)
        const __amdgpu_buffer_rsrc_t rsZ = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(a.bwd_z ? a.bwd_z : a.y), 0, (a.bwd_partial && a.bwd_relu) ? yrec : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(a.bwd_y ? a.bwd_y : a.y), 0, a.bwd_partial ? yrec : 0, 0x00020000);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            zz[mb][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsZ, (int)yoff[q], mb * 64, 0));
            yy[mb][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, (int)yoff[q], mb * 64, 0));
        }
    };
    if constexpr (PRE) {
        const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(a.addend ? a.addend : a.y), 0, a.addend ? yrec : 0, 0x00020000);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            f32x4 ad[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                ad[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)yoff[q], mb * 64, 0));
            acc[0][mb] = ad[0];          // M[0][0] -> Y[0][0]
            acc[3][mb] = -ad[1];         // M[0][3] -> -Y[0][1]
            acc[12][mb] = -ad[2];        // M[3][0] -> -Y[1][0]
            acc[15][mb] = ad[3];         // M[3][3] -> Y[1][1]
        }
    }

    const long long tm_start = (a.stamps != nullptr && bid == 0) ? __builtin_amdgcn_s_memtime() : 0;
    load_chunk(0);
    store_chunk(0);
    if (nchunks > 1) load_chunk(1);
    __syncthreads();
    // operand fragments: the k index of a lane is lq, it covers channels 2 lq, 2 lq + 1 of the
    // chunk (one 8-byte read feeds the two k-steps); 64 lanes read 512 contiguous bytes
    const int frag_v = plane_pos<WT>(tg * 16 + l15, lq);
    const int frag_u = plane_pos<WK>(kg * 32 + l15, lq);
    // fragments of xi + D are read while the MFMAs of xi run (an LDS read that an MFMA waits on
    // exposes its whole latency: with eight waves reading it is several hundred cycles)
    auto multiply = [&](const int ch) {
        const float* Vc = Vs + (ch & 1) * STAGE + frag_v;
        const float* Uc = Us + (ch & 1) * STAGE + frag_u;
        constexpr int D = 2;
        f32x2 bv[D + 1], a0[D + 1], a1[D + 1];
#pragma unroll
        for (int x = 0; x < D; ++x) {
            bv[x] = *reinterpret_cast<const f32x2*>(Vc + x * (WT * WP));
            a0[x] = *reinterpret_cast<const f32x2*>(Uc + x * (WK * WP));
            a1[x] = *reinterpret_cast<const f32x2*>(Uc + x * (WK * WP) + 16 * 4);
        }
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) {
            const int cur = xi % (D + 1), nxt = (xi + D) % (D + 1);
            if (CILRS_WINO_DBG & 8) {
                bv[nxt] = f32x2{(float)xi, 1.f}; a0[nxt] = f32x2{(float)lane, 2.f}; a1[nxt] = f32x2{3.f, (float)xi};
            } else if (xi + D < 16) {
                bv[nxt] = *reinterpret_cast<const f32x2*>(Vc + (xi + D) * (WT * WP));
                a0[nxt] = *reinterpret_cast<const f32x2*>(Uc + (xi + D) * (WK * WP));
                a1[nxt] = *reinterpret_cast<const f32x2*>(Uc + (xi + D) * (WK * WP) + 16 * 4);
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if (!(CILRS_WINO_DBG & 4)) {
                    acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[cur][s], bv[cur][s], acc[xi][0], 0, 0, 0);
                    acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[cur][s], bv[cur][s], acc[xi][1], 0, 0, 0);
                } else {
                    acc[xi][0][s] += a0[cur][s] * bv[cur][s];
                    acc[xi][1][s] += a1[cur][s] * bv[cur][s];
                }
            }
            if (xi + D < 16) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);    // 3 LDS reads,
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                      // then 4 MFMAs
        }
    };
    // chunk ch + 1 into the other stage.  KIND (compile time): 0 = a chunk with two successors
    // (store ch + 1, request ch + 2), 1 = the second to last (store the last chunk; PRE: request the
    // epilogue's operands into the chunk registers, which are dead from here on), 2 = the last.
    // The last two chunks are peeled out of the loop: inside it the compiler would have to keep the
    // prefetched operands AND the chunk registers alive together (+64 registers = spills).
    auto refill = [&](const int ch, auto kind) {
        constexpr int KIND = decltype(kind)::value;
        if constexpr (KIND == 0) {
            store_chunk((ch + 1) & 1);
            load_chunk(ch + 2);
        } else if constexpr (KIND == 1) {
            store_chunk((ch + 1) & 1);
            if constexpr (PRE) prefetch_epilogue(0);
        }
    };
    // Two phases per chunk, a barrier after each: waves 0-3 multiply while waves 4-7 store their
    // part of the next chunk, then the roles swap (one wave's MFMAs over its SIMD partner's
    // transform + LDS stores).  Measured alternatives, cycles per chunk on layer1 (block 0):
    // both waves multiply, then both refill (fair-shared pipe, transform with the pipe idle) 6.9k;
    // the same with s_setprio for the gathering wave 6.9k (MFMA arbitration ignores it); the
    // transform cut into pieces BETWEEN the MFMA groups of the gathering wave 6.9k; this form 6.3k;
    // MFMAs alone (no loads, no LDS traffic) 5.0k, of which 4.1k is the pipe's issue rate.
    const bool late = wave >= 4;
    // diagnostics (WinoArgs::stamps, tools/wino_bench.py): where block 0's waves 0 and 4 spend a
    // launch, in shader cycles: [prologue, first phase incl. its barrier, second phase, end
    // barrier, epilogue, K loop]
    long long tm_a = 0, tm_b = 0, tm_bar = 0;
    const bool stamp = a.stamps != nullptr && bid == 0 && (tid == 0 || tid == 256);
    const long long tm0 = tm_start;
    long long tm_loop0 = 0;
    if (stamp) tm_loop0 = __builtin_amdgcn_s_memtime();
    auto chunk_step = [&](const int ch, auto kind) {
        long long ta = 0, tb = 0, tc = 0;
        if (stamp) ta = __builtin_amdgcn_s_memtime();
        if constexpr (CILRS_WINO_ONE_PHASE != 0) {
            // ONE barrier per chunk: every wave multiplies chunk ch (stage ch & 1), then stores its
            // part of chunk ch + 1 into the OTHER stage -- nobody reads that stage before the
            // barrier, and nobody overwrites stage ch & 1 before every wave has passed it
            multiply(ch);
            refill(ch, kind);
            if (stamp) tb = tc = __builtin_amdgcn_s_memtime();
            __syncthreads();
            if (stamp) { tm_a += tb - ta; tm_bar += __builtin_amdgcn_s_memtime() - tc; }
        } else {
            if (late) {
                refill(ch, kind);
                __builtin_amdgcn_sched_barrier(0);
                __syncthreads();
                if (stamp) tb = __builtin_amdgcn_s_memtime();
                multiply(ch);
            } else {
                multiply(ch);
                __builtin_amdgcn_sched_barrier(0);
                __syncthreads();
                if (stamp) tb = __builtin_amdgcn_s_memtime();
                refill(ch, kind);
            }
            if (stamp) tc = __builtin_amdgcn_s_memtime();
            __syncthreads();
            if (stamp) {
                tm_a += tb - ta;
                tm_b += tc - tb;
                tm_bar += __builtin_amdgcn_s_memtime() - tc;
            }
        }
    };
    {
        int ch = 0;
        for (; ch + 2 < nchunks; ++ch) chunk_step(ch, std::integral_constant<int, 0>{});
        if (nchunks >= 2) {
            chunk_step(ch, std::integral_constant<int, 1>{});
            ++ch;
        } else if constexpr (PRE) {
            prefetch_epilogue(0);
        }
        if (nchunks >= 1) chunk_step(ch, std::integral_constant<int, 2>{});
    }
    const long long tm_loop1 = stamp ? __builtin_amdgcn_s_memtime() : 0;

    // ---- epilogue ----
    if constexpr (PRE) prefetch_epilogue(1);      // the second half's operands under the first half's transform
    // channel split: this block's partial result goes to slab `part`, raw (the reduce launch adds
    // the addend and takes the column partials of the finished tensor)
    const bool split = csplit > 1;
    float* const ydst = split ? a.slabs + (size_t)part * (ybytes / 4) : a.y;
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc((void*)ydst, 0, yrec, 0x00020000);
    const bool stats_fwd = !split && a.bn_partial != nullptr, stats_bwd = !split && a.bwd_partial != nullptr;
    const bool with_add = !PRE && !split && a.addend != nullptr;
    f32x4 ad[2][4];
    if (with_add) {
        const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.addend, 0, yrec, 0x00020000);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                ad[mb][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)yoff[q], mb * 64, 0));
    }
    f32x4 cs1[2], cs2[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        // Y = A^T M A;  A^T = [[1,1,1,0],[0,1,-1,-1]]
        f32x4 s0[4], s1[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            s0[q] = acc[0 * 4 + q][mb] + acc[1 * 4 + q][mb] + acc[2 * 4 + q][mb];
            s1[q] = acc[1 * 4 + q][mb] - acc[2 * 4 + q][mb] - acc[3 * 4 + q][mb];
        }
        f32x4 y[4];
        y[0] = s0[0] + s0[1] + s0[2];
        y[1] = s0[1] - s0[2] - s0[3];
        y[2] = s1[0] + s1[1] + s1[2];
        y[3] = s1[1] - s1[2] - s1[3];
        if (with_add) {
#pragma unroll
            for (int q = 0; q < 4; ++q) y[q] += ad[mb][q];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4w, y[q]), rsY, (int)yoff[q], mb * 64, 0);
        cs1[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
        cs2[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (stats_fwd) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = yoff[q] != 0xFFFFFFFFu ? y[q] : f32x4{0.f, 0.f, 0.f, 0.f};
                cs1[mb] += v;
                cs2[mb] += v * v;
            }
        } else if (stats_bwd) {
            const f32x4 bmean = *reinterpret_cast<const f32x4*>(a.bwd_stats + co + mb * 16);
            const f32x4 brstd = *reinterpret_cast<const f32x4*>(a.bwd_stats + a.K + co + mb * 16);
            if constexpr (!PRE) {
                const __amdgpu_buffer_rsrc_t rsZ = __builtin_amdgcn_make_buffer_rsrc((void*)a.bwd_z, 0, yrec, 0x00020000);
                const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)a.bwd_y, 0, yrec, 0x00020000);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    zz[mb][q] = f32x4{1.f, 1.f, 1.f, 1.f};
                    if (a.bwd_relu)
                        zz[mb][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsZ, (int)yoff[q], mb * 64, 0));
                    yy[mb][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, (int)yoff[q], mb * 64, 0));
                }
            }
            const bool relu = a.bwd_relu != 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool ok = yoff[q] != 0xFFFFFFFFu;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float g = (ok && (!relu || zz[mb][q][e] > 0.f)) ? y[q][e] : 0.f;
                    cs1[mb][e] += g;
                    cs2[mb][e] = fmaf(g, (yy[mb][q][e] - bmean[e]) * brstd[e], cs2[mb][e]);
                }
            }
        }
    }
    if (stamp) {
        long long* o = a.stamps + (tid == 0 ? 0 : 8);
        o[0] = tm_loop0 - tm0; o[1] = tm_a; o[2] = tm_b; o[3] = tm_bar;
        o[4] = __builtin_amdgcn_s_memtime() - tm_loop1; o[5] = tm_loop1 - tm_loop0;
    }
    // column partials of this block: [2][K][groups] like the implicit-GEMM kernel's (channel-major);
    // the 64 tiles of a channel are summed in fixed order
    if (stats_fwd || stats_bwd) {
        float* red = smem;                      // [2 which][64 tiles][64 channels] (the K loop ended on a barrier)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            const int c = kg * 32 + mb * 16 + lq * 4;
            *reinterpret_cast<f32x4*>(red + (0 * 64 + tg * 16 + l15) * 64 + c) = cs1[mb];
            *reinterpret_cast<f32x4*>(red + (1 * 64 + tg * 16 + l15) * 64 + c) = cs2[mb];
        }
        __syncthreads();
        if (tid < 128) {                        // (which, channel)
            const int which = tid >> 6, col = tid & 63;
            float s = 0.f;
#pragma unroll 8
            for (int q = 0; q < 64; ++q) s += red[(which * 64 + q) * 64 + col];
            float* dst = stats_fwd ? a.bn_partial : a.bwd_partial;
            dst[(size_t)(which * a.K + k0 + col) * a.rows + a.row0 + blk_m] = s;
        }
    }
}

// ---- the tail of a launch: 16 tiles x 64 channels per block, producer / consumer waves ------------
// A launch of the kernel above is N x tiles / 64 blocks of ~25-85 us on 256 CUs: layer1 needs 2.15
// rounds and pays 3, layer2 1.22 and pays 2.  The blocks of the last, mostly empty round are cut
// into four: this kernel gives a CU a quarter of a block's tiles (so four times as many CUs share
// the round) and runs them in about a third of a block's time.  Waves 0-3 only multiply (16 tiles x
// 16 channels x all 16 positions each, 64 accumulator registers); waves 4-7 only produce: wave 4
// gathers + transforms the patch, all four stream the U chunk (the same 32 KB as a full block --
// which is why this is the tail's kernel, not everybody's: at 16 tiles per block the L2 -> LDS
// stream of U, 27 B/clk per CU, is the bound).  Producers run TWO chunks ahead in two register
// sets (they have no accumulators); two LDS stages (80 KB), one barrier per chunk.
constexpr int QT = 16;
constexpr int QSTAGE = 16 * QT * WP + 16 * WK * WP;          // floats per stage (40 KB)
constexpr size_t kWinoQLds = (size_t)2 * QSTAGE * sizeof(float);
__device__ __forceinline__ void wino_q_body(const WinoArgs& a, float* smem, const int bid) {
    float* Vs = smem;                          // [2 stages]: [16][QT][WP] | [16][WK][WP]
    float* Us = smem + 16 * QT * WP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const bool consumer = wave < 4;
    const int TH = (a.H + 1) >> 1, TW = (a.W + 1) >> 1;
    const int tiles_img = TH * TW, total_tiles = a.N * tiles_img;
    const int nkt = a.K / WK;
    const int blk_n = bid % nkt, blk_m = bid / nkt;
    const int k0 = blk_n * WK;
    const int nchunks = a.C / WC;
    const int tile0 = a.tile_begin + blk_m * QT;

    // ---- producers ----
    const int ptid = tid - 256;                               // 0..255 for waves 4-7
    const bool gatherer = wave == 4;
    const int g_tile = (ptid >> 2) & 15, g_p = ptid & 3;
    unsigned voff[16];
    {
        const int t = tile0 + g_tile;
        const bool tv = gatherer && t < total_tiles;
        const int n = tv ? t / tiles_img : 0, rem = t - n * tiles_img;
        const int ty = rem / TW, tx = rem - ty * TW;
        const int h0 = 2 * ty - 1, w0 = 2 * tx - 1;
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const int h = h0 + (p >> 2), w = w0 + (p & 3);
            const bool ok = tv && h >= 0 && w >= 0 && h < a.H && w < a.W;
            voff[p] = ok ? (unsigned)((((n * a.H + h) * a.W + w) * a.C + g_p * 2) * 4) : 0xFFFFFFFFu;
        }
    }
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.x, 0, (int)(unsigned)((size_t)a.N * a.H * a.W * a.C * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsU = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.U, 0, (int)(unsigned)((size_t)16 * a.C * a.K * 4), 0x00020000);
    // U chunk = 2,048 float4: producer thread p takes float4 p + 256 j (j < 8): xi = 2 j + (p >> 7),
    // float4 p & 127 of the xi's contiguous [64 k][8 c] slice
    const int u_row = ptid & 63, u_half = (ptid >> 6) & 1, u_xi0 = (ptid >> 7) & 1;
    auto load_set = [&](f32x2(&d)[16], f32x4(&u)[8], const int ch) {
        const int soff = ch * WC * 4;
        if (gatherer) {
#pragma unroll
            for (int p = 0; p < 16; ++p)
                d[p] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsX, (int)voff[p], soff, 0));
        }
        const int usoff = (ch * a.K + k0) * 32;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int xi = 2 * j + u_xi0;
            const unsigned off = (unsigned)(xi * nchunks * a.K * 8 + (u_row * 2 + u_half) * 4) * 4u;
            u[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsU, (int)off, usoff, 0));
        }
    };
    auto store_set = [&](f32x2(&d)[16], const f32x4(&u)[8], const int stage) {
        if (gatherer) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x2 r0 = d[0 * 4 + j] - d[2 * 4 + j];
                const f32x2 r1 = d[1 * 4 + j] + d[2 * 4 + j];
                const f32x2 r2 = d[2 * 4 + j] - d[1 * 4 + j];
                const f32x2 r3 = d[1 * 4 + j] - d[3 * 4 + j];
                d[0 * 4 + j] = r0; d[1 * 4 + j] = r1; d[2 * 4 + j] = r2; d[3 * 4 + j] = r3;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x2 v0 = d[i * 4 + 0] - d[i * 4 + 2];
                const f32x2 v1 = d[i * 4 + 1] + d[i * 4 + 2];
                const f32x2 v2 = d[i * 4 + 2] - d[i * 4 + 1];
                const f32x2 v3 = d[i * 4 + 1] - d[i * 4 + 3];
                float* dst = Vs + stage * QSTAGE + (i * 4) * (QT * WP) + plane_pos<QT>(g_tile, g_p);
                *reinterpret_cast<f32x2*>(dst + 0 * QT * WP) = v0;
                *reinterpret_cast<f32x2*>(dst + 1 * QT * WP) = v1;
                *reinterpret_cast<f32x2*>(dst + 2 * QT * WP) = v2;
                *reinterpret_cast<f32x2*>(dst + 3 * QT * WP) = v3;
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            *reinterpret_cast<f32x4*>(Us + stage * QSTAGE + (2 * j + u_xi0) * (WK * WP) + u_half * (WK * 4) + u_row * 4) = u[j];
    };

    // ---- consumers: rows = 16 channels (k0 + 16 wave ..), columns = the 16 tiles ----
    f32x4 acc[16];
#pragma unroll
    for (int x = 0; x < 16; ++x) acc[x] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int frag_v = plane_pos<QT>(l15, lq);
    const int frag_u = plane_pos<WK>((wave & 3) * 16 + l15, lq);
    auto multiply = [&](const int ch) {
        const float* Vc = Vs + (ch & 1) * QSTAGE + frag_v;
        const float* Uc = Us + (ch & 1) * QSTAGE + frag_u;
        // pairs of positions, k-steps interleaved: consecutive MFMAs never share an accumulator
        // (40-cycle dependent latency against a 32-cycle issue); reads run two pairs ahead
        constexpr int D = 2;
        f32x2 bv[D + 1][2], av[D + 1][2];
#pragma unroll
        for (int q = 0; q < D; ++q)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                bv[q][e] = *reinterpret_cast<const f32x2*>(Vc + (2 * q + e) * (QT * WP));
                av[q][e] = *reinterpret_cast<const f32x2*>(Uc + (2 * q + e) * (WK * WP));
            }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int cur = q % (D + 1), nxt = (q + D) % (D + 1);
            if (q + D < 8) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    bv[nxt][e] = *reinterpret_cast<const f32x2*>(Vc + (2 * (q + D) + e) * (QT * WP));
                    av[nxt][e] = *reinterpret_cast<const f32x2*>(Uc + (2 * (q + D) + e) * (WK * WP));
                }
            }
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int e = 0; e < 2; ++e)
                    acc[2 * q + e] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cur][e][s], bv[cur][e][s],
                                                                         acc[2 * q + e], 0, 0, 0);
            if (q + D < 8) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
    };

    // (ONE register set for the producers: in the merged launch below the 16-tile path shares its
    //  register allocation with the 64-tile path, and two sets spilled 36 VGPRs into its loop)
    f32x2 dA[16];
    f32x4 uA[8];
    if (!consumer) {
        load_set(dA, uA, 0);
        store_set(dA, uA, 0);
        if (nchunks > 1) load_set(dA, uA, 1);
    }
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        if (consumer) {
            multiply(ch);
        } else if (ch + 1 < nchunks) {
            store_set(dA, uA, (ch + 1) & 1);
            if (ch + 2 < nchunks) load_set(dA, uA, ch + 2);
        }
        __syncthreads();
    }

    // ---- epilogue (consumers): this lane's tile and 4 consecutive channels ----
    const bool stats_fwd = a.bn_partial != nullptr, stats_bwd = a.bwd_partial != nullptr;
    f32x4 cs1 = f32x4{0.f, 0.f, 0.f, 0.f}, cs2 = cs1;
    if (consumer) {
        const int t = tile0 + l15;
        const bool tv = t < total_tiles;
        const int n = tv ? t / tiles_img : 0, rem = t - n * tiles_img;
        const int ty = rem / TW, tx = rem - ty * TW;
        const int co = k0 + wave * 16 + lq * 4;
        const int yrec = (int)(unsigned)((size_t)a.N * a.H * a.W * a.K * 4);
        const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, yrec, 0x00020000);
        unsigned yoff[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int oy = 2 * ty + (q >> 1), ox = 2 * tx + (q & 1);
            const bool ok = tv && oy < a.H && ox < a.W;
            yoff[q] = ok ? (unsigned)((((n * a.H + oy) * a.W + ox) * a.K + co) * 4) : 0xFFFFFFFFu;
        }
        f32x4 ad[4];
        if (a.addend) {
            const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.addend, 0, yrec, 0x00020000);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                ad[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)yoff[q], 0, 0));
        }
        f32x4 s0[4], s1[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            s0[q] = acc[0 * 4 + q] + acc[1 * 4 + q] + acc[2 * 4 + q];
            s1[q] = acc[1 * 4 + q] - acc[2 * 4 + q] - acc[3 * 4 + q];
        }
        f32x4 y[4];
        y[0] = s0[0] + s0[1] + s0[2];
        y[1] = s0[1] - s0[2] - s0[3];
        y[2] = s1[0] + s1[1] + s1[2];
        y[3] = s1[1] - s1[2] - s1[3];
        if (a.addend) {
#pragma unroll
            for (int q = 0; q < 4; ++q) y[q] += ad[q];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4w, y[q]), rsY, (int)yoff[q], 0, 0);
        if (stats_fwd) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = yoff[q] != 0xFFFFFFFFu ? y[q] : f32x4{0.f, 0.f, 0.f, 0.f};
                cs1 += v;
                cs2 += v * v;
            }
        } else if (stats_bwd) {
            const __amdgpu_buffer_rsrc_t rsZ = __builtin_amdgcn_make_buffer_rsrc((void*)a.bwd_z, 0, yrec, 0x00020000);
            const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)a.bwd_y, 0, yrec, 0x00020000);
            const f32x4 bmean = *reinterpret_cast<const f32x4*>(a.bwd_stats + co);
            const f32x4 brstd = *reinterpret_cast<const f32x4*>(a.bwd_stats + a.K + co);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 zz = f32x4{1.f, 1.f, 1.f, 1.f};
                if (a.bwd_relu)
                    zz = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsZ, (int)yoff[q], 0, 0));
                const f32x4 yy = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, (int)yoff[q], 0, 0));
                const bool ok = yoff[q] != 0xFFFFFFFFu;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float g = (ok && zz[e] > 0.f) ? y[q][e] : 0.f;
                    cs1[e] += g;
                    cs2[e] = fmaf(g, (yy[e] - bmean[e]) * brstd[e], cs2[e]);
                }
            }
        }
    }
    if (stats_fwd || stats_bwd) {
        float* red = smem;                      // [2 which][16 tiles][64 channels] (the K loop ended on a barrier)
        if (consumer) {
            const int c = wave * 16 + lq * 4;
            *reinterpret_cast<f32x4*>(red + (0 * QT + l15) * 64 + c) = cs1;
            *reinterpret_cast<f32x4*>(red + (1 * QT + l15) * 64 + c) = cs2;
        }
        __syncthreads();
        if (tid < 128) {
            const int which = tid >> 6, col = tid & 63;
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < QT; ++q) s += red[(which * QT + q) * 64 + col];
            float* dst = stats_fwd ? a.bn_partial : a.bwd_partial;
            dst[(size_t)(which * a.K + k0 + col) * a.rows + a.row0 + blk_m] = s;
        }
    }
}

// One launch = the 64-tile blocks followed by the 16-tile tail blocks: the tail is dispatched as the
// CUs of the last full round drain, with no kernel boundary (and, in the overlapped backward pass, no
// extra event dependency) in between.
template <bool PRE>
__global__ __launch_bounds__(WTHREADS) void conv_wino_kernel(const WinoArgs a, const WinoArgs tail,
                                                             const int nfull) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if ((int)blockIdx.x < nfull) wino_full_body<PRE>(a, smem, (int)blockIdx.x);
    else wino_q_body(tail, smem, (int)blockIdx.x - nfull);
}

// ---- weight gradient in the Winograd domain ---------------------------------------------------------
// Block = 512 threads: 64 output x 64 input channels x all 16 positions, reduction over a run of
// tiles in chunks of 8.  Wave w owns 16 output channels (w & 3) x 32 input channels (w >> 2) for
// all 16 positions (128 accumulator registers, rows = output channels): the inverse transform
// G^T dU G is register-local, the result goes to an OHWI slab with 64-byte runs.  Producers: waves
// 0-3 gather the 4x4 input patches of the chunk's 8 tiles (thread = tile x channel pair: 256
// contiguous bytes per patch pixel) and build V = B^T d B, waves 4-7 the 2x2 output-gradient
// pixels and P = A dY A^T; both land in LDS as [position][tile][64 channels + 8 pad] -- channel
// fastest, so the stores are 256-byte runs, and the operand fragment of a lane (one channel, two
// consecutive tiles) is ONE ds_read2_b32 on conflict-free banks (row pitch 72).  Same two-phase
// loop as the forward kernel.  Unlike the forward pass the split of the reduction is free, so one
// round fills the chip (no 2.15-rounds-pay-3).
constexpr int GT = 8;                        // tiles per chunk
constexpr int GP = 72;                       // LDS row pitch (floats)
constexpr int GOP = 16 * GT * GP;            // floats per operand image (36,864 B)
constexpr int GSTAGE = 2 * GOP;
constexpr size_t kWinoGLds = (size_t)2 * GSTAGE * sizeof(float);       // 147,456 B
__global__ __launch_bounds__(WTHREADS) void wino_wgrad_kernel(const WinoWgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int kg = wave & 3, cg = wave >> 2;
    const int TH = (a.H + 1) >> 1, TW = (a.W + 1) >> 1;
    const int tiles_img = TH * TW, total_tiles = a.N * tiles_img;
    const int nct = a.C / 64, nkt = a.K / 64;
    const int ct = blockIdx.x % nct, kt = (blockIdx.x / nct) % nkt, sp = blockIdx.x / (nct * nkt);
    const int c0 = ct * 64, k0 = kt * 64;
    const int t0 = sp * a.tiles_per_split;
    const int t1 = min(total_tiles, t0 + a.tiles_per_split);
    const int nchunks = t1 > t0 ? (t1 - t0 + GT - 1) / GT : 0;

    // ---- producer roles: thread = (tile j of the chunk, channel pair p) ----
    const bool vprod = tid < 256;                         // waves 0-3: V;  waves 4-7: P
    const int pj = (tid >> 5) & 7, pp = tid & 31;
    // this thread's tile of chunk 0, then + 8 tiles per chunk (incremental: no division in the loop)
    int tn, tty, ttx, tcur = t0 + pj;
    {
        const int t = min(tcur, total_tiles - 1);
        tn = t / tiles_img;
        const int rem = t - tn * tiles_img;
        tty = rem / TW;
        ttx = rem - tty * TW;
    }
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.x, 0, (int)(unsigned)((size_t)a.N * a.H * a.W * a.C * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.dy, 0, (int)(unsigned)((size_t)a.N * a.H * a.W * a.K * 4), 0x00020000);
    f32x2 d[16];                    // V producers: the 4x4 patch;  P producers: d[0..3] = the 2x2 pixels
    auto load_chunk = [&]() {       // the thread's current tile (tn, tty, ttx), valid if tcur < t1
        const bool tv = tcur < t1;
        if (vprod) {
            const int h0 = 2 * tty - 1, w0 = 2 * ttx - 1;
            const int base = (((tn * a.H + h0) * a.W + w0) * a.C + c0 + 2 * pp) * 4;
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                const int h = h0 + (p >> 2), w = w0 + (p & 3);
                const bool ok = tv && (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W;
                const unsigned off = ok ? (unsigned)(base + ((p >> 2) * a.W + (p & 3)) * a.C * 4) : 0xFFFFFFFFu;
                d[p] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsX, (int)off, 0, 0));
            }
        } else {
            const int h0 = 2 * tty, w0 = 2 * ttx;
            const int base = (((tn * a.H + h0) * a.W + w0) * a.K + k0 + 2 * pp) * 4;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int h = h0 + (p >> 1), w = w0 + (p & 1);
                const bool ok = tv && h < a.H && w < a.W;
                const unsigned off = ok ? (unsigned)(base + ((p >> 1) * a.W + (p & 1)) * a.K * 4) : 0xFFFFFFFFu;
                d[p] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsD, (int)off, 0, 0));
            }
        }
        // next chunk's tile
        tcur += GT;
        ttx += GT;
        while (ttx >= TW) { ttx -= TW; ++tty; }
        while (tty >= TH) { tty -= TH; ++tn; }
    };
    auto store_chunk = [&](const int stage) {
        float* img = smem + stage * GSTAGE + (vprod ? 0 : GOP) + pj * GP + pp * 2;
        if (vprod) {
            // V = B^T d B;  B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]]
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x2 r0 = d[0 * 4 + j] - d[2 * 4 + j];
                const f32x2 r1 = d[1 * 4 + j] + d[2 * 4 + j];
                const f32x2 r2 = d[2 * 4 + j] - d[1 * 4 + j];
                const f32x2 r3 = d[1 * 4 + j] - d[3 * 4 + j];
                d[0 * 4 + j] = r0; d[1 * 4 + j] = r1; d[2 * 4 + j] = r2; d[3 * 4 + j] = r3;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x2 v0 = d[i * 4 + 0] - d[i * 4 + 2];
                const f32x2 v1 = d[i * 4 + 1] + d[i * 4 + 2];
                const f32x2 v2 = d[i * 4 + 2] - d[i * 4 + 1];
                const f32x2 v3 = d[i * 4 + 1] - d[i * 4 + 3];
                *reinterpret_cast<f32x2*>(img + (i * 4 + 0) * (GT * GP)) = v0;
                *reinterpret_cast<f32x2*>(img + (i * 4 + 1) * (GT * GP)) = v1;
                *reinterpret_cast<f32x2*>(img + (i * 4 + 2) * (GT * GP)) = v2;
                *reinterpret_cast<f32x2*>(img + (i * 4 + 3) * (GT * GP)) = v3;
            }
        } else {
            // P = A dY A^T;  A = [[1,0],[1,1],[1,-1],[0,-1]];  d[0..3] = dY[0][0], [0][1], [1][0], [1][1]
            f32x2 r[4][2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                r[0][b] = d[b];
                r[1][b] = d[b] + d[2 + b];
                r[2][b] = d[b] - d[2 + b];
                r[3][b] = -d[2 + b];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *reinterpret_cast<f32x2*>(img + (i * 4 + 0) * (GT * GP)) = r[i][0];
                *reinterpret_cast<f32x2*>(img + (i * 4 + 1) * (GT * GP)) = r[i][0] + r[i][1];
                *reinterpret_cast<f32x2*>(img + (i * 4 + 2) * (GT * GP)) = r[i][0] - r[i][1];
                *reinterpret_cast<f32x2*>(img + (i * 4 + 3) * (GT * GP)) = -r[i][1];
            }
        }
    };

    f32x4 acc[16][2];
#pragma unroll
    for (int x = 0; x < 16; ++x)
#pragma unroll
        for (int m = 0; m < 2; ++m) acc[x][m] = f32x4{0.f, 0.f, 0.f, 0.f};

    // operand fragments: MFMA k index lq = tiles 2 lq, 2 lq + 1 of the chunk (the two k-steps)
    const int frag_p = GOP + (2 * lq) * GP + kg * 16 + l15;        // rows: output channels
    const int frag_v = (2 * lq) * GP + cg * 32 + l15;              // columns: input channels
    auto multiply = [&](const int ch) {
        const float* Pc = smem + (ch & 1) * GSTAGE + frag_p;
        const float* Vc = smem + (ch & 1) * GSTAGE + frag_v;
        constexpr int D = 2;
        f32x2 pa[D + 1], v0[D + 1], v1[D + 1];
        auto rd = [&](const int x, const int slot) {
            pa[slot] = f32x2{Pc[x * (GT * GP)], Pc[x * (GT * GP) + GP]};
            v0[slot] = f32x2{Vc[x * (GT * GP)], Vc[x * (GT * GP) + GP]};
            v1[slot] = f32x2{Vc[x * (GT * GP) + 16], Vc[x * (GT * GP) + GP + 16]};
        };
#pragma unroll
        for (int x = 0; x < D; ++x) rd(x, x);
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) {
            const int cur = xi % (D + 1), nxt = (xi + D) % (D + 1);
            if (xi + D < 16) rd(xi + D, nxt);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[cur][s], v0[cur][s], acc[xi][0], 0, 0, 0);
                acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[cur][s], v1[cur][s], acc[xi][1], 0, 0, 0);
            }
            if (xi + D < 16) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
    };

    if (nchunks > 0) {
        load_chunk();
        store_chunk(0);
        if (nchunks > 1) load_chunk();
    }
    __syncthreads();
    const bool late = wave >= 4;
    for (int ch = 0; ch < nchunks; ++ch) {
        if constexpr (CILRS_WINO_ONE_PHASE != 0) {                   // (see conv_wino_kernel)
            multiply(ch);
            if (ch + 1 < nchunks) { store_chunk((ch + 1) & 1); if (ch + 2 < nchunks) load_chunk(); }
            __syncthreads();
        } else {
        if (late) {
            if (ch + 1 < nchunks) { store_chunk((ch + 1) & 1); if (ch + 2 < nchunks) load_chunk(); }
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            multiply(ch);
        } else {
            multiply(ch);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            if (ch + 1 < nchunks) { store_chunk((ch + 1) & 1); if (ch + 2 < nchunks) load_chunk(); }
        }
        __syncthreads();
        }
    }

    // ---- epilogue: dw = G^T dU G per (k, c), register-local;  G^T = [[1,.5,.5,0],[0,.5,-.5,0],[0,.5,.5,1]] ----
    float* slab = a.slabs + (size_t)sp * a.K * 9 * a.C;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        f32x4 t[3][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 hs = 0.5f * (acc[1 * 4 + q][m] + acc[2 * 4 + q][m]);
            t[0][q] = acc[0 * 4 + q][m] + hs;
            t[1][q] = 0.5f * (acc[1 * 4 + q][m] - acc[2 * 4 + q][m]);
            t[2][q] = hs + acc[3 * 4 + q][m];
        }
        const int c = c0 + cg * 32 + m * 16 + l15;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const f32x4 hs = 0.5f * (t[p][1] + t[p][2]);
            const f32x4 g0 = t[p][0] + hs;
            const f32x4 g1 = 0.5f * (t[p][1] - t[p][2]);
            const f32x4 g2 = hs + t[p][3];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = k0 + kg * 16 + lq * 4 + r;
                float* dst = slab + ((size_t)k * 9 + p * 3) * a.C + c;
                dst[0] = g0[r];
                dst[a.C] = g1[r];
                dst[2 * a.C] = g2[r];
            }
        }
    }
}

}  // namespace

size_t wino_weight_floats(int K, int C) { return (size_t)16 * K * C; }

int launch_wino_weights(const float* w, float* U, int K, int C, int dgrad, hipStream_t s) {
    CILRS_CHECK(K % 8 == 0 && C % 8 == 0, "wino_weights: channels must be multiples of 8");
    wino_weights_kernel<<<cdiv(K * C, 256), 256, 0, s>>>(w, U, K, C, dgrad);
    CILRS_LAUNCH_CHECK();
    return 0;
}

int launch_wino_weights_all(const WinoWeightTable& t, const float* params, float* ubase, hipStream_t s) {
    if (t.n <= 0) return 0;
    for (int e = 0; e < t.n; ++e)
        CILRS_CHECK(t.K[e] % 8 == 0 && t.C[e] % 32 == 0, "wino_weights_all: K %% 8, C %% 32");
    wino_weights_all_kernel<<<t.blk_begin[t.n], 256, 0, s>>>(t, params, ubase);
    CILRS_LAUNCH_CHECK();
    return 0;
}

bool wino_supported(int C, int K, int ksize, int stride, int pad) {
    return ksize == 3 && stride == 1 && pad == 1 && C % WC == 0 && K % WK == 0;
}

int wino_groups(int N, int H, int W) { return cdiv(N * ((H + 1) / 2) * ((W + 1) / 2), WT); }

constexpr size_t kWinoLds = (size_t)2 * (16 * WT * WP + 16 * WK * WP) * sizeof(float);     // 128 KB

// (a plan calls this when it is built: the attribute must not be set for the first time inside a
//  stream capture)
int wino_prepare() {
    if (once_per_device(reinterpret_cast<const void*>(&conv_wino_kernel<false>))) {
        CILRS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino_kernel<false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWinoLds));
        CILRS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino_kernel<true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWinoLds));
    }
    return 0;
}

// How a launch is cut: `full` 64-tile groups on the main kernel, the remaining tiles as 16-tile
// groups on the tail kernel.  The tail takes the blocks of the last round when that round is at
// most a quarter full (so that its four-times-as-many blocks still fit one round); CILRS_WINO_TAIL=0
// never, =2 everything on the 16-tile kernel (experiments).
struct WinoSplit { int full, tail; };
static WinoSplit wino_split(int N, int H, int W, int K, int no_tail) {
    static const int mode = getenv("CILRS_WINO_TAIL") ? atoi(getenv("CILRS_WINO_TAIL")) : 1;
    const int cus = device_cus();
    const int tiles = N * ((H + 1) / 2) * ((W + 1) / 2);
    const int groups = cdiv(tiles, WT), nkt = K / WK;
    WinoSplit sp{groups, 0};
    if (mode == 2) sp.full = 0;
    else if (mode == 1 && !no_tail) {
        const int blocks = groups * nkt, last = blocks % cus;
        if (blocks > cus && last > 0 && 4 * last <= cus) sp.full = (blocks - last) / nkt;
    }
    sp.tail = cdiv(tiles - sp.full * WT > 0 ? tiles - sp.full * WT : 0, QT);
    return sp;
}
// Channel split of an under-filled launch.  Cost model in shader cycles per block (in-kernel
// stamps, profiles/r03_wino_dbg.log): ~6.0k per 8-channel chunk, ~10.5k prologue + epilogue; the
// reduce launch + its boundary ~25k.  layer3 at B=128 (168 blocks, 32 chunks): 1 part 202k,
// 2 parts (336 blocks = two rounds) 213k + 25k, 3 parts (504 blocks, two rounds) 149k + 25k.
static int wino_csplit(int N, int H, int W, int C, int K, size_t slab_floats) {
    const int cus = device_cus();
    const int tiles = N * ((H + 1) / 2) * ((W + 1) / 2);
    const int blocks = cdiv(tiles, WT) * (K / WK), nchunks = C / WC;
    if (blocks >= cus || (K >> 2) > 256 || 256 % (K >> 2) != 0) return 1;
    const size_t y_floats = (size_t)N * H * W * K;
    int best = 1;
    double best_t = 1e30;
    for (int sp = 1; sp <= 4; ++sp) {
        if (sp > 1 && ((size_t)sp * y_floats > slab_floats || nchunks / sp < 4)) break;
        const double t = (double)cdiv(sp * blocks, cus) * (cdiv(nchunks, sp) * 6.0e3 + 10.5e3) +
                         (sp > 1 ? 25.0e3 : 0.0);
        if (t < 0.95 * best_t) { best_t = t; best = sp; }
    }
    return best;
}
int wino_rows(int N, int H, int W, int K, int no_tail, int C, size_t slab_floats) {
    if (C > 0 && wino_csplit(N, H, W, C, K, slab_floats) > 1) return slab_reduce_rows(N * H * W, K, nullptr);
    const WinoSplit sp = wino_split(N, H, W, K, no_tail);
    return sp.full + sp.tail;
}

static int g_last_csplit = 0;
int wino_last_csplit() { return g_last_csplit; }     // parts per tile of the most recent launch (tests)

int launch_conv_wino(const WinoArgs& a_in, hipStream_t s) {
    WinoArgs a = a_in;
    CILRS_CHECK(a.x && a.U && a.y, "conv_wino: NULL tensor");
    CILRS_CHECK(a.C % WC == 0 && a.K % WK == 0, "conv_wino: C %% 8, K %% 64");
    CILRS_CHECK((size_t)a.N * a.H * a.W * a.C * 4 < (1ull << 32) &&
                    (size_t)a.N * a.H * a.W * a.K * 4 < (1ull << 32) &&
                    (size_t)16 * a.C * a.K * 4 < (1ull << 32),
                "conv_wino: tensor too large for 32-bit offsets");
    CILRS_CHECK(!(a.bn_partial && a.bwd_partial), "conv_wino: one kind of column partials per launch");
    if (wino_prepare()) return 1;
    a.csplit = (a.slabs && a.scratch_partial) ? wino_csplit(a.N, a.H, a.W, a.C, a.K, a.slab_floats) : 1;
    g_last_csplit = a.csplit;
    if (a.csplit > 1) {
        // every tile on the 64-tile kernel, csplit blocks per tile; then the fixed-order reduce
        a.tile_begin = 0; a.row0 = 0; a.rows = 0;
        const int nfull = wino_groups(a.N, a.H, a.W) * (a.K / WK) * a.csplit;
        conv_wino_kernel<false><<<nfull, WTHREADS, kWinoLds, s>>>(a, a, nfull);
        CILRS_LAUNCH_CHECK();
        const int mode = a.bwd_partial ? 1 : 0;
        float* partial = a.bwd_partial ? a.bwd_partial : a.bn_partial ? a.bn_partial : a.scratch_partial;
        return launch_slab_reduce_cols(mode, a.slabs, a.csplit, a.y, a.addend, a.N * a.H * a.W, a.K,
                                       a.bwd_z, a.bwd_y, a.bwd_stats, a.bwd_relu, partial, s);
    }
    const WinoSplit sp = wino_split(a.N, a.H, a.W, a.K, a.no_tail);
    a.rows = sp.full + sp.tail;
    WinoArgs t = a;
    a.tile_begin = 0; a.row0 = 0;
    t.tile_begin = sp.full * WT; t.row0 = sp.full; t.stamps = nullptr;
    const int nfull = sp.full * (a.K / WK), ntail = sp.tail * (a.K / WK);
    // (operands of the epilogue requested ahead of the K loop's end: wino_full_body<true>)
    if (a.addend || a.bwd_partial)
        conv_wino_kernel<true><<<nfull + ntail, WTHREADS, nfull > 0 ? kWinoLds : kWinoQLds, s>>>(a, t, nfull);
    else
        conv_wino_kernel<false><<<nfull + ntail, WTHREADS, nfull > 0 ? kWinoLds : kWinoQLds, s>>>(a, t, nfull);
    CILRS_LAUNCH_CHECK();
    return 0;
}

bool wino_wgrad_supported(int C, int K, int ksize, int stride, int pad) {
    return ksize == 3 && stride == 1 && pad == 1 && C % 64 == 0 && K % 64 == 0;
}
static void wino_wgrad_split(int N, int H, int W, int C, int K, int* splits, int* tps) {
    const int cus = device_cus();
    const int tiles = N * ((H + 1) / 2) * ((W + 1) / 2);
    const int kc = (C / 64) * (K / 64);
    int s = cus / kc;
    if (s < 1) s = 1;
    int per = cdiv(cdiv(tiles, s), GT) * GT;          // whole chunks
    if (per < GT) per = GT;
    *tps = per;
    *splits = cdiv(tiles, per);
}
size_t wino_wgrad_scratch_floats(int N, int H, int W, int C, int K) {
    int splits, tps;
    wino_wgrad_split(N, H, W, C, K, &splits, &tps);
    return (size_t)splits * K * 9 * C;
}
int launch_conv_wino_wgrad(const WinoWgradArgs& a_in, hipStream_t s) {
    WinoWgradArgs a = a_in;
    CILRS_CHECK(a.x && a.dy && a.dw && a.slabs, "conv_wino_wgrad: NULL tensor");
    CILRS_CHECK(a.C % 64 == 0 && a.K % 64 == 0, "conv_wino_wgrad: C %% 64, K %% 64");
    CILRS_CHECK((size_t)a.N * a.H * a.W * a.C * 4 < (1ull << 31) &&
                    (size_t)a.N * a.H * a.W * a.K * 4 < (1ull << 31),
                "conv_wino_wgrad: tensor too large for 32-bit offsets");
    if (once_per_device(reinterpret_cast<const void*>(&wino_wgrad_kernel))) {
        CILRS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_wgrad_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWinoGLds));
    }
    wino_wgrad_split(a.N, a.H, a.W, a.C, a.K, &a.splits, &a.tiles_per_split);
    wino_wgrad_kernel<<<a.splits * (a.C / 64) * (a.K / 64), WTHREADS, kWinoGLds, s>>>(a);
    CILRS_LAUNCH_CHECK();
    return launch_wgrad_reduce(a.slabs, a.dw, a.splits, (size_t)a.K * 9 * a.C, a.accumulate, s);
}

}  // namespace cilrs
