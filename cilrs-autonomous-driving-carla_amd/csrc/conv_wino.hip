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
// output channels x all 16 xi; wave w owns xi = 2w, 2w+1 for the whole tile (128 accumulator
// registers: 2 xi x 2x2 sub-tiles of 32x32); reduction in chunks of 8 input channels:
//   * gather + transform: four waves (one per SIMD), thread = (tile, channel pair): 16 8-byte
//     buffer loads (a patch pixel outside the image gets the offset ~0 = out of range = zeros),
//     32 packed additions, 16 8-byte LDS stores;
//   * U chunk: 32 KB, contiguous 2 KB per xi;  both are fetched one chunk ahead into registers;
//   * multiply: per xi two A and two B fragments (ds_read_b128), 16 v_mfma_f32_32x32x2_f32; two
//     LDS stages, one barrier per chunk;
//   * epilogue, per 32x32 sub-tile: accumulators -> LDS [xi][tile][channel], inverse transform by
//     (tile, channel) threads, then the same fused work as the implicit-GEMM kernel: residual /
//     addend, BatchNorm batch statistics (forward) or BatchNorm-backward reductions (data
//     gradient) as per-block column partials, 128-byte coalesced stores.
// fp32 Winograd is not bit-identical to the direct sum (other rounding points): outputs differ
// by a few 1e-6 relative, inside the 1e-4 contract (tests/test_ops_gpu.py).
#include "common.h"

namespace cilrs {
namespace {

constexpr int WT = 64;             // output tiles per block
constexpr int WK = 64;             // output channels per block
constexpr int WC = 8;              // input channels per reduction chunk
constexpr int WP = 8;              // LDS row pitch (floats) of the V / U images: no pad, so that TWO
                                   // stages fit (2 x 64 KB); a fragment read is 2-way conflicted,
                                   // which 8 reads per 32 MFMAs do not notice
constexpr int WTHREADS = 512;
// timing experiments (tools/wino_dbg.sh; results meaningless): 1 no global loads, 2 no LDS stores
// of a chunk, 4 no MFMAs, 8 no epilogue
#ifndef CILRS_WINO_DBG
#define CILRS_WINO_DBG 0
#endif

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

__global__ __launch_bounds__(WTHREADS) void conv_wino_kernel(const WinoArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int STAGE = 16 * WT * WP + 16 * WK * WP;      // floats per stage (64 KB)
    float* Vs = smem;                          // [2 stages]: [16][WT][WP] | [16][WK][WP]
    float* Us = smem + 16 * WT * WP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int TH = (a.H + 1) >> 1, TW = (a.W + 1) >> 1;
    const int tiles_img = TH * TW, total_tiles = a.N * tiles_img;
    const int nkt = a.K / WK;
    const int blk_n = blockIdx.x % nkt, blk_m = blockIdx.x / nkt;
    const int k0 = blk_n * WK;
    const int nchunks = a.C / WC;

    // ---- gather role (waves 0-3, one per SIMD): this thread's (tile, channel PAIR) and the 16
    //      patch pixels; 8-byte loads, packed additions, 8-byte LDS stores ----
    const bool gatherer = tid < 256;
    const int g_tile = (tid >> 2) & 63, g_p = tid & 3;
    unsigned voff[16];
    {
        const int t = blk_m * WT + g_tile;
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
    const int u_idx = tid & 127, u_xi0 = tid >> 7;
    const unsigned u_lds = (unsigned)((u_idx >> 1) * WP + (u_idx & 1) * 4);

    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 d[16];
    f32x4 ur[4];
    auto load_chunk = [&](const int ch) {
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
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int xi = q * 4 + u_xi0;
            const unsigned off = (unsigned)((((size_t)xi * nchunks + ch) * a.K + k0) * 8 + u_idx * 4) * 4u;
            ur[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsU, (int)off, 0, 0));
        }
    };
    auto store_chunk = [&](const int stage) {
        if (CILRS_WINO_DBG & 2) {
            asm volatile("" ::"v"(d[0]), "v"(d[15]), "v"(ur[0]), "v"(ur[3]));
            return;
        }
        // V = B^T d B;  B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]]
        if (gatherer) {
            f32x2 r[16];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                r[0 * 4 + j] = d[0 * 4 + j] - d[2 * 4 + j];
                r[1 * 4 + j] = d[1 * 4 + j] + d[2 * 4 + j];
                r[2 * 4 + j] = d[2 * 4 + j] - d[1 * 4 + j];
                r[3 * 4 + j] = d[1 * 4 + j] - d[3 * 4 + j];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x2 v0 = r[i * 4 + 0] - r[i * 4 + 2];
                const f32x2 v1 = r[i * 4 + 1] + r[i * 4 + 2];
                const f32x2 v2 = r[i * 4 + 2] - r[i * 4 + 1];
                const f32x2 v3 = r[i * 4 + 1] - r[i * 4 + 3];
                float* dst = Vs + stage * STAGE + (i * 4) * (WT * WP) + g_tile * WP + g_p * 2;
                *reinterpret_cast<f32x2*>(dst + 0 * WT * WP) = v0;
                *reinterpret_cast<f32x2*>(dst + 1 * WT * WP) = v1;
                *reinterpret_cast<f32x2*>(dst + 2 * WT * WP) = v2;
                *reinterpret_cast<f32x2*>(dst + 3 * WT * WP) = v3;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<f32x4*>(Us + stage * STAGE + (q * 4 + u_xi0) * (WK * WP) + u_lds) = ur[q];
    };

    f32x16 acc[2][2][2];             // [xi of this wave][row sub-tile][channel sub-tile]
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[x][i][j][e] = 0.f;

    // Two LDS stages: while a wave multiplies chunk ch out of stage ch & 1, the chunk after it is
    // already on its way from global memory (registers) and is transformed and stored into the
    // other stage right behind the multiplies -- ONE barrier per chunk, and the waves of a SIMD
    // drift apart so that one's LDS stores run under the other's MFMAs.
    const long long tm_start = (a.stamps != nullptr && blockIdx.x == 0) ? __builtin_amdgcn_s_memtime() : 0;
    load_chunk(0);
    store_chunk(0);
    if (nchunks > 1) load_chunk(1);
    __syncthreads();
    // The two waves of a SIMD run the same program; left alone they multiply together (half the
    // matrix pipe each) and then transform / store together (pipe idle).  Waves 4-7 therefore do
    // the two halves of an iteration in the OTHER order: one wave's MFMAs run under its partner's
    // transform + LDS stores.  (Stage ch & 1 is complete at the top of iteration ch; the stores
    // of chunk ch + 1 go to the other stage, which nobody has read since iteration ch - 1.)
    const bool late = wave >= 4;
    auto multiply = [&](const int ch) {
        const float* Vc = Vs + (ch & 1) * STAGE;
        const float* Uc = Us + (ch & 1) * STAGE;
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const int xi = 2 * wave + x;
            f32x4 af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
                af[i] = *reinterpret_cast<const f32x4*>(Vc + xi * (WT * WP) + (i * 32 + l31) * WP + lh * 4);
#pragma unroll
            for (int j = 0; j < 2; ++j)
                bf[j] = *reinterpret_cast<const f32x4*>(Uc + xi * (WK * WP) + (j * 32 + l31) * WP + lh * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        if (!(CILRS_WINO_DBG & 4))
                        acc[x][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e],
                                                                            acc[x][i][j], 0, 0, 0);
                        else acc[x][i][j][e] += af[i][e] * bf[j][e];
        }
    };
    auto refill = [&](const int ch) {                   // chunk ch + 1 into the other stage
        if (ch + 1 < nchunks) {
            store_chunk((ch + 1) & 1);
            if (ch + 2 < nchunks) load_chunk(ch + 2);
        }
    };
    // diagnostics (WinoArgs::stamps, tools/wino_bench.py): where block 0's waves 0 and 4 spend a
    // launch, in shader cycles: [prologue, multiply, refill, barrier wait, epilogue]
    long long tm_mul = 0, tm_ref = 0, tm_bar = 0;
    const bool stamp = a.stamps != nullptr && blockIdx.x == 0 && (tid == 0 || tid == 256);
    const long long tm0 = tm_start;
    long long tm_loop0 = 0;
    if (stamp) tm_loop0 = __builtin_amdgcn_s_memtime();
    for (int ch = 0; ch < nchunks; ++ch) {
        long long ta = 0, tb = 0, tc = 0, td = 0;
        if (stamp) ta = __builtin_amdgcn_s_memtime();
        if (late) {
            refill(ch);
            __builtin_amdgcn_sched_barrier(0);
            if (stamp) tb = __builtin_amdgcn_s_memtime();
            multiply(ch);
        } else {
            multiply(ch);
            __builtin_amdgcn_sched_barrier(0);
            if (stamp) tb = __builtin_amdgcn_s_memtime();
            refill(ch);
        }
        if (stamp) tc = __builtin_amdgcn_s_memtime();
        __syncthreads();
        if (stamp) {
            td = __builtin_amdgcn_s_memtime();
            if (late) { tm_ref += tb - ta; tm_mul += tc - tb; } else { tm_mul += tb - ta; tm_ref += tc - tb; }
            tm_bar += td - tc;
        }
    }
    const long long tm_loop1 = stamp ? __builtin_amdgcn_s_memtime() : 0;

    if (CILRS_WINO_DBG & 8) {
        if (acc[0][0][0][0] == 123.f) a.y[tid] = acc[1][1][1][3];
        return;
    }
    float* Ms = smem;                           // [16][32][32] = 64 KB (the K loop ended on a barrier)
    float cs1[2] = {0.f, 0.f}, cs2[2] = {0.f, 0.f};     // column partials of this thread's channel
    const int e_col = tid & 31;                 // channel within the sub-tile
    const int e_t0 = tid >> 5;                  // tiles e_t0 and e_t0 + 16 of the sub-tile
    const bool stats_fwd = a.bn_partial != nullptr, stats_bwd = a.bwd_partial != nullptr;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                float* mp = Ms + (2 * wave + x) * 1024 + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                    mp[row * 32] = acc[x][i][j][r];
                }
            }
            __syncthreads();
            const int co = k0 + j * 32 + e_col;
            float bmean = 0.f, brstd = 0.f;
            if (stats_bwd) { bmean = a.bwd_stats[co]; brstd = a.bwd_stats[a.K + co]; }
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int tl = e_t0 + 16 * half;              // tile within the sub-tile
                float m[16];
#pragma unroll
                for (int xi = 0; xi < 16; ++xi) m[xi] = Ms[xi * 1024 + tl * 32 + e_col];
                // Y = A^T M A;  A^T = [[1,1,1,0],[0,1,-1,-1]]
                float s0[4], s1[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    s0[q] = m[0 * 4 + q] + m[1 * 4 + q] + m[2 * 4 + q];
                    s1[q] = m[1 * 4 + q] - m[2 * 4 + q] - m[3 * 4 + q];
                }
                float y[4];
                y[0] = s0[0] + s0[1] + s0[2];
                y[1] = s0[1] - s0[2] - s0[3];
                y[2] = s1[0] + s1[1] + s1[2];
                y[3] = s1[1] - s1[2] - s1[3];
                const int t = blk_m * WT + i * 32 + tl;
                if (t < total_tiles) {
                    const int n = t / tiles_img, rem = t - n * tiles_img;
                    const int ty = rem / TW, tx = rem - ty * TW;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int oy = 2 * ty + (q >> 1), ox = 2 * tx + (q & 1);
                        if (oy < a.H && ox < a.W) {
                            const size_t o = ((size_t)(n * a.H + oy) * a.W + ox) * a.K + co;
                            float v = y[q];
                            if (a.addend) v += a.addend[o];
                            a.y[o] = v;
                            if (stats_fwd) {
                                cs1[j] += v;
                                cs2[j] = fmaf(v, v, cs2[j]);
                            } else if (stats_bwd) {
                                const float zz = a.bwd_relu ? a.bwd_z[o] : 1.f;
                                const float g = zz > 0.f ? v : 0.f;
                                cs1[j] += g;
                                cs2[j] = fmaf(g, (a.bwd_y[o] - bmean) * brstd, cs2[j]);
                            }
                        }
                    }
                }
            }
            __syncthreads();
        }
    }
    if (stamp) {
        long long* o = a.stamps + (tid == 0 ? 0 : 8);
        o[0] = tm_loop0 - tm0; o[1] = tm_mul; o[2] = tm_ref; o[3] = tm_bar;
        o[4] = __builtin_amdgcn_s_memtime() - tm_loop1; o[5] = tm_loop1 - tm_loop0;
    }
    // column partials of this block: [2][K][groups] like the implicit-GEMM kernel's (channel-major),
    // threads with equal e_col (16 of them) summed in fixed order
    if (stats_fwd || stats_bwd) {
        float* red = smem;                      // [2][2 j][16 t0][32 col]
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            red[((0 * 2 + j) * 16 + e_t0) * 32 + e_col] = cs1[j];
            red[((1 * 2 + j) * 16 + e_t0) * 32 + e_col] = cs2[j];
        }
        __syncthreads();
        if (tid < 128) {                        // (which, j, col)
            const int which = tid >> 6, j = (tid >> 5) & 1, col = tid & 31;
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) s += red[((which * 2 + j) * 16 + q) * 32 + col];
            float* dst = stats_fwd ? a.bn_partial : a.bwd_partial;
            const int groups = gridDim.x / nkt;
            dst[(size_t)(which * a.K + k0 + j * 32 + col) * groups + blk_m] = s;
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

bool wino_supported(int C, int K, int ksize, int stride, int pad) {
    return ksize == 3 && stride == 1 && pad == 1 && C % WC == 0 && K % WK == 0;
}

int wino_groups(int N, int H, int W) { return cdiv(N * ((H + 1) / 2) * ((W + 1) / 2), WT); }

int launch_conv_wino(const WinoArgs& a, hipStream_t s) {
    CILRS_CHECK(a.x && a.U && a.y, "conv_wino: NULL tensor");
    CILRS_CHECK(a.C % WC == 0 && a.K % WK == 0, "conv_wino: C %% 8, K %% 64");
    CILRS_CHECK((size_t)a.N * a.H * a.W * a.C * 4 < (1ull << 32) &&
                    (size_t)16 * a.C * a.K * 4 < (1ull << 32),
                "conv_wino: tensor too large for 32-bit offsets");
    CILRS_CHECK(!(a.bn_partial && a.bwd_partial), "conv_wino: one kind of column partials per launch");
    constexpr size_t lds = (size_t)2 * (16 * WT * WP + 16 * WK * WP) * sizeof(float);     // 128 KB
    static bool attr_set = false;
    if (!attr_set) {
        CILRS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    const int groups = wino_groups(a.N, a.H, a.W);
    conv_wino_kernel<<<groups * (a.K / WK), WTHREADS, lds, s>>>(a);
    CILRS_LAUNCH_CHECK();
    return 0;
}

}  // namespace cilrs
