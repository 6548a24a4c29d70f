// Convolution for single-frame inference (reference control loop, model/autonomous_drive.py:908-920:
// one 200x88 frame per tick, model.eval()).
//
// At B=1 a trunk convolution has 21..1,100 output pixels but 37 K..2.4 M weights: the work is
// streaming the weights once, and the cost is latency.  The big implicit-GEMM kernel needs a
// split-K launch plus a reduce launch per layer for that (~13 us); here ONE launch does it:
//   * one block per 16x16 (pixels x channels) output tile, 16 waves per block (64..276 blocks per
//     layer: enough CUs for the MFMA work of the zero-padded tiles to stay at a few microseconds);
//   * the 16 waves split the reduction index (filter taps x input channels), each streams its
//     slice of the activations and weights global -> registers -> v_mfma_f32_16x16x4_f32 with
//     eight k-groups (16 buffer loads) in flight, no LDS staging;
//   * the 16 partial tiles are summed through LDS in wave order (deterministic), then the folded
//     BatchNorm scale/shift, ReLU and residual add run and the tile is stored.
#include "common.h"

namespace cilrs {
namespace {

constexpr int SW = 16;            // waves per block
constexpr int SU = 10;            // k-groups (16 reduction indices each) in flight per wave
constexpr int ST = 16;            // output tile: 16 pixels x 16 channels (v_mfma_f32_16x16x4_f32)

__global__ __launch_bounds__(64 * SW) void conv_small_kernel(const ConvSmallArgs a) {
    __shared__ float red[SW][ST][ST + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, kq = lane >> 4;      // operand row / k-quad of this lane
    const int m0 = blockIdx.y * ST, n0 = blockIdx.x * ST;
    const int M = a.N * a.Ho * a.Wo, HoWo = a.Ho * a.Wo;
    const int ntaps = a.K * a.K, cgn = a.Cin >> 4;
    const int S = ntaps * cgn;                    // k-groups (16 input channels of one tap)
    const int per = (S + SW - 1) / SW;
    const int sb = wave * per, se = min(S, sb + per);

    // this lane's activation row (output pixel) and weight row (output channel)
    const int m = m0 + r;
    unsigned rowOff = 0u, rowMask = 0u;
    if (m < M) {
        const int n = m / HoWo, rem = m - n * HoWo;
        const int oh = rem / a.Wo, ow = rem - oh * a.Wo;
        const int hb = oh * a.stride - a.pad, wb = ow * a.stride - a.pad;
        rowOff = (unsigned)((((long)(n * a.H + hb) * a.W + wb) * a.Cin + kq * 4) * 4);
        for (int t = 0; t < ntaps; ++t) {
            const int h = hb + t / a.K, w = wb + t % a.K;
            if (h >= 0 && w >= 0 && h < a.H && w < a.W) rowMask |= 1u << t;
        }
    }
    const unsigned wOff = (unsigned)((((long)(n0 + r) * ntaps) * a.Cin + kq * 4) * 4);
    int tapA_v = 0;
    if (lane < ntaps) tapA_v = ((lane / a.K) * a.W + lane % a.K) * a.Cin * 4;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.x, 0, (int)(unsigned)((size_t)a.N * a.H * a.W * a.Cin * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.w, 0, (int)(unsigned)((size_t)a.Cout * ntaps * a.Cin * 4), 0x00020000);

    // MFMA operand map (16x16x4): lane l feeds A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15];
    // the lane loads 4 consecutive reduction indices and spends them in 4 MFMAs, so one k-group
    // covers 16 indices (A and B use the same permutation of k, which a dot product ignores)
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s0 = sb; s0 < se; s0 += SU) {
        f32x4 av[SU], bv[SU];
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int s = s0 + u;                 // wave-uniform
            const int tap = s / cgn, cg = s - tap * cgn;
            const bool live = s < se;
            const int tapc = live ? tap : 0;
            const unsigned toff = (unsigned)__builtin_amdgcn_readlane(tapA_v, tapc) +
                                  (unsigned)(cg * 64);
            const unsigned offA = (live && ((rowMask >> tapc) & 1u)) ? rowOff + toff : 0xFFFFFFFFu;
            const unsigned offB = live ? wOff + (unsigned)((tap * a.Cin + cg * 16) * 4) : 0xFFFFFFFFu;
            av[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)offA, 0, 0));
            bv[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, (int)offB, 0, 0));
        }
        __builtin_amdgcn_sched_barrier(0);        // all 2*SU loads in flight before the first MFMA
#pragma unroll
        for (int u = 0; u < SU; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][e], bv[u][e], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---- sum the 16 k-slices in wave order (C/D map: col = lane & 15, row = 4*(lane >> 4) + i) ----
#pragma unroll
    for (int i = 0; i < 4; ++i) red[wave][4 * kq + i][r] = acc[i];
    __syncthreads();
    if (threadIdx.x >= ST * ST) return;
    const int col = threadIdx.x & 15, row = threadIdx.x >> 4;
    const int cm = m0 + row, co = n0 + col;
    if (cm >= M || co >= a.Cout) return;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < SW; ++w) v += red[w][row][col];
    if (a.scale) v = v * a.scale[co] + a.shift[co];
    if (a.relu) v = fmaxf(v, 0.f);
    const size_t o = (size_t)cm * a.Cout + co;
    if (a.addend) v += a.addend[o];
    if (a.relu_post) v = fmaxf(v, 0.f);
    a.y[o] = v;
}

}  // namespace

int launch_conv_small(const ConvSmallArgs& a, hipStream_t s) {
    CILRS_CHECK(a.Cin % 16 == 0 && a.Cout % ST == 0 && a.K * a.K <= 16,
                "conv_small: Cin %% 16, Cout %% 16, <= 16 taps");
    CILRS_CHECK((size_t)a.N * a.H * a.W * a.Cin * 4 < (1ull << 32) &&
                    (size_t)a.Cout * a.K * a.K * a.Cin * 4 < (1ull << 32),
                "conv_small: tensor too large for 32-bit offsets");
    const int M = a.N * a.Ho * a.Wo;
    conv_small_kernel<<<dim3(a.Cout / ST, cdiv(M, ST)), 64 * SW, 0, s>>>(a);
    CILRS_LAUNCH_CHECK();
    return 0;
}

}  // namespace cilrs
