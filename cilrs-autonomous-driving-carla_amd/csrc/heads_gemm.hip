// Grouped small GEMMs for the heads (speed encoder, four control branches, speed predictor;
// reference model/autonomous_drive.py:371-387, 391-398 and their autograd backward).
//
// At B=128 these layers are 0.1 % of the step's FLOPs but used to be ~60 dependent launches of
// 5-50 us each on five side streams (1.07 ms of a 14 ms step on the timeline).  Here every layer
// of every chain that can run at the same time is ONE launch: blockIdx.z picks the chain from a
// small table, one block owns one 32x32 output tile, its four waves split the reduction index and
// combine through LDS in a fixed order (deterministic), operands go global -> registers -> MFMA
// (v_mfma_f32_32x32x2_f32, exact f32) with no LDS staging -- the matrices are L2-resident and a
// few hundred KB.
//
//   NT  C[m][n] = sum_k A[m][k] * B[n][k]  (+ bias[n], ReLU, dropout)         nn.Linear forward
//   NN  C[m][n] = sum_k A[m][k] * B[k][n]  (masked by act[m][n] > 0, scaled)  input gradient
//   TN  C[m][n] = sum_k A[k][m] * B[k][n], dbias[m] = sum_k A[k][m]           weight / bias gradient
#include "common.h"

namespace cilrs {
namespace {

__device__ __forceinline__ unsigned int hg_mix32(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    x ^= x >> 31;
    return (unsigned int)(x >> 32);
}

template <int MODE>      // 0 NT, 1 NN, 2 TN
__global__ __launch_bounds__(256) void hgemm_kernel(const HGemmArgs a) {
    __shared__ float red[4][32][33];
    __shared__ float redb[4][32];
    const HGemmGroup& g = a.g[blockIdx.z];
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    if (m0 >= g.M || n0 >= g.N) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, kh = lane >> 5;
    const int K = g.K;
    // the four waves take consecutive slices of the reduction index (multiples of 8)
    const int kw = ((K + 31) / 32) * 8;
    const int kb = wave * kw, ke = min(K, kb + kw);
    const int m = m0 + r, n = n0 + r;
    const bool m_ok = m < g.M, n_ok = n < g.N;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float bsum = 0.f;
    const float* Ap = g.A;
    const float* Bp = g.B;
    for (int k0 = kb; k0 < ke; k0 += 8) {
        const int kk = k0 + kh * 4;
        float av[4], bv[4];
        if (MODE == 0) {
            if (g.vec) {
                f32x4 t = {0.f, 0.f, 0.f, 0.f}, u = {0.f, 0.f, 0.f, 0.f};
                if (m_ok && kk < ke) t = *reinterpret_cast<const f32x4*>(Ap + (size_t)m * g.lda + kk);
                if (n_ok && kk < ke) u = *reinterpret_cast<const f32x4*>(Bp + (size_t)n * g.ldb + kk);
#pragma unroll
                for (int e = 0; e < 4; ++e) { av[e] = t[e]; bv[e] = u[e]; }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    av[e] = (m_ok && kk + e < ke) ? Ap[(size_t)m * g.lda + kk + e] : 0.f;
                    bv[e] = (n_ok && kk + e < ke) ? Bp[(size_t)n * g.ldb + kk + e] : 0.f;
                }
            }
        } else if (MODE == 1) {
            if (g.vec) {
                f32x4 t = {0.f, 0.f, 0.f, 0.f};
                if (m_ok && kk < ke) t = *reinterpret_cast<const f32x4*>(Ap + (size_t)m * g.lda + kk);
#pragma unroll
                for (int e = 0; e < 4; ++e) av[e] = t[e];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    av[e] = (m_ok && kk + e < ke) ? Ap[(size_t)m * g.lda + kk + e] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
                bv[e] = (n_ok && kk + e < ke) ? Bp[(size_t)(kk + e) * g.ldb + n] : 0.f;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                av[e] = (m_ok && kk + e < ke) ? Ap[(size_t)(kk + e) * g.lda + m] : 0.f;
                bv[e] = (n_ok && kk + e < ke) ? Bp[(size_t)(kk + e) * g.ldb + n] : 0.f;
                bsum += av[e];
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv[e], acc, 0, 0, 0);
    }
    // ---- combine the four k-slices: C/D map col = lane & 31, row = (i&3) + 8*(i>>2) + 4*kh ----
#pragma unroll
    for (int i = 0; i < 16; ++i) red[wave][(i & 3) + 8 * (i >> 2) + 4 * kh][r] = acc[i];
    if (MODE == 2) {
        bsum += __shfl_xor(bsum, 32);
        if (kh == 0) redb[wave][r] = bsum;
    }
    __syncthreads();
    const int col = threadIdx.x & 31, rq = threadIdx.x >> 5;        // 8 row groups x 4 rows
    const int cn = n0 + col;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = rq * 4 + i, cm = m0 + row;
        float v = (red[0][row][col] + red[1][row][col]) + (red[2][row][col] + red[3][row][col]);
        if (cm >= g.M || cn >= g.N) continue;
        if (MODE == 0) {
            if (g.bias) v += g.bias[cn];
            if (a.relu) v = fmaxf(v, 0.f);
            if (a.drop_p > 0.f && g.drop_stream != kNoDrop) {   // hash / order of dropout_kernel
                const unsigned i_el = (unsigned)(cm * g.N + cn);
                const unsigned int h = hg_mix32(a.seed * 0x2545F4914F6CDD1Dull +
                                                (g.drop_stream << 40) + i_el);
                const float u = (float)(h >> 8) * (1.0f / 16777216.0f);
                v = (u >= a.drop_p) ? v / (1.0f - a.drop_p) : 0.f;
            }
        } else if (MODE == 1) {
            if (g.mask) v = g.mask[(size_t)cm * g.ldmask + cn] > 0.f ? v * g.mask_scale : 0.f;
        }
        float* o = g.C + (size_t)cm * g.ldc + cn;
        *o = (MODE == 2 && a.accumulate) ? *o + v : v;
    }
    if (MODE == 2 && g.dbias && blockIdx.x == 0 && threadIdx.x < 32 && m0 + threadIdx.x < g.M) {
        const int t = threadIdx.x;
        const float s = (redb[0][t] + redb[1][t]) + (redb[2][t] + redb[3][t]);
        float* o = g.dbias + m0 + t;
        *o = a.accumulate ? *o + s : s;
    }
}

}  // namespace

int launch_hgemm(int mode, HGemmArgs& a, hipStream_t s) {
    CILRS_CHECK(a.ngroups >= 1 && a.ngroups <= kMaxCmd + 1, "hgemm: 1..%d groups", kMaxCmd + 1);
    int maxM = 0, maxN = 0;
    for (int i = 0; i < a.ngroups; ++i) {
        HGemmGroup& g = a.g[i];
        CILRS_CHECK(g.A && g.B && g.C && g.M >= 1 && g.N >= 1 && g.K >= 1, "hgemm: bad group");
        maxM = g.M > maxM ? g.M : maxM;
        maxN = g.N > maxN ? g.N : maxN;
        // float4 operand loads along k need whole, aligned quads
        const bool a_vec = g.K % 8 == 0 && g.lda % 4 == 0 && ((uintptr_t)g.A & 15) == 0;
        const bool b_vec = g.ldb % 4 == 0 && ((uintptr_t)g.B & 15) == 0;
        g.vec = (mode == 0) ? (a_vec && b_vec) : (mode == 1 ? a_vec : 0);
    }
    const dim3 grid(cdiv(maxN, 32), cdiv(maxM, 32), a.ngroups);
    if (mode == 0) hgemm_kernel<0><<<grid, 256, 0, s>>>(a);
    else if (mode == 1) hgemm_kernel<1><<<grid, 256, 0, s>>>(a);
    else hgemm_kernel<2><<<grid, 256, 0, s>>>(a);
    CILRS_LAUNCH_CHECK();
    return 0;
}

}  // namespace cilrs
