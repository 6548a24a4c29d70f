// BatchNorm2d (train + eval), ReLU, residual add, MaxPool2d(3,2,1), AdaptiveAvgPool2d(1) -- forward
// and backward -- for NHWC fp32 activations.  These are the HBM-bound pieces of the CILRS trunk
// (SURVEY.md 2b): torchvision BasicBlock = conv-BN-ReLU-conv-BN-(+id)-ReLU, stem = conv-BN-ReLU-
// maxpool, tail = avgpool (reference model/autonomous_drive.py:366-370).
//
// Semantics follow torch.nn.BatchNorm2d exactly: batch mean / BIASED variance for normalisation,
// UNBIASED variance into running_var, momentum 0.1, eps 1e-5, num_batches_tracked += 1; like
// torch's CPU kernels the per-channel statistics are accumulated in double from fp32 partial sums.
//
// Every reduction is a fixed-shape tree (per-thread rows -> LDS -> per-block partial -> serial
// double sum over blocks): deterministic, no atomics.  Channel is the fastest index, so a wave reads
// whole 256-B/1-KiB rows (float4 per lane) -- fully coalesced.
#include "common.h"

namespace cilrs {

namespace {

constexpr int kMaxPartBlocks = 1024;

// optional bf16 shadow of an fp32 tensor (the 16-bit operand of the next convolution in the bf16
// training mode): 4 elements = 8 bytes per store
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_bf16x4(__bf16* __restrict__ p, const size_t i4, const f32x4 v) {
    const bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    *reinterpret_cast<bf16x4*>(p + i4 * 4) = o;
}

// partial[0][c][blk] = sum_rows v1, partial[1][c][blk] = sum_rows v2   (channel-major)
// MODE 0: v1 = y, v2 = y*y                     (forward statistics)
// MODE 1: v1 = g, v2 = g * (y - mean) * rstd   (backward reductions), g = dz * (z > 0 if relu)
template <int MODE>
__global__ __launch_bounds__(256) void bn_colreduce_kernel(
    const float* __restrict__ y, const float* __restrict__ dz, const float* __restrict__ z,
    const float* __restrict__ stats, float* __restrict__ partial, const int M, const int C,
    const int relu, const int rows_per_block) {
    __shared__ float red[2][256 * 4];
    // rows wider than 1,024 channels (the Bottleneck variant's 2,048) are cut into column groups
    // of 1,024: blockIdx.y = group, one float4 per thread per row as before
    const int Cg = C > 1024 ? 1024 : C;    // channels this block covers
    const int c0 = blockIdx.y * Cg;
    const int cq = Cg >> 2;                // float4 quads per row (16..256)
    const int tpr = cq;                    // threads per row
    const int rpi = 256 / tpr;             // rows per iteration
    const int q = threadIdx.x % tpr, rsub = threadIdx.x / tpr;
    const int row_begin = blockIdx.x * rows_per_block;
    const int row_end = min(M, row_begin + rows_per_block);
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    f32x4 mean = {0.f, 0.f, 0.f, 0.f}, rstd = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 1) {
        mean = *reinterpret_cast<const f32x4*>(stats + c0 + q * 4);
        rstd = *reinterpret_cast<const f32x4*>(stats + C + c0 + q * 4);
    }
    auto accum = [&](int r, f32x4& a1, f32x4& a2) {
        const size_t o = (size_t)r * C + c0 + q * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(y + o);
        if (MODE == 0) {
            a1 += v;
            a2 += v * v;
        } else {
            f32x4 g = *reinterpret_cast<const f32x4*>(dz + o);
            if (relu) {
                const f32x4 zz = *reinterpret_cast<const f32x4*>(z + o);
#pragma unroll
                for (int e = 0; e < 4; ++e) g[e] = zz[e] > 0.f ? g[e] : 0.f;
            }
            a1 += g;
            a2 += g * ((v - mean) * rstd);
        }
    };
    {
        // four independent row streams per thread: 4x the bytes in flight, and a shallower
        // summation tree (fixed order => deterministic)
        f32x4 t1[4], t2[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            t1[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            t2[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        int r = row_begin + rsub;
        for (; r + 3 * rpi < row_end; r += 4 * rpi) {
#pragma unroll
            for (int u = 0; u < 4; ++u) accum(r + u * rpi, t1[u], t2[u]);
        }
        for (; r < row_end; r += rpi) accum(r, t1[0], t2[0]);
        s1 = (t1[0] + t1[1]) + (t1[2] + t1[3]);
        s2 = (t2[0] + t2[1]) + (t2[2] + t2[3]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        red[0][threadIdx.x * 4 + e] = s1[e];
        red[1][threadIdx.x * 4 + e] = s2[e];
    }
    __syncthreads();
    // thread c (< C) sums the rpi row-subgroups for channel c in a fixed order
    for (int c = threadIdx.x; c < Cg; c += 256) {
        float a1 = 0.f, a2 = 0.f;
        const int qq = c >> 2, e = c & 3;
        for (int rs = 0; rs < rpi; ++rs) {
            a1 += red[0][(rs * tpr + qq) * 4 + e];
            a2 += red[1][(rs * tpr + qq) * 4 + e];
        }
        partial[(size_t)(c0 + c) * gridDim.x + blockIdx.x] = a1;
        partial[(size_t)(C + c0 + c) * gridDim.x + blockIdx.x] = a2;
    }
}

// Sum one channel's per-block partials (channel-major layout [2][C][nblk]): one 256-thread block
// per channel.  The kernel is pure load latency (the partials were written by another kernel's
// blocks on other XCDs, so they come from the memory side): every thread issues all of its loads
// (8 per array cover 2,048 partials) before the first add, then a fixed-order double tree --
// shuffles inside a wave, LDS across the four waves (deterministic).  Returns the two sums to
// thread 0.  (One wave per channel with a 4-way unrolled loop took 5.5 us per launch at
// nblk = 2,200: nine dependent memory round trips.)
constexpr int kFinThreads = 256;
__device__ __forceinline__ bool partial_sums(const float* __restrict__ partial, const int nblk,
                                             const int C, const int c, double& s1, double& s2) {
    __shared__ double red[2][kFinThreads / 64];
    const float* r1 = partial + (size_t)c * nblk;
    const float* r2 = partial + (size_t)(C + c) * nblk;
    double p1 = 0.0, p2 = 0.0;
    for (int base = 0; base < nblk; base += 8 * kFinThreads) {
        float v1[8], v2[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int b = base + threadIdx.x + kFinThreads * u;
            v1[u] = b < nblk ? r1[b] : 0.f;
            v2[u] = b < nblk ? r2[b] : 0.f;
        }
        p1 += (((double)v1[0] + (double)v1[1]) + ((double)v1[2] + (double)v1[3])) +
              (((double)v1[4] + (double)v1[5]) + ((double)v1[6] + (double)v1[7]));
        p2 += (((double)v2[0] + (double)v2[1]) + ((double)v2[2] + (double)v2[3])) +
              (((double)v2[4] + (double)v2[5]) + ((double)v2[6] + (double)v2[7]));
    }
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) {
        p1 += __shfl_xor(p1, sft);
        p2 += __shfl_xor(p2, sft);
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = p1;
        red[1][threadIdx.x >> 6] = p2;
    }
    __syncthreads();
    s1 = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    s2 = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    __syncthreads();                 // (callers may loop over channels: red is reused)
    return threadIdx.x == 0;
}

// ---- finalize inside the apply launch ------------------------------------------------------------
// The per-channel finalize (a C-block launch of pure load latency, 5.4 us + the launch gap, 72 of
// them per train step) runs as the FIRST job of the apply kernel instead.  The apply kernel is one
// 1,024-thread workgroup per CU; workgroup b finalizes channels 4b .. 4b+3 (one per 256-thread
// group, the same fixed-order double-precision tree as the stand-alone kernel), publishes the 3-4
// coefficients with write-through (sc1) stores, drains them and arrives on one of 8 sharded device
// counters (channel & 7, one 128-byte line each).  Every workgroup first puts its first elements
// in flight, then ONE wave polls the 8 shards (8 lanes, sc1 loads) until they reach this launch's
// target, and everybody reads the coefficients with sc1 loads -- the fence-free hand-off of
// infer_b1.hip / cdna guide G16.  The counters are monotonic (the host passes the cumulative
// target, compared wrap-safe), so nothing is reset on the device: a first version with one counter,
// 1,024 polling workgroups and a departure counter for the reset cost +25 us per launch in
// same-address atomics (profiles/r03_bn_fused.log).  No deadlock: the grid is at most one
// workgroup per CU, all resident; the spin is bounded all the same, and a timeout poisons the
// output with NaN (fails loudly in every parity test and in the loss).
constexpr int kBnSpinLimit = 1 << 21;
constexpr int kBnShardStride = 32;              // ints: one 128-byte line per shard
constexpr int kApplyThreads = 1024;
__device__ __forceinline__ void st_sc1(float* p, const float v) {
    asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ int ld_int_sc1(const int* p) {
    int v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void publish_channel(int* sync, const int c) {   // the group's thread 0
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(&sync[(c & 7) * kBnShardStride], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// returns false on a timeout (block-uniform)
__device__ __forceinline__ bool wait_channels(const int* sync, const int target) {
    __shared__ int ok_s;
    if (threadIdx.x < 64) {
        int spins = 0, ok = 1;
        while (true) {
            int here = 1;
            if (threadIdx.x < 8)      // unsigned difference: wrap-safe (see infer_b1.hip grid_wait)
                here = (int)((unsigned)ld_int_sc1(&sync[threadIdx.x * kBnShardStride]) - (unsigned)target) >= 0;
            if (__all(here)) break;
            if (++spins > kBnSpinLimit) { ok = 0; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (threadIdx.x == 0) ok_s = ok;
    }
    __syncthreads();
    return ok_s != 0;
}
// partial_sums for a 1,024-thread workgroup: 256-thread group g sums channel c (valid or not, all
// groups run the same barriers); true for the group's first thread
__device__ __forceinline__ bool partial_sums4(const float* __restrict__ partial, const int nblk,
                                              const int C, const int c, const bool valid,
                                              double& s1, double& s2) {
    __shared__ double red4[2][kApplyThreads / 64];
    const int t = threadIdx.x & 255, g = threadIdx.x >> 8;
    const float* r1 = partial + (size_t)(valid ? c : 0) * nblk;
    const float* r2 = partial + (size_t)(C + (valid ? c : 0)) * nblk;
    double p1 = 0.0, p2 = 0.0;
    for (int base = 0; base < nblk; base += 8 * 256) {
        float v1[8], v2[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int b = base + t + 256 * u;
            v1[u] = b < nblk ? r1[b] : 0.f;
            v2[u] = b < nblk ? r2[b] : 0.f;
        }
        p1 += (((double)v1[0] + (double)v1[1]) + ((double)v1[2] + (double)v1[3])) +
              (((double)v1[4] + (double)v1[5]) + ((double)v1[6] + (double)v1[7]));
        p2 += (((double)v2[0] + (double)v2[1]) + ((double)v2[2] + (double)v2[3])) +
              (((double)v2[4] + (double)v2[5]) + ((double)v2[6] + (double)v2[7]));
    }
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) {
        p1 += __shfl_xor(p1, sft);
        p2 += __shfl_xor(p2, sft);
    }
    if ((threadIdx.x & 63) == 0) {
        red4[0][threadIdx.x >> 6] = p1;
        red4[1][threadIdx.x >> 6] = p2;
    }
    __syncthreads();
    s1 = (red4[0][4 * g] + red4[0][4 * g + 1]) + (red4[0][4 * g + 2] + red4[0][4 * g + 3]);
    s2 = (red4[1][4 * g] + red4[1][4 * g + 1]) + (red4[1][4 * g + 2] + red4[1][4 * g + 3]);
    __syncthreads();
    return t == 0 && valid;
}
// stats layout (floats): [0,C) mean | [C,2C) rstd | [2C,3C) w = gamma*rstd | [3C,4C) b = beta-mean*w
template <bool SC1>
__device__ __forceinline__ void fwd_finalize_channel(
    const float* __restrict__ partial, const int nblk, const int M, const int C, const int c,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* running_mean,
    float* running_var, long long* nbt, const float momentum, const float eps,
    float* __restrict__ stats) {
    double s1, s2;
    if (SC1 ? !partial_sums4(partial, nblk, C, c, c < C, s1, s2)
            : !partial_sums(partial, nblk, C, c, s1, s2)) return;
    const double mean = s1 / M;
    double var = s2 / M - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float w = gamma[c] * rstd;
    if (SC1) {
        st_sc1(stats + c, (float)mean);
        st_sc1(stats + C + c, rstd);
        st_sc1(stats + 2 * C + c, w);
        st_sc1(stats + 3 * C + c, beta[c] - (float)mean * w);
    } else {
        stats[c] = (float)mean;
        stats[C + c] = rstd;
        stats[2 * C + c] = w;
        stats[3 * C + c] = beta[c] - (float)mean * w;
    }
    if (running_mean) {
        const double unbiased = M > 1 ? var * ((double)M / (double)(M - 1)) : var;
        running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
    }
    if (nbt && c == 0) *nbt += 1;
}
__global__ __launch_bounds__(kFinThreads) void bn_fwd_finalize_kernel(
    const float* __restrict__ partial, const int nblk, const int M, const int C,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* running_mean,
    float* running_var, long long* nbt, const float momentum, const float eps,
    float* __restrict__ stats) {
    fwd_finalize_channel<false>(partial, nblk, M, C, blockIdx.x, gamma, beta, running_mean,
                                running_var, nbt, momentum, eps, stats);
}

// eval mode: stats from the running statistics
__global__ void bn_eval_stats_kernel(const float* __restrict__ gamma,
                                     const float* __restrict__ beta,
                                     const float* __restrict__ running_mean,
                                     const float* __restrict__ running_var, const float eps,
                                     const int C, float* __restrict__ stats) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float rstd = 1.0f / sqrtf(running_var[c] + eps);
    const float w = gamma[c] * rstd;
    stats[c] = running_mean[c];
    stats[C + c] = rstd;
    stats[2 * C + c] = w;
    stats[3 * C + c] = beta[c] - running_mean[c] * w;
}

// every BatchNorm layer's eval-mode scale/shift in one launch: block b = layer b
__global__ __launch_bounds__(256) void bn_eval_stats_all_kernel(const BnEvalTable t,
                                                                const float* __restrict__ params,
                                                                const float* __restrict__ running,
                                                                float* __restrict__ ws,
                                                                const float eps) {
    const int l = blockIdx.x;
    const int C = t.C[l];
    const float* gamma = params + t.gamma[l];
    const float* beta = params + t.beta[l];
    const float* rm = running + t.rm[l];
    const float* rv = running + t.rv[l];
    float* stats = ws + t.stats[l];
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float rstd = 1.0f / sqrtf(rv[c] + eps);
        const float w = gamma[c] * rstd;
        stats[c] = rm[c];
        stats[C + c] = rstd;
        stats[2 * C + c] = w;
        stats[3 * C + c] = beta[c] - rm[c] * w;
    }
}

// z = relu?( y*w + b (+ residual) )
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ y,
                                                       const float* __restrict__ stats,
                                                       const float* __restrict__ residual,
                                                       float* __restrict__ z, const size_t total4,
                                                       const int C, const int relu,
                                                       __bf16* __restrict__ z16) {
    const int cq = C >> 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
         i += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % cq);
        const f32x4 w = *reinterpret_cast<const f32x4*>(stats + 2 * C + q * 4);
        const f32x4 b = *reinterpret_cast<const f32x4*>(stats + 3 * C + q * 4);
        f32x4 v = *reinterpret_cast<const f32x4*>(y + i * 4);
        v = v * w + b;
        if (residual) v += *reinterpret_cast<const f32x4*>(residual + i * 4);
        if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        *reinterpret_cast<f32x4*>(z + i * 4) = v;
        if (z16) store_bf16x4(z16, i, v);
    }
}

// finalize + apply in one launch (see "finalize inside the apply launch"); the grid stride is a
// multiple of C/4, so a thread keeps its channel quad for the whole loop
struct BnFinArgs {
    const float* partial; int nblk; int M;
    const float* gamma; const float* beta; float* running_mean; float* running_var;
    long long* nbt; float momentum; float eps;
    int* sync; int target;
};
__global__ __launch_bounds__(kApplyThreads) void bn_finalize_apply_kernel(
    const BnFinArgs f, const float* __restrict__ y, float* __restrict__ stats,
    const float* __restrict__ residual, float* __restrict__ z, const size_t total4, const int C,
    const int relu, __bf16* __restrict__ z16) {
    for (int base = blockIdx.x * 4; base < C; base += gridDim.x * 4) {       // block-uniform trips
        const int c = base + (threadIdx.x >> 8);
        fwd_finalize_channel<true>(f.partial, f.nblk, f.M, C, c, f.gamma, f.beta, f.running_mean,
                                   f.running_var, f.nbt, f.momentum, f.eps, stats);
        if ((threadIdx.x & 255) == 0 && c < C) publish_channel(f.sync, c);
    }
    const int cq = C >> 2;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int q = (int)(i % cq);
    // kU elements per thread and trip, all loads of a trip issued before the first use (the
    // kernel is pure HBM streaming: bytes in flight per CU set the rate); the first trip's loads
    // are in flight while the channels arrive
    constexpr int kU = 4;
    f32x4 v[kU], r[kU];
    auto load_trip = [&](const size_t base) {
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const size_t j = base + u * stride;
            v[u] = r[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (j < total4) {
                v[u] = *reinterpret_cast<const f32x4*>(y + j * 4);
                if (residual) r[u] = *reinterpret_cast<const f32x4*>(residual + j * 4);
            }
        }
    };
    load_trip(i);
    const bool ok = wait_channels(f.sync, f.target);
    f32x4 w, b;
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(w), "=&v"(b) : "v"(stats + 2 * C + q * 4), "v"(stats + 3 * C + q * 4) : "memory");
    if (!ok) w = f32x4{__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")};
    while (i < total4) {
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const size_t j = i + u * stride;
            if (j < total4) {
                f32x4 o = v[u] * w + b;
                if (residual) o += r[u];
                if (relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
                }
                *reinterpret_cast<f32x4*>(z + j * 4) = o;
                if (z16) store_bf16x4(z16, j, o);
            }
        }
        i += kU * stride;
        if (i < total4) load_trip(i);
    }
}

// coef layout: [0,C) c1 = gamma*rstd | [C,2C) c2 = sum(g)/M | [2C,3C) c3 = sum(g*xhat)/M
template <bool SC1>
__device__ __forceinline__ void bwd_finalize_channel(
    const float* __restrict__ partial, const int nblk, const int M, const int C, const int c,
    const float* __restrict__ gamma, const float* __restrict__ stats, float* dgamma,
    float* dbeta, float* __restrict__ coef, const int accumulate) {
    double s1, s2;
    if (SC1 ? !partial_sums4(partial, nblk, C, c, c < C, s1, s2)
            : !partial_sums(partial, nblk, C, c, s1, s2)) return;
    const float db = (float)s1, dg = (float)s2;
    dbeta[c] = accumulate ? dbeta[c] + db : db;
    dgamma[c] = accumulate ? dgamma[c] + dg : dg;
    if (SC1) {
        st_sc1(coef + c, gamma[c] * stats[C + c]);
        st_sc1(coef + C + c, (float)(s1 / M));
        st_sc1(coef + 2 * C + c, (float)(s2 / M));
    } else {
        coef[c] = gamma[c] * stats[C + c];
        coef[C + c] = (float)(s1 / M);
        coef[2 * C + c] = (float)(s2 / M);
    }
}
__global__ __launch_bounds__(kFinThreads) void bn_bwd_finalize_kernel(
    const float* __restrict__ partial, const int nblk, const int M, const int C,
    const float* __restrict__ gamma, const float* __restrict__ stats, float* dgamma,
    float* dbeta, float* __restrict__ coef, const int accumulate) {
    bwd_finalize_channel<false>(partial, nblk, M, C, blockIdx.x, gamma, stats, dgamma, dbeta, coef,
                                accumulate);
}

// g = dz * (z>0 if relu);  dy = (g - c2 - xhat*c3) * c1;  optionally g_out = g (residual path)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(
    const float* __restrict__ dz, const float* __restrict__ z, const float* __restrict__ y,
    const float* __restrict__ stats, const float* __restrict__ coef, float* __restrict__ dy,
    float* __restrict__ g_out, const size_t total4, const int C, const int relu,
    __bf16* __restrict__ dy16) {
    const int cq = C >> 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
         i += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % cq);
        const f32x4 mean = *reinterpret_cast<const f32x4*>(stats + q * 4);
        const f32x4 rstd = *reinterpret_cast<const f32x4*>(stats + C + q * 4);
        const f32x4 c1 = *reinterpret_cast<const f32x4*>(coef + q * 4);
        const f32x4 c2 = *reinterpret_cast<const f32x4*>(coef + C + q * 4);
        const f32x4 c3 = *reinterpret_cast<const f32x4*>(coef + 2 * C + q * 4);
        f32x4 g = *reinterpret_cast<const f32x4*>(dz + i * 4);
        if (relu) {
            const f32x4 zz = *reinterpret_cast<const f32x4*>(z + i * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = zz[e] > 0.f ? g[e] : 0.f;
        }
        const f32x4 xh = (*reinterpret_cast<const f32x4*>(y + i * 4) - mean) * rstd;
        if (g_out) *reinterpret_cast<f32x4*>(g_out + i * 4) = g;
        const f32x4 d = (g - c2 - xh * c3) * c1;
        if (dy) *reinterpret_cast<f32x4*>(dy + i * 4) = d;
        if (dy16) store_bf16x4(dy16, i, d);
    }
}

struct BnBwdFinArgs {
    const float* partial; int nblk; int M;
    const float* gamma; float* dgamma; float* dbeta; int accumulate;
    int* sync; int target;
};
__global__ __launch_bounds__(kApplyThreads) void bn_bwd_finalize_apply_kernel(
    const BnBwdFinArgs f, const float* __restrict__ dz, const float* __restrict__ z,
    const float* __restrict__ y, const float* __restrict__ stats, float* __restrict__ coef,
    float* __restrict__ dy, float* __restrict__ g_out, const size_t total4, const int C,
    const int relu, __bf16* __restrict__ dy16) {
    for (int base = blockIdx.x * 4; base < C; base += gridDim.x * 4) {
        const int c = base + (threadIdx.x >> 8);
        bwd_finalize_channel<true>(f.partial, f.nblk, f.M, C, c, f.gamma, stats, f.dgamma, f.dbeta,
                                   coef, f.accumulate);
        if ((threadIdx.x & 255) == 0 && c < C) publish_channel(f.sync, c);
    }
    const int cq = C >> 2;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int q = (int)(i % cq);
    const f32x4 mean = *reinterpret_cast<const f32x4*>(stats + q * 4);
    const f32x4 rstd = *reinterpret_cast<const f32x4*>(stats + C + q * 4);
    constexpr int kU = 3;
    f32x4 g[kU], zz[kU], yy[kU];
    auto load_trip = [&](const size_t base) {
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const size_t j = base + u * stride;
            g[u] = zz[u] = yy[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (j < total4) {
                g[u] = *reinterpret_cast<const f32x4*>(dz + j * 4);
                if (relu) zz[u] = *reinterpret_cast<const f32x4*>(z + j * 4);
                yy[u] = *reinterpret_cast<const f32x4*>(y + j * 4);
            }
        }
    };
    load_trip(i);
    const bool ok = wait_channels(f.sync, f.target);
    f32x4 c1, c2, c3;
    asm volatile("global_load_dwordx4 %0, %3, off sc1\n\tglobal_load_dwordx4 %1, %4, off sc1\n\t"
                 "global_load_dwordx4 %2, %5, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(c1), "=&v"(c2), "=&v"(c3)
                 : "v"(coef + q * 4), "v"(coef + C + q * 4), "v"(coef + 2 * C + q * 4) : "memory");
    if (!ok) c1 = f32x4{__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")};
    while (i < total4) {
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const size_t j = i + u * stride;
            if (j < total4) {
                f32x4 gg = g[u];
                if (relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) gg[e] = zz[u][e] > 0.f ? gg[e] : 0.f;
                }
                const f32x4 xh = (yy[u] - mean) * rstd;
                if (g_out) *reinterpret_cast<f32x4*>(g_out + j * 4) = gg;
                const f32x4 d = (gg - c2 - xh * c3) * c1;
                if (dy) *reinterpret_cast<f32x4*>(dy + j * 4) = d;
                if (dy16) store_bf16x4(dy16, j, d);
            }
        }
        i += kU * stride;
        if (i < total4) load_trip(i);
    }
}

// ---- MaxPool2d(kernel 3, stride 2, pad 1) --------------------------------------------------
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x,
                                                          float* __restrict__ out,
                                                          unsigned char* __restrict__ argmax,
                                                          const int N, const int H, const int W,
                                                          const int C, const int Ho, const int Wo) {
    const int cq = C >> 2;
    const size_t total = (size_t)N * Ho * Wo * cq;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % cq);
        size_t p = i / cq;
        const int ow = (int)(p % Wo); p /= Wo;
        const int oh = (int)(p % Ho);
        const int n = (int)(p / Ho);
        f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int bi[4] = {0, 0, 0, 0};
        bool first = true;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int h = oh * 2 - 1 + kh;
            if (h < 0 || h >= H) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int w = ow * 2 - 1 + kw;
                if (w < 0 || w >= W) continue;
                const f32x4 v =
                    *reinterpret_cast<const f32x4*>(x + ((size_t)(n * H + h) * W + w) * C + q * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // torch: first maximum in scan order wins (val > max || isnan(val))
                    if (first || v[e] > best[e] || v[e] != v[e]) {
                        best[e] = v[e];
                        bi[e] = kh * 3 + kw;
                    }
                }
                first = false;
            }
        }
        *reinterpret_cast<f32x4*>(out + i * 4) = best;
        if (argmax) {
            uchar4 a;
            a.x = (unsigned char)bi[0]; a.y = (unsigned char)bi[1];
            a.z = (unsigned char)bi[2]; a.w = (unsigned char)bi[3];
            *reinterpret_cast<uchar4*>(argmax + i * 4) = a;
        }
    }
}

// gather form: each INPUT element sums the (<= 4) windows that selected it
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dout,
                                                          const unsigned char* __restrict__ argmax,
                                                          float* __restrict__ dx, const int N,
                                                          const int H, const int W, const int C,
                                                          const int Ho, const int Wo) {
    const int cq = C >> 2;
    const size_t total = (size_t)N * H * W * cq;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % cq);
        size_t p = i / cq;
        const int w = (int)(p % W); p /= W;
        const int h = (int)(p % H);
        const int n = (int)(p / H);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int kh = 0; kh < 3; ++kh) {
            const int t = h + 1 - kh;
            if (t < 0 || (t & 1)) continue;
            const int oh = t >> 1;
            if (oh >= Ho) continue;
            for (int kw = 0; kw < 3; ++kw) {
                const int u = w + 1 - kw;
                if (u < 0 || (u & 1)) continue;
                const int ow = u >> 1;
                if (ow >= Wo) continue;
                const size_t o = (((size_t)(n * Ho + oh) * Wo + ow) * cq + q) * 4;
                const uchar4 a = *reinterpret_cast<const uchar4*>(argmax + o);
                const f32x4 g = *reinterpret_cast<const f32x4*>(dout + o);
                const int k = kh * 3 + kw;
                if (a.x == k) acc[0] += g[0];
                if (a.y == k) acc[1] += g[1];
                if (a.z == k) acc[2] += g[2];
                if (a.w == k) acc[3] += g[3];
            }
        }
        *reinterpret_cast<f32x4*>(dx + i * 4) = acc;
    }
}

// ---- stem: BatchNorm apply + ReLU + MaxPool in one pass, and its backward ------------------------
// The stem's post-BN tensor z (144 MB at B=128) is never materialised: the forward pools
// relu(y*w + b) straight from the conv output y, and the backward rebuilds both the max-pool
// gradient (gather over the <= 4 windows that selected a pixel) and the ReLU mask (y*w + b > 0)
// on the fly.  Saves one write and three reads of z, the write + two reads of dz, and a launch.
__device__ __forceinline__ f32x4 bn_relu4(const f32x4 v, const f32x4 w, const f32x4 b) {
    f32x4 r = v * w + b;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = fmaxf(r[e], 0.f);
    return r;
}

__global__ __launch_bounds__(256) void bn_relu_maxpool_fwd_kernel(
    const float* __restrict__ y, const float* __restrict__ stats, float* __restrict__ out,
    unsigned char* __restrict__ argmax, const int N, const int H, const int W, const int C,
    const int Ho, const int Wo, __bf16* __restrict__ out16) {
    const int cq = C >> 2;
    const size_t total = (size_t)N * Ho * Wo * cq;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % cq);
        size_t p = i / cq;
        const int ow = (int)(p % Wo); p /= Wo;
        const int oh = (int)(p % Ho);
        const int n = (int)(p / Ho);
        const f32x4 sw = *reinterpret_cast<const f32x4*>(stats + 2 * C + q * 4);
        const f32x4 sb = *reinterpret_cast<const f32x4*>(stats + 3 * C + q * 4);
        f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int bi[4] = {0, 0, 0, 0};
        bool first = true;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int h = oh * 2 - 1 + kh;
            if (h < 0 || h >= H) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int w = ow * 2 - 1 + kw;
                if (w < 0 || w >= W) continue;
                const f32x4 v = bn_relu4(
                    *reinterpret_cast<const f32x4*>(y + ((size_t)(n * H + h) * W + w) * C + q * 4),
                    sw, sb);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (first || v[e] > best[e] || v[e] != v[e]) {
                        best[e] = v[e];
                        bi[e] = kh * 3 + kw;
                    }
                }
                first = false;
            }
        }
        *reinterpret_cast<f32x4*>(out + i * 4) = best;
        if (out16) store_bf16x4(out16, i, best);
        uchar4 a;
        a.x = (unsigned char)bi[0]; a.y = (unsigned char)bi[1];
        a.z = (unsigned char)bi[2]; a.w = (unsigned char)bi[3];
        *reinterpret_cast<uchar4*>(argmax + i * 4) = a;
    }
}

// g(n,h,w,c) = [y*w+b > 0] * sum over the pooling windows that selected (h,w) of dpool:
// pixel (h,w) is reached by window (oh,ow) through tap (kh,kw) when h + 1 - kh = 2*oh and
// w + 1 - kw = 2*ow; torch accumulates the windows in (kh, kw) ascending order.
// Computed for a 2x2 input patch (2a..2a+1, 2b..2b+1) at once.  A pixel's candidate windows
// depend on the parity of its coordinates (1, 2, 2 or 4 of them), so a per-pixel gather
// diverges inside a wave and issues its loads one dependent window at a time (measured: 131 us
// for the reduction pass at B=128, 1.45 TB/s); a patch
// always needs exactly the four windows (a,b) (a,b+1) (a+1,b) (a+1,b+1): every thread issues the
// same 12 independent loads (4 pixels of y, 4 windows of argmax + dpool).  Accumulation order per
// pixel is (kh ascending, kw ascending).
struct PoolPatch {
    f32x4 y[4];        // pixels (2a,2b) (2a,2b+1) (2a+1,2b) (2a+1,2b+1); zeros where outside
    f32x4 g[4];
};
__device__ __forceinline__ f32x4 pool_sel(const uchar4 a, const int k, const f32x4 d) {
    f32x4 r;
    r[0] = a.x == k ? d[0] : 0.f;
    r[1] = a.y == k ? d[1] : 0.f;
    r[2] = a.z == k ? d[2] : 0.f;
    r[3] = a.w == k ? d[3] : 0.f;
    return r;
}
__device__ __forceinline__ void pool_patch(const float* __restrict__ y,
                                           const float* __restrict__ dpool,
                                           const unsigned char* __restrict__ argmax,
                                           const f32x4 sw, const f32x4 sb, const int n,
                                           const int a, const int b, const int q, const int H,
                                           const int W, const int C, const int Ho, const int Wo,
                                           PoolPatch& pp) {
    const int cq = C >> 2;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const bool ph1 = 2 * a + 1 < H, pw1 = 2 * b + 1 < W;       // second row / column of the patch
    const bool wh1 = a + 1 < Ho, ww1 = b + 1 < Wo;             // second window row / column
    const size_t p00 = (((size_t)n * H + 2 * a) * W + 2 * b) * C + q * 4;
    pp.y[0] = *reinterpret_cast<const f32x4*>(y + p00);
    pp.y[1] = pw1 ? *reinterpret_cast<const f32x4*>(y + p00 + C) : zero;
    pp.y[2] = ph1 ? *reinterpret_cast<const f32x4*>(y + p00 + (size_t)W * C) : zero;
    pp.y[3] = ph1 && pw1 ? *reinterpret_cast<const f32x4*>(y + p00 + (size_t)W * C + C) : zero;
    const size_t w00 = (((size_t)(n * Ho + a) * Wo + b) * cq + q) * 4;
    const size_t dw = (size_t)cq * 4, dh = (size_t)Wo * cq * 4;
    const uchar4 none = {255, 255, 255, 255};
    const uchar4 a00 = *reinterpret_cast<const uchar4*>(argmax + w00);
    const uchar4 a01 = ww1 ? *reinterpret_cast<const uchar4*>(argmax + w00 + dw) : none;
    const uchar4 a10 = wh1 ? *reinterpret_cast<const uchar4*>(argmax + w00 + dh) : none;
    const uchar4 a11 = wh1 && ww1 ? *reinterpret_cast<const uchar4*>(argmax + w00 + dh + dw) : none;
    const f32x4 d00 = *reinterpret_cast<const f32x4*>(dpool + w00);
    const f32x4 d01 = ww1 ? *reinterpret_cast<const f32x4*>(dpool + w00 + dw) : zero;
    const f32x4 d10 = wh1 ? *reinterpret_cast<const f32x4*>(dpool + w00 + dh) : zero;
    const f32x4 d11 = wh1 && ww1 ? *reinterpret_cast<const f32x4*>(dpool + w00 + dh + dw) : zero;
    // tap index k = kh*3 + kw of the window that reaches the pixel
    pp.g[0] = pool_sel(a00, 4, d00);
    pp.g[1] = pool_sel(a01, 3, d01) + pool_sel(a00, 5, d00);
    pp.g[2] = pool_sel(a10, 1, d10) + pool_sel(a00, 7, d00);
    pp.g[3] = ((pool_sel(a11, 0, d11) + pool_sel(a10, 2, d10)) + pool_sel(a01, 6, d01)) +
              pool_sel(a00, 8, d00);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x4 z = pp.y[i] * sw + sb;
#pragma unroll
        for (int e = 0; e < 4; ++e) pp.g[i][e] = z[e] > 0.f ? pp.g[i][e] : 0.f;
    }
    if (!pw1) { pp.g[1] = zero; pp.g[3] = zero; }
    if (!ph1) { pp.g[2] = zero; pp.g[3] = zero; }
}

// column partials of the stem's BatchNorm backward (layout of bn_colreduce_kernel<1>); a block
// owns `patches_per_block` consecutive 2x2 patches
__global__ __launch_bounds__(256) void bn_colreduce_pool_kernel(
    const float* __restrict__ y, const float* __restrict__ dpool,
    const unsigned char* __restrict__ argmax, const float* __restrict__ stats,
    float* __restrict__ partial, const int N, const int H, const int W, const int C, const int Ho,
    const int Wo, const int patches_per_block) {
    __shared__ float red[2][256 * 4];
    const int PH = (H + 1) >> 1, PW = (W + 1) >> 1;
    const int NP = N * PH * PW;
    const int cq = C >> 2, tpr = cq, ppi = 256 / tpr;
    const int q = threadIdx.x % tpr, psub = threadIdx.x / tpr;
    const int p_begin = blockIdx.x * patches_per_block;
    const int p_end = min(NP, p_begin + patches_per_block);
    const f32x4 mean = *reinterpret_cast<const f32x4*>(stats + q * 4);
    const f32x4 rstd = *reinterpret_cast<const f32x4*>(stats + C + q * 4);
    const f32x4 sw = *reinterpret_cast<const f32x4*>(stats + 2 * C + q * 4);
    const f32x4 sb = *reinterpret_cast<const f32x4*>(stats + 3 * C + q * 4);
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    for (int p = p_begin + psub; p < p_end; p += ppi) {
        const int b = p % PW, t = p / PW, a = t % PH, n = t / PH;
        PoolPatch pp;
        pool_patch(y, dpool, argmax, sw, sb, n, a, b, q, H, W, C, Ho, Wo, pp);
        s1 += (pp.g[0] + pp.g[1]) + (pp.g[2] + pp.g[3]);
        s2 += (pp.g[0] * ((pp.y[0] - mean) * rstd) + pp.g[1] * ((pp.y[1] - mean) * rstd)) +
              (pp.g[2] * ((pp.y[2] - mean) * rstd) + pp.g[3] * ((pp.y[3] - mean) * rstd));
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
        for (int rs = 0; rs < ppi; ++rs) {
            a1 += red[0][(rs * tpr + qq) * 4 + e];
            a2 += red[1][(rs * tpr + qq) * 4 + e];
        }
        partial[(size_t)c * gridDim.x + blockIdx.x] = a1;
        partial[(size_t)(C + c) * gridDim.x + blockIdx.x] = a2;
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_pool_kernel(
    const float* __restrict__ y, const float* __restrict__ dpool,
    const unsigned char* __restrict__ argmax, const float* __restrict__ stats,
    const float* __restrict__ coef, float* __restrict__ dy, const int N, const int H, const int W,
    const int C, const int Ho, const int Wo) {
    const int cq = C >> 2;
    const int PH = (H + 1) >> 1, PW = (W + 1) >> 1;
    const size_t total = (size_t)N * PH * PW * cq;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % cq);
        size_t p = i / cq;
        const int b = (int)(p % PW); p /= PW;
        const int a = (int)(p % PH);
        const int n = (int)(p / PH);
        const f32x4 mean = *reinterpret_cast<const f32x4*>(stats + q * 4);
        const f32x4 rstd = *reinterpret_cast<const f32x4*>(stats + C + q * 4);
        const f32x4 sw = *reinterpret_cast<const f32x4*>(stats + 2 * C + q * 4);
        const f32x4 sb = *reinterpret_cast<const f32x4*>(stats + 3 * C + q * 4);
        const f32x4 c1 = *reinterpret_cast<const f32x4*>(coef + q * 4);
        const f32x4 c2 = *reinterpret_cast<const f32x4*>(coef + C + q * 4);
        const f32x4 c3 = *reinterpret_cast<const f32x4*>(coef + 2 * C + q * 4);
        PoolPatch pp;
        pool_patch(y, dpool, argmax, sw, sb, n, a, b, q, H, W, C, Ho, Wo, pp);
        const bool ph1 = 2 * a + 1 < H, pw1 = 2 * b + 1 < W;
        const size_t p00 = (((size_t)n * H + 2 * a) * W + 2 * b) * C + q * 4;
        const size_t off[4] = {p00, p00 + C, p00 + (size_t)W * C, p00 + (size_t)W * C + C};
        const bool ok[4] = {true, pw1, ph1, ph1 && pw1};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!ok[k]) continue;
            const f32x4 xh = (pp.y[k] - mean) * rstd;
            *reinterpret_cast<f32x4*>(dy + off[k]) = (pp.g[k] - c2 - xh * c3) * c1;
        }
    }
}

// ---- AdaptiveAvgPool2d((1,1)) + Flatten ------------------------------------------------------
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const float* __restrict__ x,
                                                          float* __restrict__ out, const int N,
                                                          const int HW, const int C,
                                                          const int out_ld) {
    const int cq = C >> 2;
    const int total = N * cq;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int q = i % cq, n = i / cq;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int p = 0; p < HW; ++p)
        s += *reinterpret_cast<const f32x4*>(x + ((size_t)n * HW + p) * C + q * 4);
    const float inv = (float)HW;
    *reinterpret_cast<f32x4*>(out + (size_t)n * out_ld + q * 4) = s / inv;
}

__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float* __restrict__ dout,
                                                          float* __restrict__ dx, const int N,
                                                          const int HW, const int C,
                                                          const int dout_ld) {
    const int cq = C >> 2;
    const size_t total = (size_t)N * HW * cq;
    const float inv = (float)HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % cq);
        const int n = (int)(i / ((size_t)HW * cq));
        const f32x4 g = *reinterpret_cast<const f32x4*>(dout + (size_t)n * dout_ld + q * 4);
        *reinterpret_cast<f32x4*>(dx + i * 4) = g / inv;
    }
}

// ---- 16-bit tensors end to end (bf16 training mode, round 4) --------------------------------------
// The same three passes on bf16 NHWC tensors: 8 channels = 16 bytes per lane, fp32 arithmetic, fp32
// statistics / coefficients, results rounded to bf16 once on the way out.  BatchNorm here is
// exactly the fp32 BatchNorm of the STORED (rounded) convolution output.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void cvt8(const bf16x8 v, f32x4& lo, f32x4& hi) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        lo[e] = (float)v[e];
        hi[e] = (float)v[4 + e];
    }
}
__device__ __forceinline__ bf16x8 pack8(const f32x4 lo, const f32x4 hi) {
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        o[e] = (__bf16)lo[e];
        o[4 + e] = (__bf16)hi[e];
    }
    return o;
}

// partial layout and modes of bn_colreduce_kernel; a thread owns 8 channels of its rows
template <int MODE>
__global__ __launch_bounds__(256) void bn16_colreduce_kernel(
    const __bf16* __restrict__ y, const __bf16* __restrict__ dz, const __bf16* __restrict__ z,
    const float* __restrict__ stats, float* __restrict__ partial, const int M, const int C,
    const int relu, const int rows_per_block) {
    __shared__ float red[2][256 * 8];
    const int Cg = C > 2048 ? 2048 : C;
    const int c0 = blockIdx.y * Cg;
    const int tpr = Cg >> 3;               // threads per row
    const int rpi = 256 / tpr;             // rows per iteration
    const int q = threadIdx.x % tpr, rsub = threadIdx.x / tpr;
    const int row_begin = blockIdx.x * rows_per_block;
    const int row_end = min(M, row_begin + rows_per_block);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 mlo = zero, mhi = zero, rlo = zero, rhi = zero;
    if (MODE == 1) {
        mlo = *reinterpret_cast<const f32x4*>(stats + c0 + q * 8);
        mhi = *reinterpret_cast<const f32x4*>(stats + c0 + q * 8 + 4);
        rlo = *reinterpret_cast<const f32x4*>(stats + C + c0 + q * 8);
        rhi = *reinterpret_cast<const f32x4*>(stats + C + c0 + q * 8 + 4);
    }
    f32x4 a1lo[2], a1hi[2], a2lo[2], a2hi[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) a1lo[u] = a1hi[u] = a2lo[u] = a2hi[u] = zero;
    auto accum = [&](const int r, const int u) {
        const size_t o = (size_t)r * C + c0 + q * 8;
        f32x4 vlo, vhi;
        cvt8(*reinterpret_cast<const bf16x8*>(y + o), vlo, vhi);
        if (MODE == 0) {
            a1lo[u] += vlo; a1hi[u] += vhi;
            a2lo[u] += vlo * vlo; a2hi[u] += vhi * vhi;
        } else {
            f32x4 glo, ghi;
            cvt8(*reinterpret_cast<const bf16x8*>(dz + o), glo, ghi);
            if (relu) {
                f32x4 zlo, zhi;
                cvt8(*reinterpret_cast<const bf16x8*>(z + o), zlo, zhi);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    glo[e] = zlo[e] > 0.f ? glo[e] : 0.f;
                    ghi[e] = zhi[e] > 0.f ? ghi[e] : 0.f;
                }
            }
            a1lo[u] += glo; a1hi[u] += ghi;
            a2lo[u] += glo * ((vlo - mlo) * rlo);
            a2hi[u] += ghi * ((vhi - mhi) * rhi);
        }
    };
    int r = row_begin + rsub;
    for (; r + rpi < row_end; r += 2 * rpi) {
        accum(r, 0);
        accum(r + rpi, 1);
    }
    for (; r < row_end; r += rpi) accum(r, 0);
    const f32x4 s1lo = a1lo[0] + a1lo[1], s1hi = a1hi[0] + a1hi[1];
    const f32x4 s2lo = a2lo[0] + a2lo[1], s2hi = a2hi[0] + a2hi[1];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        red[0][threadIdx.x * 8 + e] = s1lo[e];
        red[0][threadIdx.x * 8 + 4 + e] = s1hi[e];
        red[1][threadIdx.x * 8 + e] = s2lo[e];
        red[1][threadIdx.x * 8 + 4 + e] = s2hi[e];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < Cg; c += 256) {
        float a1 = 0.f, a2 = 0.f;
        const int qq = c >> 3, e = c & 7;
        for (int rs = 0; rs < rpi; ++rs) {
            a1 += red[0][(rs * tpr + qq) * 8 + e];
            a2 += red[1][(rs * tpr + qq) * 8 + e];
        }
        partial[(size_t)(c0 + c) * gridDim.x + blockIdx.x] = a1;
        partial[(size_t)(C + c0 + c) * gridDim.x + blockIdx.x] = a2;
    }
}

// z16 = round( relu?( y16 * w + b (+ residual16) ) ).  kU independent 16-byte streams per thread,
// every load of a trip in flight before the first use (pure HBM streaming).
__global__ __launch_bounds__(256) void bn16_apply_kernel(const __bf16* __restrict__ y,
                                                         const float* __restrict__ stats,
                                                         const __bf16* __restrict__ residual,
                                                         __bf16* __restrict__ z, const size_t total8,
                                                         const int C, const int relu) {
    constexpr int kU = 4;
    const int cq = C >> 3;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const int qstep = (int)(stride % (size_t)cq);
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    int q = (int)(i % (size_t)cq);
    const float* sw = stats + 2 * C;
    const float* sb = stats + 3 * C;
    for (; i < total8; i += kU * stride) {
        bf16x8 v[kU], r[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const size_t j = i + u * stride;
            if (j < total8) {
                v[u] = *reinterpret_cast<const bf16x8*>(y + j * 8);
                if (residual) r[u] = *reinterpret_cast<const bf16x8*>(residual + j * 8);
            }
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const size_t j = i + u * stride;
            if (j < total8) {
                f32x4 lo, hi;
                cvt8(v[u], lo, hi);
                lo = lo * *reinterpret_cast<const f32x4*>(sw + q * 8) +
                     *reinterpret_cast<const f32x4*>(sb + q * 8);
                hi = hi * *reinterpret_cast<const f32x4*>(sw + q * 8 + 4) +
                     *reinterpret_cast<const f32x4*>(sb + q * 8 + 4);
                if (residual) {
                    f32x4 rl, rh;
                    cvt8(r[u], rl, rh);
                    lo += rl;
                    hi += rh;
                }
                if (relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        lo[e] = fmaxf(lo[e], 0.f);
                        hi[e] = fmaxf(hi[e], 0.f);
                    }
                }
                *reinterpret_cast<bf16x8*>(z + j * 8) = pack8(lo, hi);
            }
            q += qstep;
            if (q >= cq) q -= cq;
        }
    }
}

// g = dz16 * (z16 > 0 if relu);  dy16 = round((g - c2 - xhat*c3) * c1), xhat from y16;
// optionally g_out16 = g (the residual path's gradient: a selection of bf16 values, exact)
__global__ __launch_bounds__(256) void bn16_bwd_apply_kernel(
    const __bf16* __restrict__ dz, const __bf16* __restrict__ z, const __bf16* __restrict__ y,
    const float* __restrict__ stats, const float* __restrict__ coef, __bf16* __restrict__ dy,
    __bf16* __restrict__ g_out, const size_t total8, const int C, const int relu) {
    constexpr int kU = 3;
    const int cq = C >> 3;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const int qstep = (int)(stride % (size_t)cq);
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    int q = (int)(i % (size_t)cq);
    for (; i < total8; i += kU * stride) {
        bf16x8 gv[kU], zv[kU], yv[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const size_t j = i + u * stride;
            if (j < total8) {
                gv[u] = *reinterpret_cast<const bf16x8*>(dz + j * 8);
                if (relu) zv[u] = *reinterpret_cast<const bf16x8*>(z + j * 8);
                yv[u] = *reinterpret_cast<const bf16x8*>(y + j * 8);
            }
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const size_t j = i + u * stride;
            if (j < total8) {
                f32x4 glo, ghi, ylo, yhi;
                cvt8(gv[u], glo, ghi);
                cvt8(yv[u], ylo, yhi);
                if (relu) {
                    f32x4 zlo, zhi;
                    cvt8(zv[u], zlo, zhi);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        glo[e] = zlo[e] > 0.f ? glo[e] : 0.f;
                        ghi[e] = zhi[e] > 0.f ? ghi[e] : 0.f;
                    }
                }
                if (g_out) *reinterpret_cast<bf16x8*>(g_out + j * 8) = pack8(glo, ghi);
                const f32x4 xl = (ylo - *reinterpret_cast<const f32x4*>(stats + q * 8)) *
                                 *reinterpret_cast<const f32x4*>(stats + C + q * 8);
                const f32x4 xh = (yhi - *reinterpret_cast<const f32x4*>(stats + q * 8 + 4)) *
                                 *reinterpret_cast<const f32x4*>(stats + C + q * 8 + 4);
                const f32x4 dl = (glo - *reinterpret_cast<const f32x4*>(coef + C + q * 8) -
                                  xl * *reinterpret_cast<const f32x4*>(coef + 2 * C + q * 8)) *
                                 *reinterpret_cast<const f32x4*>(coef + q * 8);
                const f32x4 dh = (ghi - *reinterpret_cast<const f32x4*>(coef + C + q * 8 + 4) -
                                  xh * *reinterpret_cast<const f32x4*>(coef + 2 * C + q * 8 + 4)) *
                                 *reinterpret_cast<const f32x4*>(coef + q * 8 + 4);
                *reinterpret_cast<bf16x8*>(dy + j * 8) = pack8(dl, dh);
            }
            q += qstep;
            if (q >= cq) q -= cq;
        }
    }
}

// AdaptiveAvgPool2d backward into a bf16 gradient tensor
__global__ __launch_bounds__(256) void avgpool_bwd16_kernel(const float* __restrict__ dout,
                                                            __bf16* __restrict__ dx, const int N,
                                                            const int HW, const int C,
                                                            const int dout_ld) {
    const int cq = C >> 3;
    const size_t total = (size_t)N * HW * cq;
    const float inv = (float)HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % cq);
        const int n = (int)(i / ((size_t)HW * cq));
        const f32x4 lo = *reinterpret_cast<const f32x4*>(dout + (size_t)n * dout_ld + q * 8);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(dout + (size_t)n * dout_ld + q * 8 + 4);
        *reinterpret_cast<bf16x8*>(dx + i * 8) = pack8(lo / inv, hi / inv);
    }
}

int grid_for(size_t total, int per_block = 256, int cap = 4096) {
    size_t b = (total + per_block - 1) / per_block;
    if (b > (size_t)cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

// one 1,024-thread workgroup per CU: every workgroup of the finalize-inside-apply kernels is resident
int apply_grid_cap() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 64;
        cus = p.multiProcessorCount > 0 ? p.multiProcessorCount : 64;
    }
    return cus;
}

struct ColPlan { int nblk; int rows_per_block; };
int col_groups(int C) { return C > 1024 ? C / 1024 : 1; }
ColPlan col_plan(int M, int C) {
    const int rpi = 256 / ((C > 1024 ? 1024 : C) >> 2);
    // >= 8 iterations of the 4-way unrolled loop per block when M allows, <= kMaxPartBlocks blocks
    int rows = cdiv(M, kMaxPartBlocks);
    const int min_rows = 4 * rpi;          // at least one 4-way unrolled iteration per block
    if (rows < min_rows) rows = min_rows;
    rows = cdiv(rows, rpi) * rpi;
    ColPlan p{cdiv(M, rows), rows};
    return p;
}

}  // namespace

size_t bn_partial_floats(int C) { return (size_t)kMaxPartBlocks * 2 * C; }

static int check_c(int C) {
    CILRS_CHECK(C % 4 == 0 && C >= 4 &&
                    ((C <= 1024 && 256 % (C / 4) == 0) || (C <= 4096 && C % 1024 == 0)),
                "batchnorm: unsupported channel count %d", C);
    return 0;
}

int launch_bn_train_fwd(const float* y, int M, int C, const float* gamma, const float* beta,
                        float* running_mean, float* running_var, long long* nbt, float momentum,
                        float eps, const float* residual, int relu, float* stats, float* partial,
                        float* z, int pre_nblk, hipStream_t s, void* z16, BnSync* sync) {
    if (check_c(C)) return 1;
    int nblk = pre_nblk;
    if (nblk <= 0) {
        const ColPlan p = col_plan(M, C);
        bn_colreduce_kernel<0><<<dim3(p.nblk, col_groups(C)), 256, 0, s>>>(y, nullptr, nullptr, nullptr, partial, M, C,
                                                      0, p.rows_per_block);
        CILRS_LAUNCH_CHECK();
        nblk = p.nblk;
    }
    if (z && sync && sync->dev && C <= 1024 && C % 8 == 0) {   // finalize as the first job of the apply launch
        const size_t total4 = (size_t)M * C / 4;
        sync->total = (int)((unsigned)sync->total + (unsigned)(C / 8));
        const BnFinArgs f{partial, nblk, M, gamma, beta, running_mean, running_var, nbt, momentum, eps,
                          sync->dev, sync->total};
        bn_finalize_apply_kernel<<<grid_for(total4, kApplyThreads, apply_grid_cap()), kApplyThreads, 0, s>>>(
            f, y, stats, residual, z, total4, C, relu, reinterpret_cast<__bf16*>(z16));
        CILRS_LAUNCH_CHECK();
        return 0;
    }
    bn_fwd_finalize_kernel<<<C, kFinThreads, 0, s>>>(partial, nblk, M, C, gamma, beta,
                                                        running_mean, running_var, nbt, momentum,
                                                        eps, stats);
    CILRS_LAUNCH_CHECK();
    if (z) {
        const size_t total4 = (size_t)M * C / 4;
        bn_apply_kernel<<<grid_for(total4), 256, 0, s>>>(y, stats, residual, z, total4, C, relu,
                                                         reinterpret_cast<__bf16*>(z16));
        CILRS_LAUNCH_CHECK();
    }
    return 0;
}

int launch_bn_eval_stats_all(const BnEvalTable& t, const float* params, const float* bn_running,
                             float* ws, float eps, hipStream_t s) {
    bn_eval_stats_all_kernel<<<t.n, 256, 0, s>>>(t, params, bn_running, ws, eps);
    CILRS_LAUNCH_CHECK();
    return 0;
}

int launch_bn_eval_fwd(const float* y, int M, int C, const float* gamma, const float* beta,
                       const float* running_mean, const float* running_var, float eps,
                       const float* residual, int relu, float* stats, float* z, hipStream_t s) {
    if (check_c(C)) return 1;
    bn_eval_stats_kernel<<<cdiv(C, 128), 128, 0, s>>>(gamma, beta, running_mean, running_var, eps,
                                                      C, stats);
    CILRS_LAUNCH_CHECK();
    const size_t total4 = (size_t)M * C / 4;
    bn_apply_kernel<<<grid_for(total4), 256, 0, s>>>(y, stats, residual, z, total4, C, relu,
                                                     nullptr);
    CILRS_LAUNCH_CHECK();
    return 0;
}

int launch_bn_bwd(const float* dz, const float* z, const float* y, int M, int C,
                  const float* gamma, const float* stats, int relu, float* dgamma, float* dbeta,
                  int accumulate, float* coef, float* partial, float* dy, float* g_out,
                  int pre_nblk, hipStream_t s, void* dy16, BnSync* sync) {
    if (check_c(C)) return 1;
    int nblk = pre_nblk;
    if (nblk <= 0) {
        const ColPlan p = col_plan(M, C);
        bn_colreduce_kernel<1><<<dim3(p.nblk, col_groups(C)), 256, 0, s>>>(y, dz, z, stats, partial, M, C, relu,
                                                      p.rows_per_block);
        CILRS_LAUNCH_CHECK();
        nblk = p.nblk;
    }
    const size_t total4 = (size_t)M * C / 4;
    if (sync && sync->dev && C <= 1024 && C % 8 == 0) {
        sync->total = (int)((unsigned)sync->total + (unsigned)(C / 8));
        const BnBwdFinArgs f{partial, nblk, M, gamma, dgamma, dbeta, accumulate, sync->dev, sync->total};
        bn_bwd_finalize_apply_kernel<<<grid_for(total4, kApplyThreads, apply_grid_cap()), kApplyThreads, 0, s>>>(
            f, dz, z, y, stats, coef, dy, g_out, total4, C, relu, reinterpret_cast<__bf16*>(dy16));
        CILRS_LAUNCH_CHECK();
        return 0;
    }
    bn_bwd_finalize_kernel<<<C, kFinThreads, 0, s>>>(partial, nblk, M, C, gamma, stats,
                                                        dgamma, dbeta, coef, accumulate);
    CILRS_LAUNCH_CHECK();
    bn_bwd_apply_kernel<<<grid_for(total4), 256, 0, s>>>(dz, z, y, stats, coef, dy, g_out, total4,
                                                         C, relu, reinterpret_cast<__bf16*>(dy16));
    CILRS_LAUNCH_CHECK();
    return 0;
}

int launch_bn_relu_maxpool_fwd(const float* y, const float* stats, float* out,
                               unsigned char* argmax, int N, int H, int W, int C, hipStream_t s,
                               void* out16) {
    CILRS_CHECK(C % 4 == 0 && argmax != nullptr, "bn_relu_maxpool: C %% 4, argmax required");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const size_t total = (size_t)N * Ho * Wo * (C / 4);
    bn_relu_maxpool_fwd_kernel<<<grid_for(total), 256, 0, s>>>(
        y, stats, out, argmax, N, H, W, C, Ho, Wo, reinterpret_cast<__bf16*>(out16));
    CILRS_LAUNCH_CHECK();
    return 0;
}

// BatchNorm backward of  maxpool(relu(bn(y)))  given d(maxpool output): dgamma, dbeta, dy
int launch_bn_bwd_pool(const float* dpool, const unsigned char* argmax, const float* y, int N,
                       int H, int W, int C, const float* gamma, const float* stats, float* dgamma,
                       float* dbeta, float* coef, float* partial, float* dy, hipStream_t s) {
    if (check_c(C)) return 1;
    CILRS_CHECK(C <= 1024, "bn_bwd_pool: at most 1024 channels (got %d)", C);
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const int M = N * H * W;
    // a block owns consecutive 2x2 patches: <= kMaxPartBlocks blocks, whole iterations of the
    // 256-thread block (256 / (C/4) patches each)
    const int NP = N * ((H + 1) / 2) * ((W + 1) / 2);
    const int ppi = 256 / (C >> 2);
    int ppb = cdiv(NP, kMaxPartBlocks);
    if (ppb < 2 * ppi) ppb = 2 * ppi;
    ppb = cdiv(ppb, ppi) * ppi;
    const int nblk = cdiv(NP, ppb);
    bn_colreduce_pool_kernel<<<nblk, 256, 0, s>>>(y, dpool, argmax, stats, partial, N, H, W, C,
                                                  Ho, Wo, ppb);
    CILRS_LAUNCH_CHECK();
    bn_bwd_finalize_kernel<<<C, kFinThreads, 0, s>>>(partial, nblk, M, C, gamma, stats, dgamma,
                                                     dbeta, coef, 0);
    CILRS_LAUNCH_CHECK();
    const size_t total = (size_t)NP * (C / 4);
    bn_bwd_apply_pool_kernel<<<grid_for(total), 256, 0, s>>>(y, dpool, argmax, stats, coef, dy, N,
                                                            H, W, C, Ho, Wo);
    CILRS_LAUNCH_CHECK();
    return 0;
}

int launch_maxpool_fwd(const float* x, float* out, unsigned char* argmax, int N, int H, int W,
                       int C, hipStream_t s) {
    CILRS_CHECK(C % 4 == 0, "maxpool: C %% 4");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const size_t total = (size_t)N * Ho * Wo * (C / 4);
    maxpool_fwd_kernel<<<grid_for(total), 256, 0, s>>>(x, out, argmax, N, H, W, C, Ho, Wo);
    CILRS_LAUNCH_CHECK();
    return 0;
}

int launch_maxpool_bwd(const float* dout, const unsigned char* argmax, float* dx, int N, int H,
                       int W, int C, hipStream_t s) {
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const size_t total = (size_t)N * H * W * (C / 4);
    maxpool_bwd_kernel<<<grid_for(total), 256, 0, s>>>(dout, argmax, dx, N, H, W, C, Ho, Wo);
    CILRS_LAUNCH_CHECK();
    return 0;
}

int launch_avgpool_fwd(const float* x, float* out, int N, int HW, int C, int out_ld,
                       hipStream_t s) {
    const int total = N * (C / 4);
    avgpool_fwd_kernel<<<cdiv(total, 256), 256, 0, s>>>(x, out, N, HW, C, out_ld);
    CILRS_LAUNCH_CHECK();
    return 0;
}

int launch_avgpool_bwd(const float* dout, float* dx, int N, int HW, int C, int dout_ld,
                       hipStream_t s) {
    const size_t total = (size_t)N * HW * (C / 4);
    avgpool_bwd_kernel<<<grid_for(total), 256, 0, s>>>(dout, dx, N, HW, C, dout_ld);
    CILRS_LAUNCH_CHECK();
    return 0;
}

// ---- 16-bit tensors end to end -------------------------------------------------------------------
static int check_c16(int C) {
    CILRS_CHECK(C % 8 == 0 && C >= 8 &&
                    ((C <= 2048 && 256 % (C / 8) == 0) || (C <= 8192 && C % 2048 == 0)),
                "batchnorm (16-bit): unsupported channel count %d", C);
    return 0;
}
static ColPlan col_plan16(int M, int C) {
    const int rpi = 256 / ((C > 2048 ? 2048 : C) >> 3);
    int rows = cdiv(M, kMaxPartBlocks);
    const int min_rows = 4 * rpi;
    if (rows < min_rows) rows = min_rows;
    rows = cdiv(rows, rpi) * rpi;
    ColPlan p{cdiv(M, rows), rows};
    return p;
}

int launch_bn16_train_fwd(const void* y16, int M, int C, const float* gamma, const float* beta,
                          float* running_mean, float* running_var, long long* nbt, float momentum,
                          float eps, const void* residual16, int relu, float* stats, float* partial,
                          void* z16, int pre_nblk, hipStream_t s) {
    if (check_c16(C)) return 1;
    const __bf16* y = reinterpret_cast<const __bf16*>(y16);
    int nblk = pre_nblk;
    if (nblk <= 0) {
        const ColPlan p = col_plan16(M, C);
        bn16_colreduce_kernel<0><<<dim3(p.nblk, C > 2048 ? C / 2048 : 1), 256, 0, s>>>(
            y, nullptr, nullptr, nullptr, partial, M, C, 0, p.rows_per_block);
        CILRS_LAUNCH_CHECK();
        nblk = p.nblk;
    }
    bn_fwd_finalize_kernel<<<C, kFinThreads, 0, s>>>(partial, nblk, M, C, gamma, beta, running_mean,
                                                     running_var, nbt, momentum, eps, stats);
    CILRS_LAUNCH_CHECK();
    if (z16) {
        const size_t total8 = (size_t)M * C / 8;
        bn16_apply_kernel<<<grid_for(total8, 256 * 4), 256, 0, s>>>(
            y, stats, reinterpret_cast<const __bf16*>(residual16), reinterpret_cast<__bf16*>(z16),
            total8, C, relu);
        CILRS_LAUNCH_CHECK();
    }
    return 0;
}

int launch_bn16_bwd(const void* dz16, const void* z16, const void* y16, int M, int C,
                    const float* gamma, const float* stats, int relu, float* dgamma, float* dbeta,
                    float* coef, float* partial, void* dy16, void* g_out16, int pre_nblk,
                    hipStream_t s) {
    if (check_c16(C)) return 1;
    CILRS_CHECK(dz16 && y16 && dy16 && (z16 || !relu), "bn16_bwd: NULL tensor");
    int nblk = pre_nblk;
    if (nblk <= 0) {
        const ColPlan p = col_plan16(M, C);
        bn16_colreduce_kernel<1><<<dim3(p.nblk, C > 2048 ? C / 2048 : 1), 256, 0, s>>>(
            reinterpret_cast<const __bf16*>(y16), reinterpret_cast<const __bf16*>(dz16),
            reinterpret_cast<const __bf16*>(z16), stats, partial, M, C, relu, p.rows_per_block);
        CILRS_LAUNCH_CHECK();
        nblk = p.nblk;
    }
    bn_bwd_finalize_kernel<<<C, kFinThreads, 0, s>>>(partial, nblk, M, C, gamma, stats, dgamma,
                                                     dbeta, coef, 0);
    CILRS_LAUNCH_CHECK();
    const size_t total8 = (size_t)M * C / 8;
    bn16_bwd_apply_kernel<<<grid_for(total8, 256 * 3), 256, 0, s>>>(
        reinterpret_cast<const __bf16*>(dz16), reinterpret_cast<const __bf16*>(z16),
        reinterpret_cast<const __bf16*>(y16), stats, coef, reinterpret_cast<__bf16*>(dy16),
        reinterpret_cast<__bf16*>(g_out16), total8, C, relu);
    CILRS_LAUNCH_CHECK();
    return 0;
}

int launch_avgpool_bwd16(const float* dout, void* dx16, int N, int HW, int C, int dout_ld,
                         hipStream_t s) {
    CILRS_CHECK(C % 8 == 0, "avgpool_bwd16: C %% 8");
    const size_t total = (size_t)N * HW * (C / 8);
    avgpool_bwd16_kernel<<<grid_for(total), 256, 0, s>>>(dout, reinterpret_cast<__bf16*>(dx16), N,
                                                        HW, C, dout_ld);
    CILRS_LAUNCH_CHECK();
    return 0;
}

}  // namespace cilrs
