// Single-frame inference as ONE persistent launch (reference control loop,
// model/autonomous_drive.py:908-920: one 200x88 frame per tick through model.eval()).
//
// At B=1 the network is ~40 dependent steps of 1-3 us of work each; as separate launches every
// step pays the dependent-kernel boundary plus a cold prologue (round 2: 55 launches, 0.47 ms).
// Here one 1,024-thread workgroup per CU stays resident and walks a stage table:
//
//     preprocess (uint8 HWC -> normalised NHWC4) | stem 7x7/s2 | max-pool | 33 convolution stages
//     (a block's 1x1 downsample shares the stage of its conv1) | three head layers
//
// separated by grid barriers.  Measured prices on MI355X (tools/grid_barrier_probe.hip,
// profiles/r03_barrier_probe.log): bare barrier 1.5 us, with a drained 1 KB sc1 hand-off 3.0 us,
// the fenced (__threadfence) form 14 us, a trivial dependent launch 2.4 us.
//
// Hand-off (cdna guide G16, "every load sc1" form, no release / acquire fences): every byte one
// stage hands to the next is STORED write-through (`sc1`, aux 16) and drained by its storing wave
// (s_waitcnt vmcnt(0)) in front of the workgroup barrier that precedes the arrival; every LOAD of
// such bytes is an `sc1` buffer load to registers (never plain, never scalar); the arrival is one
// agent-scope atomic add per workgroup on one of 8 counter shards, the poll is one `sc1` load per
// shard by 8 lanes of ONE wave, the other waves wait behind a workgroup barrier.  Weights, folded
// BatchNorm tables and the frame are written by EARLIER launches and use plain loads.  Counters are
// monotonic: the epoch base lives next to them and is advanced by block 0 at the end of a launch,
// so nothing is zeroed per call.  Every spin is bounded; a block that gives up keeps arriving at
// the remaining barriers (so nobody else hangs on it), skips the work, and the outputs are NaN.
//
// Convolution stage = conv_small.hip's scheme inside the stage loop: a 16x16 output tile per
// group of WPT waves (2..16, chosen per stage so that one pass of tile slots covers the layer), the
// waves of a group split the reduction index, operands global -> registers ->
// v_mfma_f32_16x16x4_f32, partial tiles summed through LDS in wave order (deterministic), folded
// BatchNorm / ReLU / residual epilogue by the group's first wave.
#include "common.h"

#include <string.h>

namespace cilrs {
namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int kThreads = 1024;
constexpr int kSU = 9;                       // k-groups in flight per wave (18 buffer loads)
constexpr int kDescInts = (int)(sizeof(B1Stage) / sizeof(int));
constexpr int kSpinLimit = 1 << 21;          // ~1-2 s of polling before a block gives up
static_assert(sizeof(B1Stage) % 16 == 0, "B1Stage must stay 16-byte granular");

#define RFL(x) __builtin_amdgcn_readfirstlane((int)(x))

typedef __amdgpu_buffer_rsrc_t rsrc_t;

__device__ __forceinline__ f32x4 ld_act(const rsrc_t rs, const unsigned off) {      // handed-off bytes
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 16));
}
__device__ __forceinline__ f32x4 ld_const(const rsrc_t rs, const unsigned off) {    // earlier launches
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
}
__device__ __forceinline__ float ld_const1(const rsrc_t rs, const unsigned off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, 0, 0));
}
__device__ __forceinline__ void st_act(const rsrc_t rs, const unsigned off, const f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, (int)off, 0, 16);
}
__device__ __forceinline__ void st_act1(const rsrc_t rs, const unsigned off, const float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, (int)off, 0, 16);
}
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ int xcd_chunk(const int bid, const int nwg) {
    // blocks b and b+8 share an XCD (round-robin dispatch; speed only): consecutive logical ids
    // on one XCD, so tiles that share weights or input rows meet in one L2
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

// ---- grid barrier ----------------------------------------------------------------------------
// Every wave that stored handed-off bytes has drained them before the call.
__device__ __forceinline__ void grid_barrier(int* sync, const int target, volatile int* lds_fail,
                                             int* status) {
    __syncthreads();
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        if (lane == 0)
            __hip_atomic_fetch_add(sync + (blockIdx.x & 7) * 32, 1, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        if (!*lds_fail) {
            for (int spins = 0;;) {
                int ok = 1;
                if (lane < 8) {
                    int v;
                    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)"
                                 : "=&v"(v) : "v"(sync + lane * 32) : "memory");
                    ok = (v - target) >= 0;                   // wrap-safe
                }
                if (__all(ok)) break;
                if (++spins > kSpinLimit) {
                    if (lane == 0) {
                        *lds_fail = 1;
                        __hip_atomic_store(status + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
    }
    __syncthreads();
}

// ---- convolution stage -------------------------------------------------------------------------
// MODE 0: a k-group is 16 input channels of one filter tap (Cin % 16 == 0, <= 9 taps).
// MODE 1: the stem -- 7x7 taps over the channel-padded image (Cin = 4): a k-group is 4 taps x 4
//         channels, lane quad kq takes tap 4g + kq; weights are the padded [64][49][4] copy.
#define B1_FINE(i_)                                                                     \
    do {                                                                                \
        if (fine && threadIdx.x == 0) fine[i_] = __builtin_amdgcn_s_memrealtime();      \
    } while (0)

template <int MODE>
__device__ __forceinline__ void conv_stage(const B1Stage* st, const rsrc_t rsW, const rsrc_t rsP,
                                           float* red, const int nblk, const int lb,
                                           long long* fine) {
    const int lane = threadIdx.x & 63, wave = RFL(threadIdx.x >> 6);
    const int r = lane & 15, kq = lane >> 4;
    const int wpt = RFL(st->wpt);
    const int wshift = wpt == 16 ? 4 : wpt == 8 ? 3 : wpt == 4 ? 2 : 1;
    const int groups = 16 >> wshift;
    const int grp = wave >> wshift, wig = wave & (wpt - 1);
    const int total = RFL(st->total_tiles);
    const int slots = nblk * groups;
    const int npass = (total + slots - 1) / slots;
    const int nt0 = RFL(st->c[0].ntiles);
    const rsrc_t rsB = MODE == 1 ? rsW : rsP;

    for (int pass = 0; pass < npass; ++pass) {
        int t = pass * slots + grp * nblk + lb;
        const bool active = t < total;                       // wave-uniform
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        f32x4 e_sc = {1.f, 1.f, 1.f, 1.f}, e_sh = {0.f, 0.f, 0.f, 0.f}, e_add = {0.f, 0.f, 0.f, 0.f};
        unsigned e_yoff = 0u;
        bool e_store = false;
        int e_relu = 0, e_relu_post = 0;
        if (active) {
            const int pi = t >= nt0 ? 1 : 0;
            if (pi) t -= nt0;
            const B1Conv* c = &st->c[pi];
            const int H = RFL(c->H), W = RFL(c->W), Cin = RFL(c->Cin), Wo = RFL(c->Wo);
            const int Cout = RFL(c->Cout), K = RFL(c->K), stride = RFL(c->stride), pad = RFL(c->pad);
            const int M = RFL(c->M), nmt = RFL(c->nmt), S = RFL(c->S), cshift = RFL(c->cshift);
            const unsigned x_off = (unsigned)RFL(c->x_off), w_off = (unsigned)RFL(c->w_off);
            const int nt = t / nmt, mt = t - nt * nmt;       // m fastest: tiles sharing weights adjoin
            const int m0 = mt * 16, n0 = nt * 16;
            if (wig == 0) {      // the group's epilogue wave fetches its operands up front
                const int row = lane >> 2, c4 = (lane & 3) * 4;
                const int m = m0 + row;
                e_store = m < M;
                const unsigned eo = (unsigned)((m * Cout + n0 + c4) * 4);
                e_yoff = (unsigned)RFL(c->y_off) + eo;
                if (RFL(c->has_add))
                    e_add = ld_act(rsW, e_store ? (unsigned)RFL(c->add_off) + eo : 0xFFFFFFFFu);
                e_relu = RFL(c->relu);
                e_relu_post = RFL(c->relu_post);
            }
            B1_FINE(0);
            // ---- this wave's slice of the reduction index
            const int per = (S + wpt - 1) >> wshift;
            const int sb = wig * per, se = min(S, sb + per);
            const int m = m0 + r;
            const int oh = m / Wo, ow = m - oh * Wo;
            const int hb = oh * stride - pad, wb = ow * stride - pad;
            unsigned rowOff = 0u, rowMask = 0u, wOff = 0u;
            int tapA_v = 0;
            if (MODE == 0) {
                const int ntaps = K * K;
                if (m < M)
                    for (int kh = 0; kh < K; ++kh)
                        for (int kw = 0; kw < K; ++kw) {
                            const int h = hb + kh, w = wb + kw;
                            if (h >= 0 && w >= 0 && h < H && w < W) rowMask |= 1u << (kh * K + kw);
                        }
                rowOff = x_off + (unsigned)(((hb * W + wb) * Cin + kq * 4) * 4);
                wOff = w_off + (unsigned)((((n0 + r) * ntaps) * Cin + kq * 4) * 4);
                if (lane < ntaps) tapA_v = ((lane / K) * W + lane % K) * Cin * 4;
            }
            B1_FINE(1);
            for (int s0 = sb; s0 < se; s0 += kSU) {
                f32x4 av[kSU], bv[kSU];
#pragma unroll
                for (int u = 0; u < kSU; ++u) {
                    const int s = s0 + u;                     // wave-uniform
                    const bool live = s < se;
                    unsigned offA, offB;
                    if (MODE == 0) {
                        const int tap = live ? (s >> cshift) : 0;
                        const int cg = s & ((1 << cshift) - 1);
                        const unsigned toff = (unsigned)__builtin_amdgcn_readlane(tapA_v, tap) +
                                              (unsigned)(cg * 64);
                        offA = (live && ((rowMask >> tap) & 1u)) ? rowOff + toff : 0xFFFFFFFFu;
                        offB = live ? wOff + (unsigned)((tap * Cin + cg * 16) * 4) : 0xFFFFFFFFu;
                    } else {
                        const int tap = 4 * s + kq;           // per lane
                        const int kh = tap / 7, kw = tap - 7 * kh;
                        const int h = hb + kh, w = wb + kw;
                        const bool tv = live && tap < 49;
                        const bool ok = tv && m < M && h >= 0 && w >= 0 && h < H && w < W;
                        offA = ok ? x_off + (unsigned)((h * W + w) * 16) : 0xFFFFFFFFu;
                        offB = tv ? w_off + (unsigned)(((n0 + r) * 49 + tap) * 16) : 0xFFFFFFFFu;
                    }
                    av[u] = ld_act(rsW, offA);
                    bv[u] = ld_const(rsB, offB);
                }
                __builtin_amdgcn_sched_barrier(0);    // every load in flight before the first MFMA
                B1_FINE(2);
#pragma unroll
                for (int u = 0; u < kSU; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][e], bv[u][e], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            B1_FINE(3);
            if (wig == 0) {      // folded BatchNorm of the tile's channels: lands under the LDS hand-over
                const int c4 = (lane & 3) * 4;
                e_sc = ld_const(rsW, (unsigned)RFL(c->scale_off) + (unsigned)((n0 + c4) * 4));
                e_sh = ld_const(rsW, (unsigned)RFL(c->shift_off) + (unsigned)((n0 + c4) * 4));
            }
            // C/D map of 16x16x4: col = lane & 15, row = 4 * (lane >> 4) + i
#pragma unroll
            for (int i = 0; i < 4; ++i) red[wave * 256 + (4 * kq + i) * 16 + r] = acc[i];
        }
        B1_FINE(4);
        __syncthreads();
        B1_FINE(5);
        if (active && wig == 0) {
            const int row = lane >> 2, c4 = (lane & 3) * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            for (int w = 0; w < wpt; ++w)                     // wave order: deterministic
                v += *reinterpret_cast<const f32x4*>(red + ((grp << wshift) + w) * 256 + row * 16 + c4);
            v = v * e_sc + e_sh;
            if (e_relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            v += e_add;
            if (e_relu_post) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            if (e_store) st_act(rsW, e_yoff, v);
            B1_FINE(6);
            drain_stores();
        }
        B1_FINE(7);
        __syncthreads();
    }
}

// ---- uint8 HWC frame -> normalised NHWC4 (preprocess_image, autonomous_drive.py:897-902) -------
__device__ __forceinline__ void pre_stage(const B1Stage* st, const B1Launch& a, const rsrc_t rsW,
                                          const int nblk) {
    const int npix = RFL(st->pH) * RFL(st->pW);
    const unsigned dst = (unsigned)RFL(st->dst_off);
    bool stored = false;
    for (int i = threadIdx.x * nblk + blockIdx.x; i < npix; i += kThreads * nblk) {
        const unsigned char* p = a.frame + (size_t)i * 3;
        f32x4 v;
        v[0] = ((float)p[0] / 255.0f - a.mean[0]) / a.stdv[0];
        v[1] = ((float)p[1] / 255.0f - a.mean[1]) / a.stdv[1];
        v[2] = ((float)p[2] / 255.0f - a.mean[2]) / a.stdv[2];
        v[3] = 0.f;
        st_act(rsW, dst + (unsigned)i * 16u, v);
        stored = true;
    }
    if (stored) drain_stores();
}

// ---- MaxPool2d(3, 2, 1) on NHWC (torch: first maximum in scan order wins) ---------------------
__device__ __forceinline__ void pool_stage(const B1Stage* st, const rsrc_t rsW, const int nblk) {
    const int H = RFL(st->pH), W = RFL(st->pW), C = RFL(st->pC), Ho = RFL(st->pHo), Wo = RFL(st->pWo);
    const unsigned src = (unsigned)RFL(st->src_off), dst = (unsigned)RFL(st->dst_off);
    const int cq = C >> 2;
    const int total = Ho * Wo * cq;
    bool stored = false;
    for (int i = threadIdx.x * nblk + blockIdx.x; i < total; i += kThreads * nblk) {
        const int q = i % cq, p = i / cq;
        const int ow = p % Wo, oh = p / Wo;
        f32x4 v[9];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int h = oh * 2 - 1 + kh, w = ow * 2 - 1 + kw;
                const bool ok = h >= 0 && w >= 0 && h < H && w < W;
                v[kh * 3 + kw] = ld_act(rsW, ok ? src + (unsigned)(((h * W + w) * C + q * 4) * 4)
                                                : 0xFFFFFFFFu);
            }
        f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        bool first = true;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int h = oh * 2 - 1 + kh, w = ow * 2 - 1 + kw;
                if (h < 0 || w < 0 || h >= H || w >= W) continue;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = v[kh * 3 + kw][e];
                    if (first || x > best[e] || x != x) best[e] = x;
                }
                first = false;
            }
        st_act(rsW, dst + (unsigned)i * 16u, best);
        stored = true;
    }
    if (stored) drain_stores();
}

// ---- one nn.Linear of the commanded branch + the speed head: one wave per output feature -------
// (arithmetic order of heads_small_layer_kernel / heads_small_pre_kernel in heads_optim.hip:
//  the two paths agree bit for bit given the same inputs)
__device__ __forceinline__ void head_stage(const B1Stage* st, const B1Launch& a, const rsrc_t rsW,
                                           const rsrc_t rsP, float* hx, float* s1, const int nblk) {
    const B1Head* h = &st->h;
    const int tid = threadIdx.x, lane = tid & 63, wave = RFL(tid >> 6);
    const long long cmd = a.cmd[0];
    const int k = (cmd < 0 || cmd > 3) ? 0 : (int)cmd;
    const int first = RFL(h->first), last = RFL(h->last);
    const int in0 = RFL(h->in[0]), in1 = RFL(h->in[1]);
    float* x0 = hx;
    float* x1 = hx + 640;
    if (first) {
        // a command outside 0..3: torch.gather would raise (:397-398); the host reads this word
        if (blockIdx.x == 0 && tid == 0) a.status[0] = (cmd < 0 || cmd > 3) ? 1 : 0;
        const int HW = RFL(h->featHW), C = RFL(h->featC);
        const unsigned fo = (unsigned)RFL(h->feat_off);
        if (tid < (C >> 2)) {              // AdaptiveAvgPool2d(1,1) + Flatten (:369)
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            for (int p = 0; p < HW; ++p) s += ld_act(rsW, fo + (unsigned)((p * C + tid * 4) * 4));
            *reinterpret_cast<f32x4*>(x0 + tid * 4) = s / (float)HW;
        } else if (tid >= 256 && tid < 384) {      // speed encoder layer 1 (:371-372, 391)
            const int t = tid - 256;
            const float w0 = ld_const1(rsP, (unsigned)RFL(h->se_w0) + (unsigned)t * 4u);
            const float b0 = ld_const1(rsP, (unsigned)RFL(h->se_b0) + (unsigned)t * 4u);
            s1[t] = fmaxf(fmaf(a.speed[0], w0, 0.f) + b0, 0.f);
        }
        __syncthreads();
        {                                           // speed encoder layer 2 (:373-374)
            const float xa = s1[lane], xb = s1[lane + 64];
            const unsigned w1 = (unsigned)RFL(h->se_w1), b1 = (unsigned)RFL(h->se_b1);
            for (int o = wave * 8; o < wave * 8 + 8; ++o) {
                const float wl = ld_const1(rsP, w1 + (unsigned)((o * 128 + lane) * 4));
                const float wh = ld_const1(rsP, w1 + (unsigned)((o * 128 + lane + 64) * 4));
                float v = fmaf(xb, wh, xa * wl);
#pragma unroll
                for (int sft = 32; sft > 0; sft >>= 1) v += __shfl_xor(v, sft);
                if (lane == 0) x0[C + o] = fmaxf(v + ld_const1(rsP, b1 + (unsigned)o * 4u), 0.f);
            }
        }
        x1 = x0;                                     // the speed head reads the visual half (:393)
    } else {
        if (tid < (in0 >> 2))
            *reinterpret_cast<f32x4*>(x0 + tid * 4) =
                ld_act(rsW, (unsigned)RFL(h->x_off[0]) + (unsigned)tid * 16u);
        else if (tid >= 256 && tid - 256 < (in1 >> 2))
            *reinterpret_cast<f32x4*>(x1 + (tid - 256) * 4) =
                ld_act(rsW, (unsigned)RFL(h->x_off[1]) + (unsigned)(tid - 256) * 16u);
    }
    __syncthreads();
    const int out0 = RFL(h->out[0]), out1 = RFL(h->out[1]);
    const int relu = RFL(h->relu);
    bool stored = false;
    for (int slot = wave * nblk + blockIdx.x; slot < out0 + out1; slot += 16 * nblk) {
        const int chain = slot >= out0 ? 1 : 0;
        const int o = chain ? slot - out0 : slot;
        const int in = chain ? in1 : in0;
        const int nq = in >> 2;
        const int widx = chain ? 4 : k;
        const unsigned wr = (unsigned)RFL(h->w_off[widx]) + (unsigned)(o * in) * 4u;
        const float* xr = chain ? x1 : x0;
        f32x4 wv[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int q = lane + 64 * j;
            wv[j] = ld_const(rsP, q < nq ? wr + (unsigned)q * 16u : 0xFFFFFFFFu);
        }
        const float bias = ld_const1(rsP, (unsigned)RFL(h->b_off[widx]) + (unsigned)o * 4u);
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int q = lane + 64 * j;
            if (q < nq) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(xr + q * 4);
                acc = fmaf(xv[0], wv[j][0], acc);
                acc = fmaf(xv[1], wv[j][1], acc);
                acc = fmaf(xv[2], wv[j][2], acc);
                acc = fmaf(xv[3], wv[j][3], acc);
            }
        }
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) acc += __shfl_xor(acc, sft);
        if (lane == 0) {
            float v = acc + bias;
            if (relu) v = fmaxf(v, 0.f);
            if (last) {
                if (chain) a.pred_speed[o] = v;      // read by the host after the launch
                else a.controls[o] = v;
            } else {
                st_act1(rsW, (unsigned)RFL(h->y_off[chain]) + (unsigned)o * 4u, v);
                stored = true;
            }
        }
    }
    if (stored) drain_stores();
}

__global__ __launch_bounds__(kThreads) void infer_b1_kernel(const B1Launch a) {
    __shared__ __attribute__((aligned(16))) int desc[2][kDescInts];
    __shared__ __attribute__((aligned(16))) float red[16 * 256];
    __shared__ __attribute__((aligned(16))) float hx[2 * 640];
    __shared__ float s1[128];
    __shared__ int fail;
    const int tid = threadIdx.x;
    const int nblk = gridDim.x;
    const int lb = xcd_chunk(blockIdx.x, nblk);
    const rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)a.ws, 0, (int)(unsigned)a.ws_bytes,
                                                         0x00020000);
    const rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc((void*)a.params, 0,
                                                         (int)(unsigned)a.param_bytes, 0x00020000);
    // epoch base of the monotonic counters: advanced by block 0 at the end of the previous launch
    int base;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(base) : "v"(a.sync + 8 * 32) : "memory");
    base = RFL(base);
    if (tid < kDescInts) desc[0][tid] = reinterpret_cast<const int*>(a.table)[tid];
    if (tid == 0) {
        fail = 0;
        if (blockIdx.x == 0)     // a give-up (>= 1 s of polling) cannot race this store
            __hip_atomic_store(a.status + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const int per_shard = nblk >> 3;
    const int nstages = a.nstages;
    for (int s = 0; s < nstages; ++s) {
        const B1Stage* st = reinterpret_cast<const B1Stage*>(desc[s & 1]);
        int next_word = 0;                    // next stage's descriptor travels under this stage
        if (s + 1 < nstages && tid < kDescInts)
            next_word = reinterpret_cast<const int*>(a.table + s + 1)[tid];
        const int failed = *reinterpret_cast<volatile int*>(&fail);
        if (a.stamps && blockIdx.x == 0 && tid == 0) a.stamps[s] = __builtin_amdgcn_s_memrealtime();
        if (!failed) {
            const int type = RFL(st->type);
            long long* fine = (a.stamps && blockIdx.x == 0)
                                  ? a.stamps + 2 * (kB1MaxStages + 1) + 8 * s : nullptr;
            if (type == B1_CONV) conv_stage<0>(st, rsW, rsP, red, nblk, lb, fine);
            else if (type == B1_STEM) conv_stage<1>(st, rsW, rsP, red, nblk, lb, fine);
            else if (type == B1_POOL) pool_stage(st, rsW, nblk);
            else if (type == B1_PRE) pre_stage(st, a, rsW, nblk);
            else head_stage(st, a, rsW, rsP, hx, s1, nblk);
        }
        if (a.stamps && blockIdx.x == 0 && tid == 0)
            a.stamps[kB1MaxStages + 1 + s] = __builtin_amdgcn_s_memrealtime();
        if (s + 1 < nstages) {
            if (tid < kDescInts) desc[(s + 1) & 1][tid] = next_word;
            grid_barrier(a.sync, base + (s + 1) * per_shard, &fail, a.status);
        }
    }
    if (blockIdx.x == 0 && tid == 0) {
        __hip_atomic_store(a.sync + 8 * 32, base + (nstages - 1) * per_shard, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        if (*reinterpret_cast<volatile int*>(&fail)) {
            const float nan = __builtin_nanf("");
            a.controls[0] = nan; a.controls[1] = nan; a.controls[2] = nan; a.pred_speed[0] = nan;
        }
    }
}

}  // namespace

int infer_b1_grid(int* blocks) {
    CILRS_CHECK(blocks != nullptr, "infer_b1_grid: NULL");
    int dev = 0;
    CILRS_HIP(hipGetDevice(&dev));
    hipDeviceProp_t p;
    CILRS_HIP(hipGetDeviceProperties(&p, dev));
    int per_cu = 0;
    CILRS_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(
        &per_cu, reinterpret_cast<const void*>(&infer_b1_kernel), kThreads, 0));
    // one workgroup per CU, a multiple of the 8 counter shards; every block must be resident
    *blocks = per_cu >= 1 ? (p.multiProcessorCount / 8) * 8 : 0;
    return 0;
}

int launch_infer_b1(const B1Launch& a, int blocks, hipStream_t s) {
    CILRS_CHECK(blocks >= 8 && blocks % 8 == 0, "infer_b1: grid %d is not a multiple of 8", blocks);
    CILRS_CHECK(a.nstages >= 2 && a.nstages <= kB1MaxStages, "infer_b1: bad stage count");
    CILRS_CHECK(a.ws_bytes < (1ull << 32) && a.param_bytes < (1ull << 32),
                "infer_b1: arenas must be addressable with 32-bit offsets");
    infer_b1_kernel<<<blocks, kThreads, 0, s>>>(a);
    CILRS_LAUNCH_CHECK();
    return 0;
}

}  // namespace cilrs
