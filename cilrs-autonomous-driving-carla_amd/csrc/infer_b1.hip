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
// group of WPT waves (2..16, chosen per stage so that one pass of slots covers the layer), the
// waves of a group split the reduction index, operands global -> registers ->
// v_mfma_f32_16x16x4_f32, partial tiles summed through LDS in wave order (deterministic), folded
// BatchNorm / ReLU / residual epilogue by the group's first wave.  What bounds a stage is the
// operand stream into each CU (measured ~70 GB/s per CU from L2: a 16x16 tile with reduction
// length 2,304 is 288 KB = 4 us), so
//   * layers whose tiles do not cover the CUs split the reduction index over SEVERAL workgroups
//     (layer4: 64 tiles x 4): each parks its partial tile in a slab (sc1), takes a ticket, and the
//     workgroup whose ticket is last sums the slabs in slice order (deterministic) and runs the
//     epilogue;
//   * everything of stage s+1 that does not depend on stage s -- descriptor, addresses, the
//     WEIGHT fragments, folded BatchNorm -- is computed / loaded BEFORE the grid barrier
//     (ConvPlan), under the drain of the stage's own stores and the wait for the slowest
//     workgroup; after the barrier a wave issues its activation loads and multiplies.
#include "common.h"

#include <string.h>

namespace cilrs {
namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// threadIdx.x behind an opaque move: everything a stage derives from it is recomputed per stage.
// Without it the compiler hoists a dozen lane-derived index terms of ALL stage kinds out of the
// stage loop, runs out of registers and reloads five of them from scratch inside every stage -- a
// memory round trip on each stage's critical path.
__device__ __forceinline__ int tid_now() {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}
constexpr int kThreads = 1024;
constexpr int kSU = 8;                       // k-groups per wave (A and B fragment: 16 buffer loads)
constexpr int kDescInts = (int)(sizeof(B1Stage) / sizeof(int));
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr int kSpinLimit = 1 << 21;          // ~1-2 s of polling before a block gives up
static_assert(sizeof(B1Stage) % 16 == 0, "B1Stage must stay 16-byte granular");

#define RFL(x) __builtin_amdgcn_readfirstlane((int)(x))

typedef __amdgpu_buffer_rsrc_t rsrc_t;

// (`base`: wave-uniform byte offset, added by the hardware; it takes no part in the range check, so
//  a lane offset of 0xFFFFFFFF still reads zeros)
__device__ __forceinline__ f32x4 ld_act(const rsrc_t rs, const unsigned off, const unsigned base = 0u) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, (int)base, 16));
}
__device__ __forceinline__ f32x4 ld_const(const rsrc_t rs, const unsigned off, const unsigned base = 0u) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, (int)base, 0));
}
__device__ __forceinline__ float ld_const1(const rsrc_t rs, const unsigned off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, 0, 0));
}
__device__ __forceinline__ void st_act(const rsrc_t rs, const unsigned off, const f32x4 v,
                                       const unsigned base = 0u) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, (int)off, (int)base, 16);
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

// ---- grid barrier, in two halves -----------------------------------------------------------------
// arrive: every wave that stored handed-off bytes has drained them before the call.
__device__ __forceinline__ void grid_arrive(int* sync) {
    __syncthreads();
    if (threadIdx.x == 0)
        __hip_atomic_fetch_add(sync + (blockIdx.x & 7) * 32, 1, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
}
// wait: until every workgroup has arrived `target / per_shard` times.  One wave polls the eight
// shards; its own older loads (a ConvPlan's weight fragments) return first, which costs nothing:
// the slowest workgroup is still on its way.
__device__ __forceinline__ void grid_wait(int* sync, const int target, volatile int* lds_fail,
                                          int* status) {
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        if (!*lds_fail) {
            for (int spins = 0;;) {
                int ok = 1;
                if (lane < 8) {
                    int v;
                    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)"
                                 : "=&v"(v) : "v"(sync + lane * 32) : "memory");
                    // wrap-safe: the difference is formed in UNSIGNED arithmetic (signed overflow
                    // is undefined, and hipcc folds `(v - target) >= 0` to `target <= v`, which
                    // lets every wait pass on the launch that crosses INT_MAX)
                    ok = (int)((unsigned)v - (unsigned)target) >= 0;
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

// ---- convolution stage ------------------------------------------------------------------------
// MODE 0: a k-group is 16 input channels of one filter tap (Cin % 16 == 0; 3x3 pad 1 or 1x1 pad 0).
// MODE 1: the stem -- 7x7 taps over the channel-padded image (Cin = 4): a k-group is 4 taps x 4
//         channels, lane quad kq takes tap 4g + kq; weights are the padded [64][49][4] copy.
#define B1_FINE(i_)                                                                     \
    do {                                                                                \
        if (fine && threadIdx.x == 0) fine[i_] = __builtin_amdgcn_s_memrealtime();      \
    } while (0)

// One wave's share of a convolution stage: everything that can be known before the stage's inputs
// exist.  Scalars are wave-uniform.  A unit is a 16-row x (16 * nt)-channel output tile (nt = 1 or
// 2: two channel tiles share every activation fragment) times one slice of the reduction index.
// Lane offsets are RELATIVE to the tensors' bases, which ride in the buffer instructions' scalar
// offset: consecutive stages of one shape (the second .. last convolution of a ResNet layer) keep
// the lane part of the plan and only refresh the bases (plan_rebase).
struct ConvPlan {
    int active, wig, grp, wpt, nk, nt;     // unit assigned?  wave in group, group, waves per unit,
                                           // k-groups of this wave, channel tiles per unit
    int ksplit, kj, ticket;                // workgroups per tile, this one's slice, ticket word
    int relu, relu_post, has_add;
    unsigned x_off, w_off, w_off1, y_off, add_off, scale_off, shift_off, slab_off;   // bases (bytes)
    int sb, pi;                            // first k-group of this wave; sub-problem of the stage
    unsigned wrel;                         // this lane's weight row + first k-group (relative)
    unsigned offA[kSU];                    // activation fragment offsets (relative; ~0: zeros)
    unsigned e_rel, e_bn_rel, e_slab_pitch;   // epilogue wave: output element / channel (relative)
    int e_store;
};

__device__ __forceinline__ void plan_clear(ConvPlan& p) {
    p.active = 0; p.wig = 0; p.grp = 0; p.wpt = 16; p.nk = 0; p.nt = 1;
    p.ksplit = 1; p.kj = 0; p.ticket = 0; p.relu = 0; p.relu_post = 0; p.has_add = 0;
    p.x_off = p.w_off = p.w_off1 = p.y_off = p.add_off = p.scale_off = p.shift_off = p.slab_off = 0u;
    p.sb = 0; p.pi = 0; p.wrel = 0u; p.e_rel = 0u; p.e_bn_rel = 0u; p.e_slab_pitch = 0u; p.e_store = 0;
#pragma unroll
    for (int u = 0; u < kSU; ++u) p.offA[u] = 0xFFFFFFFFu;
}

// The weight fragments of a plan: slot of (k-group u, channel tile j) = u * nt + j (host: nk * nt
// <= kSU).  OHWI weights: reduction index tap * Cin + 16 cg = 16 s, i.e. 64 s bytes into the row.
struct ConvW { f32x4 b[kSU]; };          // weight fragments, [k-group][channel tile]; one stage

template <int MODE>
__device__ __forceinline__ void plan_issue_b(const ConvPlan& p, ConvW& cw, const rsrc_t rsW,
                                             const rsrc_t rsP) {
    const rsrc_t rsB = MODE == 1 ? rsW : rsP;
    const int kq = (tid_now() & 63) >> 4;
    const int n = RFL(p.nk) * RFL(p.nt), two = RFL(p.nt) == 2;
    const unsigned w0 = (unsigned)RFL(p.w_off), w1 = (unsigned)RFL(p.w_off1);
#pragma unroll
    for (int i = 0; i < kSU; ++i) {
        if (i < n) {
            const int u = two ? (i >> 1) : i;
            const unsigned base = (two && (i & 1)) ? w1 : w0;
            unsigned off = p.wrel + (unsigned)(u * 64);
            if (MODE == 1 && 4 * (RFL(p.sb) + u) + kq >= 49) off = 0xFFFFFFFFu;
            cw.b[i] = ld_const(rsB, off, base);
        }
    }
}

// the bases of a stage and what its epilogue does: all that changes between stages of one shape
__device__ __forceinline__ void plan_bases(ConvPlan& p, const B1Conv* c) {
    const i32x4* cv = reinterpret_cast<const i32x4*>(c);
    const i32x4 q0 = cv[0], q1 = cv[1], q4 = cv[4], q5 = cv[5], q7 = cv[7];
    p.x_off = (unsigned)RFL(q0[0]); p.y_off = (unsigned)RFL(q0[1]);
    p.add_off = (unsigned)RFL(q0[2]); p.w_off = (unsigned)RFL(q0[3]);
    p.scale_off = (unsigned)RFL(q1[0]); p.shift_off = (unsigned)RFL(q1[1]);
    p.w_off1 = p.w_off + (unsigned)(16 * RFL(q4[2]));          // second channel tile: 16 rows on
    p.relu = RFL(q4[3]); p.relu_post = RFL(q5[0]); p.has_add = RFL(q5[1]);
    p.slab_off = (unsigned)RFL(q7[0]);
}

// The shape part of a plan parks in LDS between stages (registers stay free for the stage bodies):
// lane words as [word][thread] (conflict-free), wave scalars as [wave][16].
constexpr int kPlanLaneWords = kSU + 5;
__device__ __forceinline__ void plan_save(const ConvPlan& p, unsigned* lds_lane, int* lds_wave) {
    const int tid = tid_now(), wave = RFL(tid >> 6);
#pragma unroll
    for (int u = 0; u < kSU; ++u) lds_lane[u * kThreads + tid] = p.offA[u];
    lds_lane[(kSU + 0) * kThreads + tid] = p.wrel;
    lds_lane[(kSU + 1) * kThreads + tid] = p.e_rel;
    lds_lane[(kSU + 2) * kThreads + tid] = p.e_bn_rel;
    lds_lane[(kSU + 3) * kThreads + tid] = p.e_slab_pitch;
    lds_lane[(kSU + 4) * kThreads + tid] = (unsigned)p.e_store;
    if ((tid & 63) == 0) {
        int* w = lds_wave + wave * 16;
        w[0] = p.active; w[1] = p.wig; w[2] = p.grp; w[3] = p.wpt; w[4] = p.nk; w[5] = p.nt;
        w[6] = p.ksplit; w[7] = p.kj; w[8] = p.ticket; w[9] = p.sb; w[10] = p.pi;
    }
}
__device__ __forceinline__ void plan_load(ConvPlan& p, const unsigned* lds_lane, const int* lds_wave) {
    const int tid = tid_now(), wave = RFL(tid >> 6);
    const i32x4* w = reinterpret_cast<const i32x4*>(lds_wave + wave * 16);
    const i32x4 w0 = w[0], w1 = w[1], w2 = w[2];
    plan_clear(p);
    p.active = RFL(w0[0]); p.wig = RFL(w0[1]); p.grp = RFL(w0[2]); p.wpt = RFL(w0[3]);
    p.nk = RFL(w1[0]); p.nt = RFL(w1[1]); p.ksplit = RFL(w1[2]); p.kj = RFL(w1[3]);
    p.ticket = RFL(w2[0]); p.sb = RFL(w2[1]); p.pi = RFL(w2[2]);
    if (!p.active) return;
#pragma unroll
    for (int u = 0; u < kSU; ++u) p.offA[u] = lds_lane[u * kThreads + tid];
    p.wrel = lds_lane[(kSU + 0) * kThreads + tid];
    p.e_rel = lds_lane[(kSU + 1) * kThreads + tid];
    p.e_bn_rel = lds_lane[(kSU + 2) * kThreads + tid];
    p.e_slab_pitch = lds_lane[(kSU + 3) * kThreads + tid];
    p.e_store = (int)lds_lane[(kSU + 4) * kThreads + tid];
}

// Same shape as the previous stage (host flag): take the parked lane part, refresh the bases, put
// the weight fragments in flight.
template <int MODE>
__device__ __forceinline__ void plan_rebase(ConvPlan& p, ConvW& cw, const B1Stage* st,
                                            const rsrc_t rsW, const rsrc_t rsP, const bool defer_b,
                                            const unsigned* lds_lane, const int* lds_wave) {
    plan_load(p, lds_lane, lds_wave);
    if (!RFL(p.active)) return;
    plan_bases(p, &st->c[RFL(p.pi)]);
    if (!defer_b) plan_issue_b<MODE>(p, cw, rsW, rsP);
}

// defer_b: the wave loads its weight fragments later (plan_issue_b): the polling wave after its
// poll (a poll is an in-order vector load and would otherwise wait behind them).
template <int MODE>
__device__ __forceinline__ void plan_conv(ConvPlan& p, ConvW& cw, const B1Stage* st,
                                          const rsrc_t rsW, const rsrc_t rsP, const int nblk,
                                          const bool defer_b) {
    const int lane = tid_now() & 63, wave = RFL(tid_now() >> 6);
    const int r = lane & 15, kq = lane >> 4;
    const i32x4 hd = *reinterpret_cast<const i32x4*>(st);           // type, wpt, nunits0, total
    const int wpt = RFL(hd[1]), nunits0 = RFL(hd[2]), total = RFL(hd[3]);
    const int wshift = wpt == 16 ? 4 : wpt == 8 ? 3 : wpt == 4 ? 2 : 1;
    const int grp = wave >> wshift, wig = wave & (wpt - 1);
    // One pass (host: slots >= units): wave group g of the blocks takes units [g nblk, (g+1) nblk).
    // Its units are dealt to the 8 XCDs in equal contiguous chunks -- every L2 carries an eighth
    // of them, tiles that share weights meet in one L2 -- blocks b and b + 8 sharing an XCD
    // (speed only).
    const int ng = min(nblk, total - grp * nblk);              // units of this wave group
    const int q = (ng + 7) >> 3;                               // ... per XCD
    const int idx = (int)blockIdx.x >> 3, within = ((int)blockIdx.x & 7) * q + idx;
    int t = grp * nblk + within;
    plan_clear(p);
    p.active = idx < q && within < ng;
    p.wig = wig; p.grp = grp; p.wpt = wpt;
    if (!p.active) return;
    const int pi = t >= nunits0 ? 1 : 0;
    if (pi) t -= nunits0;
    p.pi = pi;
    plan_bases(p, &st->c[pi]);
    const i32x4* cv = reinterpret_cast<const i32x4*>(&st->c[pi]);
    const i32x4 q1 = cv[1], q2 = cv[2], q3 = cv[3], q4 = cv[4], q5 = cv[5], q6 = cv[6], q7 = cv[7];
    const int H = RFL(q1[2]), W = RFL(q1[3]);
    const int Cin = RFL(q2[0]), Wo = RFL(q2[1]), Cout = RFL(q2[2]), K = RFL(q2[3]);
    const int stride = RFL(q3[0]), M = RFL(q3[1]), nmt = RFL(q3[2]);
    const int S = RFL(q4[0]), cshift = RFL(q4[1]), krow4 = RFL(q4[2]);
    const unsigned wo_magic = (unsigned)RFL(q5[2]), nmt_magic = (unsigned)RFL(q5[3]);
    const int ksplit = RFL(q6[0]), sper = RFL(q6[1]), per = RFL(q6[2]);
    const unsigned ks_magic = (unsigned)RFL(q6[3]);
    const int nt = RFL(q7[3]);
    p.nt = nt;
    // unit -> (tile, k-slice); tile -> (channel tile, row tile), rows fastest
    const int tile = (int)(((unsigned)t * ks_magic) >> 20), kj = t - tile * ksplit;
    const int ntile = (int)(((unsigned)tile * nmt_magic) >> 20), mt = tile - ntile * nmt;
    const int m0 = mt * 16, n0 = ntile * 16 * nt;
    p.ksplit = ksplit; p.kj = kj; p.ticket = RFL(q7[1]) + tile;
    // ---- this wave's slice of the reduction index
    const int ub = kj * sper, ue = min(S, ub + sper);
    const int sb = ub + wig * per, se = min(ue, sb + per);
    p.nk = max(se - sb, 0);
    p.sb = sb;
    const int m = m0 + r;
    const int oh = (int)(((unsigned)m * wo_magic) >> 20), ow = m - oh * Wo;
    unsigned rowOff = 0u, rowMask = 0u;
    int tapA_v = 0, hb = 0, wb = 0;
    if (MODE == 0) {
        if (K == 3) {          // pad 1: tap (kh, kw) is inside iff 0 <= hb + kh < H, same for w
            hb = oh * stride - 1; wb = ow * stride - 1;
            const unsigned hm = (hb >= 0 ? 1u : 0u) | 2u | (hb + 2 < H ? 4u : 0u);
            const unsigned wm = (wb >= 0 ? 1u : 0u) | 2u | (wb + 2 < W ? 4u : 0u);
            rowMask = ((hm & 1u) ? wm : 0u) | ((hm & 2u) ? wm << 3 : 0u) | ((hm & 4u) ? wm << 6 : 0u);
            const int kh = (lane * 11) >> 5;                     // lane / 3 for lane < 16
            tapA_v = (kh * W + (lane - 3 * kh)) * Cin * 4;
        } else {               // 1x1, pad 0
            hb = oh * stride; wb = ow * stride;
            rowMask = 1u;
        }
        if (m >= M) rowMask = 0u;
        rowOff = (unsigned)(((hb * W + wb) * Cin + kq * 4) * 4);
        p.wrel = (unsigned)((n0 + r) * krow4 + kq * 16 + sb * 64);
    } else {
        hb = oh * 2 - 3; wb = ow * 2 - 3;
        p.wrel = (unsigned)(((n0 + r) * 49 + 4 * sb + kq) * 16);
    }
#pragma unroll
    for (int u = 0; u < kSU; ++u) {
        const int s = sb + u;                                 // wave-uniform
        if (u < p.nk) {
            if (MODE == 0) {
                const int tap = s >> cshift;
                const int cg = s & ((1 << cshift) - 1);
                const unsigned toff = (unsigned)__builtin_amdgcn_readlane(tapA_v, tap) +
                                      (unsigned)(cg * 64);
                p.offA[u] = ((rowMask >> tap) & 1u) ? rowOff + toff : 0xFFFFFFFFu;
            } else {
                const int tap = 4 * s + kq;                   // per lane
                const int kh = (tap * 37) >> 8, kw = tap - 7 * kh;       // tap / 7, tap < 56
                const int h = hb + kh, w = wb + kw;
                const bool ok = tap < 49 && m < M && h >= 0 && w >= 0 && h < H && w < W;
                p.offA[u] = ok ? (unsigned)((h * W + w) * 16) : 0xFFFFFFFFu;
            }
        }
    }
    if (!defer_b) plan_issue_b<MODE>(p, cw, rsW, rsP);
    if (wig == 0) {          // the unit's epilogue wave: output element / channel of this lane
        const int row = lane >> 2, c4 = (lane & 3) * 4;
        const int me = m0 + row;
        p.e_store = me < M;
        p.e_rel = (unsigned)((me * Cout + n0 + c4) * 4);
        p.e_slab_pitch = (unsigned)(nmt * 16 * Cout * 4);
        p.e_bn_rel = (unsigned)((n0 + c4) * 4);
    }
}

template <int WPT>
__device__ __forceinline__ f32x4 sum_partials(const float* rp) {
    constexpr int G = WPT < 8 ? WPT : 8;        // reads in flight at a time
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w0 = 0; w0 < WPT; w0 += G) {
        f32x4 t[G];
#pragma unroll
        for (int w = 0; w < G; ++w) t[w] = *reinterpret_cast<const f32x4*>(rp + (w0 + w) * 512);
        __builtin_amdgcn_sched_barrier(0);      // the reads in flight, then the adds in wave order
#pragma unroll
        for (int w = 0; w < G; ++w) v = (w0 + w == 0) ? t[0] : v + t[w];
        __builtin_amdgcn_sched_barrier(0);
    }
    return v;
}

// what the epilogue wave of a unit fetched for its (up to two) channel tiles
struct ConvEpi { f32x4 add0, add1, sc0, sc1, sh0, sh1; };

// Multiply phase (every wave): activation loads, MFMAs, partial tiles to LDS [wave][nt][16][16].
__device__ __forceinline__ void exec_conv_main(const ConvPlan& p, const ConvW& cw, ConvEpi& ep,
                                               const rsrc_t rsW, float* red, long long* fine) {
    const int lane = tid_now() & 63, wave = RFL(tid_now() >> 6);
    const int r = lane & 15, kq = lane >> 4;
    ep.add0 = ep.add1 = ep.sh0 = ep.sh1 = f32x4{0.f, 0.f, 0.f, 0.f};
    ep.sc0 = ep.sc1 = f32x4{1.f, 1.f, 1.f, 1.f};
    B1_FINE(0);
    const int nk = RFL(p.nk), nt = RFL(p.nt), wig = RFL(p.wig), has_add = RFL(p.has_add);
    const unsigned x_off = (unsigned)RFL(p.x_off), add_off = (unsigned)RFL(p.add_off);
    if (RFL(p.active)) {
        f32x4 av[kSU];
#pragma unroll
        for (int u = 0; u < kSU; ++u)
            if (u < nk) av[u] = ld_act(rsW, p.offA[u], x_off);
        if (wig == 0 && has_add) {
            const unsigned ao = p.e_store ? p.e_rel : 0xFFFFFFFFu;
            ep.add0 = ld_act(rsW, ao, add_off);
            if (nt == 2) ep.add1 = ld_act(rsW, ao, add_off + 64u);
        }
        __builtin_amdgcn_sched_barrier(0);        // every load in flight before the first MFMA
        B1_FINE(1);
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        if (nt == 1) {
#pragma unroll
            for (int u = 0; u < kSU; ++u) {
                if (u < nk) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (u & 1)
                            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][e], cw.b[u][e], acc1, 0, 0, 0);
                        else
                            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][e], cw.b[u][e], acc0, 0, 0, 0);
                    }
                }
            }
            acc0 += acc1;
        } else {
#pragma unroll
            for (int u = 0; u < kSU / 2; ++u) {
                if (u < nk) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][e], cw.b[2 * u][e], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][e], cw.b[2 * u + 1][e], acc1, 0, 0, 0);
                    }
                }
            }
        }
        B1_FINE(2);
        if (wig == 0) {          // folded BatchNorm: lands under the LDS hand-over
            const unsigned sc = (unsigned)RFL(p.scale_off), sh = (unsigned)RFL(p.shift_off);
            ep.sc0 = ld_const(rsW, p.e_bn_rel, sc);
            ep.sh0 = ld_const(rsW, p.e_bn_rel, sh);
            if (nt == 2) {
                ep.sc1 = ld_const(rsW, p.e_bn_rel, sc + 64u);
                ep.sh1 = ld_const(rsW, p.e_bn_rel, sh + 64u);
            }
        }
        // C/D map of 16x16x4: col = lane & 15, row = 4 * (lane >> 4) + i
#pragma unroll
        for (int i = 0; i < 4; ++i) red[wave * 512 + (4 * kq + i) * 16 + r] = acc0[i];
        if (nt == 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) red[wave * 512 + 256 + (4 * kq + i) * 16 + r] = acc1[i];
        }
    }
    B1_FINE(3);
    __syncthreads();
    B1_FINE(4);
}

// Epilogue of a unit (its first wave): partial tiles summed in wave order, split-K combine by
// ticket, folded BatchNorm / ReLU / residual, sc1 store.  Returns true if stores are in flight.
__device__ __forceinline__ bool exec_conv_epilogue(const ConvPlan& p, const ConvEpi& ep,
                                                   const rsrc_t rsW, const float* red, int* sync,
                                                   long long* fine) {
    const int lane = tid_now() & 63;
    const int row = lane >> 2, c4 = (lane & 3) * 4;
    const int nt = RFL(p.nt), wpt = RFL(p.wpt), grp = RFL(p.grp), ksplit = RFL(p.ksplit);
    const int kj = RFL(p.kj), ticket = RFL(p.ticket), relu = RFL(p.relu), relu_post = RFL(p.relu_post);
    const unsigned y_off = (unsigned)RFL(p.y_off), slab_off = (unsigned)RFL(p.slab_off);
    bool stored = false;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (j < nt) {
            const float* rp = red + (grp * wpt) * 512 + j * 256 + row * 16 + c4;
            f32x4 v = wpt == 16 ? sum_partials<16>(rp) : wpt == 8 ? sum_partials<8>(rp)
                      : wpt == 4 ? sum_partials<4>(rp) : sum_partials<2>(rp);
            bool finish = true;
            if (j == 0 && ksplit > 1) {      // (host: split units have one channel tile)
                // park the partial tile, take a ticket; the last arriver sums the slices in order
                if (p.e_store) st_act(rsW, p.e_rel + (unsigned)kj * p.e_slab_pitch, v, slab_off);
                drain_stores();
                int old = 0;
                if (lane == 0)
                    old = __hip_atomic_fetch_add(sync + 9 * 32 + ticket, 1, __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT);
                old = RFL(old);
                finish = old == ksplit - 1;
                if (finish) {
                    if (lane == 0)      // ready for the next launch; nobody else touches it any more
                        __hip_atomic_store(sync + 9 * 32 + ticket, 0, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                    f32x4 sl[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        sl[k] = ld_act(rsW, (k < ksplit && p.e_store)
                                                ? p.e_rel + (unsigned)k * p.e_slab_pitch : 0xFFFFFFFFu,
                                       slab_off);
                    v = sl[0];
#pragma unroll
                    for (int k = 1; k < 4; ++k)
                        if (k < ksplit) v += sl[k];
                }
            }
            if (finish) {
                v = v * (j ? ep.sc1 : ep.sc0) + (j ? ep.sh1 : ep.sh0);
                if (relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                v += j ? ep.add1 : ep.add0;
                if (relu_post) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                if (p.e_store) st_act(rsW, p.e_rel, v, y_off + 64u * j);
                stored = true;
            }
        }
    }
    B1_FINE(5);
    return stored;
}

// ---- uint8 HWC frame -> normalised NHWC4 (preprocess_image, autonomous_drive.py:897-902) -------
// The frame usually sits in PINNED HOST memory (zero-copy): every load is a PCIe read, so what
// counts is the number of transactions.  Each workgroup takes a CONTIGUOUS run of pixels, one wave
// fetches the run's bytes as coalesced dwords (a few 64-byte requests per workgroup; the first
// version read three single bytes per pixel with the pixels interleaved across workgroups: 52,800
// byte-sized host reads, 13 us for the stage), parks them in LDS and the pixel threads pick their
// three bytes from there.
// (the run lives in the kernel's 32 KB `red` array; net.hip checks that it fits)
__device__ __forceinline__ void pre_stage(const B1Stage* st, const B1Launch& a, const rsrc_t rsW,
                                          const int nblk, unsigned* pre_lds) {
    const int npix = RFL(st->pH) * RFL(st->pW);
    const unsigned dst = (unsigned)RFL(st->dst_off);
    const int per = (npix + nblk - 1) / nblk;
    const int p0 = (int)blockIdx.x * per, p1 = min(npix, p0 + per);
    const int b0 = 3 * p0, b1 = 3 * p1;
    unsigned char* bytes = reinterpret_cast<unsigned char*>(pre_lds);
    const int head = (int)((reinterpret_cast<uintptr_t>(a.frame) + (unsigned)b0) & 3);    // bytes before b0 in its dword
    if (p0 < p1) {
        // dwords [b0 - head, ...) of the frame; the last one may reach past b1 but never past the
        // frame's last dword when the frame's end is dword-aligned, else the tail goes byte-wise
        const unsigned* src = reinterpret_cast<const unsigned*>(a.frame + b0 - head);
        const int nd = (b1 - b0 + head) >> 2;
        for (int k = tid_now(); k < nd; k += kThreads) pre_lds[k] = src[k];
        for (int k = 4 * nd + tid_now(); k < b1 - b0 + head; k += kThreads) bytes[k] = a.frame[b0 - head + k];
    }
    __syncthreads();
    bool stored = false;
    for (int i = p0 + tid_now(); i < p1; i += kThreads) {
        const int off = 3 * (i - p0) + head;
        f32x4 v;
        v[0] = ((float)bytes[off + 0] / 255.0f - a.mean[0]) / a.stdv[0];
        v[1] = ((float)bytes[off + 1] / 255.0f - a.mean[1]) / a.stdv[1];
        v[2] = ((float)bytes[off + 2] / 255.0f - a.mean[2]) / a.stdv[2];
        v[3] = 0.f;
        st_act(rsW, dst + (unsigned)i * 16u, v);
        stored = true;
    }
    if (stored) drain_stores();
}

// ---- speed encoder (autonomous_drive.py:371-374, 391) ----------------------------------------
// It depends on the speed alone: the last block evaluates it AFTER it has arrived at the first
// barrier (nobody waits for it: its consumer is 36 stages away), arithmetic order of
// heads_small_pre_kernel (heads_optim.hip).
__device__ __forceinline__ void speed_encoder(const B1Stage* st, const B1Launch& a, const rsrc_t rsW,
                                              const rsrc_t rsP, float* s1) {
    const B1Head* h = &st->h;
    const int tid = tid_now(), lane = tid & 63, wave = RFL(tid >> 6);
    if (tid < 128) {
        const float w0 = ld_const1(rsP, (unsigned)RFL(h->se_w0) + (unsigned)tid * 4u);
        const float b0 = ld_const1(rsP, (unsigned)RFL(h->se_b0) + (unsigned)tid * 4u);
        s1[tid] = fmaxf(fmaf(a.speed[0], w0, 0.f) + b0, 0.f);
    }
    const unsigned w1 = (unsigned)RFL(h->se_w1), b1 = (unsigned)RFL(h->se_b1);
    float wl[8], wh[8], bb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {                  // weights do not wait for layer 1
        const int o = wave * 8 + j;
        wl[j] = ld_const1(rsP, w1 + (unsigned)((o * 128 + lane) * 4));
        wh[j] = ld_const1(rsP, w1 + (unsigned)((o * 128 + lane + 64) * 4));
        bb[j] = ld_const1(rsP, b1 + (unsigned)o * 4u);
    }
    __syncthreads();
    const float xa = s1[lane], xb = s1[lane + 64];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float v = fmaf(xb, wh[j], xa * wl[j]);
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) v += __shfl_xor(v, sft);
        if (lane == 0)
            st_act1(rsW, (unsigned)RFL(h->y_off[0]) + (unsigned)(wave * 8 + j) * 4u,
                    fmaxf(v + bb[j], 0.f));
    }
    drain_stores();
}

// ---- MaxPool2d(3, 2, 1) on NHWC (torch: first maximum in scan order wins) ---------------------
__device__ __forceinline__ void pool_stage(const B1Stage* st, const rsrc_t rsW, const int nblk) {
    const int H = RFL(st->pH), W = RFL(st->pW), C = RFL(st->pC), Ho = RFL(st->pHo), Wo = RFL(st->pWo);
    const unsigned src = (unsigned)RFL(st->src_off), dst = (unsigned)RFL(st->dst_off);
    const int cq = C >> 2;
    const int total = Ho * Wo * cq;
    bool stored = false;
    for (int i = tid_now() * nblk + blockIdx.x; i < total; i += kThreads * nblk) {
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
                                           const rsrc_t rsP, float* hx, const int nblk) {
    const B1Head* h = &st->h;
    const int tid = tid_now(), lane = tid & 63, wave = RFL(tid >> 6);
    const int cmd = RFL(__builtin_bit_cast(int, __builtin_amdgcn_raw_buffer_load_b32(
        rsW, (int)(unsigned)RFL(st->cmd_off), 0, 16)));            // parked by stage 0; -1: out of range
    const int k = cmd < 0 ? 0 : cmd;
    const int first = RFL(h->first), last = RFL(h->last);
    const int in0 = RFL(h->in[0]), in1 = RFL(h->in[1]);
    float* x0 = hx;
    float* x1 = hx + 640;
    if (first) {
        // a command outside 0..3: torch.gather would raise (:397-398); the host reads this word
        if (blockIdx.x == 0 && tid == 0 && cmd < 0) a.status[0] = 1;      // sticky: the host clears it
        const int HW = RFL(h->featHW), C = RFL(h->featC);
        const unsigned fo = (unsigned)RFL(h->feat_off);
        if (tid < (C >> 2)) {              // AdaptiveAvgPool2d(1,1) + Flatten (:369); pixel order
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            for (int p0 = 0; p0 < HW; p0 += 8) {
                f32x4 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j)        // eight loads in flight; past the end: zeros
                    v[j] = ld_act(rsW, p0 + j < HW ? fo + (unsigned)(((p0 + j) * C + tid * 4) * 4)
                                                   : 0xFFFFFFFFu);
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (p0 + j < HW) s += v[j];
            }
            *reinterpret_cast<f32x4*>(x0 + tid * 4) = s / (float)HW;
        } else if (tid >= 256 && tid < 288) {      // speed features, computed in stage 0
            *reinterpret_cast<f32x4*>(x0 + C + (tid - 256) * 4) =
                ld_act(rsW, (unsigned)RFL(h->x_off[0]) + (unsigned)(tid - 256) * 16u);
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
    // the last layer (3 + 1 outputs) stays in block 0, so that its thread 0 can post the tick's
    // completion word right behind the outputs
    const int slot0 = last ? (blockIdx.x == 0 ? wave : out0 + out1) : wave * nblk + (int)blockIdx.x;
    const int slot_step = last ? out0 + out1 : 16 * nblk;
    for (int slot = slot0; slot < out0 + out1; slot += slot_step) {
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
    if (last && blockIdx.x == 0 && a.done != nullptr) {
        // Completion word for a host that spins on pinned memory instead of waiting for the
        // stream: the four outputs are drained first (they may be in HOST memory: stores from
        // four waves), then ONE store of this tick's sequence number.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            // (writes of one device to one host buffer arrive in order)
            __hip_atomic_store(a.done, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

__device__ __forceinline__ unsigned desc_stage_cmd_off(const B1Stage* table) { return table[0].cmd_off; }
__device__ __forceinline__ const B1Stage* desc_stage(const int* desc, const int s) {
    return reinterpret_cast<const B1Stage*>(desc + s * kDescInts);
}

__global__ __launch_bounds__(kThreads) void infer_b1_kernel(const B1Launch a) {
    __shared__ __attribute__((aligned(16))) int desc[kB1MaxStages * kDescInts];
    __shared__ __attribute__((aligned(16))) float red[16 * 512];
    __shared__ __attribute__((aligned(16))) float hx[2 * 640];
    __shared__ float s1[128];
    __shared__ unsigned plan_lane[kPlanLaneWords * kThreads];
    __shared__ __attribute__((aligned(16))) int plan_wave[16 * 16];
    __shared__ int fail;
    const int tid = threadIdx.x;
    const int nblk = gridDim.x;
    const rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)a.ws, 0, (int)(unsigned)a.ws_bytes,
                                                         0x00020000);
    const rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc((void*)a.params, 0,
                                                         (int)(unsigned)a.param_bytes, 0x00020000);
    const int nstages = a.nstages;
    // epoch base of the monotonic counters: advanced by block 0 at the end of the previous launch
    int base;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(base) : "v"(a.sync + 8 * 32) : "memory");
    base = RFL(base);
    for (int i = tid; i < nstages * kDescInts; i += kThreads)      // the stage table, once
        desc[i] = reinterpret_cast<const int*>(a.table)[i];
    if (tid == 0) fail = 0;
    // the command may live in pinned HOST memory (zero-copy control loop): read it once, here, and
    // park it next to the activations -- the three head stages then do not cross PCIe
    if (blockIdx.x == 0 && tid == 0) {
        const long long c = a.cmd[0];
        st_act1(rsW, (unsigned)desc_stage_cmd_off(a.table),
                __builtin_bit_cast(float, (c < 0 || c > 3) ? -1 : (int)c));
        drain_stores();
    }
    __syncthreads();
    const int per_shard = nblk >> 3;
    const int first = a.first_stage;
    const bool poller = RFL(tid >> 6) == 0;            // wave 0 (a scalar: no divergent plans)
    // Stage 0 of a whole-frame launch is the preprocess stage: it runs OUTSIDE the stage loop so that
    // its code takes no part in the loop's register allocation (inside the loop's switch it cost
    // the convolution stages 4 more spilled VGPRs and 10 us per frame).
    int s_begin = first;
    if (first == 0 && RFL(desc_stage(desc, 0)->type) == B1_PRE) {
        if (a.stamps && blockIdx.x == 0 && tid == 0) a.stamps[0] = __builtin_amdgcn_s_memrealtime();
        pre_stage(desc_stage(desc, 0), a, rsW, nblk, reinterpret_cast<unsigned*>(red));
        if (a.stamps && blockIdx.x == 0 && tid == 0)
            a.stamps[kB1MaxStages + 1] = __builtin_amdgcn_s_memrealtime();
        if (a.stamps && tid == 0)
            a.stamps[10 * (kB1MaxStages + 1) + blockIdx.x] = __builtin_amdgcn_s_memrealtime();
        if (1 < nstages) {
            grid_arrive(a.sync);
            if ((int)blockIdx.x == nblk - 1) speed_encoder(desc_stage(desc, 0), a, rsW, rsP, s1);
        }
        s_begin = 1;
    }
    for (int s = s_begin; s < nstages; ++s) {
        const B1Stage* st = desc_stage(desc, s);
        const int type = RFL(st->type);
        int failed = RFL(*reinterpret_cast<volatile int*>(&fail));     // wave-uniform
        // The plan of this stage (addresses, weight fragments in flight) is made between this
        // workgroup's arrival at the barrier (end of the previous iteration) and its wait: while
        // the slowest workgroup is still on its way.
        const bool defer = poller && s > first;
        ConvPlan plan;
        ConvW cw;
#pragma unroll
        for (int u = 0; u < kSU; ++u) cw.b[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!failed && type == B1_CONV) {
            if (RFL(st->same_shape)) {
                plan_rebase<0>(plan, cw, st, rsW, rsP, defer, plan_lane, plan_wave);
            } else {
                plan_conv<0>(plan, cw, st, rsW, rsP, nblk, defer);
                if (s + 1 < nstages && RFL(desc_stage(desc, s + 1)->same_shape))
                    plan_save(plan, plan_lane, plan_wave);     // (each wave reads back its own words)
            }
        } else if (!failed && type == B1_STEM) {
            plan_conv<1>(plan, cw, st, rsW, rsP, nblk, defer);
        } else {
            plan_clear(plan);
        }
        if (s > first) {
            grid_wait(a.sync, (int)((unsigned)base + (unsigned)((s - first) * per_shard)), &fail, a.status);
            failed = RFL(*reinterpret_cast<volatile int*>(&fail));
            if (defer && !failed) {
                if (type == B1_CONV) plan_issue_b<0>(plan, cw, rsW, rsP);
                else if (type == B1_STEM) plan_issue_b<1>(plan, cw, rsW, rsP);
            }
        }
        if (a.stamps && blockIdx.x == 0 && tid == 0) a.stamps[s] = __builtin_amdgcn_s_memrealtime();
        long long* fine = (a.stamps && blockIdx.x == 0)
                              ? a.stamps + 2 * (kB1MaxStages + 1) + 8 * s : nullptr;
        if (!failed && (type == B1_CONV || type == B1_STEM)) {
            ConvEpi ep;
            exec_conv_main(plan, cw, ep, rsW, red, fine);
            if (RFL(plan.active) && RFL(plan.wig) == 0) {
                if (exec_conv_epilogue(plan, ep, rsW, red, a.sync, fine)) drain_stores();
            }
        } else if (!failed) {
            if (type == B1_POOL) pool_stage(st, rsW, nblk);
            else head_stage(st, a, rsW, rsP, hx, nblk);     // (these drain their own stores)
        }
        if (a.stamps && blockIdx.x == 0 && tid == 0)
            a.stamps[kB1MaxStages + 1 + s] = __builtin_amdgcn_s_memrealtime();
        if (a.stamps && tid == 0)      // diagnostics: when each workgroup was done with the stage
            a.stamps[10 * (kB1MaxStages + 1) + s * nblk + blockIdx.x] = __builtin_amdgcn_s_memrealtime();
        if (s + 1 < nstages) {
            grid_arrive(a.sync);
            if (s == first && (int)blockIdx.x == nblk - 1)
                speed_encoder(desc_stage(desc, 0), a, rsW, rsP, s1);
        }
    }
    if (blockIdx.x == 0 && tid == 0) {
        __hip_atomic_store(a.sync + 8 * 32, (int)((unsigned)base + (unsigned)((nstages - 1 - first) * per_shard)), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        if (*reinterpret_cast<volatile int*>(&fail)) {
            const float nan = __builtin_nanf("");
            a.controls[0] = nan; a.controls[1] = nan; a.controls[2] = nan; a.pred_speed[0] = nan;
            if (a.done != nullptr) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(a.done, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
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
    CILRS_CHECK(a.nstages >= 2 && a.nstages <= kB1MaxStages && (a.first_stage == 0 || a.first_stage == 1),
                "infer_b1: bad stage count");
    CILRS_CHECK(a.ws_bytes < (1ull << 32) && a.param_bytes < (1ull << 32),
                "infer_b1: arenas must be addressable with 32-bit offsets");
    infer_b1_kernel<<<blocks, kThreads, 0, s>>>(a);
    CILRS_LAUNCH_CHECK();
    return 0;
}

}  // namespace cilrs
