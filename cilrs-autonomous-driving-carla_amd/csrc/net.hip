// Network plan + C-ABI of the CILRS engine.
//
// The plan is the MI355X-side restatement of the graph the reference builds in
// CILRS.__init__/forward (model/autonomous_drive.py:361-399): ResNet-34 trunk (torchvision
// BasicBlock stacks [3,4,6,3]) -> 512-d feature; speed encoder 1->128->128; concat 640; four
// command branches 640->256->256->3 (all evaluated, then gathered by `command`); speed head
// 512->256->256->1.  Parameters live in ONE flat fp32 arena in nn.Module.parameters() order
// (conv weights OHWI); gradients / Adam moments use the same layout, so clip + Adam are single
// launches and data-parallel all-reduce works on contiguous ranges.
#include "common.h"
#include "../../include/cilrs_hip.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <string>
#include <vector>

namespace cilrs {

static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char* last_error() { return g_err; }

static int current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    return dev;
}
int device_cus() {
    static int cus[64] = {};
    const int dev = current_device();
    if (cus[dev] == 0) {
        hipDeviceProp_t p;
        cus[dev] = (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0)
                       ? p.multiProcessorCount : 256;
    }
    return cus[dev];
}
bool once_per_device(const void* key) {
    static std::map<std::pair<const void*, int>, bool> seen;
    return seen.emplace(std::make_pair(key, current_device()), true).second;
}

#ifdef CILRS_EXPERIMENTS
int experiment_env(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}
#endif

// CILRS_PRIO: static wave priorities of the GEMM kernels (common.h wave_priority)
int wave_priority_mode() {
    static const int m = experiment_env("CILRS_PRIO", 0);
    return m;
}

namespace {

// ------------------------------------------------------------------------------------------------
// Architecture: parameter arena layout
// ------------------------------------------------------------------------------------------------
struct ParamT { std::string name; size_t off, numel; int ndim; int shape[4]; };
struct BnT { std::string prefix; int C; size_t gamma, beta, rm, rv; };
struct ConvT { int cin, cout, k, stride, pad; size_t w; int bn; int group; };
struct BlockT { int conv1, conv2, conv3, down; };   // conv3 = -1: BasicBlock; else Bottleneck
struct LinT { int in, out; size_t w, b; };

// variant 0: the reference's network -- ResNet-34 trunk (BasicBlock [3,4,6,3]), 512-d features
//            (model/autonomous_drive.py:365-370)
// variant 1: BASELINE.json configs[3], "ResNet-50 backbone variant": Bottleneck [3,4,6,3] with the
//            stride on the 3x3 convolution (torchvision's ResNet-50 v1.5), 2048-d features into the
//            same heads.  The reference has no such model; parity is against the build's own CPU
//            restatement (oracle/resnet50_oracle.py).  Trains in fp32 through the same kernels;
//            the 16-bit trunks are inference-only.
struct Arch {
    int variant = 0;                    // trunk: 0 ResNet-34 (the reference), 1 ResNet-50 variant
    int ncmd = 4;                       // control branches = commands (autonomous_drive.py:362)
    int feat = 512;                     // trunk feature width (avg-pool output)
    std::vector<ParamT> params;
    std::vector<BnT> bns;
    std::vector<ConvT> convs;           // convs[0] = stem
    std::vector<BlockT> blocks;
    LinT se0, se3, br[kMaxCmd][3], sp0, sp3, sp5;
    size_t arena_floats = 0, count = 0, bn_floats = 0;
    size_t seg_begin[6], seg_end[6];    // 0 heads, 1 layer4, 2 layer3, 3 layer2, 4 layer1, 5 stem

    size_t add_param(const std::string& name, int ndim, int s0, int s1 = 1, int s2 = 1,
                     int s3 = 1) {
        ParamT p;
        p.name = name;
        p.ndim = ndim;
        p.shape[0] = s0; p.shape[1] = s1; p.shape[2] = s2; p.shape[3] = s3;
        p.numel = (size_t)s0 * s1 * s2 * s3;
        p.off = arena_floats;
        arena_floats += (p.numel + 3) / 4 * 4;      // keep every tensor 16-byte aligned
        count += p.numel;
        params.push_back(p);
        return p.off;
    }
    int add_bn(const std::string& prefix, int C) {
        BnT b;
        b.prefix = prefix;
        b.C = C;
        b.gamma = add_param(prefix + ".weight", 1, C);
        b.beta = add_param(prefix + ".bias", 1, C);
        b.rm = bn_floats;
        b.rv = bn_floats + C;
        bn_floats += 2 * (size_t)C;
        bns.push_back(b);
        return (int)bns.size() - 1;
    }
    int add_conv(const std::string& wname, const std::string& bnprefix, int cin, int cout, int k,
                 int stride, int pad, int group) {
        ConvT c;
        c.cin = cin; c.cout = cout; c.k = k; c.stride = stride; c.pad = pad; c.group = group;
        c.w = add_param(wname, 4, cout, cin, k, k);          // logical OIHW, stored OHWI
        c.bn = add_bn(bnprefix, cout);
        convs.push_back(c);
        return (int)convs.size() - 1;
    }
    LinT add_lin(const std::string& prefix, int in, int out) {
        LinT l;
        l.in = in; l.out = out;
        l.w = add_param(prefix + ".weight", 2, out, in);
        l.b = add_param(prefix + ".bias", 1, out);
        return l;
    }

    Arch(int variant_, int ncmd_) : variant(variant_), ncmd(ncmd_) {
        const size_t stem_begin = arena_floats;
        add_conv("visual_encoder.0.weight", "visual_encoder.1", 3, 64, 7, 2, 3, 0);
        size_t layer_begin[5];
        const int nblk[4] = {3, 4, 6, 3};
        const int width[4] = {64, 128, 256, 512};
        const int expansion = variant == 1 ? 4 : 1;
        int inpl = 64;
        for (int L = 0; L < 4; ++L) {
            layer_begin[L] = arena_floats;
            for (int b = 0; b < nblk[L]; ++b) {
                const std::string p =
                    "visual_encoder." + std::to_string(4 + L) + "." + std::to_string(b);
                const int stride = (b == 0 && L > 0) ? 2 : 1;
                const int outpl = width[L] * expansion;
                BlockT blk;
                if (variant == 1) {     // Bottleneck: 1x1 -> 3x3 (stride) -> 1x1 (x4)
                    blk.conv1 = add_conv(p + ".conv1.weight", p + ".bn1", inpl, width[L], 1, 1, 0,
                                         L + 1);
                    blk.conv2 = add_conv(p + ".conv2.weight", p + ".bn2", width[L], width[L], 3,
                                         stride, 1, L + 1);
                    blk.conv3 = add_conv(p + ".conv3.weight", p + ".bn3", width[L], outpl, 1, 1, 0,
                                         L + 1);
                } else {                // BasicBlock: 3x3 (stride) -> 3x3
                    blk.conv1 = add_conv(p + ".conv1.weight", p + ".bn1", inpl, width[L], 3,
                                         stride, 1, L + 1);
                    blk.conv2 = add_conv(p + ".conv2.weight", p + ".bn2", width[L], width[L], 3, 1,
                                         1, L + 1);
                    blk.conv3 = -1;
                }
                blk.down = -1;
                if (stride != 1 || inpl != outpl)
                    blk.down = add_conv(p + ".downsample.0.weight", p + ".downsample.1", inpl,
                                        outpl, 1, stride, 0, L + 1);
                blocks.push_back(blk);
                inpl = outpl;
            }
        }
        feat = inpl;
        layer_begin[4] = arena_floats;
        se0 = add_lin("speed_encoder.0", 1, 128);
        se3 = add_lin("speed_encoder.3", 128, 128);
        for (int k = 0; k < ncmd; ++k) {
            const std::string p = "control_branches." + std::to_string(k);
            br[k][0] = add_lin(p + ".0", feat + 128, 256);
            br[k][1] = add_lin(p + ".3", 256, 256);
            br[k][2] = add_lin(p + ".6", 256, 3);
        }
        sp0 = add_lin("speed_predictor.0", feat, 256);
        sp3 = add_lin("speed_predictor.3", 256, 256);
        sp5 = add_lin("speed_predictor.5", 256, 1);
        seg_begin[0] = layer_begin[4]; seg_end[0] = arena_floats;
        for (int L = 0; L < 4; ++L) {          // seg 1 = layer4 ... seg 4 = layer1
            seg_begin[4 - L] = layer_begin[L];
            seg_end[4 - L] = layer_begin[L + 1];
        }
        seg_begin[5] = stem_begin; seg_end[5] = layer_begin[0];
    }
};

constexpr int kNumVariants = 2;
// variant code = trunk | num_commands << 8 (0 in the upper bits = the reference's 4)
inline int code_trunk(int code) { return code & 0xff; }
inline int code_ncmd(int code) { return (code >> 8) == 0 ? 4 : (code >> 8); }
const Arch& arch(int code = 0) {
    static std::map<int, Arch*> cache;
    const int key = code_trunk(code) | (code_ncmd(code) << 8);
    auto it = cache.find(key);
    if (it == cache.end()) it = cache.emplace(key, new Arch(code_trunk(code), code_ncmd(code))).first;
    return *it->second;
}

const char* kGroupName[5] = {"stem", "layer1", "layer2", "layer3", "layer4"};

// ------------------------------------------------------------------------------------------------
// Per-kernel timing
// ------------------------------------------------------------------------------------------------
struct ProfRec { int label; hipEvent_t e0, e1; double flops, bytes; };
struct ProfAgg { long long calls = 0; double ms = 0, flops = 0, bytes = 0; };

struct Prof {
    bool on = false;
    std::vector<std::string> labels;
    std::map<std::string, int> index;
    std::vector<ProfAgg> agg;
    std::vector<ProfRec> pending;
    std::vector<hipEvent_t> pool;

    int label_id(const std::string& s) {
        auto it = index.find(s);
        if (it != index.end()) return it->second;
        const int id = (int)labels.size();
        labels.push_back(s);
        agg.push_back(ProfAgg());
        index[s] = id;
        return id;
    }
    hipEvent_t get_event() {
        if (!pool.empty()) {
            hipEvent_t e = pool.back();
            pool.pop_back();
            return e;
        }
        // timing events: no system-scope release at the record (hipEventDisableSystemFence is
        // meant for exactly this: the cache write-back / invalidate between two kernels is not
        // part of either kernel, and the unprofiled step does not have it)
        hipEvent_t e;
        const unsigned flags = experiment_env("CILRS_PROF_NOFENCE", 1) ? hipEventDisableSystemFence : 0u;
        if (hipEventCreateWithFlags(&e, flags) != hipSuccess) return nullptr;
        return e;
    }
    int collect() {
        for (auto& r : pending) {
            CILRS_HIP(hipEventSynchronize(r.e1));
            float ms = 0.f;
            CILRS_HIP(hipEventElapsedTime(&ms, r.e0, r.e1));
            ProfAgg& a = agg[r.label];
            a.calls += 1; a.ms += ms; a.flops += r.flops; a.bytes += r.bytes;
            pool.push_back(r.e0);
            pool.push_back(r.e1);
        }
        pending.clear();
        return 0;
    }
    ~Prof() {
        for (auto& r : pending) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
        for (auto e : pool) (void)hipEventDestroy(e);
    }
};

}  // namespace
}  // namespace cilrs

using namespace cilrs;

// ------------------------------------------------------------------------------------------------
// The plan
// ------------------------------------------------------------------------------------------------
struct ConvG { int H, W, Ho, Wo, M; size_t y, z, stats; };
constexpr int kTileCounters = 16384;

// dy ring: bn_bwd writes each conv's output gradient into the next ring slot; the weight-gradient
// GEMM that consumes it runs on side stream 0 and may lag the data-gradient chain by up to
// depth - 1 convolutions.  Measured on MI355X (tools/overlap_sweep.sh): depth 2/4/8 all give
// 14.58-14.65 ms/step, and a LOW-PRIORITY side stream (CILRS_SIDE_PRIO=1) starves the weight
// gradients outright (25.3 ms/step) -- so the defaults are depth 2, default priority.
constexpr int kDyRing = 8;
constexpr int kNumG = 5 + kDyRing - 2;
constexpr int kRingIdx[kDyRing] = {0, 4, 5, 6, 7, 8, 9, 10};
static int dy_ring_depth() {
    static const int d = experiment_env("CILRS_DY_RING", 2);
    return d < 2 ? 2 : d > kDyRing ? kDyRing : d;
}

struct cilrs_net {
    const Arch* A = nullptr;               // architecture variant of this plan
    const cilrs_adam_args* fused_adam = nullptr;   // set for the duration of cilrs_net_backward_step
    int B, H, W;
    std::vector<ConvG> cg;                 // geometry + workspace offsets (floats) per conv
    int H0, W0, H1, W1;                    // stem conv out, maxpool out
    int featHW;
    // workspace offsets (in floats unless noted)
    size_t x4, w4, pool, argmax_b /*bytes offset*/, combined, s1, p1, p2, h1[kMaxCmd], h2[kMaxCmd], all_out;
    size_t dcombined, ds1, dp1, dp2, dh1[kMaxCmd], dh2[kMaxCmd], dcomb_part[kMaxCmd + 1], d_all, speed_in,
        cmd_b /*bytes offset*/;
    size_t tile_cnt;                       // split-K ticket counters (ints), then 2 x kBnSyncInts ints:
                                           // forward / backward finalize-in-apply counters
    BnSync bn_sync[2] = {{nullptr, 0}, {nullptr, 0}};
    const void* cnt_zeroed_for = nullptr;  // workspace whose counters have been zeroed
    size_t G[kNumG];                       // gradient buffers: [1..3] fixed roles, the rest = dy ring
    size_t gmax;
    size_t bn_partial, bn_partial2, bn_coef, slabs, slabs_floats, ksplit, ksplit_floats, status_b;
    size_t ws_bytes;
    float* ws_base = nullptr;             // workspace of the current call (set by every entry)
    // side streams: independent kernels (weight-gradient vs data-gradient GEMMs, the five head
    // chains) run concurrently so one launch's tail fills with another's blocks
    bool overlap = true;
    bool streams_ready = false;
    hipStream_t side[1];                   // weight-gradient stream
    hipEvent_t fork_ev, gbuf_ev[kNumG], wprep_ev, branch_ev, heads_ev;
    bool heads_pending = false;            // the heads' weight gradients of this backward pass are on the side stream
    int side_branch = 0;                   // conv_fwd / conv_fwd16 are building a down-sample branch: no split-K scratch;
                                           // 1 = on the side stream, with column-partial scratch of its own
    bool wprep_pending = false;            // this step's weight images are being built on the side stream
    bool gbuf_pending[kNumG] = {};
    int dy_pos = 0;
    int bwd_nblk_next = 0;                 // fused BN-backward partials waiting for their BN
    BnEvalTable bn_table;
    FoldF16Table f16_table;                // fp16 inference: folded weights / biases / activations
    size_t f16_w, f16_bias, f16_act[5], f16_act_floats;
    size_t stem16_w = 0, stem16_b = 0;     // folded 16-bit stem weights [64][7][8][4] / fp32 shift
    // cached hipGraph of the uint8 inference path (fixed pointers)
    hipGraphExec_t graph_exec = nullptr;
    int graph_half = 0;
    const void* graph_key[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                nullptr};
    bool warmed = false;
    // Weight-derived inference state cached across calls: eval-mode BatchNorm scale/shift of every
    // layer + the channel-padded stem weights (prep_key), and the 16-bit folded weights / biases
    // (fold_key, fold_half).  Valid while the caller's weights key (cilrs_net_set_weights_key:
    // "parameters and BN buffers are unchanged while this value stays the same"; 0 = unknown,
    // never cache) and the buffers are the ones the state was computed from.
    uint64_t weights_key = 0, prep_key = 0, fold_key = 0, graph_wkey = 0;
    int fold_half = 0;
    const void* prep_bufs[3] = {nullptr, nullptr, nullptr};     // params, bn_running, workspace
    bool trained_fwd = false;
    float last_dropout = 0.f;
    // bf16 training mode (CILRS_PLAN_BF16_TRAIN): the trunk convolutions after the stem multiply
    // bf16 operands on v_mfma_f32_32x32x16_bf16 (fp32 accumulation, fp32 results); everything
    // else -- BatchNorm, residual adds, the stem, the heads, loss, Adam, master weights -- stays
    // fp32.  16-bit shadows (offsets in floats): whole parameter arena, transposed flipped conv
    // weights, every post-BN activation, the max-pool output, the dy ring.
    bool bf16_train = false;
    size_t w16_all = 0, wT16 = 0, pool16 = 0, slabs16 = 0, slabs16_floats = 0;
    size_t z16[kMaxConvs] = {}, wT16_off[kMaxConvs] = {}, G16[kNumG] = {};
    TransposeF16Table tr_table;
    // Winograd F(2x2, 3x3) (conv_wino.hip) for the stride-1 3x3 convolutions of fp32 train plans
    // where it beats the implicit GEMM (layers of up to 256 channels with enough blocks to fill
    // the chip; CILRS_WINO=0 turns it off): forward + data gradient.  Both transformed-filter
    // images of every such layer live in the workspace and are rebuilt by ONE launch at the top
    // of each training forward (the weights change every step).
    bool wino_on[kMaxConvs] = {};
    size_t wino_base = 0;
    WinoWeightTable wino_table;
    // persistent single-frame kernel (infer_b1.hip): stage table + barrier counters in the
    // workspace (offsets in floats; 0 = this plan has none), uploaded once per workspace
    size_t b1_table = 0, b1_sync = 0, b1_stamps = 0, b1_slabs = 0, b1_slab_floats = 0, b1_cmd = 0;
    std::vector<B1Stage> b1_host;
    int b1_blocks = -1;                    // resident grid (one workgroup per CU); -1 = not asked yet
    const void* b1_ready_for = nullptr;
    Prof prof;
};

namespace {

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Bump {
    size_t off = 0;   // floats
    size_t take(size_t floats) {
        const size_t o = off;
        off = align_up(off + floats, 64);
        return o;
    }
};

#define RUN(NET_, LBL_, FL_, BY_, ST_, CALL_)                                         \
    do {                                                                              \
        Prof& pr_ = (NET_)->prof;                                                     \
        if (pr_.on) {                                                                 \
            ProfRec r_;                                                               \
            r_.label = pr_.label_id(LBL_);                                            \
            r_.flops = (FL_); r_.bytes = (BY_);                                       \
            r_.e0 = pr_.get_event(); r_.e1 = pr_.get_event();                         \
            CILRS_CHECK(r_.e0 && r_.e1, "hipEventCreate failed");                     \
            CILRS_HIP(hipEventRecord(r_.e0, (ST_)));                                  \
            if (CALL_) return 1;                                                      \
            CILRS_HIP(hipEventRecord(r_.e1, (ST_)));                                  \
            pr_.pending.push_back(r_);                                                \
        } else {                                                                      \
            if (CALL_) return 1;                                                      \
        }                                                                             \
    } while (0)

static BnSync* bn_sync(cilrs_net* net, float* ws, int bwd);
// split-K ticket counters start at zero; every reducer block re-zeroes its own afterwards
// counters of the finalize-inside-apply BatchNorm launches (bn_pool.hip).  OFF by default: measured
// at B=128 the in-launch hand-off (partials -> coefficients -> sc1 publish -> counter -> poll ->
// sc1 read: four memory-side round trips) costs MORE than the 5.4 us finalize launch + its gap --
// BatchNorm 1.33 ms/step with the separate launches, 1.70 ms fused (profiles/r03_bn_fused.log).
// CILRS_BN_FUSED=1 selects it (tests/test_model_gpu.py runs one step that way).
static BnSync* bn_sync(cilrs_net* net, float* ws, int bwd) {
    static const int on = getenv("CILRS_BN_FUSED") ? atoi(getenv("CILRS_BN_FUSED")) : 0;
    if (!on) return nullptr;
    net->bn_sync[bwd].dev = reinterpret_cast<int*>(ws + net->tile_cnt) + kTileCounters + kBnSyncInts * bwd;
    return &net->bn_sync[bwd];
}
int zero_counters_once(cilrs_net* net, void* workspace, hipStream_t s) {
    if (net->cnt_zeroed_for == workspace) return 0;
    float* ws = reinterpret_cast<float*>(workspace);
    CILRS_HIP(hipMemsetAsync(ws + net->tile_cnt, 0, (kTileCounters + 2 * kBnSyncInts) * sizeof(int), s));
    // the int32[4] status words start at zero in every workspace (a C-ABI caller brings its own,
    // uninitialised one); afterwards the kernels only ever SET them: "since the caller last cleared"
    CILRS_HIP(hipMemsetAsync(reinterpret_cast<char*>(workspace) + net->status_b, 0, 16, s));
    net->bn_sync[0].total = net->bn_sync[1].total = 0;
    net->cnt_zeroed_for = workspace;
    return 0;
}

// Events that only order one stream of this device behind another (hipStreamWaitEvent; the host
// never inspects them).  hipEventDisableSystemFence: by default a HIP event performs a SYSTEM-scope
// release when it is recorded -- a cache write-back / invalidate so that the HOST may synchronise
// with it -- ~70 times per backward pass here (one fork and one join per weight gradient).  Without
// it the step is 0.13 ms shorter (9.37 -> 9.25 ms, tools/ab_env.sh CILRS_EVENT_NOFENCE 0 1); the
// device-side ordering of hipStreamWaitEvent is not affected (the kernels' own agent-scope
// releases make their results visible to the other stream).
unsigned stream_event_flags() {
    static const int nofence = experiment_env("CILRS_EVENT_NOFENCE", 1);
    return hipEventDisableTiming | (nofence ? hipEventDisableSystemFence : 0u);
}
int ensure_streams(cilrs_net* net) {
    if (net->streams_ready) return 0;
    int prio_least = 0, prio_greatest = 0;
    CILRS_HIP(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
    // side stream 0 carries the weight-gradient GEMMs (CILRS_SIDE_PRIO=1: lowest priority -- an
    // experiment that starves them, see kDyRing)
    static const int side_prio = experiment_env("CILRS_SIDE_PRIO", 0);
    // experiment (CILRS_SIDE_CUS=n, CILRS_SIDE_CU_MODE=0|1): confine the weight-gradient stream to n
    // CUs (mode 0: CU ids 0..n-1; mode 1: ids with (id % 32) < n / 8, i.e. n / 8 per group of 32)
    static const int side_cus = experiment_env("CILRS_SIDE_CUS", 0);
    static const int side_cu_mode = experiment_env("CILRS_SIDE_CU_MODE", 0);
    if (side_cus > 0) {
        uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int id = 0; id < 256; ++id) {
            const bool on = side_cu_mode == 0 ? id < side_cus : (id % 32) < side_cus / 8;
            if (on) mask[id >> 5] |= 1u << (id & 31);
        }
        CILRS_HIP(hipExtStreamCreateWithCUMask(&net->side[0], 8, mask));
    } else if (side_prio)
        CILRS_HIP(hipStreamCreateWithPriority(&net->side[0], hipStreamNonBlocking, prio_least));
    else
        CILRS_HIP(hipStreamCreateWithFlags(&net->side[0], hipStreamNonBlocking));
    for (int i = 0; i < kNumG; ++i)
        CILRS_HIP(hipEventCreateWithFlags(&net->gbuf_ev[i], stream_event_flags()));
    CILRS_HIP(hipEventCreateWithFlags(&net->fork_ev, stream_event_flags()));
    CILRS_HIP(hipEventCreateWithFlags(&net->wprep_ev, stream_event_flags()));
    CILRS_HIP(hipEventCreateWithFlags(&net->branch_ev, stream_event_flags()));
    CILRS_HIP(hipEventCreateWithFlags(&net->heads_ev, stream_event_flags()));
    net->streams_ready = true;
    return 0;
}
// concurrency is switched off while per-kernel timing is on (serial brackets are meaningful)
// (the CONFIGURED state: what an unprofiled step of this plan does)
bool overlap_configured(cilrs_net* net) {
    // CILRS_OVERLAP=0 serialises everything on the caller's stream (rocprofv3 per-kernel durations
    // then match the hipEvent brackets of the profile mode)
    static const int env = getenv("CILRS_OVERLAP") ? atoi(getenv("CILRS_OVERLAP")) : 1;
    return env != 0 && net->overlap;
}
bool use_overlap(cilrs_net* net) { return overlap_configured(net) && !net->prof.on; }
hipStream_t side_or(cilrs_net* net, hipStream_t main, int i) {
    return use_overlap(net) ? net->side[i] : main;
}
// gradient buffer `gi` is about to be overwritten on `main`: wait for a side-stream reader
int gbuf_acquire(cilrs_net* net, hipStream_t main, int gi) {
    if (net->gbuf_pending[gi]) {
        CILRS_HIP(hipStreamWaitEvent(main, net->gbuf_ev[gi], 0));
        net->gbuf_pending[gi] = false;
    }
    return 0;
}
// side stream 0 will read gradient buffer `gi` (just produced on `main`)
int gbuf_side_begin(cilrs_net* net, hipStream_t main) {
    if (!use_overlap(net)) return 0;
    if (ensure_streams(net)) return 1;
    CILRS_HIP(hipEventRecord(net->fork_ev, main));
    CILRS_HIP(hipStreamWaitEvent(net->side[0], net->fork_ev, 0));
    return 0;
}
int gbuf_side_end(cilrs_net* net, int gi) {
    if (!use_overlap(net)) return 0;
    CILRS_HIP(hipEventRecord(net->gbuf_ev[gi], net->side[0]));
    net->gbuf_pending[gi] = true;
    return 0;
}
int gbuf_join_all(cilrs_net* net, hipStream_t main) {
    for (int gi = 0; gi < kNumG; ++gi)
        if (gbuf_acquire(net, main, gi)) return 1;
    return 0;
}

int conv_fwd(cilrs_net* net, const ConvT& c, const ConvG& g, const float* x, int x_cin,
             const float* w, float* y, float* ws, hipStream_t s, int* bn_nblk = nullptr,
             const float* fold_stats = nullptr, int relu = 0, const float* addend = nullptr,
             int relu_post = 0) {
    const int ci = (int)(&g - &net->cg[0]);
    if (bn_nblk && !fold_stats && net->wino_on[ci]) {      // training forward on the Winograd kernel
        WinoArgs wa;
        memset(&wa, 0, sizeof(wa));
        int e = 0;
        while (net->wino_table.w[e] != (unsigned)c.w) ++e;
        wa.x = x; wa.U = ws + net->wino_base + net->wino_table.u[e]; wa.y = y;
        wa.N = net->B; wa.H = g.H; wa.W = g.W; wa.C = c.cin; wa.K = c.cout;
        wa.bn_partial = ws + net->bn_partial;
        wa.slabs = ws + net->ksplit; wa.slab_floats = net->ksplit_floats;
        wa.scratch_partial = ws + net->bn_partial;
        *bn_nblk = wino_rows(net->B, g.H, g.W, c.cout, 0, c.cin, net->ksplit_floats);
        const double flops = 2.0 * g.M * c.cout * c.k * c.k * c.cin;      // DIRECT-convolution flops
        const double bytes = 4.0 * ((double)net->B * g.H * g.W * c.cin + (double)g.M * c.cout +
                                    16.0 * c.cout * c.cin);
        RUN(net, std::string("conv_fwd.") + kGroupName[c.group], flops, bytes, s,
            launch_conv_wino(wa, s));
        return 0;
    }
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.w = w; a.y = y;
    if (fold_stats) {            // eval-mode BatchNorm (+ReLU / residual) folded into the epilogue
        a.ch_scale = fold_stats + 2 * c.cout;
        a.ch_shift = fold_stats + 3 * c.cout;
        a.relu = relu; a.addend = addend; a.relu_post = relu_post;
    }
    a.N = net->B; a.H = g.H; a.W = g.W; a.Cin = x_cin;
    a.Ho = g.Ho; a.Wo = g.Wo; a.Cout = c.cout;
    a.KH = a.KW = c.k; a.stride = c.stride; a.pad = c.pad;
    a.x_ld = x_cin; a.y_ld = c.cout; a.w_mode = 0; a.w_cin = x_cin;
    a.scratch = ws + net->ksplit; a.scratch_floats = net->ksplit_floats;
    if (net->side_branch) { a.scratch = nullptr; a.scratch_floats = 0; }     // (no split-K beside the main stream's)
    a.tile_counters = reinterpret_cast<int*>(ws + net->tile_cnt);
    a.tile_counters_cap = kTileCounters;
    a.force_cfg = -1;
    if (bn_nblk) {
        a.bn_partial = ws + (net->side_branch == 1 ? net->bn_partial2 : net->bn_partial);
        a.bn_nblk = bn_nblk; *bn_nblk = 0;
    }
    const double flops = 2.0 * g.M * c.cout * c.k * c.k * c.cin;
    const double bytes = 4.0 * ((double)net->B * g.H * g.W * c.cin + (double)g.M * c.cout +
                                (double)c.cout * c.k * c.k * c.cin);
    RUN(net, std::string("conv_fwd.") + kGroupName[c.group], flops, bytes, s,
        launch_conv_igemm(a, s));
    return 0;
}

// dx[B,H,W,cin] (+= addend) from dy[B,Ho,Wo,cout]
int conv_dgrad(cilrs_net* net, const ConvT& c, const ConvG& g, const float* dy, const float* w,
               float* dx, const float* addend, float* ws, hipStream_t s,
               const ConvG* bn_of = nullptr, int bn_relu = 1, int* bwd_nblk = nullptr) {
    const int ci = (int)(&g - &net->cg[0]);
    if (net->wino_on[ci]) {        // the same convolution with the flipped, transposed filter
        WinoArgs wa;
        memset(&wa, 0, sizeof(wa));
        int e = 0;
        while (net->wino_table.w[e] != (unsigned)c.w) ++e;
        wa.x = dy; wa.U = ws + net->wino_base + net->wino_table.ud[e]; wa.y = dx; wa.addend = addend;
        wa.N = net->B; wa.H = g.H; wa.W = g.W; wa.C = c.cout; wa.K = c.cin;
        // the 16-tile tail launch pays in the forward pass only: in the backward pass the weight
        // gradients of the side stream already fill the CUs a last round leaves idle, and a
        // second launch per data gradient costs more in boundaries than it gains (step 11.09 vs
        // 10.99 ms with tails on both sides, profiles/r03_wino_tail.log)
        // Re-measured in round 4 with both kinds of block in ONE launch (CILRS_WINO_DGRAD_TAIL=1 of
        // an experiments build): the data-gradient family 3.29 -> 3.04 ms serialised, the
        // overlapped step 9.32 -> 9.50 ms.
        // (decided by the plan's configuration, not by whether this step is being profiled: the
        //  profiled steps of bench.py must run the kernels the timed steps ran)
        wa.no_tail = (overlap_configured(net) && !experiment_env("CILRS_WINO_DGRAD_TAIL", 0)) ? 1 : 0;
        wa.slabs = ws + net->ksplit; wa.slab_floats = net->ksplit_floats;
        wa.scratch_partial = ws + net->bn_partial;
        if (bwd_nblk) *bwd_nblk = 0;
        if (bn_of && bwd_nblk) {
            wa.bwd_z = ws + bn_of->z; wa.bwd_y = ws + bn_of->y; wa.bwd_stats = ws + bn_of->stats;
            wa.bwd_relu = bn_relu; wa.bwd_partial = ws + net->bn_partial;
            *bwd_nblk = wino_rows(net->B, g.H, g.W, c.cin, wa.no_tail, c.cout, net->ksplit_floats);
        }
        const double flops = 2.0 * g.M * c.cout * c.k * c.k * c.cin;
        const double bytes = 4.0 * ((double)net->B * g.H * g.W * c.cin + (double)g.M * c.cout +
                                    16.0 * c.cout * c.cin);
        RUN(net, std::string("conv_dgrad.") + kGroupName[c.group], flops, bytes, s,
            launch_conv_wino(wa, s));
        return 0;
    }
    DgradArgs a;
    memset(&a, 0, sizeof(a));
    a.dy = dy; a.w = w; a.dx = dx; a.addend = addend;
    if (bn_of && bwd_nblk) {       // the BatchNorm layer whose output gradient this call produces
        a.bwd_z = ws + bn_of->z; a.bwd_y = ws + bn_of->y; a.bwd_stats = ws + bn_of->stats;
        a.bwd_relu = bn_relu; a.bwd_partial = ws + net->bn_partial; a.bwd_nblk = bwd_nblk;
        *bwd_nblk = 0;
    }
    a.N = net->B; a.H = g.H; a.W = g.W; a.Cin = c.cin;
    a.Ho = g.Ho; a.Wo = g.Wo; a.Cout = c.cout; a.K = c.k; a.stride = c.stride; a.pad = c.pad;
    a.dy_ld = c.cout; a.dx_ld = c.cin;
    a.scratch = ws + net->ksplit; a.scratch_floats = net->ksplit_floats;
    a.tile_counters = reinterpret_cast<int*>(ws + net->tile_cnt);
    a.tile_counters_cap = kTileCounters;
    a.force_cfg = -1;
    const double flops = 2.0 * g.M * c.cout * c.k * c.k * c.cin;
    const double bytes = 4.0 * ((double)net->B * g.H * g.W * c.cin + (double)g.M * c.cout +
                                (double)c.cout * c.k * c.k * c.cin);
    RUN(net, std::string("conv_dgrad.") + kGroupName[c.group], flops, bytes, s,
        launch_conv_dgrad(a, s));
    return 0;
}

// The weight gradient of every 3x3 / stride-1 convolution with channel counts in multiples of 64
// runs in the Winograd domain (conv_wino.hip: 1.09-1.46x the direct kernel at B=128, all four
// layers); CILRS_WINO_WGRAD=0 turns it off.  Independent of CILRS_WINO (forward / data gradient).
static bool wino_wgrad_on(const ConvT& c) {
    static const int env = getenv("CILRS_WINO_WGRAD") ? atoi(getenv("CILRS_WINO_WGRAD")) : 1;
    return env != 0 && wino_wgrad_supported(c.cin, c.cout, c.k, c.stride, c.pad);
}

int conv_wgrad(cilrs_net* net, const ConvT& c, const ConvG& g, const float* x, int x_cin,
               const float* dy, float* dw, float* ws, hipStream_t s) {
    if (x_cin == c.cin && !net->bf16_train && wino_wgrad_on(c)) {
        WinoWgradArgs wa;
        memset(&wa, 0, sizeof(wa));
        wa.x = x; wa.dy = dy; wa.dw = dw; wa.slabs = ws + net->slabs;
        wa.N = net->B; wa.H = g.H; wa.W = g.W; wa.C = c.cin; wa.K = c.cout;
        CILRS_CHECK(wino_wgrad_scratch_floats(net->B, g.H, g.W, c.cin, c.cout) <= net->slabs_floats,
                    "Winograd wgrad scratch too small");
        const double flops = 2.0 * g.M * c.cout * c.k * c.k * c.cin;      // DIRECT-convolution flops
        const double bytes = 4.0 * ((double)net->B * g.H * g.W * c.cin + (double)g.M * c.cout +
                                    (double)c.cout * c.k * c.k * c.cin);
        RUN(net, std::string("conv_wgrad.") + kGroupName[c.group], flops, bytes, s,
            launch_conv_wino_wgrad(wa, s));
        return 0;
    }
    WgradArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.dy = dy; a.dw = dw; a.slabs = ws + net->slabs;
    a.N = net->B; a.H = g.H; a.W = g.W; a.Cin = x_cin;
    a.Ho = g.Ho; a.Wo = g.Wo; a.Cout = c.cout;
    a.KH = a.KW = c.k; a.stride = c.stride; a.pad = c.pad;
    a.x_ld = x_cin; a.dy_ld = c.cout; a.Cin_dst = c.cin; a.accumulate = 0;
    CILRS_CHECK(wgrad_scratch_floats(a) <= net->slabs_floats, "wgrad scratch too small");
    const double flops = 2.0 * g.M * c.cout * c.k * c.k * c.cin;
    const double bytes = 4.0 * ((double)net->B * g.H * g.W * c.cin + (double)g.M * c.cout +
                                (double)c.cout * c.k * c.k * c.cin);
    RUN(net, std::string("conv_wgrad.") + kGroupName[c.group], flops, bytes, s,
        launch_conv_wgrad(a, s));
    return 0;
}

cilrs_half* h16(float* ws, size_t off_floats) {
    return reinterpret_cast<cilrs_half*>(ws + off_floats);
}

// the same three operators on the bf16 matrix pipe (bf16 training mode); profile labels as above.
// Since round 4 the mode keeps every trunk tensor after the stem in bf16 (activations, raw
// convolution outputs, gradients): the raw output y16 of convolution ci lives where the fp32 mode
// keeps its y (same offset, half the bytes), the post-BatchNorm tensor in z16[ci].
cilrs_half* y16_of(const cilrs_net* net, float* ws, int ci) { return h16(ws, net->cg[ci].y); }

int conv_fwd16(cilrs_net* net, const ConvT& c, const ConvG& g, int ci, const cilrs_half* x16,
               float* ws, hipStream_t s, int* bn_nblk) {
    ConvF16Args a;
    memset(&a, 0, sizeof(a));
    a.x = x16; a.w = h16(ws, net->w16_all) + c.w; a.y16 = y16_of(net, ws, ci);
    a.bn_partial = ws + (net->side_branch == 1 ? net->bn_partial2 : net->bn_partial);
    a.N = net->B; a.H = g.H; a.W = g.W; a.Cin = c.cin; a.Ho = g.Ho; a.Wo = g.Wo; a.Cout = c.cout;
    a.K = c.k; a.stride = c.stride; a.pad = c.pad; a.bf16 = 1;
    *bn_nblk = conv_f16_train_mtiles(a);
    const double flops = 2.0 * g.M * c.cout * c.k * c.k * c.cin;
    const double bytes = 2.0 * ((double)net->B * g.H * g.W * c.cin + (double)c.cout * c.k * c.k * c.cin +
                                (double)g.M * c.cout);
    RUN(net, std::string("conv_fwd.") + kGroupName[c.group], flops, bytes, s,
        launch_conv_f16_train(a, s));
    return 0;
}

// dx (16-bit, or fp32 when dx32 is given: the gradient handed to the fp32 stem) = dgrad(dy16)
// (+ addend16); bn_of: the convolution whose BatchNorm-backward reductions ride on the epilogue
int conv_dgrad16(cilrs_net* net, const ConvT& c, const ConvG& g, int ci, const cilrs_half* dy16,
                 cilrs_half* dx16, float* dx32, const cilrs_half* addend16, float* ws,
                 hipStream_t s, const ConvG* bn_of = nullptr, int* bwd_nblk = nullptr) {
    ConvF16Args a;
    memset(&a, 0, sizeof(a));
    a.x = dy16; a.w = h16(ws, net->wT16) + net->wT16_off[ci];
    if (dx32) a.y32 = dx32; else a.y16 = dx16;
    a.addend16 = addend16;
    a.N = net->B; a.H = g.Ho; a.W = g.Wo; a.Cin = c.cout;      // gathered tensor = dy
    a.Ho = g.H; a.Wo = g.W; a.Cout = c.cin;                    // enumerated grid = dx
    a.K = c.k; a.bf16 = 1;
    if (c.stride == 1) { a.stride = 1; a.pad = c.k - 1 - c.pad; }
    else { a.stride = 2; a.pad = c.pad; a.up2 = 1; }
    if (bwd_nblk) *bwd_nblk = 0;
    if (bn_of && bwd_nblk && conv_f16_train_can_fuse_bwd(a)) {   // the BatchNorm whose output gradient this is
        const int bi = (int)(bn_of - &net->cg[0]);
        a.bwd_z16 = h16(ws, net->z16[bi]); a.bwd_y16 = y16_of(net, ws, bi);
        a.bwd_stats = ws + bn_of->stats;
        a.bwd_relu = 1; a.bwd_partial = ws + net->bn_partial;
        *bwd_nblk = conv_f16_train_mtiles(a);
    }
    const double flops = 2.0 * g.M * c.cout * c.k * c.k * c.cin;
    const double bytes = 2.0 * ((double)g.M * c.cout + (double)c.cout * c.k * c.k * c.cin) +
                         (dx32 ? 4.0 : 2.0) * (double)net->B * g.H * g.W * c.cin;
    RUN(net, std::string("conv_dgrad.") + kGroupName[c.group], flops, bytes, s,
        launch_conv_f16_train(a, s));
    return 0;
}

WgradF16Args wgrad16_args(const cilrs_net* net, const ConvT& c, const ConvG& g) {
    WgradF16Args a;
    memset(&a, 0, sizeof(a));
    a.N = net->B; a.H = g.H; a.W = g.W; a.Cin = c.cin; a.Ho = g.Ho; a.Wo = g.Wo; a.Cout = c.cout;
    a.K = c.k; a.stride = c.stride; a.pad = c.pad; a.bf16 = 1;
    return a;
}

int conv_wgrad16(cilrs_net* net, const ConvT& c, const ConvG& g, const cilrs_half* x16,
                 const cilrs_half* dy16, float* dw, float* ws, hipStream_t s) {
    WgradF16Args a = wgrad16_args(net, c, g);
    a.x = x16; a.dy = dy16; a.dw = dw; a.slabs = ws + net->slabs16;
    CILRS_CHECK(wgrad_f16_scratch_floats(a) <= net->slabs16_floats, "wgrad16 scratch too small");
    const double flops = 2.0 * g.M * c.cout * c.k * c.k * c.cin;
    const double bytes = 2.0 * ((double)net->B * g.H * g.W * c.cin + (double)g.M * c.cout) +
                         4.0 * (double)c.cout * c.k * c.k * c.cin;
    RUN(net, std::string("conv_wgrad.") + kGroupName[c.group], flops, bytes, s,
        launch_wgrad_f16(a, s));
    return 0;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// C-ABI: layout queries
// ------------------------------------------------------------------------------------------------
extern "C" {

int cilrs_version(void) { return 1; }
const char* cilrs_last_error(void) { return last_error(); }
int cilrs_num_params(void) { return (int)arch().params.size(); }
int cilrs_num_bn(void) { return (int)arch().bns.size(); }
size_t cilrs_param_arena_floats(void) { return arch().arena_floats; }
size_t cilrs_param_count(void) { return arch().count; }
size_t cilrs_bn_arena_floats(void) { return arch().bn_floats; }

int cilrs_num_variants(void) { return kNumVariants; }
static bool variant_ok(int v) {
    return v >= 0 && code_trunk(v) < kNumVariants && (v >> 16) == 0 && code_ncmd(v) >= 1 &&
           code_ncmd(v) <= kMaxCmd;
}
int cilrs_variant_num_params(int variant) {
    return variant_ok(variant) ? (int)arch(variant).params.size() : -1;
}
int cilrs_variant_num_bn(int variant) {
    return variant_ok(variant) ? (int)arch(variant).bns.size() : -1;
}
size_t cilrs_variant_param_arena_floats(int variant) {
    return variant_ok(variant) ? arch(variant).arena_floats : 0;
}
size_t cilrs_variant_param_count(int variant) {
    return variant_ok(variant) ? arch(variant).count : 0;
}
size_t cilrs_variant_bn_arena_floats(int variant) {
    return variant_ok(variant) ? arch(variant).bn_floats : 0;
}
int cilrs_variant_feature_width(int variant) {
    return variant_ok(variant) ? arch(variant).feat : -1;
}

int cilrs_param_info(int i, char* name, int name_cap, size_t* offset, size_t* numel, int* ndim,
                     int* shape4) {
    return cilrs_variant_param_info(0, i, name, name_cap, offset, numel, ndim, shape4);
}

int cilrs_variant_param_info(int variant, int i, char* name, int name_cap, size_t* offset,
                             size_t* numel, int* ndim, int* shape4) {
    CILRS_CHECK(variant_ok(variant), "variant %d out of range", variant);
    const Arch& A = arch(variant);
    CILRS_CHECK(i >= 0 && i < (int)A.params.size(), "param index %d out of range", i);
    const ParamT& p = A.params[i];
    if (name && name_cap > 0) snprintf(name, name_cap, "%s", p.name.c_str());
    if (offset) *offset = p.off;
    if (numel) *numel = p.numel;
    if (ndim) *ndim = p.ndim;
    if (shape4) for (int k = 0; k < 4; ++k) shape4[k] = p.shape[k];
    return 0;
}

int cilrs_bn_info(int j, char* prefix, int prefix_cap, int* channels, size_t* rm_offset,
                  size_t* rv_offset) {
    return cilrs_variant_bn_info(0, j, prefix, prefix_cap, channels, rm_offset, rv_offset);
}

int cilrs_variant_bn_info(int variant, int j, char* prefix, int prefix_cap, int* channels,
                          size_t* rm_offset, size_t* rv_offset) {
    CILRS_CHECK(variant_ok(variant), "variant %d out of range", variant);
    const Arch& A = arch(variant);
    CILRS_CHECK(j >= 0 && j < (int)A.bns.size(), "bn index %d out of range", j);
    const BnT& b = A.bns[j];
    if (prefix && prefix_cap > 0) snprintf(prefix, prefix_cap, "%s", b.prefix.c_str());
    if (channels) *channels = b.C;
    if (rm_offset) *rm_offset = b.rm;
    if (rv_offset) *rv_offset = b.rv;
    return 0;
}

int cilrs_segment_range(int seg, size_t* begin, size_t* end) {
    return cilrs_variant_segment_range(0, seg, begin, end);
}

int cilrs_variant_segment_range(int variant, int seg, size_t* begin, size_t* end) {
    CILRS_CHECK(variant_ok(variant), "variant %d out of range", variant);
    CILRS_CHECK(seg >= 0 && seg < 6 && begin && end, "segment %d out of range", seg);
    *begin = arch(variant).seg_begin[seg];
    *end = arch(variant).seg_end[seg];
    return 0;
}

// ------------------------------------------------------------------------------------------------
// plan creation
// ------------------------------------------------------------------------------------------------
int cilrs_net_create(int batch, int height, int width, cilrs_net** out) {
    return cilrs_net_create_variant(0, batch, height, width, out);
}

int cilrs_net_create_variant(int variant, int batch, int height, int width, cilrs_net** out) {
    return cilrs_net_create_ex(variant, batch, height, width, 0u, out);
}

int cilrs_net_create_ex(int variant, int batch, int height, int width, unsigned flags,
                        cilrs_net** out) {
    CILRS_CHECK(out != nullptr, "cilrs_net_create: out is NULL");
    CILRS_CHECK((flags & ~1u) == 0, "cilrs_net_create: unknown flags 0x%x", flags);
    CILRS_CHECK(variant_ok(variant), "cilrs_net_create: variant %d out of range", variant);
    CILRS_CHECK(batch >= 1 && height >= 32 && width >= 32, "cilrs_net_create: bad geometry %d %d %d",
                batch, height, width);
    const Arch& A = arch(variant);
    const bool trainable = true;
    cilrs_net* n = new cilrs_net();
    n->A = &A;
    n->B = batch; n->H = height; n->W = width;
    n->cg.resize(A.convs.size());
    Bump bump;
    const int B = batch;
    n->x4 = bump.take((size_t)B * height * width * 4);
    n->w4 = bump.take((size_t)64 * 49 * 4);

    auto out_dim = [](int in, int k, int s, int p) { return (in + 2 * p - k) / s + 1; };
    // stem
    {
        ConvG& g = n->cg[0];
        g.H = height; g.W = width;
        g.Ho = out_dim(height, 7, 2, 3); g.Wo = out_dim(width, 7, 2, 3);
        g.M = B * g.Ho * g.Wo;
        g.y = bump.take(trainable ? (size_t)g.M * 64 : 4);
        g.z = bump.take((size_t)g.M * 64);
        g.stats = bump.take(4 * 64);
        n->H0 = g.Ho; n->W0 = g.Wo;
        n->H1 = out_dim(g.Ho, 3, 2, 1); n->W1 = out_dim(g.Wo, 3, 2, 1);
    }
    size_t gmax = (size_t)n->cg[0].M * 64;
    size_t dymax = 0;
    size_t actmax = (size_t)B * n->H1 * n->W1 * 64;       // largest trunk tensor (elements)
    n->pool = bump.take((size_t)B * n->H1 * n->W1 * 64);
    const size_t argmax_floats = ((size_t)B * n->H1 * n->W1 * 64 + 3) / 4;
    n->argmax_b = bump.take(argmax_floats) * sizeof(float);
    int h = n->H1, w = n->W1;
    size_t slabs_max = 0, ksplit_max = 0;
    auto track = [&](const ConvT& c, const ConvG& g, int x_cin) {
        if (trainable) {
            WgradArgs wa;
            memset(&wa, 0, sizeof(wa));
            wa.N = B; wa.H = g.H; wa.W = g.W; wa.Cin = x_cin; wa.Ho = g.Ho; wa.Wo = g.Wo;
            wa.Cout = c.cout; wa.KH = wa.KW = c.k; wa.stride = c.stride; wa.pad = c.pad;
            const size_t sf = wgrad_scratch_floats(wa);
            if (sf > slabs_max) slabs_max = sf;
            if (x_cin == c.cin && wino_wgrad_on(c)) {
                const size_t sw = wino_wgrad_scratch_floats(B, g.H, g.W, c.cin, c.cout);
                if (sw > slabs_max) slabs_max = sw;
            }
        }
        // split-K scratch: only worthwhile for the small-M layers
        const size_t fwd = (size_t)g.M * c.cout, bwd = (size_t)B * g.H * g.W * c.cin;
        const size_t big = fwd > bwd ? fwd : bwd;
        if (big <= (size_t)6 * 1024 * 1024 && 8 * big > ksplit_max) ksplit_max = 8 * big;
        if (fwd > actmax) actmax = fwd;
    };
    // geometry + activation buffers of one convolution (y: pre-BN output, kept for backward only)
    auto place = [&](int ci, int in_h, int in_w) {
        const ConvT& c = A.convs[ci];
        ConvG& g = n->cg[ci];
        g.H = in_h; g.W = in_w;
        g.Ho = out_dim(in_h, c.k, c.stride, c.pad); g.Wo = out_dim(in_w, c.k, c.stride, c.pad);
        g.M = B * g.Ho * g.Wo;
        g.y = bump.take(trainable ? (size_t)g.M * c.cout : 4);
        g.z = bump.take((size_t)g.M * c.cout);
        g.stats = bump.take(4 * c.cout);
        track(c, g, c.cin);
    };
    track(A.convs[0], n->cg[0], 4);
    for (const BlockT& blk : A.blocks) {
        place(blk.conv1, h, w);
        const ConvG& g1 = n->cg[blk.conv1];
        place(blk.conv2, g1.Ho, g1.Wo);
        const ConvG& g2 = n->cg[blk.conv2];
        int oh = g2.Ho, ow = g2.Wo;
        if (blk.conv3 >= 0) place(blk.conv3, g2.Ho, g2.Wo);
        if (blk.down >= 0) {
            place(blk.down, h, w);
            CILRS_CHECK(n->cg[blk.down].Ho == oh && n->cg[blk.down].Wo == ow,
                        "downsample geometry mismatch");
        }
        const ConvT& c1 = A.convs[blk.conv1];
        const size_t act = (size_t)B * h * w * c1.cin;
        if (act > gmax) gmax = act;
        // gradient buffers hold d(block input), d(block output) and every conv output's gradient
        for (int ci : {blk.conv1, blk.conv2, blk.conv3, blk.down}) {
            if (ci < 0) continue;
            const size_t o = (size_t)n->cg[ci].M * A.convs[ci].cout;
            if (o > gmax) gmax = o;
            if (o > dymax) dymax = o;                                         // trunk dy tensors
        }
        h = oh; w = ow;
    }
    n->featHW = h * w;
    const int feat = A.feat, comb = A.feat + 128;
    // heads
    n->combined = bump.take((size_t)B * comb);
    n->s1 = bump.take((size_t)B * 128);
    n->p1 = bump.take((size_t)B * 256);
    n->p2 = bump.take((size_t)B * 256);
    const int NC = A.ncmd;
    for (int k = 0; k < NC; ++k) {
        n->h1[k] = bump.take((size_t)B * 256);
        n->h2[k] = bump.take((size_t)B * 256);
    }
    n->all_out = bump.take((size_t)NC * B * 4);
    n->dcombined = bump.take(trainable ? (size_t)B * comb : 4);
    n->ds1 = bump.take(trainable ? (size_t)B * 128 : 4);
    n->dp1 = bump.take(trainable ? (size_t)B * 256 : 4);
    n->dp2 = bump.take(trainable ? (size_t)B * 256 : 4);
    for (int k = 0; k < NC; ++k) {
        n->dh1[k] = bump.take(trainable ? (size_t)B * 256 : 4);
        n->dh2[k] = bump.take(trainable ? (size_t)B * 256 : 4);
    }
    for (int k = 0; k <= NC; ++k) n->dcomb_part[k] = bump.take(trainable ? (size_t)B * comb : 4);
    n->d_all = bump.take((size_t)NC * B * 4);
    n->speed_in = bump.take((size_t)B);
    n->cmd_b = bump.take((size_t)B * 2) * sizeof(float);
    // the heads' wide linears also use the wgrad slabs
    if (trainable) {
        WgradArgs wa;
        memset(&wa, 0, sizeof(wa));
        wa.N = B; wa.H = wa.W = wa.Ho = wa.Wo = 1; wa.Cin = comb; wa.Cout = 256;
        wa.KH = wa.KW = 1; wa.stride = 1;
        const size_t sf = wgrad_scratch_floats(wa);
        if (sf > slabs_max) slabs_max = sf;
    }
    (void)feat;
    if (!trainable) { gmax = 4; dymax = 4; }
    n->gmax = gmax;
    for (int i = 0; i < 5; ++i) n->G[i] = bump.take(gmax);
    for (int i = 5; i < kNumG; ++i) n->G[i] = bump.take(dymax);
    {
        size_t need = bn_partial_floats(512);
        if (trainable)
            for (size_t ci = 0; ci < A.convs.size(); ++ci) {
                // per-tile column partials: forward (conv output) and backward (the data
                // gradient's output = this conv's input) reductions ride on the GEMM epilogues
                const size_t t = (size_t)cdiv(n->cg[ci].M, 64) * 2 * A.convs[ci].cout;
                if (t > need) need = t;
                const size_t tb = (size_t)cdiv(B * n->cg[ci].H * n->cg[ci].W, 64) * 2 *
                                  A.convs[ci].cin;
                if (ci > 0 && tb > need) need = tb;
                if (bn_partial_floats(A.convs[ci].cout) > need)
                    need = bn_partial_floats(A.convs[ci].cout);
            }
        n->bn_partial = bump.take(need);
        n->bn_partial2 = bump.take(need);      // (the down-sample branch of a block, built beside conv1)
    }
    n->bn_coef = bump.take(3 * 2048);
    n->slabs_floats = slabs_max;
    n->slabs = bump.take(slabs_max > 0 ? slabs_max : 4);
    n->ksplit_floats = ksplit_max;
    n->ksplit = bump.take(ksplit_max > 0 ? ksplit_max : 4);
    n->status_b = bump.take(64) * sizeof(float);
    n->tile_cnt = bump.take(kTileCounters + 2 * kBnSyncInts);  // + the BatchNorm finalize counters
    CILRS_CHECK((int)A.convs.size() <= kMaxConvs, "too many convolutions for the BN tables");
    n->bn_table.n = (int)A.convs.size();
    for (size_t ci = 0; ci < A.convs.size(); ++ci) {
        const BnT& b = A.bns[A.convs[ci].bn];
        n->bn_table.C[ci] = b.C;
        n->bn_table.gamma[ci] = (unsigned)b.gamma; n->bn_table.beta[ci] = (unsigned)b.beta;
        n->bn_table.rm[ci] = (unsigned)b.rm; n->bn_table.rv[ci] = (unsigned)b.rv;
        n->bn_table.stats[ci] = (unsigned)n->cg[ci].stats;
    }
    {   // 16-bit inference arenas (every conv but the stem): folded weights, biases, activations
        size_t halfs = 0, floats = 0;
        n->f16_table.n = (int)A.convs.size() - 1;
        for (size_t ci = 1; ci < A.convs.size(); ++ci) {
            const ConvT& c = A.convs[ci];
            const int e = (int)ci - 1;
            n->f16_table.cout[e] = c.cout;
            n->f16_table.krow[e] = (unsigned)(c.k * c.k * c.cin);
            n->f16_table.w[e] = (unsigned)c.w;
            n->f16_table.stats[e] = (unsigned)n->cg[ci].stats;
            n->f16_table.w16[e] = (unsigned)halfs;
            n->f16_table.bias[e] = (unsigned)floats;
            halfs += (size_t)c.cout * c.k * c.k * c.cin;
            floats += (size_t)c.cout;
        }
        n->f16_w = bump.take((halfs + 1) / 2);
        n->f16_bias = bump.take(floats);
        n->stem16_w = bump.take(64 * 224 / 2);
        n->stem16_b = bump.take(64);
        n->f16_act_floats = (actmax + 1) / 2;                              // largest trunk tensor
        for (int k = 0; k < 5; ++k) n->f16_act[k] = bump.take(n->f16_act_floats);
    }
    if (flags & 1u) {        // CILRS_PLAN_BF16_TRAIN: 16-bit shadows of the training tensors
        n->bf16_train = true;
        n->w16_all = bump.take((A.arena_floats + 1) / 2);
        size_t halfs = 0, sl = 0;
        n->tr_table.n = 0;
        for (size_t ci = 1; ci < A.convs.size(); ++ci) {
            const ConvT& c = A.convs[ci];
            const int e = n->tr_table.n++;
            n->tr_table.cout[e] = c.cout; n->tr_table.k[e] = c.k; n->tr_table.cin[e] = c.cin;
            n->tr_table.w[e] = (unsigned)c.w;
            n->tr_table.wT[e] = (unsigned)halfs;
            n->tr_table.tile_begin[e] = e == 0 ? 0 : n->tr_table.tile_begin[e];
            n->tr_table.tile_begin[e + 1] =
                n->tr_table.tile_begin[e] + c.k * c.k * (c.cout / 32) * (c.cin / 32);
            n->wT16_off[ci] = halfs;
            halfs += ((size_t)c.cout * c.k * c.k * c.cin + 7) / 8 * 8;
            n->z16[ci] = bump.take(((size_t)n->cg[ci].M * c.cout + 1) / 2);
            const size_t f = wgrad_f16_scratch_floats(wgrad16_args(n, c, n->cg[ci]));
            if (f > sl) sl = f;
        }
        n->wT16 = bump.take((halfs + 1) / 2);
        n->pool16 = bump.take(((size_t)B * n->H1 * n->W1 * 64 + 1) / 2);
        for (int i = 0; i < kNumG; ++i) n->G16[i] = bump.take((gmax + 1) / 2);
        n->slabs16_floats = sl;
        n->slabs16 = bump.take(sl > 0 ? sl : 4);
    }
    n->wino_table.n = 0;
    n->wino_table.blk_begin[0] = 0;
    {
        static const int wino_env = getenv("CILRS_WINO") ? atoi(getenv("CILRS_WINO")) : 1;
        if (trainable && !(flags & 1u) && wino_env) {
            size_t floats = 0;
            for (size_t ci = 1; ci < A.convs.size(); ++ci) {
                const ConvT& c = A.convs[ci];
                const ConvG& g = n->cg[ci];
                if (!wino_supported(c.cin, c.cout, c.k, c.stride, c.pad) ||
                    !wino_supported(c.cout, c.cin, c.k, c.stride, c.pad) || c.cin % 32 != 0 ||
                    c.cin > 256 || c.cout > 256)
                    continue;
                // one 64-tile x 64-channel block per CU: below ~half a chip of blocks the
                // implicit GEMM (split-K, smaller tiles) wins (tools/wino_bench.py)
                const int blocks = wino_groups(B, g.H, g.W) * (std::min(c.cin, c.cout) / 64);
                if (wino_env == 1 && blocks < 128) continue;
                WinoWeightTable& t = n->wino_table;
                const int e = t.n++;
                t.K[e] = c.cout; t.C[e] = c.cin; t.w[e] = (unsigned)c.w;
                t.u[e] = (unsigned)floats; floats += wino_weight_floats(c.cout, c.cin);
                t.ud[e] = (unsigned)floats; floats += wino_weight_floats(c.cout, c.cin);
                t.blk_begin[e + 1] = t.blk_begin[e] + (c.cout / 8) * (c.cin / 32);
                n->wino_on[ci] = true;
            }
            CILRS_CHECK(floats < (1ull << 32), "Winograd filter images exceed 32-bit offsets");
            if (n->wino_table.n) {
                n->wino_base = bump.take(floats);
                if (wino_prepare()) return 1;
            }
        }
    }
    if (batch == 1 && code_trunk(variant) == 0 && A.ncmd == 4) {     // (the persistent kernel is built for the reference's 4 commands)
        n->b1_table = bump.take(kB1MaxStages * sizeof(B1Stage) / sizeof(float));
        n->b1_sync = bump.take(kB1SyncInts);
        n->b1_cmd = bump.take(16);
        n->b1_stamps = bump.take(2 * (10 * (kB1MaxStages + 1) + kB1MaxStages * 512));
        // split-K partial tiles: up to 4 slices of the largest [16-row tiles][Cout] output, twice
        // (a stage may hold two convolutions)
        size_t big = 0;
        for (size_t ci = 1; ci < A.convs.size(); ++ci) {
            const size_t o = (size_t)cdiv(n->cg[ci].M, 16) * 16 * A.convs[ci].cout;
            if (o > big) big = o;
        }
        n->b1_slab_floats = 2 * 4 * big;
        n->b1_slabs = bump.take(n->b1_slab_floats);
    }
    n->ws_bytes = bump.off * sizeof(float);
    *out = n;
    return 0;
}

void cilrs_net_destroy(cilrs_net* net) {
    if (net && net->streams_ready) {
        (void)hipStreamDestroy(net->side[0]);
        for (int i = 0; i < kNumG; ++i) (void)hipEventDestroy(net->gbuf_ev[i]);
        (void)hipEventDestroy(net->fork_ev);
        (void)hipEventDestroy(net->wprep_ev);
        (void)hipEventDestroy(net->branch_ev);
        (void)hipEventDestroy(net->heads_ev);
    }
    if (net && net->graph_exec) (void)hipGraphExecDestroy(net->graph_exec);
    delete net;
}
size_t cilrs_net_workspace_bytes(const cilrs_net* net) { return net ? net->ws_bytes : 0; }
size_t cilrs_net_status_offset(const cilrs_net* net) { return net ? net->status_b : 0; }

int cilrs_net_activation_info(const cilrs_net* net, int conv, size_t* y_offset, size_t* z_offset,
                              size_t* numel, int* channels) {
    CILRS_CHECK(net != nullptr, "activation_info: net is NULL");
    // conv -1: the max-pool output (input of the first residual block)
    if (conv == -1) {
        if (y_offset) *y_offset = net->pool;
        if (z_offset) *z_offset = net->pool;
        if (numel) *numel = (size_t)net->B * net->H1 * net->W1 * 64;
        if (channels) *channels = 64;
        return 0;
    }
    CILRS_CHECK(conv >= 0 && conv < (int)net->cg.size(), "activation_info: conv %d out of range", conv);
    const ConvG& g = net->cg[conv];
    if (y_offset) *y_offset = g.y;
    if (z_offset) *z_offset = g.z;
    if (numel) *numel = (size_t)g.M * net->A->convs[conv].cout;
    if (channels) *channels = net->A->convs[conv].cout;
    return 0;
}

int cilrs_net_set_weights_key(cilrs_net* net, uint64_t key) {
    CILRS_CHECK(net != nullptr, "set_weights_key: net is NULL");
    net->weights_key = key;
    return 0;
}

int cilrs_dropout(float* a, int rows, int cols, int ld, float p, uint64_t seed, int site,
                  void* stream) {
    CILRS_CHECK(a && rows >= 1 && cols >= 1 && ld >= cols && site >= 0 && site <= 9,
                "dropout: bad argument");
    return launch_dropout(a, rows, cols, ld, p, seed, (unsigned long long)site,
                          reinterpret_cast<hipStream_t>(stream));
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
static int forward_from_x4(cilrs_net* net, const cilrs_buffers* bufs, const float* speed,
                           const int64_t* command, int train, float dropout_p, uint64_t seed,
                           float* controls, float* pred_speed, hipStream_t s, int half = 0) {
    const Arch& A = *net->A;
    float* ws = reinterpret_cast<float*>(bufs->workspace);
    net->ws_base = ws;
    if (zero_counters_once(net, bufs->workspace, s)) return 1;
    const float* P = bufs->params;
    float* R = bufs->bn_running;
    const int B = net->B;
    const float eps = 1e-5f, mom = 0.1f;
    const bool bf16t = train && net->bf16_train;      // trunk convolutions on the bf16 matrix pipe

    // (residual: fp32 tensor, or the bf16 identity in the bf16 training mode)
    auto bn = [&](int ci, const void* residual_v, int relu, int pre_nblk, hipStream_t st = nullptr,
                  bool side_branch = false) -> int {
        if (st == nullptr) st = s;
        float* const partial = ws + (side_branch ? net->bn_partial2 : net->bn_partial);
        const float* residual = reinterpret_cast<const float*>(residual_v);
        const ConvT& c = A.convs[ci];
        const ConvG& g = net->cg[ci];
        const BnT& b = A.bns[c.bn];
        const double bytes = 4.0 * g.M * c.cout * (residual ? 4.0 : 3.0);
        if (bf16t) {         // 16-bit tensors: residual is the bf16 identity
            RUN(net, std::string("bn_fwd.") + kGroupName[c.group], 0.0, bytes / 2.0, st,
                launch_bn16_train_fwd(y16_of(net, ws, ci), g.M, c.cout, P + b.gamma, P + b.beta,
                                      R + b.rm, R + b.rv,
                                      reinterpret_cast<long long*>(bufs->bn_nbt) + c.bn, mom, eps,
                                      residual, relu, ws + g.stats, partial,
                                      h16(ws, net->z16[ci]), pre_nblk, st));
        } else if (train) {
            RUN(net, std::string("bn_fwd.") + kGroupName[c.group], 0.0, bytes, st,
                launch_bn_train_fwd(ws + g.y, g.M, c.cout, P + b.gamma, P + b.beta, R + b.rm,
                                    R + b.rv, reinterpret_cast<long long*>(bufs->bn_nbt) + c.bn,
                                    mom, eps, residual, relu, ws + g.stats, partial,
                                    ws + g.z, pre_nblk, st, nullptr, side_branch ? nullptr : bn_sync(net, ws, 0)));
        } else {
            RUN(net, std::string("bn_fwd.") + kGroupName[c.group], 0.0, bytes, st,
                launch_bn_eval_fwd(ws + g.y, g.M, c.cout, P + b.gamma, P + b.beta, R + b.rm,
                                   R + b.rv, eps, residual, relu, ws + g.stats, ws + g.z, st));
        }
        return 0;
    };

    // ---- stem: conv7x7/s2 + BN + ReLU + maxpool3x3/s2 ----
    const bool same_bufs = net->prep_bufs[0] == (const void*)bufs->params &&
                           net->prep_bufs[1] == (const void*)bufs->bn_running &&
                           net->prep_bufs[2] == (const void*)bufs->workspace;
    const bool prep_cached = !train && net->weights_key != 0 && same_bufs &&
                             net->prep_key == net->weights_key;
    if (!prep_cached)
        RUN(net, "transform", 0.0, 0.0, s,
            launch_pad_cin3_to_4(P + A.convs[0].w, ws + net->w4, 64 * 49, s));
    if (train) { net->prep_key = 0; net->fold_key = 0; }     // weights / BN buffers are about to change
    const float* cur;
    if (train) {
        // this step's weight images -- the Winograd-transformed filters (forward + data-gradient
        // forms), in the bf16 mode the 16-bit arena and the transposed, tap-flipped copies the data
        // gradients read -- depend on the parameters only: they are built on the side stream while
        // the stem and the max-pool run (memory-bound launches beside a matrix-pipe-bound one); the
        // first residual block waits for them
        {
            hipStream_t ts = s;
            const bool fork = use_overlap(net) && (net->wino_table.n || bf16t);
            if (fork) {
                if (gbuf_side_begin(net, s)) return 1;
                ts = net->side[0];
            }
            if (net->wino_table.n)
                RUN(net, "transform", 0.0, 4.0 * 4.6 * (double)net->wino_table.blk_begin[net->wino_table.n] * 256, ts,
                    launch_wino_weights_all(net->wino_table, P, ws + net->wino_base, ts));
            if (bf16t) {
                RUN(net, "transform", 0.0, 6.0 * A.arena_floats, ts,
                    launch_f32_to_f16(P, h16(ws, net->w16_all), A.arena_floats, 1, ts));
                RUN(net, "transform", 0.0, 0.0, ts,
                    launch_transpose_flip_f16_all(net->tr_table, P, h16(ws, net->wT16), 1, ts));
            }
            if (fork) {
                CILRS_HIP(hipEventRecord(net->wprep_ev, ts));
                net->wprep_pending = true;
            }
        }
        int nb = 0;                             // batch statistics fused into the conv epilogue
        if (const int srows = stem_f32_rows(B, net->H, net->W)) {
            // (weights in registers, k = 7 x 22 instead of 13 x 16: stem_f32.hip)
            RUN(net, "conv_fwd.stem", 2.0 * net->cg[0].M * 64 * 147,
                16.0 * B * net->H * net->W + 4.0 * net->cg[0].M * 64, s,
                launch_stem_f32(ws + net->x4, P + A.convs[0].w, ws + net->cg[0].y,
                                ws + net->bn_partial, B, net->H, net->W, s));
            nb = srows;
        } else if (conv_fwd(net, A.convs[0], net->cg[0], ws + net->x4, 4, ws + net->w4,
                            ws + net->cg[0].y, ws, s, &nb)) {
            return 1;
        }
        // stem BatchNorm: statistics only -- its apply + ReLU is fused into the max-pool, the
        // post-BN tensor (144 MB at B=128) is never written
        {
            const ConvT& c0 = A.convs[0];
            const ConvG& g0 = net->cg[0];
            const BnT& b0 = A.bns[c0.bn];
            RUN(net, "bn_fwd.stem", 0.0, 0.0, s,
                launch_bn_train_fwd(ws + g0.y, g0.M, c0.cout, P + b0.gamma, P + b0.beta,
                                    R + b0.rm, R + b0.rv,
                                    reinterpret_cast<long long*>(bufs->bn_nbt) + c0.bn, mom, eps,
                                    nullptr, 1, ws + g0.stats, ws + net->bn_partial, nullptr, nb,
                                    s));
        }
        unsigned char* argmax = reinterpret_cast<unsigned char*>(bufs->workspace) + net->argmax_b;
        RUN(net, "maxpool", 0.0, 4.0 * net->cg[0].M * 64 * 1.25, s,
            launch_bn_relu_maxpool_fwd(ws + net->cg[0].y, ws + net->cg[0].stats, ws + net->pool,
                                       argmax, B, net->H0, net->W0, 64, s,
                                       bf16t ? (void*)h16(ws, net->pool16) : nullptr));
        // (this step's weight images were requested before the stem, on the side stream)
        if (net->wprep_pending) {
            CILRS_HIP(hipStreamWaitEvent(s, net->wprep_ev, 0));
            net->wprep_pending = false;
        }
        // ---- residual blocks: BasicBlock conv-BN-ReLU-conv-BN-(+id)-ReLU, Bottleneck with a third
        //      conv-BN pair; the identity (or downsample branch) joins at the last BatchNorm ----
        cur = ws + net->pool;
        const cilrs_half* cur16 = h16(ws, net->pool16);
        for (const BlockT& blk : A.blocks) {
            const int chain[3] = {blk.conv1, blk.conv2, blk.conv3};
            const int nchain = blk.conv3 >= 0 ? 3 : 2;
            const void* identity = bf16t ? (const void*)cur16 : (const void*)cur;
            const float* x = cur;
            const cilrs_half* x16 = cur16;
            // The down-sample branch (1x1 / stride-2 convolution + BatchNorm: three short launches
            // that leave most of the chip idle) depends on the block's input only: it is built on
            // the side stream beside conv1 / bn1 / conv2, with column-partial scratch of its own
            // and no split-K scratch; the block's last BatchNorm -- which adds it -- waits for it.
            bool branch_pending = false;
            const bool branch_aside = blk.down >= 0 && use_overlap(net) && bn_sync(net, ws, 0) == nullptr;
            auto down_branch = [&](hipStream_t st, bool aside) -> int {
                const ConvT& cd = A.convs[blk.down];
                const ConvG& gd = net->cg[blk.down];
                int nbd = 0;
                // (1: beside the main stream, own partial scratch; 2: on the main stream -- in BOTH
                //  cases without split-K, so that the plan, hence the summation order, of this
                //  convolution does not depend on whether the streams overlap)
                net->side_branch = aside ? 1 : 2;
                int rc = bf16t ? conv_fwd16(net, cd, gd, blk.down, cur16, ws, st, &nbd)
                               : conv_fwd(net, cd, gd, cur, cd.cin, P + cd.w, ws + gd.y, ws, st, &nbd);
                net->side_branch = 0;
                if (rc) return 1;
                if (bn(blk.down, nullptr, 0, nbd, st, aside)) return 1;
                identity = bf16t ? (const void*)h16(ws, net->z16[blk.down]) : (const void*)(ws + gd.z);
                return 0;
            };
            if (branch_aside) {
                if (gbuf_side_begin(net, s)) return 1;            // the side stream sees the block's input
                if (down_branch(net->side[0], true)) return 1;
                CILRS_HIP(hipEventRecord(net->branch_ev, net->side[0]));
                branch_pending = true;
            }
            for (int i = 0; i < nchain; ++i) {
                const ConvT& c = A.convs[chain[i]];
                const ConvG& g = net->cg[chain[i]];
                nb = 0;
                if (bf16t) {
                    if (conv_fwd16(net, c, g, chain[i], x16, ws, s, &nb)) return 1;
                } else {
                    if (conv_fwd(net, c, g, x, c.cin, P + c.w, ws + g.y, ws, s, &nb)) return 1;
                }
                if (i == 0 && blk.down >= 0 && !branch_aside) {
                    // (issued after conv1 so that both convolutions reading `cur` are adjacent)
                    if (bn(chain[0], nullptr, 1, nb)) return 1;
                    if (down_branch(s, false)) return 1;
                } else if (i + 1 < nchain) {
                    if (bn(chain[i], nullptr, 1, nb)) return 1;
                } else {
                    if (branch_pending) {
                        CILRS_HIP(hipStreamWaitEvent(s, net->branch_ev, 0));
                        branch_pending = false;
                    }
                    if (bn(chain[i], identity, 1, nb)) return 1;
                }
                x = ws + g.z;
                x16 = h16(ws, net->z16[chain[i]]);
            }
            cur = x;
            cur16 = x16;
        }
        if (bf16t) {         // features from the bf16 feature map
            RUN(net, "heads_fwd", 0.0, 0.0, s,
                launch_avgpool_f16(cur16, ws + net->combined, B, net->featHW, A.feat,
                                   A.feat + 128, 1, s));
            cur = nullptr;
        }
    } else {
        // eval: running statistics -> per-channel scale/shift (one launch for all 36 layers),
        // folded with ReLU / residual add into each conv's epilogue; no pre-BN tensor is stored
        if (!prep_cached) {
            RUN(net, "bn_fwd.eval", 0.0, 0.0, s,
                launch_bn_eval_stats_all(net->bn_table, P, R, ws, eps, s));
            net->prep_key = net->weights_key;
            net->prep_bufs[0] = bufs->params; net->prep_bufs[1] = bufs->bn_running;
            net->prep_bufs[2] = bufs->workspace;
            net->fold_key = 0;
        }
        if (!half) {
            if (conv_fwd(net, A.convs[0], net->cg[0], ws + net->x4, 4, ws + net->w4,
                         ws + net->cg[0].z, ws, s, nullptr, ws + net->cg[0].stats, 1)) return 1;
            RUN(net, "maxpool", 0.0, 4.0 * net->cg[0].M * 64 * 1.25, s,
                launch_maxpool_fwd(ws + net->cg[0].z, ws + net->pool, nullptr, B, net->H0,
                                   net->W0, 64, s));
        }
        cur = ws + net->pool;
        if (half) {
            // ---- fp16 trunk (BASELINE config 5): BatchNorm folded into fp16 weights, fp16 NHWC
            //      activations, v_mfma_f32_32x32x16_f16 with fp32 accumulation (infer_f16.hip) ----
            cilrs_half* w16 = reinterpret_cast<cilrs_half*>(ws + net->f16_w);
            float* b16 = ws + net->f16_bias;
            const int bf16 = half == 2;
            cilrs_half* act[5];
            for (int k = 0; k < 5; ++k) act[k] = reinterpret_cast<cilrs_half*>(ws + net->f16_act[k]);
            const bool fold_cached = net->weights_key != 0 && net->fold_key == net->weights_key &&
                                     net->fold_half == half && prep_cached;
            if (!fold_cached) {
                RUN(net, "transform", 0.0, 0.0, s,
                    launch_fold_bn_f16(net->f16_table, P, ws, w16, b16, bf16, s));
                RUN(net, "transform", 0.0, 0.0, s,
                    launch_fold_stem_f16(P + A.convs[0].w, ws + net->cg[0].stats,
                                         h16(ws, net->stem16_w), ws + net->stem16_b, bf16, s));
                net->fold_key = net->weights_key;
                net->fold_half = half;
            }
            // stem on the 16-bit pipe too (stem_f16.hip): conv 7x7/s2 + folded BN + ReLU from the
            // channel-padded fp32 image, 16-bit output parked in the (otherwise unused) fp32 stem
            // buffer, then the 16-bit max-pool straight into the first activation buffer
            {
                const ConvG& g0 = net->cg[0];
                RUN(net, "conv_fwd.stem", 2.0 * g0.M * 64 * 147, 16.0 * B * net->H * net->W +
                    2.0 * g0.M * 64, s,
                    launch_stem_f16(ws + net->x4, h16(ws, net->stem16_w), ws + net->stem16_b,
                                    h16(ws, g0.z), B, net->H, net->W, bf16, s));
                RUN(net, "maxpool", 0.0, 2.0 * g0.M * 64 * 1.25, s,
                    launch_maxpool_f16(h16(ws, g0.z), act[0], B, net->H0, net->W0, 64, bf16, s));
            }
            int ic = 0;                                  // index of the buffer holding `cur`
            auto conv16 = [&](int ci, const cilrs_half* x, const cilrs_half* residual,
                              cilrs_half* y, int relu) -> int {
                const ConvT& c = A.convs[ci];
                const ConvG& g = net->cg[ci];
                ConvF16Args a;
                memset(&a, 0, sizeof(a));
                a.x = x; a.w = w16 + net->f16_table.w16[ci - 1];
                a.bias = b16 + net->f16_table.bias[ci - 1];
                a.residual = residual; a.y = y; a.bf16 = bf16;
                a.N = B; a.H = g.H; a.W = g.W; a.Cin = c.cin; a.Ho = g.Ho; a.Wo = g.Wo;
                a.Cout = c.cout; a.K = c.k; a.stride = c.stride; a.pad = c.pad; a.relu = relu;
                const double bytes = 2.0 * ((double)B * g.H * g.W * c.cin + (double)g.M * c.cout *
                                            (residual ? 2.0 : 1.0) + (double)c.cout * c.k * c.k * c.cin);
                RUN(net, std::string("conv_fwd.") + kGroupName[c.group],
                    2.0 * g.M * c.cout * c.k * c.k * c.cin, bytes, s, launch_conv_f16(a, s));
                return 0;
            };
            for (const BlockT& blk : A.blocks) {
                // five rotating buffers: block input, two intermediates, projected identity, output
                const int it1 = (ic + 1) % 5, it2 = (ic + 2) % 5, iid = (ic + 3) % 5,
                          io = (ic + 4) % 5;
                if (conv16(blk.conv1, act[ic], nullptr, act[it1], 1)) return 1;
                const cilrs_half* identity = act[ic];
                if (blk.down >= 0) {
                    if (conv16(blk.down, act[ic], nullptr, act[iid], 0)) return 1;
                    identity = act[iid];
                }
                if (blk.conv3 >= 0) {
                    if (conv16(blk.conv2, act[it1], nullptr, act[it2], 1)) return 1;
                    if (conv16(blk.conv3, act[it2], identity, act[io], 1)) return 1;
                } else {
                    if (conv16(blk.conv2, act[it1], identity, act[io], 1)) return 1;
                }
                ic = io;
            }
            RUN(net, "heads_fwd", 0.0, 0.0, s,
                launch_avgpool_f16(act[ic], ws + net->combined, B, net->featHW, A.feat,
                                   A.feat + 128, bf16, s));
            cur = nullptr;                               // features already pooled into `combined`
        } else {
        // a layer with few output pixels (single-frame inference) takes the one-launch
        // latency kernel (conv_small.hip) instead of split-K implicit GEMM + reduce
        auto conv_eval = [&](const ConvT& c, const ConvG& g, const float* x, float* y, int relu,
                             const float* addend, int relu_post) -> int {
            // one 16-wave block per 16x16 tile: worth it while every block gets its own CU
            // CILRS_SMALL_BLOCKS / CILRS_SMALL_K: routing thresholds for tools/infer_ab.py
            static const int max_blocks =
                experiment_env("CILRS_SMALL_BLOCKS", kSmallConvBlocks);
            static const int max_k = experiment_env("CILRS_SMALL_K", kSmallConvK);
            if (cdiv(g.M, 16) * (c.cout / 16) > max_blocks || c.cin % 16 != 0 ||
                c.k * c.k * c.cin > max_k)
                return conv_fwd(net, c, g, x, c.cin, P + c.w, y, ws, s, nullptr, ws + g.stats,
                                relu, addend, relu_post);
            ConvSmallArgs a;
            memset(&a, 0, sizeof(a));
            a.x = x; a.w = P + c.w; a.y = y;
            a.scale = ws + g.stats + 2 * c.cout; a.shift = ws + g.stats + 3 * c.cout;
            a.addend = addend; a.relu = relu; a.relu_post = relu_post;
            a.N = B; a.H = g.H; a.W = g.W; a.Cin = c.cin; a.Ho = g.Ho; a.Wo = g.Wo;
            a.Cout = c.cout; a.K = c.k; a.stride = c.stride; a.pad = c.pad;
            RUN(net, std::string("conv_fwd.") + kGroupName[c.group],
                2.0 * g.M * c.cout * c.k * c.k * c.cin, 0.0, s, launch_conv_small(a, s));
            return 0;
        };
        for (const BlockT& blk : A.blocks) {
            const ConvT& c1 = A.convs[blk.conv1];
            const ConvT& c2 = A.convs[blk.conv2];
            const ConvG& g1 = net->cg[blk.conv1];
            const ConvG& g2 = net->cg[blk.conv2];
            if (conv_eval(c1, g1, cur, ws + g1.z, 1, nullptr, 0)) return 1;
            const float* identity = cur;
            if (blk.down >= 0) {
                const ConvT& cd = A.convs[blk.down];
                const ConvG& gd = net->cg[blk.down];
                if (conv_eval(cd, gd, cur, ws + gd.z, 0, nullptr, 0)) return 1;
                identity = ws + gd.z;
            }
            if (blk.conv3 >= 0) {      // Bottleneck: 1x1 -> 3x3 -> 1x1 (+ identity)
                const ConvT& c3 = A.convs[blk.conv3];
                const ConvG& g3 = net->cg[blk.conv3];
                if (conv_eval(c2, g2, ws + g1.z, ws + g2.z, 1, nullptr, 0)) return 1;
                if (conv_eval(c3, g3, ws + g2.z, ws + g3.z, 0, identity, 1)) return 1;
                cur = ws + g3.z;
            } else {
                if (conv_eval(c2, g2, ws + g1.z, ws + g2.z, 0, identity, 1)) return 1;
                cur = ws + g2.z;
            }
        }
        }
    }

    int* status = reinterpret_cast<int*>(reinterpret_cast<char*>(bufs->workspace) + net->status_b);
    const int feat = A.feat, comb = A.feat + 128;
    if (!train && B <= kHeadsSmallMaxB && A.variant == 0) {
        // ---- inference at control-loop batch sizes: 4 launches, commanded branch only ----
        RUN(net, "heads_fwd", 0.0, 0.0, s,
            launch_heads_small_pre(cur, net->featHW, speed, P + A.se0.w, P + A.se0.b, P + A.se3.w,
                                   P + A.se3.b, ws + net->combined, B, s));
        HeadsSmallArgs h;
        memset(&h, 0, sizeof(h));
        h.B = B;
        h.ncmd = A.ncmd;
        h.cmd = reinterpret_cast<const long long*>(command);
        for (int layer = 0; layer < 3; ++layer) {
            for (int k = 0; k < A.ncmd; ++k) {
                h.w[k] = P + A.br[k][layer].w;
                h.b[k] = P + A.br[k][layer].b;
            }
            const LinT& sp = layer == 0 ? A.sp0 : layer == 1 ? A.sp3 : A.sp5;
            h.w[A.ncmd] = P + sp.w;
            h.b[A.ncmd] = P + sp.b;
            h.in[0] = A.br[0][layer].in;
            h.in[1] = sp.in;
            h.x_ld = h.in[0];
            h.x[0] = layer == 0 ? ws + net->combined : layer == 1 ? ws + net->h1[0] : ws + net->h2[0];
            h.x[1] = layer == 0 ? ws + net->combined : layer == 1 ? ws + net->p1 : ws + net->p2;
            h.y[0] = layer == 0 ? ws + net->h1[0] : layer == 1 ? ws + net->h2[0] : controls;
            h.y[1] = layer == 0 ? ws + net->p1 : layer == 1 ? ws + net->p2 : pred_speed;
            h.out[0] = A.br[0][layer].out;
            h.out[1] = sp.out;
            h.y_ld[0] = h.out[0];
            h.y_ld[1] = h.out[1];
            h.relu = layer < 2;
            h.status = layer == 2 ? status : nullptr;
            RUN(net, "heads_fwd", 2.0 * B * (h.in[0] * h.out[0] + h.in[1] * h.out[1]),
                4.0 * (h.in[0] * h.out[0] + h.in[1] * h.out[1]), s,
                launch_heads_small_layer(h, s));
        }
        net->trained_fwd = false;
        net->last_dropout = 0.f;
        return 0;
    }

    // ---- avgpool + flatten -> combined[:, 0:512] ----
    if (cur != nullptr)
        RUN(net, "heads_fwd", 0.0, 0.0, s,
            launch_avgpool_fwd(cur, ws + net->combined, B, net->featHW, feat, comb, s));

    if (train) {   // backward needs the inputs of the heads
        if (launch_keep_head_inputs(speed, reinterpret_cast<const long long*>(command), ws + net->speed_in,
                                    reinterpret_cast<long long*>(reinterpret_cast<char*>(bufs->workspace) +
                                                                 net->cmd_b), B, s)) return 1;
    }
    const float pdrop = train ? dropout_p : 0.f;
    // Every layer of every chain that can run at the same time is ONE grouped launch
    // (heads_gemm.hip): 7 launches for the whole head instead of ~25 on five side streams.
    auto fwd_group = [&](HGemmGroup& g, const float* x, int x_ld, const LinT& l, float* y, int y_ld,
                         unsigned long long drop_stream) {
        memset(&g, 0, sizeof(g));
        g.A = x; g.lda = x_ld; g.B = P + l.w; g.ldb = l.in; g.bias = P + l.b;
        g.C = y; g.ldc = y_ld; g.M = B; g.N = l.out; g.K = l.in; g.drop_stream = drop_stream;
    };
    auto run_fwd = [&](HGemmArgs& h, int n, int relu) -> int {
        h.ngroups = n; h.relu = relu; h.accumulate = 0; h.drop_p = pdrop; h.seed = seed;
        double fl = 0.0;
        for (int i = 0; i < n; ++i) fl += 2.0 * h.g[i].M * h.g[i].N * h.g[i].K;
        RUN(net, "heads_fwd", fl, 0.0, s, launch_hgemm(0, h, s));
        return 0;
    };
    HGemmArgs h;
    // ---- speed encoder (autonomous_drive.py:371-374, 391) ----
    memset(&h, 0, sizeof(h));
    fwd_group(h.g[0], speed, 1, A.se0, ws + net->s1, 128, 0);
    if (run_fwd(h, 1, 1)) return 1;
    fwd_group(h.g[0], ws + net->s1, 128, A.se3, ws + net->combined + feat, comb, kNoDrop);
    if (run_fwd(h, 1, 1)) return 1;
    // ---- the branches (all evaluated, :394-396) + speed predictor (:383-387, 393), layer by layer;
    //      dropout streams: 1 + 2k / 2 + 2k for branch k, 2 NC + 1 for the speed predictor (= 9 for
    //      the reference's four commands)
    const int NC = A.ncmd;
    for (int k = 0; k < NC; ++k)
        fwd_group(h.g[k], ws + net->combined, comb, A.br[k][0], ws + net->h1[k], 256, 1 + 2 * k);
    fwd_group(h.g[NC], ws + net->combined, comb, A.sp0, ws + net->p1, 256, 2 * NC + 1);
    if (run_fwd(h, NC + 1, 1)) return 1;
    for (int k = 0; k < NC; ++k)
        fwd_group(h.g[k], ws + net->h1[k], 256, A.br[k][1], ws + net->h2[k], 256, 2 + 2 * k);
    fwd_group(h.g[NC], ws + net->p1, 256, A.sp3, ws + net->p2, 256, kNoDrop);
    if (run_fwd(h, NC + 1, 1)) return 1;
    for (int k = 0; k < NC; ++k)
        fwd_group(h.g[k], ws + net->h2[k], 256, A.br[k][2], ws + net->all_out + (size_t)k * B * 4,
                  4, kNoDrop);
    fwd_group(h.g[NC], ws + net->p2, 256, A.sp5, pred_speed, 1, kNoDrop);
    if (run_fwd(h, NC + 1, 0)) return 1;
    // gathered by command (:397-398)
    RUN(net, "heads_fwd", 0.0, 0.0, s,
        launch_branch_gather(ws + net->all_out, reinterpret_cast<const long long*>(command),
                             controls, B, NC, status, s));
    net->trained_fwd = train != 0;
    net->last_dropout = pdrop;
    return 0;
}

static int check_bufs(const cilrs_net* net, const cilrs_buffers* bufs, bool need_grads) {
    CILRS_CHECK(net != nullptr && bufs != nullptr, "net / buffers missing");
    CILRS_CHECK(bufs->params && bufs->bn_running && bufs->bn_nbt && bufs->workspace,
                "cilrs_buffers has NULL members");
    CILRS_CHECK(!need_grads || bufs->grads, "gradient arena missing");
    CILRS_CHECK(((uintptr_t)bufs->params & 15) == 0 && ((uintptr_t)bufs->workspace & 255) == 0,
                "params must be 16-byte and workspace 256-byte aligned");
    return 0;
}

int cilrs_net_forward(cilrs_net* net, const cilrs_buffers* bufs, const float* image, long sn,
                      long sc, long sh, long sw, const float* speed, const int64_t* command,
                      int train, float dropout_p, uint64_t seed, float* controls,
                      float* pred_speed, void* stream) {
    if (check_bufs(net, bufs, false)) return 1;
    CILRS_CHECK(image && speed && command && controls && pred_speed, "forward: NULL tensor");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    float* ws = reinterpret_cast<float*>(bufs->workspace);
    RUN(net, "transform", 0.0, 0.0, s,
        launch_nchw3_to_nhwc4(image, ws + net->x4, net->B, net->H, net->W, sn, sc, sh, sw, s));
    return forward_from_x4(net, bufs, speed, command, train, dropout_p, seed, controls,
                           pred_speed, s);
}

int cilrs_net_forward_u8(cilrs_net* net, const cilrs_buffers* bufs, const uint8_t* frames,
                         const float* speed, const int64_t* command, float* controls,
                         float* pred_speed, void* stream) {
    if (check_bufs(net, bufs, false)) return 1;
    CILRS_CHECK(frames && speed && command && controls && pred_speed, "forward_u8: NULL tensor");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    float* ws = reinterpret_cast<float*>(bufs->workspace);
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    RUN(net, "transform", 0.0, 0.0, s,
        launch_u8hwc_to_nhwc4(frames, ws + net->x4, (size_t)net->B * net->H * net->W, mean, stdv,
                              s));
    return forward_from_x4(net, bufs, speed, command, 0, 0.f, 0, controls, pred_speed, s);
}

int cilrs_net_forward_camera(cilrs_net* net, const cilrs_buffers* bufs, const uint8_t* frames,
                             int src_h, int src_w, int pixel_stride, long row_stride,
                             long frame_stride, const float* speed, const int64_t* command,
                             float* controls, float* pred_speed, void* stream) {
    if (check_bufs(net, bufs, false)) return 1;
    CILRS_CHECK(frames && speed && command && controls && pred_speed,
                "forward_camera: NULL tensor");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    float* ws = reinterpret_cast<float*>(bufs->workspace);
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    RUN(net, "transform", 0.0, 0.0, s,
        launch_camera_to_nhwc4(frames, ws + net->x4, net->B, src_h, src_w, pixel_stride,
                               row_stride, frame_stride, net->H, net->W, mean, stdv, s));
    return forward_from_x4(net, bufs, speed, command, 0, 0.f, 0, controls, pred_speed, s);
}

int cilrs_net_forward_u8_f16(cilrs_net* net, const cilrs_buffers* bufs, const uint8_t* frames,
                             const float* speed, const int64_t* command, float* controls,
                             float* pred_speed, void* stream) {
    if (check_bufs(net, bufs, false)) return 1;
    CILRS_CHECK(frames && speed && command && controls && pred_speed,
                "forward_u8_f16: NULL tensor");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    float* ws = reinterpret_cast<float*>(bufs->workspace);
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    RUN(net, "transform", 0.0, 0.0, s,
        launch_u8hwc_to_nhwc4(frames, ws + net->x4, (size_t)net->B * net->H * net->W, mean, stdv,
                              s));
    return forward_from_x4(net, bufs, speed, command, 0, 0.f, 0, controls, pred_speed, s, 1);
}

int cilrs_net_forward_u8_bf16(cilrs_net* net, const cilrs_buffers* bufs, const uint8_t* frames,
                              const float* speed, const int64_t* command, float* controls,
                              float* pred_speed, void* stream) {
    if (check_bufs(net, bufs, false)) return 1;
    CILRS_CHECK(frames && speed && command && controls && pred_speed,
                "forward_u8_bf16: NULL tensor");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    float* ws = reinterpret_cast<float*>(bufs->workspace);
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    RUN(net, "transform", 0.0, 0.0, s,
        launch_u8hwc_to_nhwc4(frames, ws + net->x4, (size_t)net->B * net->H * net->W, mean, stdv,
                              s));
    return forward_from_x4(net, bufs, speed, command, 0, 0.f, 0, controls, pred_speed, s, 2);
}

// ------------------------------------------------------------------------------------------------
// single-frame inference as ONE persistent launch (infer_b1.hip)
// ------------------------------------------------------------------------------------------------
// eval-mode weight-derived state (the same two kernels and cache keys as the eager eval path)
static int eval_prep(cilrs_net* net, const cilrs_buffers* bufs, hipStream_t s) {
    const Arch& A = *net->A;
    float* ws = reinterpret_cast<float*>(bufs->workspace);
    const bool same_bufs = net->prep_bufs[0] == (const void*)bufs->params &&
                           net->prep_bufs[1] == (const void*)bufs->bn_running &&
                           net->prep_bufs[2] == (const void*)bufs->workspace;
    if (net->weights_key != 0 && same_bufs && net->prep_key == net->weights_key) return 0;
    RUN(net, "transform", 0.0, 0.0, s,
        launch_pad_cin3_to_4(bufs->params + A.convs[0].w, ws + net->w4, 64 * 49, s));
    RUN(net, "bn_fwd.eval", 0.0, 0.0, s,
        launch_bn_eval_stats_all(net->bn_table, bufs->params, bufs->bn_running, ws, 1e-5f, s));
    net->prep_key = net->weights_key;
    net->prep_bufs[0] = bufs->params; net->prep_bufs[1] = bufs->bn_running;
    net->prep_bufs[2] = bufs->workspace;
    net->fold_key = 0;
    return 0;
}

// Stage table of the persistent kernel for this plan on a grid of `nblk` workgroups.
static int b1_build(cilrs_net* net, int nblk) {
    const Arch& A = *net->A;
    std::vector<B1Stage>& T = net->b1_host;
    T.clear();
    auto fb = [](size_t floats) { return (unsigned)(floats * sizeof(float)); };
    bool magic_ok = true;
    auto magic20 = [&](int d, int xmax) {        // x / d == (x * magic) >> 20 for 0 <= x <= xmax
        const unsigned mg = ((1u << 20) + (unsigned)d - 1u) / (unsigned)d;
        for (int x = 0; x <= xmax; ++x)
            if ((unsigned)(((unsigned long long)x * mg) >> 20) != (unsigned)(x / d) ||
                (unsigned long long)x * mg >= (1ull << 32))
                magic_ok = false;
        return mg;
    };
    auto conv_desc = [&](int ci, size_t x, bool has_add, size_t add, int relu, int relu_post) {
        const ConvT& c = A.convs[ci];
        const ConvG& g = net->cg[ci];
        B1Conv d;
        memset(&d, 0, sizeof(d));
        d.x_off = fb(x); d.y_off = fb(g.z); d.add_off = has_add ? fb(add) : 0u;
        d.scale_off = fb(g.stats + 2 * (size_t)c.cout); d.shift_off = fb(g.stats + 3 * (size_t)c.cout);
        d.H = g.H; d.W = g.W; d.Wo = g.Wo; d.Cout = c.cout; d.K = c.k;
        d.stride = c.stride; d.M = g.M; d.nmt = cdiv(g.M, 16);
        d.ntiles = d.nmt * (c.cout / 16);
        d.relu = relu; d.relu_post = relu_post; d.has_add = has_add ? 1 : 0;
        if (ci == 0) {      // stem: channel-padded image, padded weights in the workspace
            d.Cin = 4; d.w_off = fb(net->w4); d.S = cdiv(c.k * c.k, 4); d.cshift = 0;
        } else {
            d.Cin = c.cin; d.w_off = fb(c.w);
            const int cgn = c.cin / 16;
            d.S = c.k * c.k * cgn;
            while ((1 << d.cshift) < cgn) ++d.cshift;
        }
        d.krow4 = c.k * c.k * d.Cin * 4;
        d.wo_magic = magic20(g.Wo, d.nmt * 16 + 16);
        d.nmt_magic = magic20(d.nmt, d.ntiles);
        return d;
    };
    auto check_conv = [&](int ci) -> bool {
        const ConvT& c = A.convs[ci];
        const int cgn = c.cin / 16;
        return c.cin % 16 == 0 && c.cout % 16 == 0 && (cgn & (cgn - 1)) == 0 &&
               ((c.k == 3 && c.pad == 1) || (c.k == 1 && c.pad == 0));
    };
    // Shape of one stage: waves per unit (wpt), workgroups per tile (split of the reduction index,
    // ks) and channel tiles per unit (nt) of its one or two convolutions.  All units in ONE pass of
    // slots; a wave holds at most kB1MaxK weight fragments.  Cost = k-groups the busiest CU streams
    // after the barrier (activation fragments; the weight fragments are in flight before it and
    // count half; 2 KB each at ~70 GB/s per CU: 35 per microsecond) + ~50 for a ticketed combine.
    constexpr int kB1MaxK = 8;
    bool plan_ok = true;
    auto push_conv_stage = [&](int type, const B1Conv& c0, const B1Conv* c1) {
        B1Stage st;
        memset(&st, 0, sizeof(st));
        st.type = type;
        struct Opt { int ks, nt; };
        const Opt opts[5] = {{1, 1}, {2, 1}, {3, 1}, {4, 1}, {1, 2}};
        int best_cost = 1 << 30, bw = 0, b0 = 0, b1 = 0;
        auto units_of = [&](const B1Conv& c, const Opt& o) { return c.ntiles / o.nt * o.ks; };
        auto feasible = [&](const B1Conv& c, const Opt& o, int w) {
            if (o.nt == 2 && (c.Cout % 32 != 0 || type == B1_STEM)) return false;
            if (o.ks > 1 && (type == B1_STEM || o.ks > c.S)) return false;
            return cdiv(cdiv(c.S, o.ks), w) * o.nt <= kB1MaxK;
        };
        for (int w : {16, 8, 4, 2})
            for (int i0 = 0; i0 < 5; ++i0)
                for (int i1 = 0; i1 < (c1 ? 5 : 1); ++i1) {
                    if (!feasible(c0, opts[i0], w) || (c1 && !feasible(*c1, opts[i1], w))) continue;
                    const int units = units_of(c0, opts[i0]) + (c1 ? units_of(*c1, opts[i1]) : 0);
                    if (units > nblk * (16 / w)) continue;
                    auto load = [&](const B1Conv& c, const Opt& o) {
                        const int sp = cdiv(c.S, o.ks);
                        return 2 * sp + sp * o.nt;          // activations + half the weights, x2
                    };
                    int l = load(c0, opts[i0]);
                    if (c1 && load(*c1, opts[i1]) > l) l = load(*c1, opts[i1]);
                    const int cost = cdiv(units, nblk) * l / 2 +
                                     ((opts[i0].ks > 1 || (c1 && opts[i1].ks > 1)) ? 50 : 0);
                    if (cost < best_cost) { best_cost = cost; bw = w; b0 = i0; b1 = i1; }
                }
        if (bw == 0) { plan_ok = false; bw = 2; }
        auto fin = [&](B1Conv d, const Opt& o, int ticket0, size_t slab) {
            d.ksplit = o.ks; d.nt = o.nt;
            d.sper = cdiv(d.S, o.ks); d.per = cdiv(d.sper, bw);
            d.nunits = d.ntiles / o.nt * o.ks;
            d.ks_magic = magic20(o.ks, d.nunits);
            d.nmt_magic = magic20(d.nmt, d.ntiles);
            d.ticket0 = ticket0;
            d.slab_off = fb(slab);
            if (o.ks > 1 && ticket0 + d.ntiles > kB1Tickets) plan_ok = false;
            if (o.ks > 1 && (size_t)o.ks * d.nmt * 16 * d.Cout > net->b1_slab_floats / 2) plan_ok = false;
            return d;
        };
        st.c[0] = fin(c0, opts[b0], 0, net->b1_slabs);
        if (c1) st.c[1] = fin(*c1, opts[b1], c0.ntiles, net->b1_slabs + net->b1_slab_floats / 2);
        st.wpt = bw;
        st.nunits0 = st.c[0].nunits;
        st.total_units = st.c[0].nunits + (c1 ? st.c[1].nunits : 0);
        // same lane-level plan as the previous stage?  (everything but the tensor bases, the
        // epilogue flags and the BatchNorm tables equal)
        if (!T.empty() && T.back().type == B1_CONV && type == B1_CONV && !c1 &&
            T.back().total_units == T.back().nunits0 && T.back().wpt == st.wpt) {
            B1Conv p0 = T.back().c[0], n0 = st.c[0];
            p0.x_off = p0.y_off = p0.add_off = p0.w_off = p0.scale_off = p0.shift_off = 0;
            n0.x_off = n0.y_off = n0.add_off = n0.w_off = n0.scale_off = n0.shift_off = 0;
            p0.relu = p0.relu_post = p0.has_add = n0.relu = n0.relu_post = n0.has_add = 0;
            st.same_shape = memcmp(&p0, &n0, sizeof(B1Conv)) == 0;
        }
        if (experiment_env("CILRS_B1_FINE", 0))
            fprintf(stderr, "b1 stage %2d: wpt %2d units %4d (ks %d nt %d per %d | ks %d nt %d per %d) same %d\n",
                    (int)T.size(), st.wpt, st.total_units, st.c[0].ksplit, st.c[0].nt, st.c[0].per,
                    c1 ? st.c[1].ksplit : 0, c1 ? st.c[1].nt : 0, c1 ? st.c[1].per : 0, st.same_shape);
        T.push_back(st);
    };
    {   // uint8 frame -> normalised NHWC4
        B1Stage st;
        memset(&st, 0, sizeof(st));
        st.type = B1_PRE; st.pH = net->H; st.pW = net->W; st.dst_off = fb(net->x4);
        // (a workgroup parks its run of the frame's bytes in 32 KB of LDS: infer_b1.hip pre_stage)
        CILRS_CHECK(3 * cdiv(net->H * net->W, nblk) + 8 <= 16 * 512 * 4,
                    "persistent kernel: frame too large for %d workgroups", nblk);
        // ... and the speed encoder, evaluated by one block beside the pixel work
        st.h.se_w0 = fb(A.se0.w); st.h.se_b0 = fb(A.se0.b);
        st.h.se_w1 = fb(A.se3.w); st.h.se_b1 = fb(A.se3.b);
        st.h.y_off[0] = fb(net->s1);
        st.cmd_off = fb(net->b1_cmd);
        T.push_back(st);
    }
    push_conv_stage(B1_STEM, conv_desc(0, net->x4, false, 0, 1, 0), nullptr);
    {   // max-pool 3x3/s2/p1
        B1Stage st;
        memset(&st, 0, sizeof(st));
        st.type = B1_POOL; st.src_off = fb(net->cg[0].z); st.dst_off = fb(net->pool);
        st.pH = net->H0; st.pW = net->W0; st.pC = 64; st.pHo = net->H1; st.pWo = net->W1;
        T.push_back(st);
    }
    size_t cur = net->pool;
    for (const BlockT& blk : A.blocks) {
        CILRS_CHECK(blk.conv3 < 0, "infer_b1: BasicBlock networks only");
        CILRS_CHECK(check_conv(blk.conv1) && check_conv(blk.conv2) &&
                        (blk.down < 0 || check_conv(blk.down)),
                    "infer_b1: unsupported convolution shape");
        const B1Conv c1 = conv_desc(blk.conv1, cur, false, 0, 1, 0);
        size_t identity = cur;
        if (blk.down >= 0) {
            const B1Conv cd = conv_desc(blk.down, cur, false, 0, 0, 0);
            push_conv_stage(B1_CONV, c1, &cd);
            identity = net->cg[blk.down].z;
        } else {
            push_conv_stage(B1_CONV, c1, nullptr);
        }
        push_conv_stage(B1_CONV, conv_desc(blk.conv2, net->cg[blk.conv1].z, true, identity, 0, 1),
                        nullptr);
        cur = net->cg[blk.conv2].z;
    }
    // heads: commanded branch (chain 0) and speed predictor (chain 1), layer by layer
    for (int layer = 0; layer < 3; ++layer) {
        B1Stage st;
        memset(&st, 0, sizeof(st));
        st.type = B1_HEAD;
        st.cmd_off = fb(net->b1_cmd);
        B1Head& h = st.h;
        for (int k = 0; k < 4; ++k) {
            h.w_off[k] = fb(A.br[k][layer].w);
            h.b_off[k] = fb(A.br[k][layer].b);
        }
        const LinT& sp = layer == 0 ? A.sp0 : layer == 1 ? A.sp3 : A.sp5;
        h.w_off[4] = fb(sp.w); h.b_off[4] = fb(sp.b);
        h.in[0] = A.br[0][layer].in; h.in[1] = sp.in;
        h.out[0] = A.br[0][layer].out; h.out[1] = sp.out;
        h.relu = layer < 2; h.first = layer == 0; h.last = layer == 2;
        h.x_off[0] = fb(layer == 0 ? net->s1 : layer == 1 ? net->h1[0] : net->h2[0]);
        h.x_off[1] = fb(layer == 1 ? net->p1 : net->p2);
        h.y_off[0] = fb(layer == 0 ? net->h1[0] : net->h2[0]);
        h.y_off[1] = fb(layer == 0 ? net->p1 : net->p2);
        if (layer == 0) {
            h.feat_off = fb(cur); h.featHW = net->featHW; h.featC = A.feat;
            CILRS_CHECK(A.feat == 512 && h.in[0] == 640 && h.in[1] == 512,
                        "infer_b1: head geometry");
        }
        CILRS_CHECK(h.in[0] % 4 == 0 && h.in[1] % 4 == 0 && h.in[0] <= 768 && h.in[1] <= 768,
                    "infer_b1: head width");
        T.push_back(st);
    }
    CILRS_CHECK(magic_ok, "infer_b1: division constants do not cover this geometry");
    CILRS_CHECK(plan_ok, "infer_b1: no one-pass tiling of a stage on %d workgroups", nblk);
    CILRS_CHECK(A.convs[0].k == 7 && A.convs[0].pad == 3 && A.convs[0].stride == 2,
                "infer_b1: stem geometry");
    CILRS_CHECK((int)T.size() <= kB1MaxStages, "infer_b1: %d stages", (int)T.size());
    return 0;
}

static int b1_launch(cilrs_net* net, const cilrs_buffers* bufs, const uint8_t* frame,
                     int first_stage, const float* speed, const int64_t* command, float* controls,
                     float* pred_speed, void* stream, int* done = nullptr, int seq = 0) {
    if (check_bufs(net, bufs, false)) return 1;
    CILRS_CHECK((frame || first_stage == 1) && speed && command && controls && pred_speed,
                "forward_u8_b1: NULL tensor");
    CILRS_CHECK(net->B == 1 && net->A->variant == 0 && net->b1_table != 0,
                "forward_u8_b1: the persistent kernel serves the reference network at batch 1");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    float* ws = reinterpret_cast<float*>(bufs->workspace);
    net->ws_base = ws;
    if (net->b1_blocks < 0) {
        int blocks = 0;
        if (infer_b1_grid(&blocks)) return 1;
        CILRS_CHECK(blocks >= 8, "forward_u8_b1: the device cannot keep the persistent grid resident");
        if (b1_build(net, blocks)) return 1;
        net->b1_blocks = blocks;
    }
    if (zero_counters_once(net, bufs->workspace, s)) return 1;     // (also: status words of a fresh workspace)
    if (eval_prep(net, bufs, s)) return 1;
    if (net->b1_ready_for != bufs->workspace) {
        CILRS_HIP(hipMemsetAsync(ws + net->b1_sync, 0, kB1SyncInts * sizeof(int), s));
            CILRS_HIP(hipMemcpyAsync(ws + net->b1_table, net->b1_host.data(),
                                 net->b1_host.size() * sizeof(B1Stage), hipMemcpyHostToDevice, s));
        net->b1_ready_for = bufs->workspace;
    }
    B1Launch a;
    memset(&a, 0, sizeof(a));
    a.table = reinterpret_cast<const B1Stage*>(ws + net->b1_table);
    a.nstages = (int)net->b1_host.size();
    a.first_stage = first_stage;
    a.ws = ws; a.ws_bytes = net->ws_bytes;
    a.params = bufs->params; a.param_bytes = net->A->arena_floats * sizeof(float);
    a.done = done; a.seq = seq;
    a.frame = frame; a.speed = speed; a.cmd = reinterpret_cast<const long long*>(command);
    a.controls = controls; a.pred_speed = pred_speed;
    a.sync = reinterpret_cast<int*>(ws + net->b1_sync);
    a.status = reinterpret_cast<int*>(reinterpret_cast<char*>(bufs->workspace) + net->status_b);
    // CILRS_B1_STAMPS=1: block 0 records its clock at every stage (cilrs_net_b1_stage_us)
    static const int stamps_on = getenv("CILRS_B1_STAMPS") ? atoi(getenv("CILRS_B1_STAMPS")) : 0;
    a.stamps = stamps_on ? reinterpret_cast<long long*>(ws + net->b1_stamps) : nullptr;
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    for (int i = 0; i < 3; ++i) { a.mean[i] = mean[i]; a.stdv[i] = stdv[i]; }
    RUN(net, "infer_b1", 2.0 * 2.798e9 / 2.0, 0.0, s, launch_infer_b1(a, net->b1_blocks, s));
    net->trained_fwd = false;
    net->last_dropout = 0.f;
    return 0;
}

int cilrs_net_forward_u8_b1(cilrs_net* net, const cilrs_buffers* bufs, const uint8_t* frame,
                            const float* speed, const int64_t* command, float* controls,
                            float* pred_speed, void* stream) {
    return b1_launch(net, bufs, frame, 0, speed, command, controls, pred_speed, stream);
}

int cilrs_net_forward_camera_b1(cilrs_net* net, const cilrs_buffers* bufs, const uint8_t* frame,
                                int src_h, int src_w, int pixel_stride, long row_stride,
                                const float* speed, const int64_t* command, float* controls,
                                float* pred_speed, int sync, void* stream) {
    if (check_bufs(net, bufs, false)) return 1;
    CILRS_CHECK(frame != nullptr, "forward_camera_b1: NULL frame");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    float* ws = reinterpret_cast<float*>(bufs->workspace);
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    RUN(net, "transform", 0.0, 0.0, s,
        launch_camera_to_nhwc4(frame, ws + net->x4, 1, src_h, src_w, pixel_stride, row_stride,
                               (long)src_h * row_stride, net->H, net->W, mean, stdv, s));
    if (b1_launch(net, bufs, nullptr, 1, speed, command, controls, pred_speed, stream)) return 1;
    if (sync) CILRS_HIP(hipStreamSynchronize(s));
    return 0;
}

int cilrs_net_forward_u8_b1_post(cilrs_net* net, const cilrs_buffers* bufs, const uint8_t* frame,
                                 const float* speed, const int64_t* command, float* controls,
                                 float* pred_speed, int* done, int seq, void* stream) {
    CILRS_CHECK(done != nullptr, "forward_u8_b1_post: NULL completion word");
    return b1_launch(net, bufs, frame, 0, speed, command, controls, pred_speed, stream, done, seq);
}

int cilrs_net_forward_u8_b1_sync(cilrs_net* net, const cilrs_buffers* bufs, const uint8_t* frame,
                                 const float* speed, const int64_t* command, float* controls,
                                 float* pred_speed, void* stream) {
    if (cilrs_net_forward_u8_b1(net, bufs, frame, speed, command, controls, pred_speed, stream))
        return 1;
    CILRS_HIP(hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream)));
    return 0;
}

int cilrs_net_b1_stage_us(cilrs_net* net, const cilrs_buffers* bufs, float* start_us,
                          float* work_us, int cap) {
    CILRS_CHECK(net && bufs && bufs->workspace && start_us && work_us, "b1_stage_us: NULL");
    CILRS_CHECK(net->b1_table != 0 && net->b1_blocks > 0, "b1_stage_us: no persistent launch yet");
    const int n = (int)net->b1_host.size();
    CILRS_CHECK(cap >= n, "b1_stage_us: need room for %d stages", n);
    std::vector<long long> h(10 * (kB1MaxStages + 1));
    CILRS_HIP(hipMemcpy(h.data(), reinterpret_cast<float*>(bufs->workspace) + net->b1_stamps,
                        h.size() * sizeof(long long), hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) {          // 100 MHz clock
        start_us[i] = (float)((h[i] - h[0]) * 0.01);
        work_us[i] = (float)((h[kB1MaxStages + 1 + i] - h[i]) * 0.01);
    }
    if (experiment_env("CILRS_B1_FINE", 0)) {         // when each workgroup finished each stage (us after its start)
        const int nb = net->b1_blocks;
        std::vector<long long> d((size_t)n * nb);
        CILRS_HIP(hipMemcpy(d.data(), reinterpret_cast<long long*>(reinterpret_cast<float*>(bufs->workspace) +
                                                                  net->b1_stamps) + 10 * (kB1MaxStages + 1),
                            d.size() * sizeof(long long), hipMemcpyDeviceToHost));
        for (int i = 0; i < n; ++i) {
            std::vector<double> t(nb);
            for (int b = 0; b < nb; ++b) t[b] = (d[(size_t)i * nb + b] - h[i]) * 0.01;
            std::vector<double> srt = t;
            std::sort(srt.begin(), srt.end());
            int worst = 0;
            for (int b = 0; b < nb; ++b) if (t[b] > t[worst]) worst = b;
            fprintf(stderr, "stage %2d done: min %5.2f p50 %5.2f p90 %5.2f max %5.2f (block %d; block 0 %5.2f)\n",
                    i, srt[0], srt[nb / 2], srt[nb * 9 / 10], srt[nb - 1], worst, t[0]);
            if (experiment_env("CILRS_B1_DUMP", 0) && (i == 4 || i == 12 || i == 20 || i == 32)) {
                fprintf(stderr, "stage %2d per block:", i);
                for (int b = 0; b < nb; ++b) fprintf(stderr, " %.2f", t[b]);
                fprintf(stderr, "\n");
            }
        }
    }
    if (experiment_env("CILRS_B1_FINE", 0))           // conv stages: block 0 / wave 0 inside the stage
        for (int i = 0; i < n; ++i) {
            const long long* f = &h[2 * (kB1MaxStages + 1) + 8 * i];
            fprintf(stderr, "stage %2d fine:", i);
            for (int k = 0; k < 8; ++k) fprintf(stderr, " %6.2f", (f[k] - h[i]) * 0.01);
            fprintf(stderr, "\n");
        }
    return 0;
}

int cilrs_net_b1_set_epoch(cilrs_net* net, const cilrs_buffers* bufs, int value, void* stream) {
    CILRS_CHECK(net && bufs && bufs->workspace, "b1_set_epoch: NULL");
    CILRS_CHECK(net->b1_table != 0 && net->b1_ready_for == bufs->workspace,
                "b1_set_epoch: no persistent launch on this workspace yet");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    CILRS_HIP(hipStreamSynchronize(s));
    // the eight arrival shards and the epoch base are equal between launches: re-base all nine
    std::vector<int> h(9 * 32, 0);
    for (int i = 0; i < 9; ++i) h[i * 32] = value;
    CILRS_HIP(hipMemcpy(reinterpret_cast<float*>(bufs->workspace) + net->b1_sync, h.data(),
                        h.size() * sizeof(int), hipMemcpyHostToDevice));
    return 0;
}

int cilrs_net_wino_convs(cilrs_net* net) { return net ? net->wino_table.n : 0; }

int cilrs_net_b1_stages(cilrs_net* net) {
    if (!net || net->b1_table == 0) return 0;
    if (net->b1_blocks < 0) return -1;             // not launched yet
    return (int)net->b1_host.size();
}

// Same as cilrs_net_forward_u8, replayed from a cached hipGraph (one launch per frame instead of
// ~80): the B=1 control-loop path (autonomous_drive.py:908-920) is launch-latency bound.  The
// graph is re-captured when any pointer changes.  `stream` must not be the legacy NULL stream.
static int forward_u8_graph(cilrs_net* net, const cilrs_buffers* bufs, const uint8_t* frames,
                            const float* speed, const int64_t* command, float* controls,
                            float* pred_speed, void* stream, int half) {
    auto eager = half == 2 ? cilrs_net_forward_u8_bf16
                 : half ? cilrs_net_forward_u8_f16 : cilrs_net_forward_u8;
    if (check_bufs(net, bufs, false)) return 1;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    CILRS_CHECK(s != nullptr, "forward_u8_graph: capture needs a non-default stream");
    const void* key[8] = {bufs->params, bufs->bn_running, bufs->workspace, frames, speed, command,
                          controls, pred_speed};
    bool same = net->graph_exec != nullptr && net->graph_half == half &&
                net->graph_wkey == net->weights_key;
    for (int i = 0; i < 8 && same; ++i) same = key[i] == net->graph_key[i];
    if (!same) {
        // first call eager: function attributes, side streams, events; and whenever the weights
        // key moved, so that the weight-derived state is rebuilt OUTSIDE the capture and the graph
        // holds only the per-frame kernels
        // ... and whenever the 16-bit fold state is stale (another mode ran in between): the fold
        // kernels must not be captured into the per-frame graph
        const bool fold_stale = half && (net->fold_key != net->weights_key || net->fold_half != half);
        if (!net->warmed || net->prep_key != net->weights_key || net->weights_key == 0 || fold_stale) {
            if (ensure_streams(net)) return 1;
            if (eager(net, bufs, frames, speed, command, controls, pred_speed, stream)) return 1;
            CILRS_HIP(hipStreamSynchronize(s));
            net->warmed = true;
        }
        const bool prof = net->prof.on;
        net->prof.on = false;
        hipGraph_t graph = nullptr;
        CILRS_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        const int rc = eager(net, bufs, frames, speed, command, controls, pred_speed, stream);
        const hipError_t e = hipStreamEndCapture(s, &graph);
        net->prof.on = prof;
        CILRS_CHECK(rc == 0, "forward_u8_graph: capture failed: %s", last_error());
        CILRS_CHECK(e == hipSuccess && graph != nullptr, "hipStreamEndCapture failed: %s",
                    hipGetErrorString(e));
        if (net->graph_exec) { (void)hipGraphExecDestroy(net->graph_exec); net->graph_exec = nullptr; }
        const hipError_t e2 = hipGraphInstantiate(&net->graph_exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        CILRS_CHECK(e2 == hipSuccess, "hipGraphInstantiate failed: %s", hipGetErrorString(e2));
        for (int i = 0; i < 8; ++i) net->graph_key[i] = key[i];
        net->graph_half = half;
        net->graph_wkey = net->weights_key;
    }
    CILRS_HIP(hipGraphLaunch(net->graph_exec, s));
    return 0;
}

int cilrs_net_forward_u8_graph(cilrs_net* net, const cilrs_buffers* bufs, const uint8_t* frames,
                               const float* speed, const int64_t* command, float* controls,
                               float* pred_speed, void* stream) {
    return forward_u8_graph(net, bufs, frames, speed, command, controls, pred_speed, stream, 0);
}

int cilrs_net_forward_u8_f16_graph(cilrs_net* net, const cilrs_buffers* bufs,
                                   const uint8_t* frames, const float* speed,
                                   const int64_t* command, float* controls, float* pred_speed,
                                   void* stream) {
    return forward_u8_graph(net, bufs, frames, speed, command, controls, pred_speed, stream, 1);
}

int cilrs_net_forward_u8_bf16_graph(cilrs_net* net, const cilrs_buffers* bufs,
                                    const uint8_t* frames, const float* speed,
                                    const int64_t* command, float* controls, float* pred_speed,
                                    void* stream) {
    return forward_u8_graph(net, bufs, frames, speed, command, controls, pred_speed, stream, 2);
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
static int backward_heads(cilrs_net* net, const cilrs_buffers* bufs, const float* dcontrols,
                          const float* dps, const int64_t* command_unused, hipStream_t s);

int cilrs_net_backward(cilrs_net* net, const cilrs_buffers* bufs, const float* dcontrols,
                       const float* dpred_speed, int seg_begin, int seg_end, void* stream) {
    if (check_bufs(net, bufs, true)) return 1;
    CILRS_CHECK(net->trained_fwd, "backward needs a preceding train-mode forward on this plan");
    CILRS_CHECK(0 <= seg_begin && seg_begin <= seg_end && seg_end <= 6, "bad segment range");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const Arch& A = *net->A;
    float* ws = reinterpret_cast<float*>(bufs->workspace);
    net->ws_base = ws;
    const float* P = bufs->params;
    float* Gp = bufs->grads;
    const int B = net->B;
    auto last_conv = [&](const BlockT& blk) { return blk.conv3 >= 0 ? blk.conv3 : blk.conv2; };
    // cilrs_net_backward_step: segment `seg`'s gradients are complete once the main stream reaches
    // this point (BatchNorm / head gradients) and the side stream has run what it holds (weight
    // gradients): its Adam update goes on the side stream behind both
    auto segment_done = [&](int seg) -> int {
        const cilrs_adam_args* o = net->fused_adam;
        if (!o) return 0;
        const size_t b = A.seg_begin[seg], n = A.seg_end[seg] - A.seg_begin[seg];
        hipStream_t st = s;
        if (use_overlap(net)) {
            if (gbuf_side_begin(net, s)) return 1;       // side waits for the main stream's part
            st = net->side[0];
        }
        RUN(net, "adam", 0.0, 28.0 * n, st,
            launch_adam(bufs->params + b, Gp + b, o->exp_avg + b, o->exp_avg_sq + b, n, o->lr,
                        o->beta1, o->beta2, o->eps, o->weight_decay, (long long)o->step, nullptr,
                        o->grad_scale, st));
        return 0;
    };

    for (int seg = seg_begin; seg < seg_end; ++seg) {
        if (seg == 0) {
            CILRS_CHECK(dcontrols && dpred_speed, "backward: output gradients missing");
            net->bwd_nblk_next = 0;
            if (backward_heads(net, bufs, dcontrols, dpred_speed, nullptr, s)) return 1;
            // d visual -> avgpool backward -> grad of the last block's output, in G[3]
            if (net->bf16_train)
                RUN(net, "heads_bwd", 0.0, 0.0, s,
                    launch_avgpool_bwd16(ws + net->dcombined, h16(ws, net->G16[3]), B, net->featHW,
                                         A.feat, A.feat + 128, s));
            else
                RUN(net, "heads_bwd", 0.0, 0.0, s,
                    launch_avgpool_bwd(ws + net->dcombined, ws + net->G[3], B, net->featHW, A.feat,
                                       A.feat + 128, s));
            if (segment_done(seg)) return 1;
            continue;
        }
        if (seg >= 1 && seg <= 4) {
            const int layer = 5 - seg;                 // 4,3,2,1
            // blocks of this layer, last to first; gradient of the block output is in G[3]
            int first = 0;
            const int nblk[4] = {3, 4, 6, 3};
            for (int L = 1; L < layer; ++L) first += nblk[L - 1];
            for (int bi = first + nblk[layer - 1] - 1; bi >= first; --bi) {
                const BlockT& blk = A.blocks[bi];
                // conv-BN pairs of the main branch, in forward order (2: BasicBlock, 3: Bottleneck)
                const int chain[3] = {blk.conv1, blk.conv2, blk.conv3};
                const int nchain = blk.conv3 >= 0 ? 3 : 2;
                const ConvT& c1 = A.convs[blk.conv1];
                const ConvG& g1 = net->cg[blk.conv1];
                // block input activation
                const float* xin;
                if (bi == 0) xin = ws + net->pool;
                else xin = ws + net->cg[last_conv(A.blocks[bi - 1])].z;
                const std::string grp = kGroupName[c1.group];
                // weight gradients run on side stream 0, concurrently with the data gradients
                const bool bf16t = net->bf16_train;
                // gradient buffers by index: [3] d(block input / output), [1] identity-path
                // gradient, [2] d(input of a main-branch convolution), ring slots = dy tensors;
                // fp32 mode: ws + G[i]; bf16 mode: the bf16 buffer G16[i]
                auto Gf = [&](int gi) { return ws + net->G[gi]; };
                auto Gh = [&](int gi) { return h16(ws, net->G16[gi]); };
                // (x: the conv's input activation; x16: its bf16 form)
                auto wgrad_side = [&](const ConvT& c, const ConvG& g, const float* x,
                                      const cilrs_half* x16, int gi, float* dwdst) -> int {
                    if (gbuf_side_begin(net, s)) return 1;
                    if (bf16t) {
                        if (conv_wgrad16(net, c, g, x16, Gh(gi), dwdst, ws, side_or(net, s, 0)))
                            return 1;
                    } else {
                        if (conv_wgrad(net, c, g, x, c.cin, Gf(gi), dwdst, ws, side_or(net, s, 0)))
                            return 1;
                    }
                    return gbuf_side_end(net, gi);
                };
                // data gradient of conv `ci`: dy (buffer gi) -> buffer go (+ buffer gadd, -1: none);
                // out32: the bf16 mode's last data gradient of the trunk, handed to the fp32 stem
                auto dgrad = [&](int ci, int gi, int go, int gadd, const ConvG* bn_of, int* nbp,
                                 bool out32 = false) -> int {
                    const ConvT& c = A.convs[ci];
                    const ConvG& g = net->cg[ci];
                    if (bf16t)
                        return conv_dgrad16(net, c, g, ci, Gh(gi), out32 ? nullptr : Gh(go),
                                            out32 ? Gf(go) : nullptr, gadd >= 0 ? Gh(gadd) : nullptr,
                                            ws, s, bn_of, nbp);
                    return conv_dgrad(net, c, g, Gf(gi), P + c.w, Gf(go), gadd >= 0 ? Gf(gadd) : nullptr,
                                      ws, s, bn_of, 1, nbp);
                };
                // BatchNorm backward of conv `ci`: dz (buffer gz) -> dy (buffer gdy), masked
                // gradient -> buffer gg (-1: not needed)
                auto bnb = [&](int ci, int gz, int relu, int gdy, int gg, int pre_nblk,
                               double passes) -> int {
                    const ConvT& c = A.convs[ci];
                    const ConvG& g = net->cg[ci];
                    const BnT& b = A.bns[c.bn];
                    if (bf16t) {
                        RUN(net, "bn_bwd." + grp, 0.0, 2.0 * g.M * c.cout * passes, s,
                            launch_bn16_bwd(Gh(gz), relu ? h16(ws, net->z16[ci]) : nullptr,
                                            y16_of(net, ws, ci), g.M, c.cout, P + b.gamma,
                                            ws + g.stats, relu, Gp + b.gamma, Gp + b.beta,
                                            ws + net->bn_coef, ws + net->bn_partial, Gh(gdy),
                                            gg >= 0 ? Gh(gg) : nullptr, pre_nblk, s));
                    } else {
                        RUN(net, "bn_bwd." + grp, 0.0, 4.0 * g.M * c.cout * passes, s,
                            launch_bn_bwd(Gf(gz), relu ? ws + g.z : nullptr, ws + g.y, g.M, c.cout,
                                          P + b.gamma, ws + g.stats, relu, Gp + b.gamma,
                                          Gp + b.beta, 0, ws + net->bn_coef, ws + net->bn_partial,
                                          Gf(gdy), gg >= 0 ? Gf(gg) : nullptr, pre_nblk, s, nullptr,
                                          bn_sync(net, ws, 1)));
                    }
                    return 0;
                };
                auto z16_of = [&](int ci) { return h16(ws, net->z16[ci]); };
                const cilrs_half* xin16 =
                    bi == 0 ? h16(ws, net->pool16) : z16_of(last_conv(A.blocks[bi - 1]));
                auto next_ring = [&]() {
                    const int gi = kRingIdx[net->dy_pos];
                    net->dy_pos = (net->dy_pos + 1) % dy_ring_depth();
                    return gi;
                };
                // 1. out = relu(bn_last(y_last) + identity): masked grad -> [1], dy_last -> ga
                int ga = next_ring();
                if (gbuf_acquire(net, s, ga) || gbuf_acquire(net, s, 1)) return 1;
                if (bnb(chain[nchain - 1], 3, 1, ga, 1, net->bwd_nblk_next, 8.0)) return 1;
                net->bwd_nblk_next = 0;
                // 2. walk the main branch backwards: dW_i (side), d(input of conv_i) -> [2], then
                //    a = relu(bn_{i-1}(y_{i-1})): dy_{i-1} -> the next dy buffer
                for (int i = nchain - 1; i >= 1; --i) {
                    const ConvT& c = A.convs[chain[i]];
                    const ConvG& g = net->cg[chain[i]];
                    const ConvG& gp = net->cg[chain[i - 1]];
                    if (wgrad_side(c, g, ws + gp.z, z16_of(chain[i - 1]), ga, Gp + c.w)) return 1;
                    if (gbuf_acquire(net, s, 2)) return 1;
                    int nbp = 0;    // the previous BatchNorm's reductions ride on this dgrad's epilogue
                    if (dgrad(chain[i], ga, 2, -1, &gp, &nbp)) return 1;
                    ga = next_ring();
                    if (gbuf_acquire(net, s, ga)) return 1;
                    if (bnb(chain[i - 1], 2, 1, ga, -1, nbp, 7.0)) return 1;
                }
                // 3. dW1 (side)
                if (wgrad_side(c1, g1, xin, xin16, ga, Gp + c1.w)) return 1;
                if (blk.down < 0) {
                    // 4. dx = dgrad(conv1) + identity grad [1] -> [3]
                    // ... and carries the reductions of the previous block's last BatchNorm
                    const ConvG* prev = bi > 0 ? &net->cg[last_conv(A.blocks[bi - 1])] : nullptr;
                    if (dgrad(blk.conv1, ga, 3, 1, prev, &net->bwd_nblk_next, bi == 0)) return 1;
                } else {
                    const ConvT& cd = A.convs[blk.down];
                    const ConvG& gd = net->cg[blk.down];
                    if (dgrad(blk.conv1, ga, 3, -1, nullptr, nullptr)) return 1;
                    // 5. identity = bn_d(conv_d(x)) (no ReLU): dy_d -> a dy buffer
                    const int gdn = next_ring();
                    if (gbuf_acquire(net, s, gdn)) return 1;
                    if (bnb(blk.down, 1, 0, gdn, -1, 0, 6.0)) return 1;
                    if (wgrad_side(cd, gd, xin, xin16, gdn, Gp + cd.w)) return 1;
                    // 6. dx += dgrad(conv_d)
                    if (dgrad(blk.down, gdn, 3, 3, nullptr, nullptr, bi == 0)) return 1;
                }
            }
            // (the segment's weight gradients are complete when this call returns its work: the
            //  side stream joins at the end of the call -- between the segments of ONE call the
            //  data-gradient chain does not wait for the weight gradients any more)
            if (segment_done(seg)) return 1;
            continue;
        }
        // seg == 5: stem.  G[3] holds d(maxpool output)
        {
            const ConvT& c0 = A.convs[0];
            const ConvG& g0 = net->cg[0];
            const BnT& b0 = A.bns[c0.bn];
            const unsigned char* argmax =
                reinterpret_cast<const unsigned char*>(bufs->workspace) + net->argmax_b;
            // max-pool backward + ReLU mask are rebuilt inside the BatchNorm-backward passes
            if (gbuf_acquire(net, s, 1)) return 1;           // G[1] is rewritten below
            RUN(net, "bn_bwd.stem", 0.0, 4.0 * g0.M * 64 * 3.0, s,
                launch_bn_bwd_pool(ws + net->G[3], argmax, ws + g0.y, B, net->H0, net->W0, 64,
                                   P + b0.gamma, ws + g0.stats, Gp + b0.gamma, Gp + b0.beta,
                                   ws + net->bn_coef, ws + net->bn_partial, ws + net->G[1], s));
            // the stem's weight gradient runs on the main stream and uses the same slab scratch as
            // the weight gradients of the side stream: those must have finished (between the
            // segments of one call nothing else joins the two streams any more)
            if (gbuf_join_all(net, s)) return 1;
            const size_t sw = stem_wgrad_f32_scratch_floats(B, net->H, net->W);
            if (sw > 0 && sw <= net->slabs_floats) {
                // (the reduction over pixels on the matrix pipe, dy from global memory, the input
                //  rows in LDS: stem_f32.hip)
                RUN(net, "conv_wgrad.stem", 2.0 * g0.M * 64 * 147,
                    16.0 * B * net->H * net->W + 4.0 * g0.M * 64, s,
                    launch_stem_wgrad_f32(ws + net->x4, ws + net->G[1], Gp + c0.w, ws + net->slabs, B,
                                          net->H, net->W, s));
            } else if (conv_wgrad(net, c0, g0, ws + net->x4, 4, ws + net->G[1], Gp + c0.w, ws, s)) {
                return 1;
            }
            if (segment_done(seg)) return 1;
        }
    }
    // everything the side stream still runs (weight gradients, fused Adam launches) joins here
    if (gbuf_join_all(net, s)) return 1;
    if (net->heads_pending) {
        CILRS_HIP(hipStreamWaitEvent(s, net->heads_ev, 0));
        net->heads_pending = false;
    }
    if (net->fused_adam && use_overlap(net)) {
        CILRS_HIP(hipEventRecord(net->fork_ev, net->side[0]));
        CILRS_HIP(hipStreamWaitEvent(s, net->fork_ev, 0));
    }
    return 0;
}

// Backward of every segment with torch.optim.Adam.step() (notebook/notebook.ipynb:555) fused in:
// the update of a segment's parameter range is enqueued -- on the weight-gradient stream, behind
// that segment's weight gradients -- as soon as the segment's gradients are complete, so the
// HBM-bound update of layer4 (13 M of the 21 M parameters) runs under the matrix-pipe-bound data
// gradients of layers 3..1 instead of after the stem.  Same arithmetic per element as
// cilrs_adam_step over the whole arena (no gradient clipping: the global norm needs every
// gradient first -- callers that clip use cilrs_net_backward + cilrs_grad_sqnorm + cilrs_adam_step).
int cilrs_net_backward_step(cilrs_net* net, const cilrs_buffers* bufs, const float* dcontrols,
                            const float* dpred_speed, const cilrs_adam_args* opt, void* stream) {
    CILRS_CHECK(net && opt && opt->exp_avg && opt->exp_avg_sq, "backward_step: NULL argument");
    CILRS_CHECK(opt->step >= 1, "backward_step: step must be >= 1");
    net->fused_adam = opt;
    const int rc = cilrs_net_backward(net, bufs, dcontrols, dpred_speed, 0, 6, stream);
    net->fused_adam = nullptr;
    return rc;
}

static int backward_heads(cilrs_net* net, const cilrs_buffers* bufs, const float* dcontrols,
                          const float* dps, const int64_t*, hipStream_t s) {
    const Arch& A = *net->A;
    const int feat = A.feat, comb = A.feat + 128;
    float* ws = reinterpret_cast<float*>(bufs->workspace);
    const float* P = bufs->params;
    float* Gp = bufs->grads;
    const int B = net->B;
    const float dscale = net->last_dropout > 0.f ? 1.0f / (1.0f - net->last_dropout) : 1.0f;
    const long long* cmd = reinterpret_cast<const long long*>(
        reinterpret_cast<const char*>(bufs->workspace) + net->cmd_b);
    float* dcomb = ws + net->dcombined;

    // weight + bias gradient of layer l:  dW[out][in] = dy^T x,  db = colsum(dy)
    auto wg = [&](HGemmGroup& g, const float* dy, int dy_ld, const float* x, int x_ld,
                  const LinT& l) {
        memset(&g, 0, sizeof(g));
        g.A = dy; g.lda = dy_ld; g.B = x; g.ldb = x_ld; g.C = Gp + l.w; g.ldc = l.in;
        g.dbias = Gp + l.b; g.M = l.out; g.N = l.in; g.K = B;
    };
    // input gradient of layer l:  dx = (dy W) masked by act > 0, scaled
    auto dg = [&](HGemmGroup& g, const float* dy, int dy_ld, const LinT& l, float* dx, int dx_ld,
                  const float* act, int act_ld, float scale) {
        memset(&g, 0, sizeof(g));
        g.A = dy; g.lda = dy_ld; g.B = P + l.w; g.ldb = l.in; g.C = dx; g.ldc = dx_ld;
        g.mask = act; g.ldmask = act_ld; g.mask_scale = scale; g.M = B; g.N = l.in; g.K = l.out;
    };
    // mode 2 = weight + bias gradients: nothing on the main stream reads them -- they go to the side
    // stream (idle at this point of the backward pass, and these launches occupy a handful of
    // CUs), behind the output gradient they multiply; cilrs_net_backward joins them at its end
    auto run = [&](int mode, HGemmArgs& h, int n) -> int {
        h.ngroups = n; h.relu = 0; h.accumulate = 0; h.drop_p = 0.f; h.seed = 0;
        double fl = 0.0;
        for (int i = 0; i < n; ++i) fl += 2.0 * h.g[i].M * h.g[i].N * h.g[i].K;
        hipStream_t st = s;
        if (mode == 2 && use_overlap(net)) {
            if (gbuf_side_begin(net, s)) return 1;
            st = net->side[0];
        }
        RUN(net, "heads_bwd", fl, 0.0, st, launch_hgemm(mode, h, st));
        if (st != s) {
            CILRS_HIP(hipEventRecord(net->heads_ev, st));
            net->heads_pending = true;
        }
        return 0;
    };
    HGemmArgs h;
    memset(&h, 0, sizeof(h));

    // ---- only the commanded branch of each frame receives gradient (gather) ----
    const int NC = A.ncmd;
    RUN(net, "heads_bwd", 0.0, 0.0, s,
        launch_branch_scatter(dcontrols, cmd, ws + net->d_all, B, NC, s));
    const float* d_out[kMaxCmd];
    for (int k = 0; k < NC; ++k) d_out[k] = ws + net->d_all + (size_t)k * B * 4;

    // ---- output layers (256 -> 3 per branch, 256 -> 1 speed predictor) ----
    for (int k = 0; k < NC; ++k) wg(h.g[k], d_out[k], 4, ws + net->h2[k], 256, A.br[k][2]);
    wg(h.g[NC], dps, 1, ws + net->p2, 256, A.sp5);
    if (run(2, h, NC + 1)) return 1;
    for (int k = 0; k < NC; ++k)
        dg(h.g[k], d_out[k], 4, A.br[k][2], ws + net->dh2[k], 256, ws + net->h2[k], 256, dscale);
    dg(h.g[NC], dps, 1, A.sp5, ws + net->dp2, 256, ws + net->p2, 256, 1.0f);
    if (run(1, h, NC + 1)) return 1;
    // ---- middle layers (256 -> 256) ----
    for (int k = 0; k < NC; ++k) wg(h.g[k], ws + net->dh2[k], 256, ws + net->h1[k], 256, A.br[k][1]);
    wg(h.g[NC], ws + net->dp2, 256, ws + net->p1, 256, A.sp3);
    if (run(2, h, NC + 1)) return 1;
    for (int k = 0; k < NC; ++k)
        dg(h.g[k], ws + net->dh2[k], 256, A.br[k][1], ws + net->dh1[k], 256, ws + net->h1[k], 256,
           dscale);
    dg(h.g[NC], ws + net->dp2, 256, A.sp3, ws + net->dp1, 256, ws + net->p1, 256, dscale);
    if (run(1, h, NC + 1)) return 1;
    // ---- first layers (feat+128 -> 256 per branch; feat -> 256 speed predictor, visual half
    //      only; 640 / 512 for the reference's ResNet-34 trunk) ----
    for (int k = 0; k < NC; ++k)
        wg(h.g[k], ws + net->dh1[k], 256, ws + net->combined, comb, A.br[k][0]);
    wg(h.g[NC], ws + net->dp1, 256, ws + net->combined, comb, A.sp0);
    if (run(2, h, NC + 1)) return 1;
    for (int k = 0; k < NC; ++k)
        dg(h.g[k], ws + net->dh1[k], 256, A.br[k][0], ws + net->dcomb_part[k], comb, nullptr, 0, 1.f);
    dg(h.g[NC], ws + net->dp1, 256, A.sp0, ws + net->dcomb_part[NC], comb, nullptr, 0, 1.f);
    if (run(1, h, NC + 1)) return 1;
    // d combined = sum of the branch contributions (comb wide) + speed predictor (feat wide)
    SumParts parts;
    memset(&parts, 0, sizeof(parts));
    parts.n = NC;
    for (int k = 0; k < NC; ++k) parts.p[k] = ws + net->dcomb_part[k];
    parts.tail = ws + net->dcomb_part[NC];
    RUN(net, "heads_bwd", 0.0, 0.0, s, launch_sum_parts(parts, dcomb, B, comb, feat, s));
    // ---- speed encoder (the speed half of `combined`) ----
    RUN(net, "heads_bwd", 0.0, 0.0, s,
        launch_relu_mask(dcomb + feat, ws + net->combined + feat, B, 128, comb, comb, 1.0f, s));
    wg(h.g[0], dcomb + feat, comb, ws + net->s1, 128, A.se3);
    if (run(2, h, 1)) return 1;
    dg(h.g[0], dcomb + feat, comb, A.se3, ws + net->ds1, 128, ws + net->s1, 128, dscale);
    if (run(1, h, 1)) return 1;
    wg(h.g[0], ws + net->ds1, 128, ws + net->speed_in, 1, A.se0);
    if (run(2, h, 1)) return 1;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// loss / optimiser / profiling
// ------------------------------------------------------------------------------------------------
int cilrs_loss_fwd_bwd(const float* controls, const float* target_controls,
                       const float* pred_speed, const float* target_speed, int batch, int kind,
                       const float* weights4_host, float grad_scale, float* dcontrols,
                       float* dpred_speed, float* loss_out, void* stream) {
    CILRS_CHECK(controls && target_controls && pred_speed && target_speed && weights4_host &&
                    loss_out, "loss: NULL argument");
    CILRS_CHECK(batch >= 1, "loss: batch must be >= 1");
    return launch_loss(controls, target_controls, pred_speed, target_speed, batch, kind,
                       weights4_host, grad_scale, dcontrols, dpred_speed, loss_out,
                       reinterpret_cast<hipStream_t>(stream));
}

size_t cilrs_sqnorm_scratch_bytes(void) { return sqnorm_scratch_bytes(); }

int cilrs_grad_sqnorm(const float* grads, size_t n, float max_norm, void* scratch, float* out2,
                      void* stream) {
    CILRS_CHECK(grads && scratch && out2, "grad_sqnorm: NULL argument");
    return launch_grad_sqnorm(grads, n, max_norm, reinterpret_cast<double*>(scratch), out2,
                              reinterpret_cast<hipStream_t>(stream));
}

int cilrs_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                    size_t n, double lr, double beta1, double beta2, double eps,
                    double weight_decay, int64_t step, const float* clip_out2, float grad_scale,
                    void* stream) {
    CILRS_CHECK(params && grads && exp_avg && exp_avg_sq, "adam: NULL argument");
    return launch_adam(params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay,
                       (long long)step, clip_out2, grad_scale,
                       reinterpret_cast<hipStream_t>(stream));
}

int cilrs_stem_conv_fwd(const float* x4, const float* w, float* y, float* bn_partial, int N, int H,
                        int W, int* partial_rows, void* stream) {
    CILRS_CHECK(x4 && w && y, "stem_conv_fwd: NULL argument");
    const int rows = stem_f32_rows(N, H, W);
    CILRS_CHECK(rows > 0, "stem_conv_fwd: geometry %dx%dx%d not served", N, H, W);
    if (partial_rows) *partial_rows = rows;
    return launch_stem_f32(x4, w, y, bn_partial, N, H, W, reinterpret_cast<hipStream_t>(stream));
}

size_t cilrs_stem_conv_wgrad_scratch_floats(int N, int H, int W) {
    return stem_wgrad_f32_scratch_floats(N, H, W);
}
int cilrs_stem_conv_wgrad(const float* x4, const float* dy, float* dw, float* scratch,
                          size_t scratch_floats, int N, int H, int W, void* stream) {
    CILRS_CHECK(x4 && dy && dw && scratch, "stem_conv_wgrad: NULL argument");
    const size_t need = stem_wgrad_f32_scratch_floats(N, H, W);
    CILRS_CHECK(need > 0, "stem_conv_wgrad: geometry %dx%dx%d not served", N, H, W);
    CILRS_CHECK(scratch_floats >= need, "stem_conv_wgrad: scratch too small");
    return launch_stem_wgrad_f32(x4, dy, dw, scratch, N, H, W, reinterpret_cast<hipStream_t>(stream));
}

int cilrs_scale(float* x, size_t n, const float* clip_out2, float c, void* stream) {
    CILRS_CHECK(x != nullptr, "scale: NULL argument");
    return launch_scale(x, n, clip_out2, c, reinterpret_cast<hipStream_t>(stream));
}

int cilrs_augment_u8(const uint8_t* frames, const cilrs_aug_params* params, int batch, int height,
                     int width, float* out_f32, uint8_t* out_u8, void* stream) {
    CILRS_CHECK(frames && params && (out_f32 || out_u8), "augment: NULL argument");
    static_assert(sizeof(cilrs_aug_params) == 112, "cilrs_aug_params layout");
    return launch_augment_u8(frames, params, batch, height, width, out_f32, out_u8,
                             reinterpret_cast<hipStream_t>(stream));
}

// op-level nn.Linear (the kernels the heads launch, one group)
int cilrs_linear_fwd(const float* x, const float* w, const float* bias, float* y, int batch,
                     int in_features, int out_features, int x_ld, int y_ld, int relu,
                     void* stream) {
    CILRS_CHECK(x && w && y && batch >= 1 && in_features >= 1 && out_features >= 1,
                "linear_fwd: bad argument");
    HGemmArgs h;
    memset(&h, 0, sizeof(h));
    HGemmGroup& g = h.g[0];
    g.A = x; g.lda = x_ld; g.B = w; g.ldb = in_features; g.bias = bias; g.C = y; g.ldc = y_ld;
    g.M = batch; g.N = out_features; g.K = in_features; g.drop_stream = kNoDrop;
    h.ngroups = 1; h.relu = relu;
    return launch_hgemm(0, h, reinterpret_cast<hipStream_t>(stream));
}

int cilrs_linear_bwd(const float* dy, const float* x, const float* w, const float* act,
                     float act_scale, float* dx, float* dw, float* db, int batch,
                     int in_features, int out_features, int dy_ld, int x_ld, int dx_ld,
                     int act_ld, void* stream) {
    CILRS_CHECK(dy && x && w && batch >= 1 && in_features >= 1 && out_features >= 1,
                "linear_bwd: bad argument");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    HGemmArgs h;
    if (dw) {
        memset(&h, 0, sizeof(h));
        HGemmGroup& g = h.g[0];
        g.A = dy; g.lda = dy_ld; g.B = x; g.ldb = x_ld; g.C = dw; g.ldc = in_features;
        g.dbias = db; g.M = out_features; g.N = in_features; g.K = batch;
        h.ngroups = 1;
        if (launch_hgemm(2, h, s)) return 1;
    }
    if (dx) {
        memset(&h, 0, sizeof(h));
        HGemmGroup& g = h.g[0];
        g.A = dy; g.lda = dy_ld; g.B = w; g.ldb = in_features; g.C = dx; g.ldc = dx_ld;
        g.mask = act; g.ldmask = act_ld; g.mask_scale = act_scale;
        g.M = batch; g.N = in_features; g.K = out_features;
        h.ngroups = 1;
        if (launch_hgemm(1, h, s)) return 1;
    }
    return 0;
}

int cilrs_eval_acc_doubles(void) { return kEvalAccDoubles; }

int cilrs_eval_accumulate(const float* controls, const float* pred_speed,
                          const float* target_controls, const float* target_speed,
                          const int64_t* command, int batch, double* acc, float* steer_abs_err,
                          void* stream) {
    CILRS_CHECK(controls && pred_speed && target_controls && target_speed && command && acc,
                "eval_accumulate: NULL argument");
    CILRS_CHECK(batch >= 1, "eval_accumulate: empty batch");
    return launch_eval_accumulate(controls, target_controls, pred_speed, target_speed,
                                  reinterpret_cast<const long long*>(command), batch, acc,
                                  steer_abs_err, reinterpret_cast<hipStream_t>(stream));
}

int cilrs_net_profile_enable(cilrs_net* net, int on) {
    CILRS_CHECK(net != nullptr, "profile: net is NULL");
    net->prof.on = on != 0;
    return 0;
}
int cilrs_net_profile_collect(cilrs_net* net) {
    CILRS_CHECK(net != nullptr, "profile: net is NULL");
    return net->prof.collect();
}
int cilrs_net_profile_count(const cilrs_net* net) { return net ? (int)net->prof.labels.size() : 0; }
int cilrs_net_profile_entry(const cilrs_net* net, int i, char* label, int label_cap,
                            long long* calls, double* total_ms, double* total_flops,
                            double* total_bytes) {
    CILRS_CHECK(net && i >= 0 && i < (int)net->prof.labels.size(), "profile: bad index");
    if (label && label_cap > 0) snprintf(label, label_cap, "%s", net->prof.labels[i].c_str());
    const ProfAgg& a = net->prof.agg[i];
    if (calls) *calls = a.calls;
    if (total_ms) *total_ms = a.ms;
    if (total_flops) *total_flops = a.flops;
    if (total_bytes) *total_bytes = a.bytes;
    return 0;
}
int cilrs_net_profile_reset(cilrs_net* net) {
    CILRS_CHECK(net != nullptr, "profile: net is NULL");
    if (net->prof.collect()) return 1;
    for (auto& a : net->prof.agg) a = ProfAgg();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// op-level entry points (unit parity tests)
// ------------------------------------------------------------------------------------------------
int cilrs_conv2d_fwd(const float* x, const float* w, float* y, int N, int H, int W, int Cin,
                     int Cout, int KH, int KW, int stride, int pad, int force_cfg,
                     int force_splitk, float* scratch, size_t scratch_floats, void* stream) {
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.w = w; a.y = y;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin;
    a.Ho = (H + 2 * pad - KH) / stride + 1; a.Wo = (W + 2 * pad - KW) / stride + 1; a.Cout = Cout;
    a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad;
    a.x_ld = Cin; a.y_ld = Cout; a.w_mode = 0; a.w_cin = Cin;
    a.scratch = scratch; a.scratch_floats = scratch_floats;
    a.force_cfg = force_cfg; a.force_splitk = force_splitk;
    return launch_conv_igemm(a, reinterpret_cast<hipStream_t>(stream));
}

int cilrs_conv2d_dgrad(const float* dy, const float* w, float* dx, const float* addend, int N,
                       int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                       int force_cfg, int force_splitk, float* scratch, size_t scratch_floats,
                       void* stream) {
    CILRS_CHECK(KH == KW, "dgrad: square kernels only");
    DgradArgs a;
    memset(&a, 0, sizeof(a));
    a.dy = dy; a.w = w; a.dx = dx; a.addend = addend;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin;
    a.Ho = (H + 2 * pad - KH) / stride + 1; a.Wo = (W + 2 * pad - KW) / stride + 1;
    a.Cout = Cout; a.K = KH; a.stride = stride; a.pad = pad;
    a.dy_ld = Cout; a.dx_ld = Cin;
    a.scratch = scratch; a.scratch_floats = scratch_floats;
    a.force_cfg = force_cfg; a.force_splitk = force_splitk;
    return launch_conv_dgrad(a, reinterpret_cast<hipStream_t>(stream));
}

static WgradArgs make_wgrad(const float* x, const float* dy, float* dw, float* scratch, int N,
                            int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                            int Cin_dst) {
    WgradArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.dy = dy; a.dw = dw; a.slabs = scratch;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin;
    a.Ho = (H + 2 * pad - KH) / stride + 1; a.Wo = (W + 2 * pad - KW) / stride + 1; a.Cout = Cout;
    a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad;
    a.x_ld = Cin; a.dy_ld = Cout; a.Cin_dst = Cin_dst; a.accumulate = 0;
    return a;
}

size_t cilrs_conv2d_wgrad_scratch_floats(int N, int H, int W, int Cin, int Cout, int KH, int KW,
                                         int stride, int pad) {
    return wgrad_scratch_floats(
        make_wgrad(nullptr, nullptr, nullptr, nullptr, N, H, W, Cin, Cout, KH, KW, stride, pad, Cin));
}

int cilrs_conv2d_wgrad(const float* x, const float* dy, float* dw, float* scratch, int N, int H,
                       int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                       int Cin_dst, void* stream) {
    return launch_conv_wgrad(
        make_wgrad(x, dy, dw, scratch, N, H, W, Cin, Cout, KH, KW, stride, pad, Cin_dst),
        reinterpret_cast<hipStream_t>(stream));
}

// ---- the same three convolution operators on the 16-bit matrix pipe (operands rounded to bf16 /
//      fp16, fp32 accumulation and results): the kernels of the bf16 training mode, op by op ----
static size_t up8(size_t v) { return (v + 7) / 8 * 8; }
size_t cilrs_conv2d_16_scratch_halfs(int N, int H, int W, int Cin, int Cout, int K, int stride,
                                     int pad) {
    const int Ho = (H + 2 * pad - K) / stride + 1, Wo = (W + 2 * pad - K) / stride + 1;
    return up8((size_t)N * H * W * Cin) + up8((size_t)N * Ho * Wo * Cout) +
           2 * up8((size_t)Cout * K * K * Cin);
}

int cilrs_conv2d_fwd_16(const float* x, const float* w, float* y, float* bn_partial, int N, int H,
                        int W, int Cin, int Cout, int K, int stride, int pad, int bf16,
                        void* scratch16, void* stream) {
    CILRS_CHECK(x && w && y && scratch16, "conv2d_fwd_16: NULL argument");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    cilrs_half* x16 = reinterpret_cast<cilrs_half*>(scratch16);
    cilrs_half* w16 = x16 + up8((size_t)N * H * W * Cin);
    if (launch_f32_to_f16(x, x16, (size_t)N * H * W * Cin, bf16, s)) return 1;
    if (launch_f32_to_f16(w, w16, (size_t)Cout * K * K * Cin, bf16, s)) return 1;
    ConvF16Args a;
    memset(&a, 0, sizeof(a));
    a.x = x16; a.w = w16; a.y32 = y; a.bn_partial = bn_partial;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.K = K; a.stride = stride; a.pad = pad;
    a.Ho = (H + 2 * pad - K) / stride + 1; a.Wo = (W + 2 * pad - K) / stride + 1;
    a.bf16 = bf16;
    return launch_conv_f16_train(a, s);
}

int cilrs_conv2d_dgrad_16(const float* dy, const float* w, float* dx, const float* addend, int N,
                          int H, int W, int Cin, int Cout, int K, int stride, int pad, int bf16,
                          void* scratch16, void* stream) {
    CILRS_CHECK(dy && w && dx && scratch16, "conv2d_dgrad_16: NULL argument");
    CILRS_CHECK(stride == 1 || stride == 2, "conv2d_dgrad_16: stride must be 1 or 2");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int Ho = (H + 2 * pad - K) / stride + 1, Wo = (W + 2 * pad - K) / stride + 1;
    cilrs_half* dy16 = reinterpret_cast<cilrs_half*>(scratch16);
    cilrs_half* wT16 = dy16 + up8((size_t)N * Ho * Wo * Cout);
    if (launch_f32_to_f16(dy, dy16, (size_t)N * Ho * Wo * Cout, bf16, s)) return 1;
    if (launch_transpose_flip_f16(w, wT16, Cout, K, Cin, bf16, s)) return 1;
    ConvF16Args a;
    memset(&a, 0, sizeof(a));
    a.x = dy16; a.w = wT16; a.y32 = dx; a.addend32 = addend;
    a.N = N; a.H = Ho; a.W = Wo; a.Cin = Cout;       // gathered tensor = dy
    a.Ho = H; a.Wo = W; a.Cout = Cin;                // enumerated grid = dx
    a.K = K; a.bf16 = bf16;
    if (stride == 1) { a.stride = 1; a.pad = K - 1 - pad; }
    else { a.stride = 2; a.pad = pad; a.up2 = 1; }
    return launch_conv_f16_train(a, s);
}

static WgradF16Args make_wgrad16(const void* x16, const void* dy16, float* dw, float* slabs, int N,
                                 int H, int W, int Cin, int Cout, int K, int stride, int pad,
                                 int bf16) {
    WgradF16Args a;
    memset(&a, 0, sizeof(a));
    a.x = x16; a.dy = dy16; a.dw = dw; a.slabs = slabs;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.K = K; a.stride = stride; a.pad = pad;
    a.Ho = (H + 2 * pad - K) / stride + 1; a.Wo = (W + 2 * pad - K) / stride + 1;
    a.bf16 = bf16;
    return a;
}

size_t cilrs_conv2d_wgrad_16_scratch_floats(int N, int H, int W, int Cin, int Cout, int K,
                                            int stride, int pad) {
    return wgrad_f16_scratch_floats(
        make_wgrad16(nullptr, nullptr, nullptr, nullptr, N, H, W, Cin, Cout, K, stride, pad, 1));
}

int cilrs_conv2d_wgrad_16(const float* x, const float* dy, float* dw, float* scratch32, int N,
                          int H, int W, int Cin, int Cout, int K, int stride, int pad, int bf16,
                          void* scratch16, void* stream) {
    CILRS_CHECK(x && dy && dw && scratch32 && scratch16, "conv2d_wgrad_16: NULL argument");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int Ho = (H + 2 * pad - K) / stride + 1, Wo = (W + 2 * pad - K) / stride + 1;
    cilrs_half* x16 = reinterpret_cast<cilrs_half*>(scratch16);
    cilrs_half* dy16 = x16 + up8((size_t)N * H * W * Cin);
    if (launch_f32_to_f16(x, x16, (size_t)N * H * W * Cin, bf16, s)) return 1;
    if (launch_f32_to_f16(dy, dy16, (size_t)N * Ho * Wo * Cout, bf16, s)) return 1;
    return launch_wgrad_f16(
        make_wgrad16(x16, dy16, dw, scratch32, N, H, W, Cin, Cout, K, stride, pad, bf16), s);
}

// ---- the bf16 training mode's operators on 16-bit tensors (round 4: activations and gradients
//      live in bf16 end to end), op by op ----
int cilrs_conv2d_train_16(const void* x16, const void* w16, void* y16, float* y32,
                          const void* addend16, float* bn_partial, const void* bwd_z16,
                          const void* bwd_y16, const float* bwd_stats, int bwd_relu,
                          float* bwd_partial, int N, int H, int W, int Cin, int Ho, int Wo, int Cout,
                          int K, int stride, int pad, int up2, int bf16, int* partial_rows,
                          void* stream) {
    CILRS_CHECK(x16 && w16 && (y16 || y32), "conv2d_train_16: NULL argument");
    ConvF16Args a;
    memset(&a, 0, sizeof(a));
    a.x = reinterpret_cast<const cilrs_half*>(x16);
    a.w = reinterpret_cast<const cilrs_half*>(w16);
    a.y16 = reinterpret_cast<cilrs_half*>(y16);
    a.y32 = y16 ? nullptr : y32;
    a.addend16 = reinterpret_cast<const cilrs_half*>(addend16);
    a.bn_partial = bn_partial;
    a.bwd_z16 = reinterpret_cast<const cilrs_half*>(bwd_z16);
    a.bwd_y16 = reinterpret_cast<const cilrs_half*>(bwd_y16);
    a.bwd_stats = bwd_stats; a.bwd_relu = bwd_relu; a.bwd_partial = bwd_partial;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout;
    a.K = K; a.stride = stride; a.pad = pad; a.up2 = up2; a.bf16 = bf16;
    CILRS_CHECK(!bwd_partial || conv_f16_train_can_fuse_bwd(a),
                "conv2d_train_16: BatchNorm-backward partials are not available for this launch");
    if (partial_rows) *partial_rows = conv_f16_train_mtiles(a);
    return launch_conv_f16_train(a, reinterpret_cast<hipStream_t>(stream));
}

int cilrs_bn16_train_fwd(const void* y16, int M, int C, const float* gamma, const float* beta,
                         float* running_mean, float* running_var, int64_t* nbt, float momentum,
                         float eps, const void* residual16, int relu, float* stats, float* partial,
                         void* z16, int pre_rows, void* stream) {
    CILRS_CHECK(y16 && gamma && beta && stats && partial, "bn16_train_fwd: NULL argument");
    return launch_bn16_train_fwd(y16, M, C, gamma, beta, running_mean, running_var,
                                 reinterpret_cast<long long*>(nbt), momentum, eps, residual16, relu,
                                 stats, partial, z16, pre_rows, reinterpret_cast<hipStream_t>(stream));
}

int cilrs_bn16_bwd(const void* dz16, const void* z16, const void* y16, int M, int C,
                   const float* gamma, const float* stats, int relu, float* dgamma, float* dbeta,
                   float* coef3c, float* partial, void* dy16, void* g_out16, int pre_rows,
                   void* stream) {
    CILRS_CHECK(gamma && stats && dgamma && dbeta && coef3c && partial, "bn16_bwd: NULL argument");
    return launch_bn16_bwd(dz16, z16, y16, M, C, gamma, stats, relu, dgamma, dbeta, coef3c, partial,
                           dy16, g_out16, pre_rows, reinterpret_cast<hipStream_t>(stream));
}

// Winograd F(2x2,3x3) forms of the two operators above (3x3 / stride 1 / pad 1): filter transform
// + fused convolution.  scratch: cilrs_conv2d_wino_scratch_floats(Cin, Cout) floats.
size_t cilrs_conv2d_wino_scratch_floats(int Cin, int Cout) { return wino_weight_floats(Cout, Cin); }

int cilrs_conv2d_wino_fwd(const float* x, const float* w, float* y, int N, int H, int W, int Cin,
                          int Cout, float* scratch, void* stream) {
    CILRS_CHECK(x && w && y && scratch, "conv2d_wino_fwd: NULL argument");
    CILRS_CHECK(wino_supported(Cin, Cout, 3, 1, 1), "conv2d_wino_fwd: Cin %% 8, Cout %% 64");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (launch_wino_weights(w, scratch, Cout, Cin, 0, s)) return 1;
    WinoArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.U = scratch; a.y = y; a.N = N; a.H = H; a.W = W; a.C = Cin; a.K = Cout;
    return launch_conv_wino(a, s);
}

size_t cilrs_conv2d_wino_wgrad_scratch_floats(int N, int H, int W, int Cin, int Cout) {
    return wino_wgrad_scratch_floats(N, H, W, Cin, Cout);
}
int cilrs_conv2d_wino_wgrad(const float* x, const float* dy, float* dw, int N, int H, int W, int Cin,
                            int Cout, float* scratch, size_t scratch_floats, void* stream) {
    CILRS_CHECK(x && dy && dw && scratch, "conv2d_wino_wgrad: NULL argument");
    CILRS_CHECK(wino_wgrad_supported(Cin, Cout, 3, 1, 1), "conv2d_wino_wgrad: Cin %% 64, Cout %% 64");
    CILRS_CHECK(scratch_floats >= wino_wgrad_scratch_floats(N, H, W, Cin, Cout),
                "conv2d_wino_wgrad: scratch too small");
    WinoWgradArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.dy = dy; a.dw = dw; a.slabs = scratch;
    a.N = N; a.H = H; a.W = W; a.C = Cin; a.K = Cout;
    return launch_conv_wino_wgrad(a, reinterpret_cast<hipStream_t>(stream));
}

static long long* g_wino_stamps = nullptr;
int cilrs_conv2d_wino_stamps(long long* stamps16) { g_wino_stamps = stamps16; return 0; }

int cilrs_wino_filter_transform(const float* w, float* U, int Cin, int Cout, int dgrad, void* stream) {
    CILRS_CHECK(w && U, "wino_filter_transform: NULL argument");
    return launch_wino_weights(w, U, Cout, Cin, dgrad, reinterpret_cast<hipStream_t>(stream));
}

int cilrs_conv2d_wino_pre(const float* x, const float* U, float* y, const float* addend, int N, int H,
                          int W, int Cred, int Cout, void* stream) {
    CILRS_CHECK(x && U && y, "conv2d_wino_pre: NULL argument");
    WinoArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.U = U; a.y = y; a.addend = addend; a.N = N; a.H = H; a.W = W; a.C = Cred; a.K = Cout;
    a.stamps = g_wino_stamps;        // diagnostics: cilrs_conv2d_wino_stamps (tools/wino_bench.py)
    return launch_conv_wino(a, reinterpret_cast<hipStream_t>(stream));
}

int cilrs_conv2d_wino_split(const float* x, const float* U, float* y, const float* addend,
                            float* bn_partial, int N, int H, int W, int Cred, int Cout, float* slabs,
                            size_t slab_floats, int* csplit, int* partial_rows, void* stream) {
    CILRS_CHECK(x && U && y && bn_partial && slabs, "conv2d_wino_split: NULL argument");
    WinoArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.U = U; a.y = y; a.addend = addend; a.N = N; a.H = H; a.W = W; a.C = Cred; a.K = Cout;
    a.bn_partial = bn_partial; a.slabs = slabs; a.slab_floats = slab_floats;
    a.scratch_partial = bn_partial;
    if (partial_rows) *partial_rows = wino_rows(N, H, W, Cout, 0, Cred, slab_floats);
    const int rc = launch_conv_wino(a, reinterpret_cast<hipStream_t>(stream));
    if (csplit) *csplit = wino_last_csplit();
    return rc;
}

int cilrs_conv2d_wino_dgrad(const float* dy, const float* w, float* dx, const float* addend, int N,
                            int H, int W, int Cin, int Cout, float* scratch, void* stream) {
    CILRS_CHECK(dy && w && dx && scratch, "conv2d_wino_dgrad: NULL argument");
    CILRS_CHECK(wino_supported(Cout, Cin, 3, 1, 1), "conv2d_wino_dgrad: Cout %% 8, Cin %% 64");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (launch_wino_weights(w, scratch, Cout, Cin, 1, s)) return 1;
    WinoArgs a;
    memset(&a, 0, sizeof(a));
    a.x = dy; a.U = scratch; a.y = dx; a.addend = addend;
    a.N = N; a.H = H; a.W = W; a.C = Cout; a.K = Cin;
    return launch_conv_wino(a, s);
}

size_t cilrs_bn_partial_floats(int C) { return bn_partial_floats(C); }

int cilrs_bn_train_fwd(const float* y, int M, int C, const float* gamma, const float* beta,
                       float* running_mean, float* running_var, int64_t* nbt, float momentum,
                       float eps, const float* residual, int relu, float* stats, float* partial,
                       float* z, void* stream) {
    return launch_bn_train_fwd(y, M, C, gamma, beta, running_mean, running_var,
                               reinterpret_cast<long long*>(nbt), momentum, eps, residual, relu,
                               stats, partial, z, 0, reinterpret_cast<hipStream_t>(stream));
}
int cilrs_bn_eval_fwd(const float* y, int M, int C, const float* gamma, const float* beta,
                      const float* running_mean, const float* running_var, float eps,
                      const float* residual, int relu, float* stats, float* z, void* stream) {
    return launch_bn_eval_fwd(y, M, C, gamma, beta, running_mean, running_var, eps, residual,
                              relu, stats, z, reinterpret_cast<hipStream_t>(stream));
}
int cilrs_bn_bwd(const float* dz, const float* z, const float* y, int M, int C,
                 const float* gamma, const float* stats, int relu, float* dgamma, float* dbeta,
                 float* coef3c, float* partial, float* dy, float* g_out, void* stream) {
    return launch_bn_bwd(dz, z, y, M, C, gamma, stats, relu, dgamma, dbeta, 0, coef3c, partial, dy,
                         g_out, 0, reinterpret_cast<hipStream_t>(stream));
}
int cilrs_maxpool_fwd(const float* x, float* out, uint8_t* argmax, int N, int H, int W, int C,
                      void* stream) {
    return launch_maxpool_fwd(x, out, argmax, N, H, W, C, reinterpret_cast<hipStream_t>(stream));
}
int cilrs_maxpool_bwd(const float* dout, const uint8_t* argmax, float* dx, int N, int H, int W,
                      int C, void* stream) {
    return launch_maxpool_bwd(dout, argmax, dx, N, H, W, C, reinterpret_cast<hipStream_t>(stream));
}

}  // extern "C"
