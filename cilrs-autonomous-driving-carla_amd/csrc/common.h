// Shared declarations for the CILRS gfx950 kernels.  Everything here is fp32: the reference path
// (model/autonomous_drive.py:361-399, notebook/notebook.ipynb:504-555) runs in fp32 and parity is
// stated at 1e-4 fp32, so the matrix work uses the exact-f32 MFMA forms
// (v_mfma_f32_32x32x2_f32), which are bitwise an fmaf chain.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/cilrs_hip.h"

namespace cilrs {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- error plumbing (C-ABI: int status + cilrs_last_error()) -------------------------------
void set_error(const char* fmt, ...);
const char* last_error();

#define CILRS_CHECK(cond, ...)                   \
    do {                                         \
        if (!(cond)) {                           \
            ::cilrs::set_error(__VA_ARGS__);     \
            return 1;                            \
        }                                        \
    } while (0)

#define CILRS_HIP(expr)                                                                 \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess) {                                                         \
            ::cilrs::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),   \
                               __FILE__, __LINE__);                                     \
            return 1;                                                                   \
        }                                                                               \
    } while (0)

#define CILRS_LAUNCH_CHECK()                                                            \
    do {                                                                                \
        hipError_t e_ = hipGetLastError();                                              \
        if (e_ != hipSuccess) {                                                         \
            ::cilrs::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), \
                               __FILE__, __LINE__);                                     \
            return 1;                                                                   \
        }                                                                               \
    } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Per-DEVICE launch state (one process may drive several devices): the CU count of the current
// device, and "is this the first time `key` (a kernel) is prepared on the current device" for
// function attributes such as the dynamic-LDS limit.
int device_cus();
bool once_per_device(const void* key);

// Tuning / diagnostic switches whose measured result is a recorded negative
// are COMPILED OUT of the product library: experiment_env() is the constant default unless the
// library is built with -DCILRS_EXPERIMENTS (make CXXEXTRA=-DCILRS_EXPERIMENTS; tools/README.md).
// Switches a test exercises (CILRS_WINO, CILRS_WINO_TAIL, CILRS_WINO_WGRAD, CILRS_OVERLAP,
// CILRS_BN_FUSED, CILRS_SPLITK_INKERNEL, CILRS_CONV16_TILE, CILRS_B1_STAMPS) stay run-time.
#ifdef CILRS_EXPERIMENTS
int experiment_env(const char* name, int dflt);
#else
static inline int experiment_env(const char*, int dflt) { return dflt; }
#endif
static inline size_t cdivz(size_t a, size_t b) { return (a + b - 1) / b; }

// ---- implicit-GEMM convolution (conv_igemm.hip) --------------------------------------------
// One kernel family serves:
//   * forward conv           y[m][co]  = sum_{tap,ci} x[pix(m,tap)][ci] * W[co][tap][ci]
//   * data gradient (dgrad)  dx[m][ci] = sum_{tap,co} dy[pix'(m,tap)][co] * W[co][flip(tap)][ci]
//   * the 128..640-wide linear layers of the heads (1x1 "convs" over a [B,1,1,C] image)
// Activations are NHWC, weights OHWI ([Cout][KH][KW][Cin], i.e. torch channels_last memory).
// Up to four sub-problems run by ONE launch: the output-parity classes of a stride-2 data gradient.
// They share x / w / y, the channel counts and the output tensor; each has its own enumerated
// output grid, its own (at most four) filter taps and its own place in the output.
struct ConvMulti {
    int n;                 // 0 = a single problem (everything below unused)
    int tile_begin[5];     // cumulative block count per class (class c owns [c], [c+1])
    int Ho[4], Wo[4], out_h0[4], out_w0[4], ntaps[4];
    int tap_dh[4][4], tap_dw[4][4], tap_w[4][4];
};

struct ConvArgs {
    const float* x;        // gathered operand: [N][H][W] pixels, x_ld floats apart, Cin used
    const float* w;        // weights (OHWI of the FORWARD conv)
    float* y;              // [N][out_H][out_W] pixels, y_ld floats apart, Cout columns written
    const float* addend;   // optional, y's layout: y = acc + addend (may alias y)
    const float* bias;     // optional [Cout]
    const float* ch_scale; // optional [Cout] per-channel affine applied first: y = acc*scale+shift
    const float* ch_shift; //   (eval-mode BatchNorm folded into the conv epilogue)
    int relu_post;         // ReLU after the addend (BasicBlock: relu(bn(conv) + identity))
    const float* mask;     // optional, pixel-indexed like y, mask_ld apart:
                           //   y = (mask > 0) ? y * mask_scale : 0
    int N, H, W, Cin;      // geometry of the tensor the A-gather reads
    int Ho, Wo, Cout;      // ENUMERATED output grid (M = N*Ho*Wo rows), Cout columns
    int KH, KW;            // filter size (generic-tap path and weight row pitch)
    int stride, pad;       // enumerated output pixel -> base input pixel: o*stride - pad
    int x_ld, y_ld, mask_ld;
    int w_mode;            // 0: B is k-contiguous (forward); 1: dgrad (B rows = forward Cout)
    int w_cin;             // forward conv's Cin (OHWI row pitch), used in w_mode 1
    int relu;
    float mask_scale;
    // where enumerated output pixel (n, oh, ow) lands in y / addend / mask:
    //   pixel (n, oh*out_sh + out_h0, ow*out_sw + out_w0) of an [N][out_H][out_W] tensor
    int out_H, out_W, out_sh, out_sw, out_h0, out_w0;
    // filter taps visited (uniform path): input pixel = base + (tap_dh, tap_dw), weights of
    // forward tap tap_w.  ntaps == 0 on entry => launcher fills the dense KHxKW table.
    int ntaps;
    int tap_dh[16], tap_dw[16], tap_w[16];
    float* bn_partial;     // optional: per-M-tile column sums / sums of squares of the raw output,
                           // [tile][2][Cout] (BatchNorm batch statistics fused into the conv)
    int* bn_nblk;          // host out: tiles written (0 = not fused, e.g. split-K was chosen)
    // optional: BatchNorm-backward reductions of the layer this (data-gradient) output feeds,
    // fused into the epilogue: per M-tile [2][Cout] partial sums of g and g*xhat
    const float* bwd_z; const float* bwd_y; const float* bwd_stats; int bwd_relu;
    float* bwd_partial; int* bwd_nblk;     // host out: tiles written (0 = not fused)
    int* tile_counters;    // optional: zeroed ticket counters (one per output tile) for the
    int tile_counters_cap; //   in-kernel split-K reduction; NULL => carved from `scratch`
    float* scratch;        // optional split-K scratch (>= 2*M*y_ld floats to be considered)
    size_t scratch_floats;
    int force_cfg;         // -1 auto; 0/1/2: 128x128, 128x64, 64x64 register-staged; 3/4/5: the
                           // same tiles fed by LDS-DMA (tests/tuning)
    int force_splitk;      // 0 auto
    int splitk;            // set by the launcher
    int prio_mode;         // set by the launcher: wave_priority() mode (CILRS_PRIO)
    ConvMulti multi;       // set by launch_conv_dgrad (stride 2)
};
int launch_conv_igemm(const ConvArgs& a, hipStream_t s);
// fixed-order sum of dense [M][C] slabs (+ addend) with BatchNorm column partials [2][C][rows]
int slab_reduce_rows(int M, int C, int* rows_per_block);
int launch_slab_reduce_cols(int mode, const float* slabs, int splits, float* y, const float* addend,
                            int M, int C, const float* bwd_z, const float* bwd_y,
                            const float* bwd_stats, int bwd_relu, float* partial, hipStream_t s);
// Data gradient of a forward conv (stride 1 or 2): dx[N,H,W,Cin] = dgrad(dy[N,Ho,Wo,Cout]) (+addend).
// Stride 2 is decomposed into the four output-parity classes, each a dense stride-1 problem over
// its own tap subset (no multiplications by the zeros of an up-sampled dy).
struct DgradArgs {
    const float* dy; const float* w; float* dx; const float* addend;
    const float* mask; int mask_ld; float mask_scale;
    int N, H, W, Cin, Ho, Wo, Cout, K, stride, pad;
    int dy_ld, dx_ld;
    float* scratch; size_t scratch_floats; int force_cfg, force_splitk;
    int* tile_counters; int tile_counters_cap;     // see ConvArgs
    const float* bwd_z; const float* bwd_y; const float* bwd_stats; int bwd_relu;
    float* bwd_partial; int* bwd_nblk;
};
int launch_conv_dgrad(const DgradArgs& a, hipStream_t s);

// ---- static wave priorities --------------------------------------------------------------------
// Blocks that share a CU run the same K loop: [index math, LDS write, barrier, first LDS reads]
// (~700 cycles with the matrix pipe idle for that wave) then a burst of dependent MFMAs.  With
// equal priorities the SIMD's arbiter hands the matrix pipe round-robin to every wave that has an
// MFMA ready, so co-resident waves finish their bursts together and sit out their gaps TOGETHER
// (tools/occupancy_probe.py: a launch costs KT x (0.29 + 0.46 n) us for n blocks per CU -- the
// 0.29 never overlaps).  Distinct priorities per co-resident block break the convoy: the highest
// one runs undisturbed and the others fill its gaps.
//   mode 1: (blockIdx.x / 256) & 3   -- blocks are dealt to the 256 CUs in order
//   mode 2: hardware wave slot & 3   -- HW_ID.wave_id, the slot this wave occupies on its SIMD
#if defined(__HIPCC__)
__device__ __forceinline__ void wave_priority(const int mode) {
    if (mode == 0) return;
    int p;
    if (mode == 1) p = (int)(blockIdx.x >> 8) & 3;
    else p = (int)__builtin_amdgcn_s_getreg((3 << 11) | 4) & 3;      // HW_REG_HW_ID[3:0]
    p = __builtin_amdgcn_readfirstlane(p);
    if (p == 0) __builtin_amdgcn_s_setprio(0);
    else if (p == 1) __builtin_amdgcn_s_setprio(1);
    else if (p == 2) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(3);
}
// Experiment (CILRS_PRIO = 16 + units, tools/occupancy_probe.py): instead of priorities, START the
// co-resident blocks out of phase -- block k of a CU sleeps k x units x 64 cycles before its K loop
__device__ __forceinline__ void wave_stagger(const int mode) {
    if (mode < 16) return;
    const int units = mode - 16;
    const int k = (int)((blockIdx.x >> 8) % 5u);
    for (int i = 0; i < k * units; ++i) __builtin_amdgcn_s_sleep(1);
}
#endif
int wave_priority_mode();      // CILRS_PRIO (default: see net.hip)

// ---- weight gradient (conv_wgrad.hip) --------------------------------------------------------
struct WgradArgs {
    const float* x;        // forward input  [N][H][W] pixels, x_ld apart, Cin used
    const float* dy;       // output grad    [N*Ho*Wo][dy_ld], Cout used
    float* dw;             // [Cout][KH][KW][Cin_dst]  (OHWI)
    float* slabs;          // scratch: splits * Cout*KH*KW*Cin floats
    int N, H, W, Cin;
    int Ho, Wo, Cout;
    int KH, KW, stride, pad;
    int x_ld, dy_ld;
    int Cin_dst;           // Cin of dw (3 for the stem whose x is channel-padded to 4)
    int accumulate;        // dw += result instead of dw = result
    int prio_mode;         // wave_priority() mode; launch_conv_wgrad fills it from CILRS_PRIO
};
size_t wgrad_scratch_floats(const WgradArgs& a);
int launch_conv_wgrad(const WgradArgs& a, hipStream_t s);

// ---- BatchNorm / pooling (bn_pool.hip) --------------------------------------------------------
// stats: 4*C floats (mean | rstd | w | b); coef: 3*C floats; partial: bn_partial_floats(C) floats
size_t bn_partial_floats(int C);
// counters of the finalize-inside-apply launches: 8 shards x 32 ints of device memory, zero when
// `total` is zero; the launchers advance `total` (cumulative arrivals per shard) -- one object per
// stream of BatchNorm launches, never shared between two launches that may overlap
struct BnSync { int* dev; int total; };
constexpr int kBnSyncInts = 8 * 32;
// pre_nblk > 0: `partial` already holds pre_nblk per-tile partial sums (fused into the conv)
int launch_bn_train_fwd(const float* y, int M, int C, const float* gamma, const float* beta,
                        float* running_mean, float* running_var, long long* nbt, float momentum,
                        float eps, const float* residual, int relu, float* stats, float* partial,
                        float* z, int pre_nblk, hipStream_t s, void* z16 = nullptr,
                        BnSync* sync = nullptr);
// (sync != NULL: the per-channel finalize runs inside the apply launch instead of a launch of its
//  own -- bn_pool.hip, "finalize inside the apply launch")
// (z16 / dy16 / out16: optional bf16 shadow of the fp32 result -- the 16-bit operand of the next
//  convolution in the bf16 training mode; launch_bn_bwd: dy may then be NULL)
// eval-mode scale/shift of up to kMaxConvs BatchNorm layers in one launch (offsets in floats);
// 36 layers in the ResNet-34 network, 53 in the ResNet-50 variant
constexpr int kMaxConvs = 56;
struct BnEvalTable {
    int n;
    int C[kMaxConvs];
    unsigned gamma[kMaxConvs], beta[kMaxConvs];      // into the parameter arena
    unsigned rm[kMaxConvs], rv[kMaxConvs];           // into the BN buffer arena
    unsigned stats[kMaxConvs];         // into the workspace (4*C floats: mean|rstd|w|b)
};
int launch_bn_eval_stats_all(const BnEvalTable& t, const float* params, const float* bn_running,
                             float* ws, float eps, hipStream_t s);
int launch_bn_eval_fwd(const float* y, int M, int C, const float* gamma, const float* beta,
                       const float* running_mean, const float* running_var, float eps,
                       const float* residual, int relu, float* stats, float* z, hipStream_t s);
// pre_nblk > 0: `partial` already holds pre_nblk per-tile partial sums (fused into the dgrad)
int launch_bn_bwd(const float* dz, const float* z, const float* y, int M, int C,
                  const float* gamma, const float* stats, int relu, float* dgamma, float* dbeta,
                  int accumulate, float* coef, float* partial, float* dy, float* g_out,
                  int pre_nblk, hipStream_t s, void* dy16 = nullptr, BnSync* sync = nullptr);
// The same passes on bf16 NHWC tensors (bf16 training mode: 16-bit activations and gradients end to
// end, fp32 statistics and coefficients; C % 8 == 0).  y16 / z16 / residual16 / dz16 / dy16 /
// g_out16 are bf16; BatchNorm is the fp32 BatchNorm of the stored y16.
int launch_bn16_train_fwd(const void* y16, int M, int C, const float* gamma, const float* beta,
                          float* running_mean, float* running_var, long long* nbt, float momentum,
                          float eps, const void* residual16, int relu, float* stats, float* partial,
                          void* z16, int pre_nblk, hipStream_t s);
int launch_bn16_bwd(const void* dz16, const void* z16, const void* y16, int M, int C,
                    const float* gamma, const float* stats, int relu, float* dgamma, float* dbeta,
                    float* coef, float* partial, void* dy16, void* g_out16, int pre_nblk,
                    hipStream_t s);
int launch_avgpool_bwd16(const float* dout, void* dx16, int N, int HW, int C, int dout_ld,
                         hipStream_t s);
// stem: BatchNorm apply + ReLU + max-pool without materialising the post-BN tensor, and its backward
int launch_bn_relu_maxpool_fwd(const float* y, const float* stats, float* out,
                               unsigned char* argmax, int N, int H, int W, int C, hipStream_t s,
                               void* out16 = nullptr);
int launch_bn_bwd_pool(const float* dpool, const unsigned char* argmax, const float* y, int N,
                       int H, int W, int C, const float* gamma, const float* stats, float* dgamma,
                       float* dbeta, float* coef, float* partial, float* dy, hipStream_t s);
int launch_maxpool_fwd(const float* x, float* out, unsigned char* argmax, int N, int H, int W,
                       int C, hipStream_t s);
int launch_maxpool_bwd(const float* dout, const unsigned char* argmax, float* dx, int N, int H,
                       int W, int C, hipStream_t s);
int launch_avgpool_fwd(const float* x, float* out, int N, int HW, int C, int out_ld,
                       hipStream_t s);
int launch_avgpool_bwd(const float* dout, float* dx, int N, int HW, int C, int dout_ld,
                       hipStream_t s);

// ---- heads, loss, optimiser, transforms (heads_optim.hip) -------------------------------------
int launch_nchw3_to_nhwc4(const float* x, float* out, int N, int H, int W, long sn, long sc,
                          long sh, long sw, hipStream_t s);
int launch_u8hwc_to_nhwc4(const unsigned char* x, float* out, size_t npix, const float* mean,
                          const float* stdv, hipStream_t s);
int launch_camera_to_nhwc4(const unsigned char* src, float* out, int B, int sh, int sw,
                           int pix_stride, long row_stride, long frame_stride, int H, int W,
                           const float* mean, const float* stdv, hipStream_t s);
int launch_pad_cin3_to_4(const float* w3, float* w4, int n_taps_total, hipStream_t s);
int launch_linear_small_fwd(const float* x, const float* w, const float* bias, float* y, int B,
                            int in, int out, int x_ld, int y_ld, int relu, hipStream_t s);
int launch_linear_small_bwd(const float* dy, const float* x, const float* w, const float* act,
                            float act_scale, float* dx, float* dw, float* db, int B, int in,
                            int out, int dy_ld, int x_ld, int dx_ld, int act_ld, int accumulate,
                            hipStream_t s);
int launch_colsum(const float* dy, float* db, int B, int out, int dy_ld, int accumulate,
                  hipStream_t s);
int launch_relu_mask(float* d, const float* act, int B, int cols, int d_ld, int act_ld,
                     float scale, hipStream_t s);
// The reference builds one control branch per command (autonomous_drive.py:362, 380-381); every
// caller passes 4.  Plans are generic up to kMaxCmd branches (a plan parameter, encoded in the
// variant code: trunk | num_commands << 8, 0 meaning 4).
constexpr int kMaxCmd = 8;
struct SumParts { const float* p[kMaxCmd]; int n; const float* tail; };   // n branch parts + the speed head's
int launch_sum_parts(const SumParts& parts, float* out, int B, int cols, int cols_tail, hipStream_t s);
int launch_dropout(float* a, int B, int cols, int ld, float p, unsigned long long seed,
                   unsigned long long stream, hipStream_t s);
// small-batch inference heads (eval, B <= kHeadsSmallMaxB): only the commanded branch of each
// sample is evaluated (the reference evaluates all four and gathers, autonomous_drive.py:394-398;
// the discarded ones do not influence the result).  Chain 0 = control branches, 1 = speed head.
constexpr int kHeadsSmallMaxB = 16;
struct HeadsSmallArgs {
    const float* x[2];       // input rows per chain, leading dimension x_ld
    const float* w[kMaxCmd + 1];   // branch 0..ncmd-1 weights [out][in], then (index ncmd) the speed-head layer
    const float* b[kMaxCmd + 1];
    int ncmd;
    float* y[2];
    int y_ld[2];
    int out[2];
    int in[2];               // input features per chain
    int x_ld, relu, B;
    const long long* cmd;    // [B] command per row
    int* status;             // set to 1 on a command outside 0..ncmd-1 (may be NULL)
};
int launch_heads_small_pre(const float* feat, int HW, const float* speed, const float* w0,
                           const float* b0, const float* w1, const float* b1, float* combined,
                           int B, hipStream_t s);
int launch_heads_small_layer(const HeadsSmallArgs& a, hipStream_t s);
int launch_augment_u8(const unsigned char* frames, const cilrs_aug_params* params, int B, int H,
                      int W, float* out_f32, unsigned char* out_u8, hipStream_t s);
// ---- single-frame inference convolution (conv_small.hip) --------------------------------------
struct ConvSmallArgs {
    const float* x; const float* w; float* y;      // NHWC / OHWI / NHWC, dense
    const float* scale; const float* shift;        // folded eval-mode BatchNorm per output channel
    const float* addend;                           // residual, layout of y (may be NULL)
    int relu, relu_post;
    int N, H, W, Cin, Ho, Wo, Cout, K, stride, pad;
};
int launch_conv_small(const ConvSmallArgs& a, hipStream_t s);
// measured (tools/f16_probe.py): at B=1 it beats split-K + reduce on layers 2-3 (144 / 80 tiles,
// reduction length <= 2304) and loses on layer1 (276 tiles: mostly idle 16-wave blocks), on layer4
// (reduction length 4608: two dependent rounds of loads per wave) and on layer3 at B=4 (320 tiles)
constexpr int kSmallConvBlocks = 256;              // 16x16 tiles up to which it is used
constexpr int kSmallConvK = 2304;                  // reduction length up to which it is used

// ---- Winograd F(2x2, 3x3) convolution (conv_wino.hip): 3x3 / stride 1 / pad 1 ------------------
struct WinoArgs {
    const float* x;          // [N][H][W][C] NHWC (forward: activations; data gradient: dy)
    const float* U;          // transformed filters [16][C/8][K][8] (launch_wino_weights)
    float* y;                // [N][H][W][K]
    const float* addend;     // optional, y's layout: y = conv + addend
    int N, H, W, C, K;
    float* bn_partial;       // optional: BatchNorm batch statistics of the output, per-block column
                             // partials [2][K][wino_groups] (sum | sum of squares)
    // optional (data gradient): BatchNorm-backward reductions of the layer this output feeds,
    // [2][K][wino_groups] partials of g and g * xhat (as ConvArgs::bwd_*)
    const float* bwd_z; const float* bwd_y; const float* bwd_stats; int bwd_relu;
    float* bwd_partial;
    long long* stamps;       // diagnostics: block 0's waves 0 / 4 write 2 x 8 cycle counts (NULL: off)
    // channel split (an under-filled launch: fewer blocks than CUs): the reduction over the C input
    // channels is cut into `csplit` parts on csplit x as many blocks, each writes its partial
    // result to a slab, the slabs are summed in fixed order by launch_slab_reduce_cols (which also
    // applies the addend and emits the column partials).  slabs: caller's scratch of slab_floats
    // floats (>= csplit * N*H*W*K to be considered); scratch_partial: where the reduce may park
    // column partials nobody asked for.  csplit is filled in by the launcher.
    float* slabs; size_t slab_floats; float* scratch_partial; int csplit;
    // filled in by launch_conv_wino (a launch = full 64-tile blocks + a tail of 16-tile blocks):
    int no_tail;             // caller: 1 = all tiles on the 64-tile kernel (one launch)
    int tile_begin;          // first output tile of this kernel's block 0
    int row0, rows;          // column-partial row of block 0 / rows of the whole launch
};
size_t wino_weight_floats(int K, int C);
// U from OHWI weights w[K][3][3][C]; dgrad = 1: the filter of the data gradient (taps flipped,
// channel roles swapped: reduction over K, C output channels)
int launch_wino_weights(const float* w, float* U, int K, int C, int dgrad, hipStream_t s);
bool wino_supported(int C, int K, int ksize, int stride, int pad);
int wino_groups(int N, int H, int W);        // 64-tile groups of a launch
// column-partial rows launch_conv_wino writes for these sizes (full groups + 16-tile tail groups)
// (C > 0 and slab_floats: the channel split may change the answer)
int wino_rows(int N, int H, int W, int K, int no_tail = 0, int C = 0, size_t slab_floats = 0);
int launch_conv_wino(const WinoArgs& a, hipStream_t s);
int wino_last_csplit();                      // parts per tile of the most recent launch
// Weight gradient of the same convolutions in the Winograd domain:
//   dU_xi[k][c] = sum over tiles of (A dY A^T)_xi[tile][k] * (B^T d B)_xi[tile][c],  dw = G^T dU G
// (16 products per tile, channel pair and tap set instead of 36).  The tile range is split over
// `splits` blocks per 64 x 64 channel tile (chosen so that one round fills the chip); each block
// writes an OHWI slab, summed in slab order by launch_wgrad_reduce.
struct WinoWgradArgs {
    const float* x;          // [N][H][W][C] input activations of the convolution
    const float* dy;         // [N][H][W][K] output gradient
    float* dw;               // [K][3][3][C]
    float* slabs;            // wino_wgrad_scratch_floats floats
    int N, H, W, C, K;
    int accumulate;
    // filled in by the launcher:
    int splits, tiles_per_split;
};
bool wino_wgrad_supported(int C, int K, int ksize, int stride, int pad);
size_t wino_wgrad_scratch_floats(int N, int H, int W, int C, int K);
int launch_conv_wino_wgrad(const WinoWgradArgs& a, hipStream_t s);
int launch_wgrad_reduce(const float* slabs, float* dw, int splits, size_t n, int accumulate,
                        hipStream_t s);
int wino_prepare();                          // one-time kernel attribute (call outside stream capture)
// Both filter forms of every Winograd convolution of a network in ONE launch (the weights change
// every optimiser step): block = 8 output x 32 input channels of one layer, staged through LDS so
// that both images are written in 256-byte / 1-KB contiguous runs.
struct WinoWeightTable {
    int n;
    int K[kMaxConvs], C[kMaxConvs];
    unsigned w[kMaxConvs];           // OHWI weights in the parameter arena (floats)
    unsigned u[kMaxConvs];           // forward form  [16][C/8][K][8], floats from `ubase`
    unsigned ud[kMaxConvs];          // data-gradient form [16][K/8][C][8]
    int blk_begin[kMaxConvs + 1];    // prefix sums of (K/8) * (C/32)
};
int launch_wino_weights_all(const WinoWeightTable& t, const float* params, float* ubase, hipStream_t s);

// ---- persistent single-frame inference kernel (infer_b1.hip) ---------------------------------
// The whole eval forward of ONE frame (reference control loop, model/autonomous_drive.py:908-920)
// as ONE launch: one 1,024-thread workgroup per CU walks a table of stages (preprocess, stem,
// max-pool, 33 convolution stages, three head layers) separated by in-launch grid barriers.
// All byte offsets are 32-bit: activations / folded BatchNorm tables / padded stem weights live in
// the plan's workspace, every other weight in the parameter arena.
struct B1Conv {              // 32 ints, 16-byte aligned inside B1Stage (read as eight 128-bit words)
    unsigned x_off, y_off, add_off, w_off;       // bytes (w: arena; the stem's padded copy: ws)
    unsigned scale_off, shift_off;               // folded BatchNorm per output channel (ws bytes)
    int H, W;
    int Cin, Wo, Cout, K;                        // K = 3 (pad 1), 1 (pad 0) or 7 (stem, pad 3)
    int stride, M, nmt, ntiles;                  // output pixels, 16-row tiles, 16x16 output tiles
    int S, cshift, krow4, relu;                  // k-groups per tile; log2(Cin/16); K*K*Cin*4
    int relu_post, has_add;
    unsigned wo_magic, nmt_magic;                // x / Wo == (x * wo_magic) >> 20 on the ranges used
    int ksplit, sper, per;                       // workgroups per tile (split of the reduction index),
    unsigned ks_magic;                           //   k-groups per workgroup / per wave; x / ksplit magic
    unsigned slab_off;                           // ksplit > 1: partial tiles [ksplit][nmt*16][Cout] (ws)
    int ticket0, nunits, nt;                     // first ticket word; units = ntiles / nt * ksplit;
                                                 // channel tiles per unit (1, or 2 sharing the
                                                 // activation fragments; then ksplit == 1)
};
struct B1Head {              // one nn.Linear of the commanded branch (chain 0) + speed head (chain 1)
    unsigned w_off[5], b_off[5];     // branch 0..3, then the speed-predictor layer (arena bytes)
    unsigned x_off[2], y_off[2];     // workspace bytes: input rows / outputs of the two chains
    int in[2], out[2], relu, first, last;
    // first layer only: avg-pool source + speed encoder (autonomous_drive.py:369-374, 390-392)
    unsigned feat_off; int featHW, featC;
    unsigned se_w0, se_b0, se_w1, se_b1;
};
enum { B1_PRE = 0, B1_CONV = 1, B1_STEM = 2, B1_POOL = 3, B1_HEAD = 4 };
struct B1Stage {
    int type, wpt, nunits0, total_units;   // conv stages: waves per unit (tile x k-slice), units of
                                           // sub-problem 0, units of both
    B1Conv c[2];
    B1Head h;
    unsigned src_off, dst_off; int pH, pW, pC, pHo, pWo;    // B1_PRE / B1_POOL geometry
    unsigned cmd_off;        // where stage 0 parks the command for the head stages (ws bytes)
    int same_shape;          // conv stage laid out exactly like the previous one (only bases differ)
    int pad_[3];
};
constexpr int kB1MaxStages = 48;
constexpr int kB1Tickets = 512;            // split-K arrival tickets (one per output tile of a stage)
constexpr int kB1SyncInts = 9 * 32 + kB1Tickets;   // 8 counter shards + the epoch base (one 128-B
                                                   // line each), then the tickets
struct B1Launch {
    const B1Stage* table; int nstages;     // device table
    int first_stage;                       // 0; 1 = the normalised NHWC4 image is already in place
    float* ws; size_t ws_bytes;
    const float* params; size_t param_bytes;
    const unsigned char* frame; const float* speed; const long long* cmd;
    float* controls; float* pred_speed;
    int* sync;                              // kB1SyncInts ints, zeroed once
    int* status;                            // [0] bad command, [1] grid barrier gave up
    int* done; int seq;                     // optional: word (pinned host memory) that receives `seq`
                                            // right after the outputs are written
    long long* stamps;                      // optional [2][kB1MaxStages + 1]: block 0's 100 MHz clock at
                                            // every stage start / work end (diagnostics; NULL = off)
    float mean[3], stdv[3];
};
int infer_b1_grid(int* blocks);             // resident grid size (0: kernel cannot run here)
int launch_infer_b1(const B1Launch& a, int blocks, hipStream_t s);

// ---- fp16 inference trunk (infer_f16.hip) -------------------------------------------------------
typedef _Float16 cilrs_half;
struct ConvF16Args {
    const cilrs_half* x;          // [N][H][W][Cin] fp16
    const cilrs_half* w;          // [Cout][K][K][Cin] fp16, BatchNorm scale folded in
    const float* bias;            // [Cout] folded BatchNorm shift
    const cilrs_half* residual;   // [N][Ho][Wo][Cout] or NULL
    cilrs_half* y;                // [N][Ho][Wo][Cout]
    int N, H, W, Cin, Ho, Wo, Cout, K, stride, pad, relu;
    int bf16;                     // 0: the 16-bit buffers hold fp16, 1: bf16
    // ---- training use (16-bit operands, fp32 results; launch_conv_f16_train) ----
    float* y32;                   // [N][Ho][Wo][Cout] fp32 raw result (no bias / ReLU)
    const float* addend32;        // optional fp32 tensor of y32's shape added to the result
    float* bn_partial;            // optional per-M-tile column sums / sums of squares of the raw
                                  // result, channel-major [2][Cout][M-tiles] (as ConvArgs)
    // optional (data gradient, stride 1): BatchNorm-backward reductions of the layer whose output
    // gradient this launch produces, per M-tile [2][Cout] partials of g and g * xhat with
    // g = result * (z > 0 if bwd_relu), xhat = (y - mean) * rstd  (as ConvArgs::bwd_*)
    const float* bwd_z; const float* bwd_y; const float* bwd_stats; int bwd_relu;
    float* bwd_partial;
    int up2;                      // 1: x is the output gradient of a stride-2 convolution and the
                                  // enumerated grid [Ho][Wo] its INPUT: tap (kh', kw') of the
                                  // flipped filter reads x at ((h + pad - kh) / 2, (w + pad - kw) / 2),
                                  // kh = K-1-kh', when both differences are even (pad = forward pad)
    // ---- 16-bit tensors end to end (the bf16 training mode since round 4): the raw result is
    //      ROUNDED to 16 bits on its way out, y16 = round(acc + addend16), and everything derived
    //      from it -- BatchNorm batch statistics (bn_partial) and the BatchNorm-backward
    //      reductions (bwd_*16) -- is computed from the rounded values, so that BatchNorm is
    //      exactly "fp32 BatchNorm of the stored 16-bit tensor" (what autocast-style training does)
    cilrs_half* y16;              // 16-bit raw result instead of y32 (exactly one of the two)
    const cilrs_half* addend16;   // optional 16-bit addend (either output type)
    const cilrs_half* bwd_z16; const cilrs_half* bwd_y16;   // 16-bit forms of bwd_z / bwd_y
    // ---- stride-2 data gradient by output-parity class (up2 == 2): the enumerated grid is the
    //      sub-grid (h, w) = (2 i + ph, 2 j + pw) of the convolution's INPUT, its taps the
    //      cls_ntaps filter taps of that parity (cls_tap[t] = flipped-tap index into w,
    //      cls_dh/dw[t] = offset of the gathered dy pixel from (i, j)); all four classes in ONE
    //      launch, tile ranges cls_tile_begin[0..4]
    int cls_Ho[4], cls_Wo[4], cls_ph[4], cls_pw[4], cls_ntaps[4], cls_tile_begin[5];
    int cls_tap[4][4], cls_dh[4][4], cls_dw[4][4];
};
// the same implicit GEMM with fp32 output: forward (w = 16-bit copy of the OHWI weights), data
// gradient (w = transposed, tap-flipped 16-bit copy; stride 1: pad = K-1-pad_fwd; stride 2: up2)
int launch_conv_f16_train(const ConvF16Args& a, hipStream_t s);
int conv_f16_train_mtiles(const ConvF16Args& a);         // rows of the column partials it writes
int launch_conv16_large(const ConvF16Args& a, hipStream_t s);   // conv16.hip; -1: not taken
void conv_f16_up2_classes(ConvF16Args& c);               // fills the parity-class tables (up2 = 2)
bool conv_f16_train_can_fuse_bwd(const ConvF16Args& a);  // BatchNorm-backward partials possible?
// wT[ci][K-1-kh][K-1-kw][co] = (16-bit) w[co][kh][kw][ci]
int launch_transpose_flip_f16(const float* w, void* wT, int Cout, int K, int Cin, int bf16,
                              hipStream_t s);
struct TransposeF16Table {
    int n;
    int cout[kMaxConvs], k[kMaxConvs], cin[kMaxConvs];
    unsigned w[kMaxConvs];        // fp32 weights in the parameter arena (floats)
    unsigned wT[kMaxConvs];       // 16-bit elements into the transposed-weight arena
    int tile_begin[kMaxConvs + 1];   // prefix sums of K*K*(Cout/32)*(Cin/32) 32x32 tiles
};
int launch_transpose_flip_f16_all(const TransposeF16Table& t, const float* params, void* wT16,
                                  int bf16, hipStream_t s);
struct FoldF16Table {
    int n;
    int cout[kMaxConvs];
    unsigned krow[kMaxConvs];     // K*K*Cin
    unsigned w[kMaxConvs];        // fp32 weights in the parameter arena
    unsigned stats[kMaxConvs];    // eval-mode BN stats in the workspace (mean|rstd|scale|shift)
    unsigned w16[kMaxConvs];      // halfs into the folded-weight arena
    unsigned bias[kMaxConvs];     // floats into the folded-bias arena
};
int launch_conv_f16(const ConvF16Args& a, hipStream_t s);
int launch_fold_bn_f16(const FoldF16Table& t, const float* params, const float* ws, void* w16,
                       float* bias, int bf16, hipStream_t s);
int launch_f32_to_f16(const float* x, void* y, size_t n, int bf16, hipStream_t s);
int launch_avgpool_f16(const void* x, float* out, int N, int HW, int C, int out_ld, int bf16,
                       hipStream_t s);

// fp32 stem of the training step (stem_f32.hip): conv 7x7/s2 on the channel-padded image with the
// weights in registers and k = 7 x (21 + 1); bn_partial [2][64][stem_f32_rows()] or NULL.
// stem_f32_rows() == 0: geometry not served (callers keep the implicit GEMM)
int stem_f32_rows(int N, int H, int W);
int launch_stem_f32(const float* x4, const float* w, float* y, float* bn_partial, int N, int H, int W,
                    hipStream_t s);
// its weight gradient: dw = OHWI [64][7][7][3], overwritten; scratch: stem_wgrad_f32_scratch_floats()
// floats (0: geometry not served -- rows of 100 or 200 output pixels only)
size_t stem_wgrad_f32_scratch_floats(int N, int H, int W);
int launch_stem_wgrad_f32(const float* x4, const float* dy, float* dw, float* scratch, int N, int H,
                          int W, hipStream_t s);
// 16-bit stem of the serving path (stem_f16.hip): conv 7x7/s2 + folded BN + ReLU on the channel-
// padded fp32 image -> 16-bit [N][Ho][Wo][64]; max-pool 3x3/s2/p1 on 16-bit NHWC
int launch_fold_stem_f16(const float* w, const float* stats, void* w16, float* bias, int bf16,
                         hipStream_t s);
int launch_stem_f16(const float* x4, const void* w16, const float* bias, void* z, int N, int H,
                    int W, int bf16, hipStream_t s);
int launch_maxpool_f16(const void* x, void* out, int N, int H, int W, int C, int bf16,
                       hipStream_t s);

// 16-bit weight gradient (wgrad_f16.hip): dw fp32 OHWI from 16-bit NHWC x and dy
struct WgradF16Args {
    const void* x;         // [N][H][W][Cin] 16-bit
    const void* dy;        // [N][Ho][Wo][Cout] 16-bit
    float* dw;             // [Cout][K][K][Cin] fp32
    float* slabs;          // wgrad_f16_scratch_floats() floats
    int N, H, W, Cin, Ho, Wo, Cout, K, stride, pad;
    int bf16;
};
size_t wgrad_f16_scratch_floats(const WgradF16Args& a);
int launch_wgrad_f16(const WgradF16Args& a, hipStream_t s);

// grouped small GEMMs of the heads (heads_gemm.hip): mode 0 NT (linear forward), 1 NN (input
// gradient), 2 TN (weight + bias gradient); one launch covers up to five chains
struct HGemmGroup {
    const float* A; const float* B; float* C;
    const float* bias;                 // NT: added per column (may be NULL)
    const float* mask;                 // NN: activation, gradient kept where > 0 (may be NULL)
    float* dbias;                      // TN: column sums of A (may be NULL)
    unsigned long long drop_stream;    // NT: dropout stream id of this chain's layer (kNoDrop: none)
    float mask_scale;                  // NN: factor applied where the mask is > 0
    int M, N, K, lda, ldb, ldc, ldmask;
    int vec;                           // set by the launcher
};
struct HGemmArgs {
    HGemmGroup g[kMaxCmd + 1];
    int ngroups, relu, accumulate;
    float drop_p;
    unsigned long long seed;
};
constexpr unsigned long long kNoDrop = ~0ull;
int launch_hgemm(int mode, HGemmArgs& a, hipStream_t s);
constexpr int kEvalAccDoubles = 72;
int launch_eval_accumulate(const float* pc, const float* tc, const float* ps, const float* ts,
                           const long long* cmd, int B, double* acc, float* steer_err,
                           hipStream_t s);
int launch_branch_gather(const float* all_out, const long long* cmd, float* controls, int B,
                         int nbranch, int* status, hipStream_t s);
int launch_branch_scatter(const float* dcontrols, const long long* cmd, float* d_all, int B,
                          int nbranch, hipStream_t s);
int launch_loss(const float* pc, const float* tc, const float* ps, const float* ts, int B,
                int kind, const float* w, float grad_scale, float* dpc, float* dps, float* out,
                hipStream_t s);
size_t sqnorm_scratch_bytes();
int launch_grad_sqnorm(const float* g, size_t n, float max_norm, double* partial, float* out,
                       hipStream_t s);
int launch_adam(float* p, const float* g, float* m, float* v, size_t n, double lr, double beta1,
                double beta2, double eps, double wd, long long step, const float* clip_out,
                float gscale, hipStream_t s);
int launch_scale(float* g, size_t n, const float* coef_ptr, float c, hipStream_t s);
int launch_keep_head_inputs(const float* speed, const long long* cmd, float* speed_dst,
                            long long* cmd_dst, int B, hipStream_t s);

}  // namespace cilrs
