/* C-ABI of libcilrs_hip.so -- the MI355X (gfx950) engine behind the CILRS operator boundary.
 *
 * The reference has no FFI: its boundary for this path is the torch.nn.Module `CILRS`
 * (model/autonomous_drive.py:361-399 == notebook/notebook.ipynb:440-477), its loss
 * (notebook/notebook.ipynb:504-527) and torch.optim.Adam + clip_grad_norm_
 * (notebook/notebook.ipynb:533-534, 553-555).  Each entry point below names the reference
 * interface it replaces.  The Python mirror of that nn.Module (cilrs_mi355.CILRS) binds these
 * symbols with ctypes -- see INTEGRATION.md for the stub a reference maintainer would add.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller (PyTorch-ROCm's allocator) and only
 *    borrowed for the call, except `out_*` scalars explicitly marked host;
 *  - `stream` is a hipStream_t; every function only ENQUEUES work and never synchronises
 *    (cilrs_net_profile_collect excepted);
 *  - return value 0 = ok; non-zero = error, text in cilrs_last_error() (thread-local);
 *  - activations are NHWC fp32, conv weights OHWI fp32 (torch channels_last memory), all
 *    arithmetic fp32 (exact-f32 MFMA), reductions deterministic.
 */
#ifndef CILRS_HIP_H
#define CILRS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cilrs_net cilrs_net;

int cilrs_version(void);
const char* cilrs_last_error(void);

/* ---- parameter / buffer arena layout (replaces nn.Module.parameters()/state_dict() order,
 *      autonomous_drive.py:497 strict key contract) ----------------------- */
/* number of parameter tensors (142), of BatchNorm layers (36) */
int cilrs_num_params(void);
int cilrs_num_bn(void);
/* floats in the parameter arena (each tensor 16-byte aligned; >= 22,421,453) and exact count */
size_t cilrs_param_arena_floats(void);
size_t cilrs_param_count(void);
/* tensor i: state_dict name, float offset into the arena, numel, logical torch shape (ndim<=4).
 * Conv weights are stored OHWI at that offset (logical OIHW shape reported). */
int cilrs_param_info(int i, char* name, int name_cap, size_t* offset, size_t* numel, int* ndim,
                     int* shape4);
/* BatchNorm j: module prefix ("visual_encoder.1", ...), channels, float offsets of running_mean /
 * running_var inside the BN buffer arena; num_batches_tracked lives at int64 index j */
int cilrs_bn_info(int j, char* prefix, int prefix_cap, int* channels, size_t* rm_offset,
                  size_t* rv_offset);
size_t cilrs_bn_arena_floats(void);

/* ---- architecture variants ----------------------------------------------------------------------
 * variant 0: the reference's network (ResNet-34 trunk, autonomous_drive.py:365-370) -- everything
 *            above describes it.
 * variant 1: BASELINE.json configs[3] "ResNet-50 backbone variant": torchvision-style Bottleneck
 *            stacks [3,4,6,3] (stride on the 3x3 convolution), 2048-d features into the same
 *            speed encoder / four branches / speed predictor (first Linear layers 2176 and 2048
 *            wide).  The reference has no such model: parity is against the build's own CPU
 *            restatement (oracle/resnet50_oracle.py).  Trains in fp32 (train-mode forward,
 *            backward, the same segments); the fp16 / bf16 trunks are inference-only.
 * Number of commands: the reference's constructor builds one control branch per command
 * (model/autonomous_drive.py:362, 380-381).  Wherever a `variant` is passed it is the architecture
 * CODE  trunk | num_commands << 8  (trunk 0 or 1 as above; num_commands 1..8; 0 in the upper bits
 * = the 4 every caller of the reference passes).  Parameter names, arena offsets and gradient
 * segments follow the code; the persistent single-frame kernel exists for 4 commands only. */
int cilrs_num_variants(void);
int cilrs_variant_num_params(int variant);
int cilrs_variant_num_bn(int variant);
size_t cilrs_variant_param_arena_floats(int variant);
size_t cilrs_variant_param_count(int variant);
size_t cilrs_variant_bn_arena_floats(int variant);
int cilrs_variant_feature_width(int variant);
int cilrs_variant_param_info(int variant, int i, char* name, int name_cap, size_t* offset,
                             size_t* numel, int* ndim, int* shape4);
int cilrs_variant_bn_info(int variant, int j, char* prefix, int prefix_cap, int* channels,
                          size_t* rm_offset, size_t* rv_offset);

/* ---- network plan ----------------------------------------------------------------------------- */
typedef struct {
    float* params;          /* parameter arena                               */
    float* grads;           /* gradient arena, same layout (may be NULL for inference) */
    float* bn_running;      /* BN running_mean / running_var arena            */
    int64_t* bn_nbt;        /* [36] num_batches_tracked                       */
    void* workspace;        /* cilrs_net_workspace_bytes() bytes              */
} cilrs_buffers;

/* plan for a fixed batch / frame size (reference: B x 3 x 88 x 200) */
int cilrs_net_create(int batch, int height, int width, cilrs_net** out);
int cilrs_net_create_variant(int variant, int batch, int height, int width, cilrs_net** out);
/* The same with plan options.  CILRS_PLAN_BF16_TRAIN: "bf16 MFMA path" training (BASELINE.json
 * configs[3]) -- in train mode the trunk convolutions after the stem (forward, data gradient,
 * weight gradient) multiply bf16-rounded activations / weights / output gradients on
 * v_mfma_f32_32x32x16_bf16 with fp32 accumulation; BatchNorm, residual adds, the stem, the heads,
 * the loss, Adam and the master weights stay fp32.  Not the reference's arithmetic (it trains in
 * fp32): outputs and gradients agree with the fp32 path to bf16 rounding (~1e-2), not to 1e-4.
 * Eval-mode entry points are unaffected.  The workspace grows by the 16-bit shadow tensors. */
#define CILRS_PLAN_BF16_TRAIN 1u
int cilrs_net_create_ex(int variant, int batch, int height, int width, unsigned flags,
                        cilrs_net** out);
void cilrs_net_destroy(cilrs_net* net);
size_t cilrs_net_workspace_bytes(const cilrs_net* net);

/* CILRS.forward(image, speed, command) -> (controls[B,3], pred_speed[B])
 * (autonomous_drive.py:389-399).  image: f32 logical NCHW [B,3,H,W] with element strides
 * (sn,sc,sh,sw); speed f32 [B]; command int64 [B] in {0..3}.
 * train != 0: BatchNorm uses batch statistics and updates running stats (model.train());
 * train == 0: running statistics (model.eval()).  dropout_p applies only when train != 0. */
int cilrs_net_forward(cilrs_net* net, const cilrs_buffers* bufs, const float* image, long sn,
                      long sc, long sh, long sw, const float* speed, const int64_t* command,
                      int train, float dropout_p, uint64_t seed, float* controls,
                      float* pred_speed, void* stream);

/* Byte offset, inside the workspace, of the plan's int32[4] status words.  The library zeroes them
 * ONCE per workspace (the first entry point that sees a workspace pointer); after that the kernels
 * only ever SET them, so a word means "since the caller last cleared it" (sticky): read them after
 * synchronising the stream(s) the forwards ran on, and clear them with a 16-byte memset ordered
 * after those forwards.  Word 0 becomes 1 when a `command` value lies outside {0..num_commands-1}
 * -- the case in which the reference's torch.gather (autonomous_drive.py:397-398) raises; the
 * kernels then use branch 0 so nothing faults.  Word 1: a grid barrier of the persistent
 * single-frame launch gave up (outputs are NaN).  The host mirror reads the words at its next
 * synchronisation (Predictor: with the outputs; Trainer.losses() / validate()) and raises like
 * torch does. */
size_t cilrs_net_status_offset(const cilrs_net* net);

/* Inference between weight updates (the control loop, autonomous_drive.py:908-920, calls forward
 * once per simulator tick on weights loaded once, :496-498): a promise by the caller that the
 * parameter arena and the BatchNorm buffers are unchanged for as long as `key` keeps its value.
 * The plan then keeps what it derives from them -- every layer's eval-mode BatchNorm scale/shift,
 * the channel-padded stem weights, the 16-bit folded weights -- instead of recomputing it on
 * every eval forward (two to three launches per frame).  key 0 (the default) = no promise.
 * The Python mirror derives the key from the arenas' torch version counters plus a counter it
 * bumps on every train-mode forward and optimiser step. */
int cilrs_net_set_weights_key(cilrs_net* net, uint64_t key);

/* Where a train-mode forward left the tensors backward re-reads, as float offsets into the
 * workspace (NHWC, dense): convolution `conv` (parameter order: 0 stem, then every block's conv1,
 * conv2[, conv3][, downsample]) -- y = its raw output (input of its BatchNorm), z = after BatchNorm
 * (+ residual) (+ ReLU); conv -1: the max-pool output.  The stem's z is not materialised in train
 * mode (BatchNorm + ReLU are fused into the max-pool).  Test / diagnostics aid: the parity tests
 * count the ReLU decisions on which the engine and the oracle differ. */
int cilrs_net_activation_info(const cilrs_net* net, int conv, size_t* y_offset, size_t* z_offset,
                              size_t* numel, int* channels);

/* nn.Dropout(p) in training mode exactly as the fused heads apply it (inverted dropout, keep
 * where hash(seed, site, row * cols + col) >= p, kept values divided by 1 - p), in place over
 * a [rows][cols] matrix with row pitch ld.  `site` names the Dropout module:
 *   0 speed_encoder.2;  1 + 2k control_branches.k.2;  2 + 2k control_branches.k.5  (k = 0..3);
 *   9 speed_predictor.2      (autonomous_drive.py:371-387).
 * Applied to a matrix of ones it returns the mask (times 1/(1-p)) a train-mode forward with the
 * same seed used: the parity tests feed that mask to the oracle functionally. */
int cilrs_dropout(float* a, int rows, int cols, int ld, float p, uint64_t seed, int site,
                  void* stream);

/* Same, fed with uint8 RGB HWC frames [B,H,W,3]: fuses preprocess_image's /255, HWC->CHW and
 * Normalize(mean,std) (autonomous_drive.py:897-902; the cv2.resize is the caller's). */
int cilrs_net_forward_u8(cilrs_net* net, const cilrs_buffers* bufs, const uint8_t* frames,
                         const float* speed, const int64_t* command, float* controls,
                         float* pred_speed, void* stream);

/* Same, fed with raw camera frames of any size: fuses preprocess_image completely --
 * cv2.resize(frame, (W, H)) (8-bit INTER_LINEAR, restated; cv2 is absent from the build image so
 * this step is parity-unpinned), /255, HWC->CHW, Normalize (autonomous_drive.py:897-902).
 * frames: [B] images of src_h x src_w pixels, pixel_stride 3 or 4 bytes (the CARLA camera hands
 * over 600x800 BGRA and the agent keeps bytes 0..2 of each pixel, :868-872), row_stride and
 * frame_stride in bytes. */
int cilrs_net_forward_camera(cilrs_net* net, const cilrs_buffers* bufs, const uint8_t* frames,
                             int src_h, int src_w, int pixel_stride, long row_stride,
                             long frame_stride, const float* speed, const int64_t* command,
                             float* controls, float* pred_speed, void* stream);

/* cilrs_net_forward_u8 with the BasicBlock trunk in fp16 (batched serving, BASELINE config 5):
 * BatchNorm folded into fp16 weights on every call, fp16 NHWC activations, fp16 MFMA with fp32
 * accumulation; the stem and the heads stay fp32.  Outputs agree with the fp32 path to ~1e-3. */
int cilrs_net_forward_u8_f16(cilrs_net* net, const cilrs_buffers* bufs, const uint8_t* frames,
                             const float* speed, const int64_t* command, float* controls,
                             float* pred_speed, void* stream);
/* ... replayed from a cached hipGraph (same rules as cilrs_net_forward_u8_graph; one cached graph
 * per plan, re-captured when the mode or a pointer changes). */
int cilrs_net_forward_u8_f16_graph(cilrs_net* net, const cilrs_buffers* bufs,
                                   const uint8_t* frames, const float* speed,
                                   const int64_t* command, float* controls, float* pred_speed,
                                   void* stream);

/* The same with the trunk in bf16 (v_mfma_f32_32x32x16_bf16, fp32 accumulation): the "bf16 MFMA
 * path" of BASELINE.json configs[3]; works for both variants.  bf16 keeps 8 significant bits:
 * outputs agree with the fp32 path to ~1e-2. */
int cilrs_net_forward_u8_bf16(cilrs_net* net, const cilrs_buffers* bufs, const uint8_t* frames,
                              const float* speed, const int64_t* command, float* controls,
                              float* pred_speed, void* stream);
int cilrs_net_forward_u8_bf16_graph(cilrs_net* net, const cilrs_buffers* bufs,
                                    const uint8_t* frames, const float* speed,
                                    const int64_t* command, float* controls, float* pred_speed,
                                    void* stream);

/* cilrs_net_forward_u8 replayed from a cached hipGraph (re-captured when a pointer changes);
 * `stream` must be a non-default stream.  Single-frame control loop: predict_controls,
 * autonomous_drive.py:908-920. */
int cilrs_net_forward_u8_graph(cilrs_net* net, const cilrs_buffers* bufs, const uint8_t* frames,
                               const float* speed, const int64_t* command, float* controls,
                               float* pred_speed, void* stream);

/* cilrs_net_forward_u8 for ONE frame as ONE persistent launch (csrc/infer_b1.hip): the agent's
 * per-tick call `controls, pred_speed = self.model(img_t, speed_t, cmd_t)` under model.eval() and
 * torch.no_grad() (predict_controls, model/autonomous_drive.py:908-920).  One 1,024-thread
 * workgroup per CU walks a stage table (preprocess, stem, max-pool, the 16 BasicBlocks, avg-pool,
 * speed encoder, the COMMANDED branch and the speed head) separated by in-launch grid barriers;
 * no hipGraph, no per-layer launches.  Plans of batch 1 of the reference network only.  The device
 * must be able to keep one workgroup per CU resident: a second persistent launch running at the
 * same time on the same device (another stream, another process) can stall both until the bounded
 * barrier spin gives up -- then status word 1 is set and the outputs are NaN.  Status word 0 is
 * SET (never cleared: sticky, see cilrs_net_status_offset) by a call whose command lies outside
 * 0..3, where the reference's torch.gather raises. */
int cilrs_net_forward_u8_b1(cilrs_net* net, const cilrs_buffers* bufs, const uint8_t* frame,
                            const float* speed, const int64_t* command, float* controls,
                            float* pred_speed, void* stream);
/* The agent's whole tick on a raw camera frame (preprocess_image + model call,
 * model/autonomous_drive.py:897-920): the fused resize / normalise transform of
 * cilrs_net_forward_camera, then the persistent launch from its second stage.  `frame` may be
 * pinned host memory; sync != 0 ends with hipStreamSynchronize(stream). */
int cilrs_net_forward_camera_b1(cilrs_net* net, const cilrs_buffers* bufs, const uint8_t* frame,
                                int src_h, int src_w, int pixel_stride, long row_stride,
                                const float* speed, const int64_t* command, float* controls,
                                float* pred_speed, int sync, void* stream);
/* ... that posts its completion itself: right after the four outputs the launch stores `seq` into
 * `done` (a word of pinned host memory next to the outputs).  A control loop that keeps frame,
 * outputs and this word in ONE pinned buffer launches, spins on the word and reads the outputs --
 * no stream synchronisation on the tick's critical path (it saves the completion-signal and
 * wake-up latency, ~10 us of a ~0.3 ms tick).  The launch still completes on `stream` as usual. */
int cilrs_net_forward_u8_b1_post(cilrs_net* net, const cilrs_buffers* bufs, const uint8_t* frame,
                                 const float* speed, const int64_t* command, float* controls,
                                 float* pred_speed, int* done, int seq, void* stream);
/* ... followed by hipStreamSynchronize(stream): one library call per control-loop tick when the
 * frame / speed / command and the outputs live in pinned host memory (the kernel reads and writes
 * them in place; no copy commands). */
int cilrs_net_forward_u8_b1_sync(cilrs_net* net, const cilrs_buffers* bufs, const uint8_t* frame,
                                 const float* speed, const int64_t* command, float* controls,
                                 float* pred_speed, void* stream);
/* how many convolutions of this plan's TRAIN step run on the Winograd F(2x2,3x3) kernel (forward
 * and data gradient each; csrc/conv_wino.hip): the stride-1 3x3 layers of an fp32 train plan with
 * up to 256 channels and at least half a chip of 64-tile x 64-channel blocks.  CILRS_WINO=0 in the
 * environment of the process turns the path off (0 here), CILRS_WINO=2 drops the block-count rule. */
int cilrs_net_wino_convs(cilrs_net* net);
/* number of stages (= grid barriers + 1) of that launch; -1 before the first call, 0 if the plan
 * has no persistent path */
int cilrs_net_b1_stages(cilrs_net* net);
/* Re-base the launch's monotonic barrier counters (the eight arrival shards and the epoch word,
 * which are equal between launches) to `value`, after synchronising `stream`.  The counters are
 * compared wrap-safe (unsigned difference), so a long-running control loop never needs this; it
 * exists so that a test can place them just below INT_MAX and watch the launch cross the wrap
 * (tests/test_model_gpu.py), and as a maintenance hook. */
int cilrs_net_b1_set_epoch(cilrs_net* net, const cilrs_buffers* bufs, int value, void* stream);
/* diagnostics (library started with CILRS_B1_STAMPS=1): block 0's clock of the LAST persistent
 * launch -- start_us[i] = when stage i began (relative to stage 0), work_us[i] = how long block 0
 * worked in it before it entered the grid barrier.  Synchronous (one device->host copy). */
int cilrs_net_b1_stage_us(cilrs_net* net, const cilrs_buffers* bufs, float* start_us,
                          float* work_us, int cap);

/* CILRSLoss.forward + its gradient (notebook/notebook.ipynb:514-527).
 * kind 1: w0*L1(steer)+w1*L1(throttle)+w2*L1(brake)+w3*MSE(speed)     (executed config B)
 * kind 0: MSE(controls[B,3]) + w3*MSE(speed)                            (documented config A)
 * loss_out[6] (device) = total, control, steer, throttle, brake, speed.
 * dcontrols[B,3] / dpred_speed[B] = d total / d prediction, times grad_scale. */
int cilrs_loss_fwd_bwd(const float* controls, const float* target_controls,
                       const float* pred_speed, const float* target_speed, int batch, int kind,
                       const float* weights4_host, float grad_scale, float* dcontrols,
                       float* dpred_speed, float* loss_out, void* stream);

/* loss.backward() (notebook/notebook.ipynb:552) for the graph recorded by the last train-mode
 * cilrs_net_forward: writes (overwrites) every parameter gradient of the segments
 * [seg_begin, seg_end) into bufs->grads.  Segments, in execution order:
 * 0 heads, 1 layer4, 2 layer3, 3 layer2, 4 layer1, 5 stem  (so data-parallel callers can
 * all-reduce a finished segment's gradient range while the next one runs). */
int cilrs_net_backward(cilrs_net* net, const cilrs_buffers* bufs, const float* dcontrols,
                       const float* dpred_speed, int seg_begin, int seg_end, void* stream);
/* float range [begin,end) of the gradient arena that segment `seg` produces */
int cilrs_segment_range(int seg, size_t* begin, size_t* end);
/* the same for architecture variant `variant` (0 = the reference's ResNet-34, 1 = ResNet-50) */
int cilrs_variant_segment_range(int variant, int seg, size_t* begin, size_t* end);

/* torch.nn.utils.clip_grad_norm_ (notebook/notebook.ipynb:553-554): out[0] = total L2 norm,
 * out[1] = min(1, max_norm/(norm+1e-6)) (1 when max_norm <= 0); device results, no sync.
 * scratch: cilrs_sqnorm_scratch_bytes() bytes. */
size_t cilrs_sqnorm_scratch_bytes(void);
int cilrs_grad_sqnorm(const float* grads, size_t n, float max_norm, void* scratch, float* out2,
                      void* stream);

/* torch.optim.Adam.step() with coupled L2 weight decay (notebook/notebook.ipynb:533-534, 555)
 * over a flat arena; `step` is the 1-based step count; clip_out2 (may be NULL) is the device
 * result of cilrs_grad_sqnorm whose [1] scales the gradients; grad_scale multiplies them too
 * (1/world_size for data parallel sums). */
int cilrs_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                    size_t n, double lr, double beta1, double beta2, double eps,
                    double weight_decay, int64_t step, const float* clip_out2, float grad_scale,
                    void* stream);
int cilrs_scale(float* x, size_t n, const float* clip_out2, float c, void* stream);

/* op-level stem convolution of the TRAINING step (visual_encoder.0 = torchvision resnet34.conv1,
 * model/autonomous_drive.py:366; trained by notebook/notebook.ipynb:549-555): conv 7x7 / stride 2 /
 * pad 3, fp32, x4 = the channel-padded NHWC image [N,H,W,4], w = OHWI [64,7,7,3], y = [N,Ho,Wo,64];
 * bn_partial (may be NULL) receives the [2][64][*partial_rows] column partials (sum, sum of
 * squares) the following BatchNorm reduces.  Widths up to 445 pixels. */
int cilrs_stem_conv_fwd(const float* x4, const float* w, float* y, float* bn_partial, int N, int H,
                        int W, int* partial_rows, void* stream);

/* ... and its weight gradient (loss.backward() through that layer, notebook/notebook.ipynb:552):
 * dw = OHWI [64,7,7,3] (overwritten) from x4 and dy = [N,Ho,Wo,64]; scratch of
 * cilrs_stem_conv_wgrad_scratch_floats() floats.  Served geometries: rows of 100 or 200 output
 * pixels (the reference's 200x88 frames and the 400x176 variant); _scratch_floats() returns 0
 * otherwise and cilrs_conv2d_wgrad on the channel-padded image is the general path. */
size_t cilrs_stem_conv_wgrad_scratch_floats(int N, int H, int W);
int cilrs_stem_conv_wgrad(const float* x4, const float* dy, float* dw, float* scratch,
                          size_t scratch_floats, int N, int H, int W, void* stream);

/* loss.backward() + optimizer.step() (notebook/notebook.ipynb:552, 555) in ONE call for steps
 * without gradient clipping: cilrs_net_backward over all six segments, and the Adam update of a
 * segment's parameter range enqueued as soon as that segment's gradients are complete (on the
 * plan's weight-gradient stream, so the HBM-bound update of layer4's 13 M parameters runs under
 * the data gradients of layers 3..1).  Same arithmetic per element as cilrs_adam_step; when the
 * call returns its work to `stream`, parameters, moments and gradients are final.  A step that
 * clips (nb:553-554) needs the global norm first: cilrs_net_backward + cilrs_grad_sqnorm +
 * cilrs_adam_step. */
typedef struct {
    float* exp_avg;         /* arena-shaped first / second moments                     */
    float* exp_avg_sq;
    double lr, beta1, beta2, eps, weight_decay;
    int64_t step;           /* 1-based step count                                       */
    float grad_scale;       /* multiplies the gradients (1 / world_size for summed DP gradients) */
} cilrs_adam_args;
int cilrs_net_backward_step(cilrs_net* net, const cilrs_buffers* bufs, const float* dcontrols,
                            const float* dpred_speed, const cilrs_adam_args* opt, void* stream);

/* ---- training input pipeline on the device ------------------------------------
 * The reference's per-frame CPU augmentation (albumentations Compose, notebook/notebook.ipynb:387-394:
 * RandomBrightnessContrast p.5, HueSaturationValue p.3, GaussianBlur p.2, GaussNoise p.3,
 * CoarseDropout p.2) + /255 + Normalize (:412-414) as one kernel over a uint8 batch.  The random
 * parameters of each sample are drawn by the caller; a step is disabled by its *_on / 0 value.
 * albumentations / cv2 are absent from the build image: the steps restate their published
 * behaviour and are parity-unpinned against the libraries themselves. */
typedef struct {
    uint64_t noise_seed;     /* counter-based RNG key of this sample's Gaussian noise            */
    int rbc_on;              /* v' = trunc(clip(v * alpha + beta255))                             */
    float alpha, beta255;
    int hsv_on;              /* 8-bit HSV (H in half-degrees): H+hue mod 180, S+sat, V+val clipped */
    float hue, sat, val;
    int blur_k;              /* 0/1 = off, 3 or 5 = Gaussian taps; blur_w = centre, +-1, +-2      */
    float blur_w[3];
    float noise_std255;      /* 0 = off; sigma in grey levels                                      */
    int nholes;              /* 0..3 rectangles [y0,y1) x [x0,x1) set to 0                         */
    int hole_y0[3], hole_x0[3], hole_y1[3], hole_x1[3];
    int reserved;
} cilrs_aug_params;          /* 112 bytes */

/* frames uint8 [B,H,W,3] (device), params [B] (device) -> out_f32 [B,H,W,3] normalised floats
 * (feed it to cilrs_net_forward as the NCHW view with strides (H*W*3, 1, W*3, 3)) and/or out_u8
 * [B,H,W,3] augmented bytes; either output may be NULL. */
int cilrs_augment_u8(const uint8_t* frames, const cilrs_aug_params* params, int batch, int height,
                     int width, float* out_f32, uint8_t* out_u8, void* stream);

/* ---- evaluation report accumulators (metrics schema evaluation_report.json:1-73; the reference
 *      ships the report, not the code that made it) ---------------------------------------------
 * acc: cilrs_eval_acc_doubles() doubles on the device, zeroed by the caller before the first
 * batch and ADDED to by every call:
 *   [0..31]  channel c in (steer, throttle, brake, speed) x {n, Sp, St, Spt, Spp, Stt, S|d|, Sdd}
 *   [32..67] command k x {n, S|d_steer|, S|d_throttle|, S|d_brake|, then steer Sp, St, Spt, Spp, Stt}
 *   [68..71] rows with |d_steer| <= 0.01, 0.02, 0.05, 0.1
 * steer_abs_err (may be NULL): [batch] floats, |steer error| per row (for the percentiles). */
int cilrs_eval_acc_doubles(void);
int cilrs_eval_accumulate(const float* controls, const float* pred_speed,
                          const float* target_controls, const float* target_speed,
                          const int64_t* command, int batch, double* acc, float* steer_abs_err,
                          void* stream);

/* ---- per-kernel timing (hipEvents on the launch stream; feeds bench.py's roofline) ---------- */
int cilrs_net_profile_enable(cilrs_net* net, int on);
/* synchronises the recorded events and folds them into the per-label table */
int cilrs_net_profile_collect(cilrs_net* net);
int cilrs_net_profile_count(const cilrs_net* net);
int cilrs_net_profile_entry(const cilrs_net* net, int i, char* label, int label_cap,
                            long long* calls, double* total_ms, double* total_flops,
                            double* total_bytes);
int cilrs_net_profile_reset(cilrs_net* net);

/* ---- op-level entry points (unit parity tests; same kernels the plan launches) -------------- */
/* nn.Conv2d forward, NHWC x [N,H,W,Cin] (Cin % 4 == 0), OHWI w, y [N,Ho,Wo,Cout] (Cout % 64 == 0)
 * force_cfg: -1 auto, 0/1/2 tile config; force_splitk: 0 auto; scratch may be NULL */
int cilrs_conv2d_fwd(const float* x, const float* w, float* y, int N, int H, int W, int Cin,
                     int Cout, int KH, int KW, int stride, int pad, int force_cfg,
                     int force_splitk, float* scratch, size_t scratch_floats, void* stream);
/* d(loss)/dx of the same conv: dy [N,Ho,Wo,Cout] -> dx [N,H,W,Cin] (+ addend if non-NULL) */
int cilrs_conv2d_dgrad(const float* dy, const float* w, float* dx, const float* addend, int N,
                       int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                       int force_cfg, int force_splitk, float* scratch, size_t scratch_floats,
                       void* stream);
/* d(loss)/dw (OHWI, Cin_dst channels kept); scratch from cilrs_conv2d_wgrad_scratch_floats */
size_t cilrs_conv2d_wgrad_scratch_floats(int N, int H, int W, int Cin, int Cout, int KH, int KW,
                                         int stride, int pad);
int cilrs_conv2d_wgrad(const float* x, const float* dy, float* dw, float* scratch, int N, int H,
                       int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                       int Cin_dst, void* stream);

/* ---- the same operators on the 16-bit matrix pipe (BASELINE.json configs[3] "bf16 MFMA path",
 * training side): x / w / dy are fp32 tensors as above, rounded to bf16 (bf16 = 1) or fp16
 * (bf16 = 0) into `scratch16` (cilrs_conv2d_16_scratch_halfs() 16-bit elements, 16-byte aligned),
 * multiplied on v_mfma_f32_32x32x16_* with fp32 accumulation; results are fp32.  Cin and Cout
 * multiples of 64, square filters K x K (<= 16 taps), stride 1 or 2.  Not the reference's
 * arithmetic (it trains in fp32): checked against the same product of the ROUNDED operands. */
size_t cilrs_conv2d_16_scratch_halfs(int N, int H, int W, int Cin, int Cout, int K, int stride,
                                     int pad);
/* y = conv(x, w); bn_partial (optional): per-64-row-tile column sums / sums of squares of y,
 * channel-major [2][Cout][ceil(M/64)] */
int cilrs_conv2d_fwd_16(const float* x, const float* w, float* y, float* bn_partial, int N, int H,
                        int W, int Cin, int Cout, int K, int stride, int pad, int bf16,
                        void* scratch16, void* stream);
int cilrs_conv2d_dgrad_16(const float* dy, const float* w, float* dx, const float* addend, int N,
                          int H, int W, int Cin, int Cout, int K, int stride, int pad, int bf16,
                          void* scratch16, void* stream);
size_t cilrs_conv2d_wgrad_16_scratch_floats(int N, int H, int W, int Cin, int Cout, int K,
                                            int stride, int pad);
int cilrs_conv2d_wgrad_16(const float* x, const float* dy, float* dw, float* scratch32, int N,
                          int H, int W, int Cin, int Cout, int K, int stride, int pad, int bf16,
                          void* scratch16, void* stream);

/* The bf16 training mode's operators on 16-bit tensors (round 4: every trunk tensor after the stem
 * -- activations, raw convolution outputs, gradients -- is stored in bf16; fp32 accumulation,
 * statistics and coefficients).  No reference counterpart (the reference trains in fp32,
 * notebook/notebook.ipynb:549-555); defined by oracle/bf16_emulation.py.
 *
 * cilrs_conv2d_train_16: implicit-GEMM convolution of the 16-bit NHWC tensor x16 [N][H][W][Cin] with
 * 16-bit weights w16 [Cout][K][K][Cin] over the output grid [N][Ho][Wo] (forward: the OHWI weights
 * rounded; data gradient: x16 = dy, w16 = the transposed, tap-flipped weights, pad = K-1-pad_fwd,
 * or up2 = 1 with stride 2 for the gradient of a stride-2 convolution).  Result = acc (+ addend16),
 * ROUNDED to 16 bits into y16, or fp32 into y32 when y16 is NULL.  bn_partial (optional):
 * BatchNorm batch statistics of the stored result as column partials [2][Cout][rows];
 * bwd_partial (optional, with bwd_z16 / bwd_y16 / bwd_stats): BatchNorm-backward reductions of the
 * produced gradient, g = result * (z > 0 if bwd_relu), partials of g and g * xhat.  *partial_rows
 * receives the number of partial rows (the launch's 64- or 128-row tiles). */
int cilrs_conv2d_train_16(const void* x16, const void* w16, void* y16, float* y32,
                          const void* addend16, float* bn_partial, const void* bwd_z16,
                          const void* bwd_y16, const float* bwd_stats, int bwd_relu,
                          float* bwd_partial, int N, int H, int W, int Cin, int Ho, int Wo, int Cout,
                          int K, int stride, int pad, int up2, int bf16, int* partial_rows,
                          void* stream);
/* BatchNorm2d (training) on bf16 NHWC tensors: statistics of y16 (from `partial` when pre_rows > 0:
 * rows written by cilrs_conv2d_train_16), z16 = round(relu?(bn(y16) (+ residual16))); and its
 * backward: dy16 = round(BatchNorm backward of g = dz16 * (z16 > 0 if relu)), g_out16 = g.
 * Semantics of cilrs_bn_train_fwd / cilrs_bn_bwd otherwise (nn.BatchNorm2d, nb:440-477). */
int cilrs_bn16_train_fwd(const void* y16, int M, int C, const float* gamma, const float* beta,
                         float* running_mean, float* running_var, int64_t* nbt, float momentum,
                         float eps, const void* residual16, int relu, float* stats, float* partial,
                         void* z16, int pre_rows, void* stream);
int cilrs_bn16_bwd(const void* dz16, const void* z16, const void* y16, int M, int C,
                   const float* gamma, const float* stats, int relu, float* dgamma, float* dbeta,
                   float* coef3c, float* partial, void* dy16, void* g_out16, int pre_rows,
                   void* stream);
/* nn.BatchNorm2d training forward (+ optional residual add, ReLU); stats: 4*C floats out */
/* The same 3x3 / stride 1 / pad 1 convolution (nn.Conv2d(C, K, 3, 1, 1, bias=False) of
 * torchvision's BasicBlock: conv1 of every non-first block and every conv2) and its data gradient
 * by Winograd F(2x2, 3x3) on the fp32 matrix pipe: 16 multiplications per 2x2 output tile and
 * channel instead of 36.  fp32 throughout; results differ from the direct sum by rounding
 * (a few 1e-6 relative).  w: OHWI like everywhere; scratch: transformed filters,
 * cilrs_conv2d_wino_scratch_floats(Cin, Cout) floats.  Cin %% 8 == 0, Cout %% 64 == 0 (dgrad: the
 * roles swap). */
size_t cilrs_conv2d_wino_scratch_floats(int Cin, int Cout);
int cilrs_conv2d_wino_fwd(const float* x, const float* w, float* y, int N, int H, int W, int Cin,
                          int Cout, float* scratch, void* stream);
/* the two halves on their own: U = transformed filters ([16][Cred/8][Cout][8]; dgrad = 1: the
 * data gradient's filter, reduction over the forward Cout), and the convolution on a ready U */
int cilrs_wino_filter_transform(const float* w, float* U, int Cin, int Cout, int dgrad, void* stream);
/* diagnostics: 16 int64 device words that block 0's waves 0 and 4 of every later
 * cilrs_conv2d_wino_pre launch fill with shader-cycle counts [prologue, multiply, refill, barrier
 * wait, epilogue, K loop] (NULL: off) */
int cilrs_conv2d_wino_stamps(long long* stamps16);
int cilrs_conv2d_wino_pre(const float* x, const float* U, float* y, const float* addend, int N, int H,
                          int W, int Cred, int Cout, void* stream);
/* cilrs_conv2d_wino_pre with the launch plan of the train step: an under-filled launch (fewer 64-tile
 * x 64-channel blocks than CUs: layer3 at B=128) is cut along the reduction over the Cred input
 * channels into up to four parts per tile, each part writes a slab, and a fixed-order reduce sums
 * them, adds the addend and emits the BatchNorm column partials [2][Cout][*partial_rows].
 * slabs: scratch of slab_floats floats; *csplit = parts per tile the launch used (1: not split). */
int cilrs_conv2d_wino_split(const float* x, const float* U, float* y, const float* addend,
                            float* bn_partial, int N, int H, int W, int Cred, int Cout, float* slabs,
                            size_t slab_floats, int* csplit, int* partial_rows, void* stream);
int cilrs_conv2d_wino_dgrad(const float* dy, const float* w, float* dx, const float* addend, int N,
                            int H, int W, int Cin, int Cout, float* scratch, void* stream);
/* Weight gradient of the same convolution in the Winograd domain (cuDNN conv bwd-filter, i.e. the
 * `loss.backward()` of notebook/notebook.ipynb:551 for a 3x3 / stride-1 layer): dw[Cout][3][3][Cin]
 * from x[N][H][W][Cin] and dy[N][H][W][Cout]; Cin % 64 == 0, Cout % 64 == 0.  scratch:
 * cilrs_conv2d_wino_wgrad_scratch_floats floats (per-split slabs, summed in fixed order). */
size_t cilrs_conv2d_wino_wgrad_scratch_floats(int N, int H, int W, int Cin, int Cout);
int cilrs_conv2d_wino_wgrad(const float* x, const float* dy, float* dw, int N, int H, int W, int Cin,
                            int Cout, float* scratch, size_t scratch_floats, void* stream);
size_t cilrs_bn_partial_floats(int C);
int cilrs_bn_train_fwd(const float* y, int M, int C, const float* gamma, const float* beta,
                       float* running_mean, float* running_var, int64_t* nbt, float momentum,
                       float eps, const float* residual, int relu, float* stats, float* partial,
                       float* z, void* stream);
int cilrs_bn_eval_fwd(const float* y, int M, int C, const float* gamma, const float* beta,
                      const float* running_mean, const float* running_var, float eps,
                      const float* residual, int relu, float* stats, float* z, void* stream);
int cilrs_bn_bwd(const float* dz, const float* z, const float* y, int M, int C,
                 const float* gamma, const float* stats, int relu, float* dgamma, float* dbeta,
                 float* coef3c, float* partial, float* dy, float* g_out, void* stream);
/* nn.Linear forward  y = relu?(x W^T + b)  and backward (autonomous_drive.py:371-387; the heads'
 * grouped kernels with one group): x [batch][in] (row pitch x_ld), w [out][in], y [batch][out].
 * backward: dw [out][in] = dy^T x, db [out] = column sums of dy (either may be NULL together),
 * dx [batch][in] = (dy W), kept where act > 0 and multiplied by act_scale when act != NULL. */
int cilrs_linear_fwd(const float* x, const float* w, const float* bias, float* y, int batch,
                     int in_features, int out_features, int x_ld, int y_ld, int relu,
                     void* stream);
int cilrs_linear_bwd(const float* dy, const float* x, const float* w, const float* act,
                     float act_scale, float* dx, float* dw, float* db, int batch,
                     int in_features, int out_features, int dy_ld, int x_ld, int dx_ld,
                     int act_ld, void* stream);
int cilrs_maxpool_fwd(const float* x, float* out, uint8_t* argmax, int N, int H, int W, int C,
                      void* stream);
int cilrs_maxpool_bwd(const float* dout, const uint8_t* argmax, float* dx, int N, int H, int W,
                      int C, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CILRS_HIP_H */
