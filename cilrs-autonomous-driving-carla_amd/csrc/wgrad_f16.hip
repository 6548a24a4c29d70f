// Weight gradient of a convolution on the 16-bit matrix pipe (the "bf16 MFMA path" of BASELINE.json
// configs[3], training side): dW[co][kh][kw][ci] (fp32, OHWI) = sum over output pixels p of
// dy[p][co] * x[pix(p, kh, kw)][ci], both operands 16-bit NHWC (bf16 or fp16), fp32 accumulation.
//
// The reduction index (output pixels) is the SLOW index of both operands, so neither is a
// k-contiguous MFMA operand: tiles are staged as plain [pixel][channel] row copies (16-byte global
// loads, a gathered input pixel outside the image reads zeros through the buffer range check) and
// read back TRANSPOSED with ds_read_b64_tr_b16 (cdna_hip_programming.md T10): a 16-lane group reads
// a 4-pixel x 16-channel block and every lane receives its channel's 4 pixels -- two reads make
// the 8 k-values per lane of v_mfma_f32_32x32x16_*.  Row pitch 192 (320) bytes: the four rows of a block
// land on four disjoint bank groups.
//
// One block = 64 output channels x 64 columns (one filter tap, 64 input channels) x one K-slab of
// output pixels; partial tiles go to fp32 slabs summed in slab order by the fp32 path's reduce
// kernel (deterministic, no atomics).
#include "common.h"

#include <stdlib.h>

namespace cilrs {
namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef short s8 __attribute__((ext_vector_type(8)));

constexpr int WPIX = 64;            // output pixels per K-step

template <typename T> struct WVec8;
template <> struct WVec8<_Float16> { typedef h8 type; };
template <> struct WVec8<__bf16> { typedef b8 type; };
__device__ __forceinline__ f32x16 wmfma(const h8 a, const h8 b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 wmfma(const b8 a, const b8 b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// BT = 64: one 32x32 accumulator per wave, 48 KB LDS (three blocks per CU).
// BT = 128 (Cin and Cout multiples of 128): four accumulators per wave, 80 KB LDS (two blocks per
// CU) -- twice the flops per byte staged and two transposed reads per MFMA instead of four.
template <int BT> constexpr int wpitch() { return BT == 64 ? 96 : 160; }   // 192 B / 320 B rows

template <typename T, int BT>
__global__ __launch_bounds__(256, 2) void wgrad_f16_kernel(const WgradF16Args a, const int Mpix,
                                                           const int splits, const int steps_per) {
    typedef typename WVec8<T>::type v8;
    constexpr int PITCH = wpitch<BT>();
    constexpr int TM = BT / 64, TN = BT / 64;               // MFMA tiles per wave
    constexpr int CPR = BT / 8, RPP = 256 / CPR, PASSES = WPIX / RPP;   // loader shape
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* As = reinterpret_cast<T*>(smem_raw);                 // [2][WPIX][PITCH]  dy tile
    T* Bs = As + 2 * WPIX * PITCH;                          // [2][WPIX][PITCH]  gathered x tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;                // wave tile: co (BT/2) wm.., col (BT/2) wn..

    const int tiles_ci = a.Cin / BT, taps = a.K * a.K;
    const int ntiles_col = taps * tiles_ci, ntiles_co = a.Cout / BT;
    int b = blockIdx.x;
    const int tcol = b % ntiles_col; b /= ntiles_col;
    const int tco = b % ntiles_co;
    const int split = b / ntiles_co;
    const int tap = tcol / tiles_ci, ci0 = (tcol - tap * tiles_ci) * BT, co0 = tco * BT;
    const int kh = tap / a.K, kw = tap - kh * a.K;
    const int nsteps_total = (Mpix + WPIX - 1) / WPIX;
    const int st_begin = split * steps_per;
    const int st_end = min(nsteps_total, st_begin + steps_per);
    const int nt = max(0, st_end - st_begin);

    // loader: thread -> pixel rows r0 + RPP i of the step and 16-byte chunk kq of the tile row
    const int kq = tid % CPR, r0 = tid / CPR;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.dy, 0, (int)(unsigned)((size_t)Mpix * a.Cout * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.x, 0, (int)(unsigned)((size_t)a.N * a.H * a.W * a.Cin * 2), 0x00020000);
    // pixel cursors of this thread's rows (advanced by WPIX per step without divisions)
    int pn[PASSES], poh[PASSES], pow_[PASSES];
    const int HoWo = a.Ho * a.Wo;
#pragma unroll
    for (int i = 0; i < PASSES; ++i) {
        const int p = st_begin * WPIX + r0 + RPP * i;
        pn[i] = p / HoWo;
        const int rem = p - pn[i] * HoWo;
        poh[i] = rem / a.Wo;
        pow_[i] = rem - poh[i] * a.Wo;
    }
    int ld_p = st_begin * WPIX;        // first pixel of the next step to load
    auto load_step = [&](f32x4(&ra)[PASSES], f32x4(&rb)[PASSES]) {
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            const int p = ld_p + r0 + RPP * i;
            const bool live = p < Mpix;
            const unsigned offA = live ? (unsigned)(((size_t)p * a.Cout + co0 + kq * 8) * 2) : 0xFFFFFFFFu;
            const int ih = poh[i] * a.stride - a.pad + kh, iw = pow_[i] * a.stride - a.pad + kw;
            const bool ok = live && ih >= 0 && iw >= 0 && ih < a.H && iw < a.W;
            const unsigned offB =
                ok ? (unsigned)((((size_t)(pn[i] * a.H + ih) * a.W + iw) * a.Cin + ci0 + kq * 8) * 2)
                   : 0xFFFFFFFFu;
            ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)offA, 0, 0));
            rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, (int)offB, 0, 0));
            // advance this row's cursor by WPIX pixels
            pow_[i] += WPIX;
            while (pow_[i] >= a.Wo) { pow_[i] -= a.Wo; ++poh[i]; }
            while (poh[i] >= a.Ho) { poh[i] -= a.Ho; ++pn[i]; }
        }
        ld_p += WPIX;
    };
    auto store_step = [&](int buf, const f32x4(&ra)[PASSES], const f32x4(&rb)[PASSES]) {
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            *reinterpret_cast<f32x4*>(&As[(buf * WPIX + r0 + RPP * i) * PITCH + kq * 8]) = ra[i];
            *reinterpret_cast<f32x4*>(&Bs[(buf * WPIX + r0 + RPP * i) * PITCH + kq * 8]) = rb[i];
        }
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // transposed-read addresses: lane (group g, q, p) supplies row q, columns 4p..4p+3 of the block
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int trow = 8 * (g >> 1) + q;                       // + 16 * kstep (+ 4 for the high half)
    const int tcolA = wm * (BT / 2) + 16 * (g & 1) + 4 * pp; // co within the tile (+ 32 i)
    const int tcolB = wn * (BT / 2) + 16 * (g & 1) + 4 * pp; // ci within the tile (+ 32 j)
    auto tr = [&](const T* p) -> s4 {
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s4*)(p));
    };
    auto compute = [&](int buf) {
        const T* Ab = As + buf * WPIX * PITCH;
        const T* Bb = Bs + buf * WPIX * PITCH;
#pragma unroll
        for (int ks = 0; ks < WPIX / 16; ++ks) {
            const int row = 16 * ks + trow;
            v8 av[TM], bv[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const s4 lo = tr(Ab + row * PITCH + tcolA + 32 * i);
                const s4 hi = tr(Ab + (row + 4) * PITCH + tcolA + 32 * i);
                const s8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                av[i] = __builtin_bit_cast(v8, v);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const s4 lo = tr(Bb + row * PITCH + tcolB + 32 * j);
                const s4 hi = tr(Bb + (row + 4) * PITCH + tcolB + 32 * j);
                const s8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                bv[j] = __builtin_bit_cast(v8, v);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = wmfma(av[i], bv[j], acc[i][j]);
        }
    };

    f32x4 ra0[PASSES], rb0[PASSES], ra1[PASSES], rb1[PASSES];
    if (nt > 0) load_step(ra0, rb0);
    if (nt > 1) load_step(ra1, rb1);
    if (nt > 0) store_step(0, ra0, rb0);
    __syncthreads();
    for (int it = 0; it < nt; it += 2) {
        if (it + 2 < nt) load_step(ra0, rb0);
        compute(0);
        if (it + 1 < nt) store_step(1, ra1, rb1);
        __syncthreads();
        if (it + 1 >= nt) break;
        if (it + 3 < nt) load_step(ra1, rb1);
        compute(1);
        if (it + 2 < nt) store_step(0, ra0, rb0);
        __syncthreads();
    }
    // ---- partial tile -> slab [split][co][taps*Cin] (C/D map: col = lane & 31, row = (r&3) +
    //      8 (r>>2) + 4 (lane>>5)); a single split writes the gradient itself ----
    const size_t ncols = (size_t)taps * a.Cin;
    float* slab = splits == 1 ? a.dw : a.slabs + (size_t)split * a.Cout * ncols;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = tap * a.Cin + ci0 + wn * (BT / 2) + 32 * j + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wm * (BT / 2) + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
                slab[(size_t)co * ncols + col] = acc[i][j][r];
            }
        }
}

// sum of the slabs in a fixed order.  Small tensors are cut into many slabs (a 64 x 64 1x1
// convolution: 1,024 float4 outputs, up to 512 slabs): one thread per output walking all of them
// was 40 us of dependent loads on four workgroups.  32 outputs x 8 slab groups per block: a thread
// adds the slabs k = g, g + 8, ... of its output, thread group 0 the eight partial sums.
__global__ __launch_bounds__(256) void wgrad_f16_reduce_kernel(const float* __restrict__ slabs,
                                                               float* __restrict__ dw,
                                                               const int splits, const size_t n4) {
    __shared__ f32x4 part[8][32];
    const int o = threadIdx.x & 31, g = threadIdx.x >> 5;
    for (size_t base = (size_t)blockIdx.x * 32; base < n4; base += (size_t)gridDim.x * 32) {   // (block-uniform)
        const size_t i = base + o;
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        if (i < n4) {
            int sidx = g;
            for (; sidx + 8 < splits; sidx += 16) {
                acc[0] += *reinterpret_cast<const f32x4*>(slabs + ((size_t)sidx * n4 + i) * 4);
                acc[1] += *reinterpret_cast<const f32x4*>(slabs + ((size_t)(sidx + 8) * n4 + i) * 4);
            }
            if (sidx < splits)
                acc[0] += *reinterpret_cast<const f32x4*>(slabs + ((size_t)sidx * n4 + i) * 4);
        }
        part[g][o] = acc[0] + acc[1];
        __syncthreads();
        if (g == 0 && i < n4) {
            f32x4 r = part[0][o];
#pragma unroll
            for (int q = 1; q < 8; ++q) r += part[q][o];
            *reinterpret_cast<f32x4*>(dw + i * 4) = r;
        }
        __syncthreads();
    }
}

struct W16Plan { int bt, splits, steps_per; };
W16Plan wplan16(const WgradF16Args& a) {
    const int Mpix = a.N * a.Ho * a.Wo;
    const int nsteps = cdiv(Mpix, WPIX);
    // CILRS_W16_TILE=64 keeps the small tile everywhere (A/B)
    static const int force = experiment_env("CILRS_W16_TILE", 0);
    const int bt = (a.Cin % 128 == 0 && a.Cout % 128 == 0 && force != 64) ? 128 : 64;
    const int tiles = a.K * a.K * (a.Cin / bt) * (a.Cout / bt);
    // blocks aimed at per launch (CILRS_W16_TARGET).  Measured with tools/bf16_train_probe.py, whole
    // weight-gradient time per step, ResNet-50 variant B=64 176x400 / ResNet-34 B=128 (64-wide
    // tile): target 256: 5.41 / 2.54 ms, 512: 3.67 / 1.84, 768: 3.63 / 2.00, 1536: 3.78 / 2.9,
    // 3072: 4.32 / -- (the slabs are fp32: every extra split writes and re-reads a whole copy of
    // the gradient)
    static const int target = experiment_env("CILRS_W16_TARGET", 512);
    int splits = cdiv(target, tiles);
    const int max_splits = nsteps / 4 > 0 ? nsteps / 4 : 1;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    const int per = cdiv(nsteps, splits);
    W16Plan p{bt, cdiv(nsteps, per), per};
    return p;
}

}  // namespace

size_t wgrad_f16_scratch_floats(const WgradF16Args& a) {
    const W16Plan p = wplan16(a);
    return (size_t)p.splits * a.Cout * a.K * a.K * a.Cin;
}

template <typename T, int BT>
static int launch_wgrad_f16_t(const WgradF16Args& a, const W16Plan& p, int Mpix, hipStream_t s) {
    constexpr size_t lds = (size_t)4 * WPIX * wpitch<BT>() * 2;
    static bool attr_set = false;
    if (!attr_set) {
        CILRS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_f16_kernel<T, BT>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    const int tiles = a.K * a.K * (a.Cin / BT) * (a.Cout / BT);
    wgrad_f16_kernel<T, BT><<<tiles * p.splits, 256, lds, s>>>(a, Mpix, p.splits, p.steps_per);
    CILRS_LAUNCH_CHECK();
    return 0;
}

int launch_wgrad_f16(const WgradF16Args& a, hipStream_t s) {
    CILRS_CHECK(a.Cin % 64 == 0 && a.Cout % 64 == 0 && a.K * a.K <= 16,
                "wgrad_f16: Cin %% 64, Cout %% 64, <= 16 taps");
    CILRS_CHECK(a.slabs != nullptr && ((uintptr_t)a.slabs & 15) == 0 && ((uintptr_t)a.dw & 15) == 0 &&
                    ((uintptr_t)a.x & 15) == 0 && ((uintptr_t)a.dy & 15) == 0,
                "wgrad_f16: scratch missing / operands misaligned");
    const int Mpix = a.N * a.Ho * a.Wo;
    CILRS_CHECK((size_t)a.N * a.H * a.W * a.Cin * 2 < (1ull << 32) &&
                    (size_t)Mpix * a.Cout * 2 < (1ull << 32),
                "wgrad_f16: tensor larger than 4 GB");
    const W16Plan p = wplan16(a);
    int rc;
    if (p.bt == 128)
        rc = a.bf16 ? launch_wgrad_f16_t<__bf16, 128>(a, p, Mpix, s)
                    : launch_wgrad_f16_t<_Float16, 128>(a, p, Mpix, s);
    else
        rc = a.bf16 ? launch_wgrad_f16_t<__bf16, 64>(a, p, Mpix, s)
                    : launch_wgrad_f16_t<_Float16, 64>(a, p, Mpix, s);
    if (rc) return rc;
    if (p.splits > 1) {
        const size_t n4 = (size_t)a.Cout * a.K * a.K * a.Cin / 4;
        const int blocks = (int)((n4 + 31) / 32 < 4096 ? (n4 + 31) / 32 : 4096);
        wgrad_f16_reduce_kernel<<<blocks, 256, 0, s>>>(a.slabs, a.dw, p.splits, n4);
        CILRS_LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace cilrs
