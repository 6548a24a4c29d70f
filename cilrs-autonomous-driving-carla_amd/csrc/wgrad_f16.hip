// Weight gradient of a convolution on the 16-bit matrix pipe (the "bf16 MFMA path" of BASELINE.json
// configs[3], training side): dW[co][kh][kw][ci] (fp32, OHWI) = sum over output pixels p of
// dy[p][co] * x[pix(p, kh, kw)][ci], both operands 16-bit NHWC (bf16 or fp16), fp32 accumulation.
//
// The reduction index (output pixels) is the SLOW index of both operands, so neither is a
// k-contiguous MFMA operand: tiles are staged as plain [pixel][channel] row copies (16-byte global
// loads, a gathered input pixel outside the image reads zeros through the buffer range check) and
// read back TRANSPOSED with ds_read_b64_tr_b16 (cdna_hip_programming.md T10): a 16-lane group reads
// a 4-pixel x 16-channel block and every lane receives its channel's 4 pixels -- two reads make
// the 8 k-values per lane of v_mfma_f32_32x32x16_*.  Row pitch 192 bytes: the four rows of a block
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
constexpr int WPITCH = 96;          // halfs per LDS row (192 bytes)

template <typename T> struct WVec8;
template <> struct WVec8<_Float16> { typedef h8 type; };
template <> struct WVec8<__bf16> { typedef b8 type; };
__device__ __forceinline__ f32x16 wmfma(const h8 a, const h8 b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 wmfma(const b8 a, const b8 b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

template <typename T>
__global__ __launch_bounds__(256, 2) void wgrad_f16_kernel(const WgradF16Args a, const int Mpix,
                                                           const int splits, const int steps_per) {
    typedef typename WVec8<T>::type v8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* As = reinterpret_cast<T*>(smem_raw);                 // [2][WPIX][WPITCH]  dy tile
    T* Bs = As + 2 * WPIX * WPITCH;                         // [2][WPIX][WPITCH]  gathered x tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;                // wave tile: co 32 wm.., col 32 wn..

    const int tiles_ci = a.Cin / 64, taps = a.K * a.K;
    const int ntiles_col = taps * tiles_ci, ntiles_co = a.Cout / 64;
    int b = blockIdx.x;
    const int tcol = b % ntiles_col; b /= ntiles_col;
    const int tco = b % ntiles_co;
    const int split = b / ntiles_co;
    const int tap = tcol / tiles_ci, ci0 = (tcol - tap * tiles_ci) * 64, co0 = tco * 64;
    const int kh = tap / a.K, kw = tap - kh * a.K;
    const int nsteps_total = (Mpix + WPIX - 1) / WPIX;
    const int st_begin = split * steps_per;
    const int st_end = min(nsteps_total, st_begin + steps_per);
    const int nt = max(0, st_end - st_begin);

    // loader: thread -> pixel rows r0, r0 + 32 of the step and 16-byte chunk kq of the 128-byte row
    const int kq = tid & 7, r0 = tid >> 3;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.dy, 0, (int)(unsigned)((size_t)Mpix * a.Cout * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.x, 0, (int)(unsigned)((size_t)a.N * a.H * a.W * a.Cin * 2), 0x00020000);
    // pixel cursors of the two rows (advanced by WPIX per step without divisions)
    int pn[2], poh[2], pow_[2];
    const int HoWo = a.Ho * a.Wo;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int p = st_begin * WPIX + r0 + 32 * i;
        pn[i] = p / HoWo;
        const int rem = p - pn[i] * HoWo;
        poh[i] = rem / a.Wo;
        pow_[i] = rem - poh[i] * a.Wo;
    }
    int ld_p = st_begin * WPIX;        // first pixel of the next step to load
    auto load_step = [&](f32x4(&ra)[2], f32x4(&rb)[2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int p = ld_p + r0 + 32 * i;
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
    auto store_step = [&](int buf, const f32x4(&ra)[2], const f32x4(&rb)[2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<f32x4*>(&As[(buf * WPIX + r0 + 32 * i) * WPITCH + kq * 8]) = ra[i];
            *reinterpret_cast<f32x4*>(&Bs[(buf * WPIX + r0 + 32 * i) * WPITCH + kq * 8]) = rb[i];
        }
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // transposed-read addresses: lane (group g, q, p) supplies row q, columns 4p..4p+3 of the block
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int trow = 8 * (g >> 1) + q;                       // + 16 * kstep (+ 4 for the high half)
    const int tcolA = wm * 32 + 16 * (g & 1) + 4 * pp;       // co within the tile
    const int tcolB = wn * 32 + 16 * (g & 1) + 4 * pp;       // ci within the tile
    auto tr = [&](const T* p) -> s4 {
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s4*)(p));
    };
    auto compute = [&](int buf) {
        const T* Ab = As + buf * WPIX * WPITCH;
        const T* Bb = Bs + buf * WPIX * WPITCH;
#pragma unroll
        for (int ks = 0; ks < WPIX / 16; ++ks) {
            const int row = 16 * ks + trow;
            const s4 alo = tr(Ab + row * WPITCH + tcolA), ahi = tr(Ab + (row + 4) * WPITCH + tcolA);
            const s4 blo = tr(Bb + row * WPITCH + tcolB), bhi = tr(Bb + (row + 4) * WPITCH + tcolB);
            const s8 av = {alo[0], alo[1], alo[2], alo[3], ahi[0], ahi[1], ahi[2], ahi[3]};
            const s8 bv = {blo[0], blo[1], blo[2], blo[3], bhi[0], bhi[1], bhi[2], bhi[3]};
            acc = wmfma(__builtin_bit_cast(v8, av), __builtin_bit_cast(v8, bv), acc);
        }
    };

    f32x4 ra0[2], rb0[2], ra1[2], rb1[2];
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
    //      8 (r>>2) + 4 (lane>>5)) ----
    const size_t ncols = (size_t)taps * a.Cin;
    float* slab = a.slabs + (size_t)split * a.Cout * ncols;
    const int col = tap * a.Cin + ci0 + wn * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        slab[(size_t)co * ncols + col] = acc[r];
    }
}

// sum of the slabs in slab order (float4 per thread)
__global__ __launch_bounds__(256) void wgrad_f16_reduce_kernel(const float* __restrict__ slabs,
                                                               float* __restrict__ dw,
                                                               const int splits, const size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (size_t)gridDim.x * blockDim.x) {
        f32x4 acc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        int sidx = 0;
        for (; sidx + 3 < splits; sidx += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                acc[u] += *reinterpret_cast<const f32x4*>(slabs + ((size_t)(sidx + u) * n4 + i) * 4);
        }
        for (; sidx < splits; ++sidx)
            acc[0] += *reinterpret_cast<const f32x4*>(slabs + ((size_t)sidx * n4 + i) * 4);
        *reinterpret_cast<f32x4*>(dw + i * 4) = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    }
}

struct W16Plan { int splits, steps_per; };
W16Plan wplan16(const WgradF16Args& a) {
    const int Mpix = a.N * a.Ho * a.Wo;
    const int nsteps = cdiv(Mpix, WPIX);
    const int tiles = a.K * a.K * (a.Cin / 64) * (a.Cout / 64);
    // blocks aimed at per launch (CILRS_W16_TARGET).  Measured with tools/bf16_train_probe.py, whole
    // weight-gradient time per step, ResNet-50 variant B=64 176x400 / ResNet-34 B=128: target 256:
    // 5.41 / 2.54 ms, 512: 3.67 / 1.84, 768: 3.63 / 2.00, 1536: 3.78 / 2.9, 3072: 4.32 / -- (the
    // slabs are fp32: every extra split writes and re-reads a whole copy of the gradient)
    static const int target = getenv("CILRS_W16_TARGET") ? atoi(getenv("CILRS_W16_TARGET")) : 512;
    int splits = cdiv(target, tiles);
    const int max_splits = nsteps / 4 > 0 ? nsteps / 4 : 1;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    const int per = cdiv(nsteps, splits);
    W16Plan p{cdiv(nsteps, per), per};
    return p;
}

}  // namespace

size_t wgrad_f16_scratch_floats(const WgradF16Args& a) {
    const W16Plan p = wplan16(a);
    return (size_t)p.splits * a.Cout * a.K * a.K * a.Cin;
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
    const int tiles = a.K * a.K * (a.Cin / 64) * (a.Cout / 64);
    constexpr size_t lds = (size_t)4 * WPIX * WPITCH * 2;
    if (a.bf16)
        wgrad_f16_kernel<__bf16><<<tiles * p.splits, 256, lds, s>>>(a, Mpix, p.splits, p.steps_per);
    else
        wgrad_f16_kernel<_Float16><<<tiles * p.splits, 256, lds, s>>>(a, Mpix, p.splits, p.steps_per);
    CILRS_LAUNCH_CHECK();
    const size_t n4 = (size_t)a.Cout * a.K * a.K * a.Cin / 4;
    const int blocks = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
    wgrad_f16_reduce_kernel<<<blocks, 256, 0, s>>>(a.slabs, a.dw, p.splits, n4);
    CILRS_LAUNCH_CHECK();
    return 0;
}

}  // namespace cilrs
