// Weight gradient of the convolutions / wide linear layers on the exact-f32 matrix pipe.
//
// Counterpart of what autograd's loss.backward() (reference notebook/notebook.ipynb:552) dispatches
// for every nn.Conv2d / nn.Linear weight of CILRS.
//
// GEMM view:  dW[co][tap][ci] = sum_{p} dy[p][co] * x[pix(p,tap)][ci],   K = p = N*Ho*Wo output
// pixels (140,800 for layer1 at B=128 ... 2,688 for layer4), M = Cout, N = Cin per filter tap.
// Both operands are pixel-major in HBM (NHWC), i.e. the REDUCTION index is the slow one, which
// is exactly the MFMA operand order: lane l of v_mfma_f32_32x32x2_f32 feeds A[i=l&31][k=l>>5],
// so a wave reads 32 consecutive channels of pixel k (lanes 0-31) and of pixel k+1 (lanes 32-63)
// from an LDS tile that is a plain copy of the NHWC rows -- conflict-free ds_read_b32, no
// transposes anywhere.  The result lands in OHWI order, the parameters' own memory order.
//
// K is split across blocks (grid.z); partial slabs are summed in a fixed order by a second kernel
// (deterministic, no float atomics).
#include "common.h"

#include <stdlib.h>

namespace cilrs {

namespace {

constexpr int BKP = 32;   // pixels per K-tile

template <int BT, bool TAP_UNIFORM, bool PIN>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const WgradArgs a, const int Mpix,
                                                            const int splits, const int tiles_ci,
                                                            const int Ncols, const int ntiles_x,
                                                            const int ntiles_co,
                                                            const int xcd_group) {
    // block tile BT(co) x BT(ci-columns); 4 waves as 2x2, wave tile (BT/2)^2
    constexpr int WT = BT / 2, T = WT / 32;
    constexpr int QPR = BT / 4;            // float4 quads per tile row
    constexpr int RPP = 256 / QPR;         // rows per pass
    constexpr int PASSES = BKP / RPP;
    constexpr int PITCH = BT + 4;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                      // [2][BKP][PITCH]  dy tile
    float* Bs = smem + 2 * BKP * PITCH;    // [2][BKP][PITCH]  gathered x tile

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    wave_priority(a.prio_mode < 16 ? a.prio_mode : 0);

    // 1-D grid over (column tile, co tile, K-slab), column tile fastest.  The hardware deals
    // consecutive block ids round-robin over the 8 XCDs (each with its own L2), so in plain order
    // the column tiles of ONE K-slab -- which all read the same dy rows and, tap-shifted, the
    // same x rows -- land on all eight L2s and every slab crosses the fabric up to eight times.
    // xcd_group != 0 (default) hands each XCD a contiguous run of logical ids instead, so a slab's
    // tiles share one L2.  Measured on MI355X at B=128 (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE per
    // launch): 263 MB -> 69 MB.  Alone on the chip the kernels then run 7-15 % slower (the
    // re-reads were Infinity-Cache hits at ~2.3 TB/s, far from a bound, and nine blocks marching
    // through the same lines of one L2 in lock-step cost more than they save); in the training
    // step, where they share the chip with the data-gradient chain, the step is 0.7 % FASTER
    // (11.79 vs 11.88 ms, interleaved A/B on one box): the fabric traffic they no longer generate
    // is the other stream's.  CILRS_WGRAD_XCD=0 restores plain order.
    int logical = (int)blockIdx.x;
    if (xcd_group) {
        const int nwg = (int)gridDim.x, bid = (int)blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        logical = base + (bid >> 3);
    }
    const int ntile = logical % ntiles_x;  // column tile: (tap, ci-tile) or flattened columns
    const int co0 = ((logical / ntiles_x) % ntiles_co) * BT;
    const int z = logical / (ntiles_x * ntiles_co);

    // K (pixel) range of this split, in whole K-tiles
    const int KT = (Mpix + BKP - 1) / BKP;
    const int per = (KT + splits - 1) / splits;
    const int kt_begin = z * per;
    const int kt_end = min(KT, kt_begin + per);
    const int nt = max(kt_end - kt_begin, 0);

    const int q = tid % QPR, rr = tid / QPR;

    // column (tap, ci) owned by this thread's quad -- fixed across the K loop
    int tap, ci, col0;
    bool col_ok;
    if constexpr (TAP_UNIFORM) {
        tap = ntile / tiles_ci;
        ci = (ntile - tap * tiles_ci) * BT + q * 4;
        col0 = tap * a.Cin + (ntile - tap * tiles_ci) * BT;
        col_ok = true;
    } else {
        col0 = ntile * BT;
        const int j = col0 + q * 4;
        tap = j / a.Cin;
        ci = j - tap * a.Cin;
        col_ok = j < Ncols;
    }
    const int kh = tap / a.KW, kw = tap - kh * a.KW;
    const int HoWo = a.Ho * a.Wo;

    // pixel cursor of each pass row for the NEXT tile to load: (n, oh, ow), advanced by BKP
    // pixels per tile without divisions: BKP = dn*Ho*Wo + doh*Wo + dow
    const int dn = BKP / HoWo, drem = BKP - dn * HoWo;
    const int doh = drem / a.Wo, dow = drem - doh * a.Wo;
    int cn[PASSES], coh[PASSES], cow[PASSES], cp[PASSES];
#pragma unroll
    for (int i = 0; i < PASSES; ++i) {
        const int p = kt_begin * BKP + rr + RPP * i;
        cp[i] = p;
        cn[i] = p / HoWo;
        const int rem = p - cn[i] * HoWo;
        coh[i] = rem / a.Wo;
        cow[i] = rem - coh[i] * a.Wo;
    }
    // buffer loads: wave-uniform base + 32-bit byte offset; a pixel past the end of dy is out of
    // range by construction and a padded / out-of-image x pixel gets offset ~0, so the hardware
    // returns zeros and the loads carry no branches
    const __amdgpu_buffer_rsrc_t rsDy = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.dy, 0, (int)(unsigned)((size_t)Mpix * a.dy_ld * sizeof(float)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.x, 0, (int)(unsigned)((size_t)a.N * a.H * a.W * a.x_ld * sizeof(float)),
        0x00020000);
    const unsigned dyq = (unsigned)(co0 + q * 4) * 4u;
    const unsigned dy_pitch = (unsigned)a.dy_ld * 4u;

    auto load_tile = [&](f32x4(&ra)[PASSES], f32x4(&rb)[PASSES]) {
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            const int h = coh[i] * a.stride - a.pad + kh, w = cow[i] * a.stride - a.pad + kw;
            const bool ok = col_ok && cp[i] < Mpix && h >= 0 && w >= 0 && h < a.H && w < a.W;
            const unsigned xo =
                ok ? (unsigned)(((cn[i] * a.H + h) * a.W + w) * a.x_ld + ci) * 4u : 0xFFFFFFFFu;
            ra[i] = __builtin_bit_cast(
                f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                           rsDy, (int)((unsigned)cp[i] * dy_pitch + dyq), 0, 0));
            rb[i] = __builtin_bit_cast(
                f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, (int)xo, 0, 0));
            // advance the cursor by BKP pixels
            cp[i] += BKP;
            cow[i] += dow;
            int c1 = cow[i] >= a.Wo;
            cow[i] -= c1 ? a.Wo : 0;
            coh[i] += doh + c1;
            int c2 = coh[i] >= a.Ho;
            coh[i] -= c2 ? a.Ho : 0;
            cn[i] += dn + c2;
        }
    };
    auto store_tile = [&](int buf, const f32x4(&ra)[PASSES], const f32x4(&rb)[PASSES]) {
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            const int row = rr + RPP * i;
            *reinterpret_cast<f32x4*>(As + (buf * BKP + row) * PITCH + q * 4) = ra[i];
            *reinterpret_cast<f32x4*>(Bs + (buf * BKP + row) * PITCH + q * 4) = rb[i];
        }
    };

    f32x16 acc[T][T];
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < T; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // T == 2: the two 32-wide MFMA tiles of a wave take the EVEN / ODD channels of its 64, so one
    // ds_read_b64 per operand feeds four MFMAs (tile t, row i <-> channel 2i + t)
    auto compute = [&](int buf) {
        if constexpr (T == 2) {
            const float* Ab = As + buf * BKP * PITCH + lh * PITCH + wm * WT + 2 * l31;
            const float* Bb = Bs + buf * BKP * PITCH + lh * PITCH + wn * WT + 2 * l31;
#pragma unroll
            for (int s = 0; s < BKP / 2; ++s) {
                const float2 av = *reinterpret_cast<const float2*>(Ab + 2 * s * PITCH);
                const float2 bv = *reinterpret_cast<const float2*>(Bb + 2 * s * PITCH);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.y, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.x, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[1][1], 0, 0, 0);
            }
            // pin the LDS reads two steps ahead of the MFMAs that use them
            if constexpr (PIN) {
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
                for (int s = 0; s < BKP / 2; ++s) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                    if (s + 2 < BKP / 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
            }
        } else {
            const float* Ab = As + buf * BKP * PITCH + lh * PITCH + wm * WT + l31;
            const float* Bb = Bs + buf * BKP * PITCH + lh * PITCH + wn * WT + l31;
#pragma unroll
            for (int s = 0; s < BKP / 2; ++s) {
                const float af = Ab[2 * s * PITCH], bf = Bb[2 * s * PITCH];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc[0][0], 0, 0, 0);
            }
            if constexpr (PIN) {
                __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
                for (int s = 0; s < BKP / 2; ++s) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (s + 4 < BKP / 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
            }
        }
    };

    // loads run two K-tiles ahead of the MFMAs (see conv_igemm.hip)
    f32x4 ra0[PASSES], rb0[PASSES], ra1[PASSES], rb1[PASSES];
    int it = 0;
    if (nt >= 4) {
        // branch-free steady state (see conv_igemm.hip): exact wait counts, loads issued before
        // the multiplies they hide behind
        load_tile(ra0, rb0);
        load_tile(ra1, rb1);
        store_tile(0, ra0, rb0);
        __syncthreads();
        for (; it + 3 < nt; it += 2) {
            load_tile(ra0, rb0);
            __builtin_amdgcn_sched_barrier(0);
            compute(0);
            store_tile(1, ra1, rb1);
            __syncthreads();
            load_tile(ra1, rb1);
            __builtin_amdgcn_sched_barrier(0);
            compute(1);
            store_tile(0, ra0, rb0);
            __syncthreads();
        }
    } else {
        if (nt > 0) load_tile(ra0, rb0);
        if (nt > 1) load_tile(ra1, rb1);
        if (nt > 0) store_tile(0, ra0, rb0);
        __syncthreads();
    }
    for (; it < nt; it += 2) {
        if (it + 2 < nt) load_tile(ra0, rb0);
        compute(0);
        if (it + 1 < nt) store_tile(1, ra1, rb1);
        __syncthreads();
        if (it + 1 >= nt) break;
        if (it + 3 < nt) load_tile(ra1, rb1);
        compute(1);
        if (it + 2 < nt) store_tile(0, ra0, rb0);
        __syncthreads();
    }

    // slab[z][co][col]  (col = tap*Cin + ci, OHWI order)
    float* out = a.slabs + (size_t)z * a.Cout * Ncols;
    if constexpr (T == 2) {
        const int col = col0 + wn * WT + 2 * l31;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int co = co0 + wm * WT + 2 * row + i;
                float2 v;
                v.x = acc[i][0][r];
                v.y = acc[i][1][r];
                *reinterpret_cast<float2*>(out + (size_t)co * Ncols + col) = v;
            }
    } else {
        const int col = col0 + wn * WT + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int co = co0 + wm * WT + row;
            if (col < Ncols) out[(size_t)co * Ncols + col] = acc[0][0][r];
        }
    }
}

// Sum the K-split slabs (fixed order => deterministic).  VEC = 4: float4 per thread, four
// independent slab streams in flight, and for small outputs with many slabs (e.g. a 1x1
// downsample: 8 K floats x 220 slabs) G thread groups share one output and combine through LDS so
// the chip stays full.  VEC = 1 handles the channel-padded stem.
template <int VEC>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slabs,
                                                           float* __restrict__ dw,
                                                           const int splits, const size_t n_src,
                                                           const int cin, const int cin_dst,
                                                           const int accumulate, const int G) {
    if constexpr (VEC == 4) {
        __shared__ f32x4 red[256];
        const int opb = 256 / G;                       // outputs per block
        const int g = threadIdx.x / opb, ol = threadIdx.x - g * opb;
        const size_t nvec = n_src / 4;
        const size_t iv = (size_t)blockIdx.x * opb + ol;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (iv < nvec) {
            const float* base = slabs + iv * 4;
            f32x4 acc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            int sidx = g;
            for (; sidx + 3 * G < splits; sidx += 4 * G) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    acc[u] += *reinterpret_cast<const f32x4*>(base + (size_t)(sidx + u * G) * n_src);
            }
            for (; sidx < splits; sidx += G)
                acc[0] += *reinterpret_cast<const f32x4*>(base + (size_t)sidx * n_src);
            v = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        }
        if (G > 1) {
            red[threadIdx.x] = v;
            __syncthreads();
            if (g == 0) {
                for (int k = 1; k < G; ++k) v += red[k * opb + ol];
            }
        }
        if (g == 0 && iv < nvec) {
            f32x4* o = reinterpret_cast<f32x4*>(dw + iv * 4);      // cin_dst == cin here
            *o = accumulate ? *o + v : v;
        }
    } else {
        // scalar outputs (the channel-padded stem: 12.5 K outputs x up to 512 slabs): the same
        // G-groups-per-output shape, so ~800 blocks share the slab reads instead of 49
        __shared__ float red1[256];
        const int opb = 256 / G;
        const int g = threadIdx.x / opb, ol = threadIdx.x - g * opb;
        const size_t idx = (size_t)blockIdx.x * opb + ol;
        float v = 0.f;
        if (idx < n_src) {
            const float* base = slabs + idx;
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            int sidx = g;
            for (; sidx + 3 * G < splits; sidx += 4 * G) {
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[u] += base[(size_t)(sidx + u * G) * n_src];
            }
            for (; sidx < splits; sidx += G) acc[0] += base[(size_t)sidx * n_src];
            v = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        }
        if (G > 1) {
            red1[threadIdx.x] = v;
            __syncthreads();
            if (g == 0) {
                for (int k = 1; k < G; ++k) v += red1[k * opb + ol];
            }
        }
        if (g == 0 && idx < n_src) {
            const int ci = (int)(idx % cin);
            if (ci < cin_dst) {
                const size_t o = (idx / cin) * cin_dst + ci;
                dw[o] = accumulate ? dw[o] + v : v;
            }
        }
    }
}

struct WPlan { int bt; bool uniform; int tiles_ci; int ntiles; int ncols; int splits; };

// Tile and split choice.  The output (Cout x taps*Cin) is tiny next to K = N*Ho*Wo, so K must be
// split across blocks; every split costs a slab written and read back.  Model (microseconds):
// the busiest CU runs ceil(blocks/256) blocks in co-resident groups at the rates measured by
// tools/mfma_probe.hip (derated), plus the slab traffic and the reduce launch.
double wplan_cost(double flops, int tiles, int splits, int bt, size_t out_floats) {
    static const float r128[2] = {105.f, 120.f};                 // 2 blocks/CU fit (67 KB LDS)
    static const float r64[4] = {60.f, 85.f, 96.f, 103.f};       // 4 blocks/CU fit (35 KB LDS)
    const int occ = bt == 128 ? 2 : 4;
    const float* rate = bt == 128 ? r128 : r64;
    const long blocks = (long)tiles * splits;
    const long per_cu = (blocks + 255) / 256;
    const double unit = flops / (double)blocks;                  // flops per block
    const long full = per_cu / occ, rem = per_cu % occ;
    double t = 0.0;
    if (full) t += full * unit * 256.0 * occ / (rate[occ - 1] * 1e6);
    if (rem) t += unit * 256.0 * rem / (rate[rem - 1] * 1e6);
    t += (full + (rem ? 1 : 0)) * 3.0;
    t += 4.0 + (double)out_floats * 4.0 * (2.0 * splits + 1.0) / 3.5e6;   // slabs out + back
    return t;
}

WPlan plan(const WgradArgs& a) {
    WPlan best;
    const int taps = a.KH * a.KW;
    const int Mpix = a.N * a.Ho * a.Wo;
    const int KT = cdiv(Mpix, BKP);
    const bool uniform = (a.Cin % 64) == 0;
    const size_t out_floats = (size_t)a.Cout * taps * a.Cin;
    const double flops = 2.0 * Mpix * (double)out_floats;
    static const int env_bt = experiment_env("CILRS_WGRAD_BT", 0);
    static const int env_target = experiment_env("CILRS_WGRAD_TARGET", 0);
    double bc = 1e30;
    for (int bt = 64; bt <= 128; bt += 64) {
        if (bt == 128 && !(uniform && a.Cin % 128 == 0 && a.Cout % 128 == 0)) continue;
        if (env_bt && bt != env_bt) continue;
        WPlan p;
        p.bt = bt; p.uniform = uniform; p.ncols = taps * a.Cin;
        if (uniform) { p.tiles_ci = a.Cin / bt; p.ntiles = taps * p.tiles_ci; }
        else { p.tiles_ci = 1; p.ntiles = cdiv(p.ncols, bt); }
        const int tiles = p.ntiles * (a.Cout / bt);
        const int max_splits = KT / 4 > 0 ? KT / 4 : 1;
        // candidate block budgets: whole numbers of blocks per CU
        for (int budget = 256; budget <= 2048; budget += 256) {
            if (env_target && budget != ((env_target + 255) / 256) * 256) continue;
            int splits = budget / tiles;
            if (splits < 1) splits = 1;
            if (splits > max_splits) splits = max_splits;
            const int per = cdiv(KT, splits);
            splits = cdiv(KT, per);                  // drop empty trailing splits
            const double c = wplan_cost(flops, tiles, splits, bt, out_floats);
            if (c < bc) { bc = c; best = p; best.splits = splits; }
        }
    }
    return best;
}

}  // namespace

size_t wgrad_scratch_floats(const WgradArgs& a) {
    const WPlan p = plan(a);
    return (size_t)p.splits * a.Cout * p.ncols;
}

int launch_conv_wgrad(const WgradArgs& a_in, hipStream_t s) {
    WgradArgs a = a_in;
    a.prio_mode = wave_priority_mode();
    CILRS_CHECK(a.Cout % 64 == 0, "conv_wgrad: Cout=%d must be a multiple of 64", a.Cout);
    CILRS_CHECK(a.Cin % 4 == 0 && a.x_ld % 4 == 0 && a.dy_ld % 4 == 0,
                "conv_wgrad: Cin/x_ld/dy_ld must be multiples of 4");
    CILRS_CHECK(a.slabs != nullptr, "conv_wgrad: scratch slabs missing");
    // both operands are addressed with 32-bit byte offsets (buffer loads)
    CILRS_CHECK((size_t)a.N * a.H * a.W * a.x_ld * sizeof(float) < (1ull << 32) &&
                    (size_t)a.N * a.Ho * a.Wo * a.dy_ld * sizeof(float) < (1ull << 32),
                "conv_wgrad: tensor larger than 4 GB");
    CILRS_CHECK(((uintptr_t)a.x & 15) == 0 && ((uintptr_t)a.dy & 15) == 0 &&
                    ((uintptr_t)a.slabs & 15) == 0,
                "conv_wgrad: pointers must be 16-byte aligned");
    const WPlan p = plan(a);
    const int Mpix = a.N * a.Ho * a.Wo;
    const int ntiles_co = a.Cout / p.bt;
    static const int xcd = experiment_env("CILRS_WGRAD_XCD", 1);
    dim3 grid(p.ntiles * ntiles_co * p.splits, 1, 1);
    const size_t lds = (size_t)4 * BKP * (p.bt + 4) * sizeof(float);
    // CILRS_WGRAD_PIN=0: let the compiler place the LDS reads (A/B switch for tools/conv_bench.py)
    static const bool pin = experiment_env("CILRS_WGRAD_PIN", 1) != 0;
    if (p.bt == 128) {
        static bool attr = false;
        if (!attr) {
            CILRS_HIP(hipFuncSetAttribute(
                reinterpret_cast<const void*>(&conv_wgrad_kernel<128, true, true>),
                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            CILRS_HIP(hipFuncSetAttribute(
                reinterpret_cast<const void*>(&conv_wgrad_kernel<128, true, false>),
                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr = true;
        }
        if (pin)
            conv_wgrad_kernel<128, true, true><<<grid, 256, lds, s>>>(a, Mpix, p.splits,
                                                                      p.tiles_ci, p.ncols, p.ntiles, ntiles_co, xcd);
        else
            conv_wgrad_kernel<128, true, false><<<grid, 256, lds, s>>>(a, Mpix, p.splits,
                                                                       p.tiles_ci, p.ncols, p.ntiles, ntiles_co, xcd);
    } else if (p.uniform) {
        if (pin)
            conv_wgrad_kernel<64, true, true><<<grid, 256, lds, s>>>(a, Mpix, p.splits,
                                                                     p.tiles_ci, p.ncols, p.ntiles, ntiles_co, xcd);
        else
            conv_wgrad_kernel<64, true, false><<<grid, 256, lds, s>>>(a, Mpix, p.splits,
                                                                      p.tiles_ci, p.ncols, p.ntiles, ntiles_co, xcd);
    } else {
        conv_wgrad_kernel<64, false, false><<<grid, 256, lds, s>>>(a, Mpix, p.splits, p.tiles_ci,
                                                                   p.ncols, p.ntiles, ntiles_co, xcd);
    }
    CILRS_LAUNCH_CHECK();
    const size_t n_src = (size_t)a.Cout * p.ncols;
    if (a.Cin == a.Cin_dst && n_src % 4 == 0 && ((uintptr_t)a.dw & 15) == 0) {
        const size_t nvec = n_src / 4;
        int G = 1;                                    // thread groups per output (power of two)
        while (G < 16 && nvec * G < 131072 && 8 * G <= p.splits) G *= 2;
        const int opb = 256 / G;
        wgrad_reduce_kernel<4><<<(int)((nvec + opb - 1) / opb), 256, 0, s>>>(
            a.slabs, a.dw, p.splits, n_src, a.Cin, a.Cin_dst, a.accumulate, G);
    } else {
        int G = 1;
        while (G < 16 && n_src * G < 262144 && 8 * G <= p.splits) G *= 2;
        const int opb = 256 / G;
        wgrad_reduce_kernel<1><<<(int)((n_src + opb - 1) / opb), 256, 0, s>>>(
            a.slabs, a.dw, p.splits, n_src, a.Cin, a.Cin_dst, a.accumulate, G);
    }
    CILRS_LAUNCH_CHECK();
    return 0;
}

// slabs[splits][n] -> dw[n] in slab order (n % 4 == 0, 16-byte aligned): the reduce of
// launch_conv_wgrad on its own, for the Winograd weight gradient (conv_wino.hip)
int launch_wgrad_reduce(const float* slabs, float* dw, int splits, size_t n, int accumulate,
                        hipStream_t s) {
    CILRS_CHECK(n % 4 == 0 && ((uintptr_t)dw & 15) == 0 && ((uintptr_t)slabs & 15) == 0,
                "wgrad_reduce: n %% 4, 16-byte aligned buffers");
    const size_t nvec = n / 4;
    int G = 1;
    while (G < 16 && nvec * G < 131072 && 8 * G <= splits) G *= 2;
    const int opb = 256 / G;
    wgrad_reduce_kernel<4><<<(int)((nvec + opb - 1) / opb), 256, 0, s>>>(slabs, dw, splits, n, 0, 0,
                                                                        accumulate, G);
    CILRS_LAUNCH_CHECK();
    return 0;
}

}  // namespace cilrs
