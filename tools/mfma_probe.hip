// Where does the fp32-MFMA pipe go idle?  Synthetic ladder (not product code): a 64x64-tile-like
// loop body per wave (16 dependent v_mfma_f32_32x32x2_f32 per "K-tile"), with ingredients of the
// conv kernel's main loop switched on one by one.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool LDSR, bool BAR, bool LDSW, bool GLD>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ g, float* out, int tiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int i = tid; i < 2 * 128 * 36; i += 256) smem[i] = (float)(i & 7) * 0.01f;
    __syncthreads();
    f32x4 a = {1.f, 2.f, 3.f, 4.f}, b = {0.5f, 0.25f, 0.125f, 1.f};
    f32x4 r0 = {0, 0, 0, 0}, r1 = r0, r2 = r0, r3 = r0;
    const float* gp = g + ((size_t)blockIdx.x * 256 + tid) * 4;
    for (int t = 0; t < tiles; ++t) {
        const int buf = t & 1;
        if (GLD) {
            const float* p = gp + (size_t)(t & 63) * 1048576;
            r0 = *(const f32x4*)(p); r1 = *(const f32x4*)(p + 262144);
            r2 = *(const f32x4*)(p + 524288); r3 = *(const f32x4*)(p + 786432);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 af[NACC], bf;
            if (LDSR) {
#pragma unroll
                for (int i = 0; i < NACC; ++i)
                    af[i] = *(const f32x4*)(smem + buf * 64 * 36 + ((wave & 1) * 32 + (lane & 31) + (i & 1)) * 36 + q * 8 + (lane >> 5) * 4);
                bf = *(const f32x4*)(smem + 128 * 36 + buf * 64 * 36 + ((wave >> 1) * 32 + (lane & 31)) * 36 + q * 8 + (lane >> 5) * 4);
            } else {
#pragma unroll
                for (int i = 0; i < NACC; ++i) af[i] = a;
                bf = b;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < NACC; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[e], acc[i], 0, 0, 0);
        }
        if (LDSW) {
            float* w = smem + (buf ^ 1) * 64 * 36 + (tid >> 3) * 36 + (tid & 7) * 4;
            *(f32x4*)(w) = GLD ? r0 : a; *(f32x4*)(w + 32 * 36) = GLD ? r1 : b;
            *(f32x4*)(w + 128 * 36) = GLD ? r2 : a; *(f32x4*)(w + 128 * 36 + 32 * 36) = GLD ? r3 : b;
        } else if (GLD) {
            asm volatile("" ::"v"(r0), "v"(r1), "v"(r2), "v"(r3));
        }
        if (BAR) __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + tid] = s;
}

// LDS-DMA variant: the next tile goes global -> LDS directly (no VGPR round trip, no ds_write);
// STAGES LDS buffers, loads issued STAGES-1 tiles ahead; QS = 8-k groups per tile (4 = BK 32)
template <int NACC, int STAGES, int QS>
__global__ __launch_bounds__(256) void probe_glds(const float* __restrict__ g, float* out, int tiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int TILE_F = 128 * 8 * QS;     // floats per stage: 128 rows x QS*8 k
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int i = tid; i < STAGES * TILE_F; i += 256) smem[i] = (float)(i & 7) * 0.01f;
    __syncthreads();
    const float* gp = g + ((size_t)blockIdx.x * 256 + tid) * 4;
    auto issue = [&](int t) {
        float* dst = smem + (t % STAGES) * TILE_F;
        const float* p = gp + (size_t)(t & 15) * 4194304;    // + j*256Ki + block/thread < 64 Mi floats
#pragma unroll
        for (int j = 0; j < QS; ++j)    // 4*QS/4... each instr: 256 thr x 16 B = 4 KB; tile = QS*4 KB
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + j * 262144),
                                             (__attribute__((address_space(3))) void*)(dst + j * 1024 + wave * 256), 16, 0, 0);
    };
    for (int t = 0; t < STAGES - 1; ++t) issue(t);
    for (int t = 0; t < tiles; ++t) {
        // wait for tile t (leave STAGES-2 tiles in flight), then make it visible to all waves
        if (STAGES == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (STAGES == 3) { if (QS == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
        else { if (QS == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); }
        __builtin_amdgcn_s_barrier();
        issue(t + STAGES - 1);
        const float* buf = smem + (t % STAGES) * TILE_F;
#pragma unroll
        for (int q = 0; q < QS; ++q) {
            f32x4 af[NACC], bf;
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                af[i] = *(const f32x4*)(buf + (((wave & 1) * 32 + (lane & 31) + (i & 1)) * 8 * QS + ((q * 2 + (lane >> 5)) ^ ((lane >> 1) & 7)) * 4) % TILE_F);
            bf = *(const f32x4*)(buf + ((64 + (wave >> 1) * 32 + (lane & 31)) * 8 * QS + ((q * 2 + (lane >> 5)) ^ ((lane >> 1) & 7)) * 4) % TILE_F);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < NACC; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[e], acc[i], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int NACC, int STAGES, int QS>
void run_glds(const char* name, const float* g, float* out, int blocks_per_cu) {
    const int tiles = 400 * 4 / QS, blocks = 256 * blocks_per_cu;
    const size_t lds = (size_t)STAGES * 128 * 8 * QS * 4;
    hipFuncSetAttribute((const void*)&probe_glds<NACC, STAGES, QS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe_glds<NACC, STAGES, QS><<<blocks, 256, lds>>>(g, out, tiles);
    hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) probe_glds<NACC, STAGES, QS><<<blocks, 256, lds>>>(g, out, tiles);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    const double flops = (double)blocks * 4 * tiles * 4 * QS * NACC * 4096.0;
    printf("glds %-22s st=%d qs=%d lds=%3zuKB blocks/CU=%d acc=%d  %8.1f us  %6.1f TF  (%s)\n", name, STAGES, QS, lds / 1024, blocks_per_cu, NACC, ms * 1e3, flops / ms / 1e9, hipGetErrorString(hipGetLastError()));
}

template <int NACC, bool LDSR, bool BAR, bool LDSW, bool GLD>
void run(const char* name, const float* g, float* out, int blocks_per_cu) {
    const int tiles = 400, blocks = 256 * blocks_per_cu;
    const size_t lds = 2 * 128 * 36 * 4;   // 36.9 KB -> <= 4 blocks / CU
    hipFuncSetAttribute((const void*)&probe<NACC, LDSR, BAR, LDSW, GLD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<NACC, LDSR, BAR, LDSW, GLD><<<blocks, 256, lds>>>(g, out, tiles);
    hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) probe<NACC, LDSR, BAR, LDSW, GLD><<<blocks, 256, lds>>>(g, out, tiles);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    const double flops = (double)blocks * 4 * tiles * 16 * NACC * 4096.0;
    printf("%-34s blocks/CU=%d acc=%d  %8.1f us  %6.1f TF\n", name, blocks_per_cu, NACC, ms * 1e3, flops / ms / 1e9);
}

int main(int argc, char** argv) {
    const bool only_glds = argc > 1;
    float *g, *out;
    hipMalloc(&g, (size_t)64 * 1048576 * 4 + 1024 * 256 * 16 * 4); hipMemset(g, 0, (size_t)64 * 1048576 * 4);
    hipMalloc(&out, 1024 * 256 * 4 * 4);
    for (int bpc = 1; bpc <= 4 && !only_glds; ++bpc) {
        run<1, false, false, false, false>("mfma only", g, out, bpc);
        run<1, true, false, false, false>("+ds_read_b128", g, out, bpc);
        run<1, true, true, false, false>("+ds_read +barrier", g, out, bpc);
        run<1, true, true, true, false>("+ds_read +barrier +ds_write", g, out, bpc);
        run<1, true, true, true, true>("+ds_read +barrier +ds_write +gld", g, out, bpc);
        run<1, false, true, false, false>("mfma +barrier", g, out, bpc);
    }
    for (int bpc = 1; bpc <= 2 && !only_glds; ++bpc) {
        run<2, false, false, false, false>("mfma only", g, out, bpc);
        run<2, true, true, true, true>("+ds_read +barrier +ds_write +gld", g, out, bpc);
        run<4, false, false, false, false>("mfma only", g, out, bpc);
        run<4, true, true, true, true>("+ds_read +barrier +ds_write +gld", g, out, bpc);
    }
    for (int bpc = 1; bpc <= 4; ++bpc) {
        run_glds<1, 2, 4>("c2-like", g, out, bpc);
        run_glds<1, 3, 4>("c2-like", g, out, bpc);
        run_glds<1, 4, 4>("c2-like", g, out, bpc);
        run_glds<1, 3, 8>("c2-like BK64", g, out, bpc);
    }
    for (int bpc = 1; bpc <= 3; ++bpc) {
        run_glds<2, 3, 4>("acc2", g, out, bpc);
        run_glds<4, 3, 4>("acc4", g, out, bpc);
        run_glds<4, 3, 8>("acc4 BK64", g, out, bpc);
    }
    return 0;
}
