// Price list for the persistent single-frame kernel (csrc/infer_b1.hip): what one in-launch
// grid-wide hand-off costs on MI355X, in the exact form that kernel uses.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/grid_barrier_probe tools/grid_barrier_probe.hip
//   tools/bin/grid_barrier_probe [blocks] [threads] [stages]
//
// Each stage: every block writes a 1 KB record (16-B `sc1` stores, one wave), drains it, arrives
// on a counter, waits until all blocks have arrived, then reads ANOTHER block's record with `sc1`
// loads and checks every word (records are double-buffered: stage s + 2 reuses the slot of stage
// s only after every block has passed barrier s + 1, i.e. finished reading; the hand-off of cdna guide G16 without release / acquire fences:
// all payload stores and loads are `sc1`, the counter is agent-scope atomics, the poll is an `sc1`
// load by one wave, the other waves wait behind a workgroup barrier).
// Variants: 0 = one counter; 1 = 8 counters (block & 7), polled by 8 lanes of one wave;
//           2 = variant 1 + `__threadfence()` on both sides (the fenced form, for comparison);
//           3 = variant 1 without any payload (the bare barrier);
//           16 / 32 / 64 = variant 1 with that many counters (polled by as many lanes).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kShardStride = 32;      // ints: one 128-B line per shard

__device__ __forceinline__ void store_sc1(float* p, f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ f32x4 load_sc1(const float* p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ int load_int_sc1(const int* p) {
    int v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return v;
}

template <int VARIANT>
__global__ __launch_bounds__(1024) void probe(int* counters, float* records, int stages, int base,
                                              int* errors, long long* cycles) {
    const int nblk = gridDim.x, b = blockIdx.x, tid = threadIdx.x;
    const int nshard = VARIANT == 0 ? 1 : VARIANT >= 16 ? VARIANT : 8;
    const int per_shard = nblk / nshard;
    int bad = 0;
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int s = 1; s <= stages; ++s) {
        if (VARIANT != 3 && tid < 64) {
            f32x4 v = {(float)(s * 1000 + b), (float)tid, (float)s, 1.f};
            store_sc1(records + ((size_t)(s & 1) * nblk * 256) + ((size_t)b * 64 + tid) * 4, v);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (tid < 64) {
            if (VARIANT == 2 && tid == 0) __threadfence();
            if (tid == 0)
                __hip_atomic_fetch_add(&counters[(VARIANT == 0 ? 0 : (b % nshard)) * kShardStride], 1,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int target = base + s * per_shard;
            int spins = 0;
            while (true) {
                int ok = 1;
                if (tid < nshard) ok = (load_int_sc1(&counters[tid * kShardStride]) - target) >= 0;
                if (__all(ok)) break;
                if (++spins > (1 << 22)) { bad |= 1 << 30; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (VARIANT == 2 && tid == 0) __threadfence();
        }
        __syncthreads();
        if (VARIANT != 3 && tid >= 64 && tid < 128) {      // a different wave than the writer
            const int src = (b + 37 * s) % nblk, l = tid - 64;
            const f32x4 v = load_sc1(records + ((size_t)(s & 1) * nblk * 256) + ((size_t)src * 64 + l) * 4);
            if (v[0] != (float)(s * 1000 + src) || v[1] != (float)l || v[2] != (float)s) ++bad;
        }
        // uneven load: some blocks dawdle (arrival skew is part of the price)
        if (VARIANT != 3 && (b % 17) == (s % 17)) __builtin_amdgcn_s_sleep(20);
    }
    const long long t1 = __builtin_amdgcn_s_memrealtime();
    if (bad) atomicAdd(errors, bad & 0xffff ? 1 : 0x10000);
    if (tid == 0 && b == 0) cycles[0] = t1 - t0;
}

template <int V>
void run(const char* name, int blocks, int threads, int stages, int* counters, float* records,
         int* errors, long long* cycles) {
    CK(hipMemset(counters, 0, 64 * kShardStride * sizeof(int)));
    CK(hipMemset(errors, 0, sizeof(int)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int base = 0;
    const int per_shard = V == 0 ? blocks : V >= 16 ? blocks / V : blocks / 8;
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
        CK(hipEventRecord(e0));
        probe<V><<<blocks, threads>>>(counters, records, stages, base, errors, cycles);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        base += stages * per_shard;
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    int herr = 0; long long hc = 0;
    CK(hipMemcpy(&herr, errors, sizeof(int), hipMemcpyDeviceToHost));
    CK(hipMemcpy(&hc, cycles, sizeof(long long), hipMemcpyDeviceToHost));
    printf("%-44s blocks %d x %d: %.2f us per stage (event), %.2f us in-kernel (100 MHz clock), errors 0x%x\n",
           name, blocks, threads, best * 1e3 / stages, hc * 0.01 / stages, herr);
}

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 256;
    const int threads = argc > 2 ? atoi(argv[2]) : 1024;
    const int stages = argc > 3 ? atoi(argv[3]) : 200;
    int* counters; float* records; int* errors; long long* cycles;
    CK(hipMalloc(&counters, 64 * kShardStride * sizeof(int)));
    CK(hipMalloc(&records, (size_t)blocks * 2048));   // two generations: a record is rewritten two barriers later
    CK(hipMalloc(&errors, sizeof(int)));
    CK(hipMalloc(&cycles, sizeof(long long)));
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    printf("%s, %d CUs\n", p.name, p.multiProcessorCount);
    run<3>("bare barrier, 8 sharded counters", blocks, threads, stages, counters, records, errors, cycles);
    run<0>("1 KB sc1 record + barrier, one counter", blocks, threads, stages, counters, records, errors, cycles);
    run<1>("1 KB sc1 record + barrier, 8 sharded counters", blocks, threads, stages, counters, records, errors, cycles);
    run<2>("same + __threadfence() on both sides", blocks, threads, stages, counters, records, errors, cycles);
    run<16>("1 KB sc1 record + barrier, 16 sharded counters", blocks, threads, stages, counters, records, errors, cycles);
    run<32>("1 KB sc1 record + barrier, 32 sharded counters", blocks, threads, stages, counters, records, errors, cycles);
    run<64>("1 KB sc1 record + barrier, 64 sharded counters", blocks, threads, stages, counters, records, errors, cycles);
    // launch-boundary reference: the same number of trivial dependent kernels
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        for (int s = 0; s < stages; ++s)
            probe<3><<<blocks, threads>>>(counters, records, 0, 0, errors, cycles);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep == 2) printf("%-44s blocks %d x %d: %.2f us per launch (eager, back to back)\n",
                             "trivial dependent kernels", blocks, threads, ms * 1e3 / stages);
    }
    return 0;
}
