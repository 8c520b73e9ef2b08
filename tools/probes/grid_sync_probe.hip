// Round-2 design probe (no product code): what does a device-wide phase boundary cost on MI355X?
//   1. back-to-back dependent tiny kernels inside a hipGraph   -> per-node cost of a kernel boundary
//   2. the same kernels launched eagerly on one stream          -> host launch rate
//   3. a persistent grid (one 512-thread workgroup per CU) separated by software grid barriers
//      (monotonic counter in global memory, one atomic per workgroup + polling)     -> per-barrier cost
//   4. like 3 with a hierarchical barrier: one counter per XCD (blockIdx % 8, the round-robin workgroup -> XCD
//      mapping), the last arriver of each XCD bumps the global counter
// The decode step has ~224 kernel boundaries (7 per layer); DESIGN.md section 5 estimates ~5 us of cold start per
// boundary.  A persistent per-layer / per-step kernel pays (3) or (4) instead and can prefetch the next phase's weights
// before the barrier.  Build + run:  hipcc --offload-arch=gfx950 -O3 -o grid_sync_probe grid_sync_probe.hip && ./grid_sync_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define CK(x)                                                                                                  \
    do {                                                                                                       \
        hipError_t e_ = (x);                                                                                   \
        if (e_ != hipSuccess) {                                                                                \
            printf("%s -> %s\n", #x, hipGetErrorString(e_));                                                   \
            return 1;                                                                                          \
        }                                                                                                      \
    } while (0)

__global__ void tiny_kernel(float* p)
{
    if (threadIdx.x == 0) {
        p[blockIdx.x] += 1.f;  // one dependent read-modify-write per workgroup: the next kernel must see it
    }
}

// flat barrier: every workgroup adds 1 to a monotonic counter and polls until all `nwg` of this round arrived
__device__ __forceinline__ void grid_barrier_flat(unsigned* counter, unsigned nwg, unsigned round)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        atomicAdd(counter, 1u);
        const unsigned target = nwg * (round + 1);
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
        }
        __threadfence();
    }
    __syncthreads();
}

// hierarchical: 8 per-XCD counters (workgroups are dealt round-robin to the XCDs), last arriver bumps the global one
__device__ __forceinline__ void grid_barrier_xcd(unsigned* xcd_counters, unsigned* global_counter, unsigned nwg, unsigned round)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned xcd    = blockIdx.x & 7u;
        const unsigned per    = (nwg + 7u - xcd) / 8u;  // workgroups of this XCD
        const unsigned ticket = atomicAdd(&xcd_counters[xcd * 32], 1u);  // 128-byte stride between the counters
        if (ticket == per * (round + 1) - 1) {
            atomicAdd(global_counter, 1u);
        }
        const unsigned target = 8u * (round + 1);
        while (__hip_atomic_load(global_counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
        }
        __threadfence();
    }
    __syncthreads();
}

template<int MODE>
__global__ __launch_bounds__(512) void persistent_kernel(float* p, unsigned* counters, int rounds)
{
    for (int r = 0; r < rounds; ++r) {
        if (threadIdx.x == 0) {
            p[blockIdx.x] += 1.f;
        }
        if (MODE == 0) {
            grid_barrier_flat(counters, gridDim.x, (unsigned)r);
        }
        else {
            grid_barrier_xcd(counters + 32, counters, gridDim.x, (unsigned)r);
        }
    }
}

static float ms_between(hipEvent_t a, hipEvent_t b)
{
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device: %s, %d CUs\n", prop.gcnArchName, cus);
    float*    p = nullptr;
    unsigned* c = nullptr;
    CK(hipMalloc(&p, 4096 * sizeof(float)));
    CK(hipMalloc(&c, 4096 * sizeof(unsigned)));
    CK(hipMemset(p, 0, 4096 * sizeof(float)));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int N = 224;

    // 1. graph of N dependent tiny kernels (256 workgroups x 512 threads, like a decode GEMM grid)
    hipGraph_t     g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < N; ++i) {
        tiny_kernel<<<cus, 512, 0, st>>>(p);
    }
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int w = 0; w < 3; ++w) {
        CK(hipGraphLaunch(ge, st));
    }
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int w = 0; w < 10; ++w) {
        CK(hipGraphLaunch(ge, st));
    }
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    printf("hipGraph, %d dependent tiny kernels: %.2f us per kernel boundary\n", N, ms_between(e0, e1) * 1e3f / (10.f * N));

    // 2. eager
    CK(hipEventRecord(e0, st));
    for (int w = 0; w < 10 * N; ++w) {
        tiny_kernel<<<cus, 512, 0, st>>>(p);
    }
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    printf("eager launches:                        %.2f us per kernel\n", ms_between(e0, e1) * 1e3f / (10.f * N));

    // 3. / 4. persistent grid + software barriers (needs every workgroup resident: one per CU)
    for (int mode = 0; mode < 2; ++mode) {
        const int rounds = 2000;
        CK(hipMemsetAsync(c, 0, 4096 * sizeof(unsigned), st));
        if (mode == 0) {
            persistent_kernel<0><<<cus, 512, 0, st>>>(p, c, 10);
        }
        else {
            persistent_kernel<1><<<cus, 512, 0, st>>>(p, c, 10);
        }
        CK(hipMemsetAsync(c, 0, 4096 * sizeof(unsigned), st));
        CK(hipEventRecord(e0, st));
        if (mode == 0) {
            persistent_kernel<0><<<cus, 512, 0, st>>>(p, c, rounds);
        }
        else {
            persistent_kernel<1><<<cus, 512, 0, st>>>(p, c, rounds);
        }
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        printf("persistent grid, %s barrier:  %.2f us per barrier (%d workgroups)\n", mode ? "per-XCD + global" : "flat           ",
               ms_between(e0, e1) * 1e3f / rounds, cus);
    }
    std::vector<float> h(4);
    CK(hipMemcpy(h.data(), p, 16, hipMemcpyDeviceToHost));
    printf("check: p[0] = %.0f\n", h[0]);
    return 0;
}
