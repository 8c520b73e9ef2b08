// Round-2 probe (no product code): what does ONE node of a dependent hipGraph chain cost as a function of the launch
// geometry?  The decode GEMM launches 128..224 workgroups of 1024 threads with 128 KB of dynamic LDS; an empty kernel of
// that geometry was measured at 3.1 - 4.5 us per node (profiles/r02_bench_gemm_hbm_cold.txt, arm abl32), against the
// 1.45 us the hardware guide quotes for trivial 256-thread workgroups.  This probe separates workgroup size, LDS size and
// workgroup count.   hipcc --offload-arch=gfx950 -O3 -o launch_floor_probe launch_floor_probe.hip && ./launch_floor_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CK(x)                                                                                                  \
    do {                                                                                                       \
        hipError_t e_ = (x);                                                                                   \
        if (e_ != hipSuccess) {                                                                                \
            printf("%s -> %s\n", #x, hipGetErrorString(e_));                                                   \
            return 1;                                                                                          \
        }                                                                                                      \
    } while (0)

struct Args {
    float* p;
    int    touch;     // 1: thread 0 of every workgroup does a dependent read-modify-write
    int    pad[24];   // kernel-argument block of the decode GEMM's size
};

template<int T>
__global__ __launch_bounds__(T) void node_kernel(Args a)
{
    extern __shared__ char smem[];
    if (a.touch && threadIdx.x == 0) {
        a.p[blockIdx.x] += 1.f;
    }
    if (a.touch == 7) {
        smem[threadIdx.x] = 1;  // keeps the LDS allocation alive
    }
}

template<int T>
static int run(int wgs, int lds, int touch, float* buf, hipStream_t st)
{
    CK(hipFuncSetAttribute((const void*)node_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
    Args a{};
    a.p     = buf;
    a.touch = touch;
    const int nodes = 64;
    hipGraph_t     g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < nodes; ++i) {
        node_kernel<T><<<wgs, T, lds, st>>>(a);
    }
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) {
        CK(hipGraphLaunch(ge, st));
    }
    CK(hipStreamSynchronize(st));
    const int reps = 10;
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) {
        CK(hipGraphLaunch(ge, st));
    }
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("threads %4d  wgs %4d  lds %6d  touch %d : %6.2f us / node\n", T, wgs, lds, touch, ms * 1000.f / (reps * nodes));
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
    return 0;
}

int main()
{
    hipStream_t st;
    CK(hipStreamCreate(&st));
    float* buf;
    CK(hipMalloc(&buf, 1 << 20));
    CK(hipMemset(buf, 0, 1 << 20));
    const int ldss[3] = {0, 65536, 131072};
    const int wgss[4] = {128, 224, 256, 512};
    for (int touch = 0; touch < 2; ++touch) {
        for (int li = 0; li < 3; ++li) {
            for (int wi = 0; wi < 4; ++wi) {
                if (run<256>(wgss[wi], ldss[li], touch, buf, st)) return 1;
                if (run<512>(wgss[wi], ldss[li], touch, buf, st)) return 1;
                if (run<1024>(wgss[wi], ldss[li], touch, buf, st)) return 1;
            }
        }
    }
    // many small workgroups instead of few large ones (same wave count as 224 x 1024)
    if (run<256>(896, 32768, 1, buf, st)) return 1;
    if (run<256>(1024, 32768, 1, buf, st)) return 1;
    if (run<64>(3584, 0, 1, buf, st)) return 1;
    return 0;
}
