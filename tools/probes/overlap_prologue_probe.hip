// Round-6 probe (no product code): what does a successor GAIN when it is launched without the dependency edge, sits beside
// its predecessor (<= 64 KB of LDS each), has its argument fetch and first weight units in flight under the predecessor's tail, and gates only
// its activation loads on the predecessor's completion counter?  Round 5's probe (any_order_probe.hip) measured when such a successor is READY
// (1.80 us after the predecessor's last end against 1.95 us for a plain dependent launch); this one measures the moment its first MFMA could
// issue: weights (32 KB per workgroup, cold, independent of the predecessor) AND the first activation stage (64 KB of the predecessor's output,
// system-fresh: sc1 loads of write-through stores) landed.
//   producer: 256 workgroups x 256 threads, 64 KB LDS; streams 96 KB, writes its 2 KB slice of x (512 KB) write-through, drains, bumps a counter
//   consumer: 256 workgroups x 256 threads, 64 KB LDS; mode 0 in-order launch (kernel boundary; no counter wait needed, polls once);
//             mode 1 hipExtAnyOrderLaunch, weights requested BEFORE the wait, x after it; mode 2 any-order, weights requested after the wait too
//   hipcc --offload-arch=gfx950 -O2 -o overlap_prologue_probe overlap_prologue_probe.hip && ./overlap_prologue_probe
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <vector>

#define CK(x)                                                                                                  \
    do {                                                                                                       \
        hipError_t e_ = (x);                                                                                   \
        if (e_ != hipSuccess) {                                                                                \
            printf("%s -> %s\n", #x, hipGetErrorString(e_));                                                   \
            return 1;                                                                                          \
        }                                                                                                      \
    } while (0)

constexpr int kWgs = 256, kThreads = 256, kLds = 64 << 10;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(kThreads) void producer(const u32x4* src, size_t vec_per_wg, u32x4* x, unsigned* counter, uint64_t* end_stamp, float* sink)
{
    extern __shared__ char smem[];
    const u32x4* p   = src + (size_t)blockIdx.x * vec_per_wg;
    u32x4        acc = {0, 0, 0, 0};
    for (size_t i = threadIdx.x; i < vec_per_wg; i += kThreads) {
        acc ^= __builtin_nontemporal_load(p + i);
    }
    if (acc.x == 0x12345678u) {
        sink[threadIdx.x] = (float)acc.y;
        smem[threadIdx.x] = 1;
    }
    // 2 KB slice of x = 128 vectors: threads 0 .. 127, write-through (sc1), then drain
    if (threadIdx.x < 128) {
        u32x4* dst = x + (size_t)blockIdx.x * 128 + threadIdx.x;
        acc.x |= 1u;
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(acc) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        end_stamp[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template<int MODE>
__global__ __launch_bounds__(kThreads) void consumer(const u32x4* weights, const u32x4* x, unsigned* counter, unsigned target, uint64_t* start_stamp,
                                                     uint64_t* ready_stamp, uint64_t* done_stamp, unsigned* timeouts, float* sink)
{
    extern __shared__ char smem[];
    __shared__ int go;
    if (threadIdx.x == 0) {
        start_stamp[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    }
    // "weights": 32 KB per workgroup, cold, independent of the predecessor: 8 x 16 B per thread
    const u32x4* wp = weights + (size_t)blockIdx.x * 2048 + threadIdx.x;
    u32x4        w[8];
    if (MODE != 2) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            w[i] = __builtin_nontemporal_load(wp + i * 256);
        }
    }
    if (threadIdx.x == 0) {
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
        bool           ok = false;
        while (__builtin_amdgcn_s_memrealtime() - t0 < 200000) {  // 2 ms: a spin must never hang the box
            if (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) {
                ok = true;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        if (!ok) {
            atomicAdd(timeouts, 1u);
        }
        ready_stamp[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
        go                      = 1;
    }
    __syncthreads();
    if (MODE == 2) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            w[i] = __builtin_nontemporal_load(wp + i * 256);
        }
    }
    // first activation stage: 64 KB of x (every workgroup the same 64 KB, as every GEMM workgroup reads the same activations): 16 x 16 B per thread,
    // agent-scope (sc1) loads of the predecessor's write-through stores
    u32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const unsigned long long* q = (const unsigned long long*)(x + (size_t)i * 256 + threadIdx.x);
        const unsigned long long  lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long  hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        acc.x ^= (unsigned)lo, acc.y ^= (unsigned)(lo >> 32), acc.z ^= (unsigned)hi, acc.w ^= (unsigned)(hi >> 32);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        acc ^= w[i];
    }
    if (acc.x == 0x12345678u && go == 7) {
        sink[threadIdx.x] = (float)acc.y;
        smem[threadIdx.x] = 1;
    }
    __syncthreads();  // every wave's loads have been consumed: weights and the activation stage are on chip
    if (threadIdx.x == 0) {
        done_stamp[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    }
}

int main()
{
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const size_t bytes_per_wg = 96 << 10;
    const size_t vec_per_wg   = bytes_per_wg / 16;
    const int    pairs        = 32;
    u32x4 *      src, *weights, *x;
    CK(hipMalloc(&src, (size_t)kWgs * bytes_per_wg * 4));
    CK(hipMemset(src, 1, (size_t)kWgs * bytes_per_wg * 4));
    CK(hipMalloc(&weights, (size_t)kWgs * 32768 * 4));  // 4 distinct regions
    CK(hipMemset(weights, 2, (size_t)kWgs * 32768 * 4));
    CK(hipMalloc(&x, 512 << 10));
    unsigned *counter, *timeouts;
    uint64_t *a_end, *b_start, *b_ready, *b_done;
    float*    sink;
    CK(hipMalloc(&counter, 4));
    CK(hipMalloc(&timeouts, 4));
    CK(hipMalloc(&a_end, 8 * kWgs * pairs));
    CK(hipMalloc(&b_start, 8 * kWgs * pairs));
    CK(hipMalloc(&b_ready, 8 * kWgs * pairs));
    CK(hipMalloc(&b_done, 8 * kWgs * pairs));
    CK(hipMalloc(&sink, 4096));
    CK(hipFuncSetAttribute((const void*)producer, hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
    CK(hipFuncSetAttribute((const void*)consumer<0>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
    CK(hipFuncSetAttribute((const void*)consumer<1>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
    CK(hipFuncSetAttribute((const void*)consumer<2>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const char* names[3] = {"in-order launch (kernel boundary), weights + x requested at workgroup start",
                            "hipExtAnyOrderLaunch: weights requested before the counter wait, x after it",
                            "hipExtAnyOrderLaunch: weights AND x requested after the counter wait"};
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemset(counter, 0, 4));
            CK(hipMemset(timeouts, 0, 4));
            CK(hipMemset(a_end, 0, 8 * kWgs * pairs));
            CK(hipMemset(b_start, 0, 8 * kWgs * pairs));
            CK(hipMemset(b_ready, 0, 8 * kWgs * pairs));
            CK(hipMemset(b_done, 0, 8 * kWgs * pairs));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < pairs; ++i) {
                const u32x4* s_i = src + (size_t)(i & 3) * kWgs * vec_per_wg;
                const u32x4* w_i = weights + (size_t)(i & 3) * kWgs * 2048;
                producer<<<kWgs, kThreads, kLds, st>>>(s_i, vec_per_wg, x, counter, a_end + (size_t)i * kWgs, sink);
                unsigned     target = (unsigned)(i + 1) * kWgs;
                uint64_t*    bs = b_start + (size_t)i * kWgs;
                uint64_t*    br = b_ready + (size_t)i * kWgs;
                uint64_t*    bd = b_done + (size_t)i * kWgs;
                const u32x4* xc = x;
                void*        args[] = {&w_i, &xc, &counter, &target, &bs, &br, &bd, &timeouts, &sink};
                if (mode == 0) {
                    CK(hipExtLaunchKernel((const void*)consumer<0>, dim3(kWgs), dim3(kThreads), args, kLds, st, nullptr, nullptr, 0));
                }
                else if (mode == 1) {
                    CK(hipExtLaunchKernel((const void*)consumer<1>, dim3(kWgs), dim3(kThreads), args, kLds, st, nullptr, nullptr, hipExtAnyOrderLaunch));
                }
                else {
                    CK(hipExtLaunchKernel((const void*)consumer<2>, dim3(kWgs), dim3(kThreads), args, kLds, st, nullptr, nullptr, hipExtAnyOrderLaunch));
                }
                CK(hipGetLastError());
            }
            CK(hipEventRecord(e1, st));
            CK(hipDeviceSynchronize());
            if (rep == 0) {
                continue;
            }
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<uint64_t> ae(kWgs * pairs), bs(kWgs * pairs), br(kWgs * pairs), bd(kWgs * pairs);
            unsigned              to = 0;
            CK(hipMemcpy(ae.data(), a_end, 8 * kWgs * pairs, hipMemcpyDeviceToHost));
            CK(hipMemcpy(bs.data(), b_start, 8 * kWgs * pairs, hipMemcpyDeviceToHost));
            CK(hipMemcpy(br.data(), b_ready, 8 * kWgs * pairs, hipMemcpyDeviceToHost));
            CK(hipMemcpy(bd.data(), b_done, 8 * kWgs * pairs, hipMemcpyDeviceToHost));
            CK(hipMemcpy(&to, timeouts, 4, hipMemcpyDeviceToHost));
            double early = 0, ready = 0, done_last = 0, done_mean = 0;
            for (int i = 1; i < pairs; ++i) {  // (pair 0: cold everything)
                const uint64_t a_last = *std::max_element(ae.begin() + i * kWgs, ae.begin() + (i + 1) * kWgs);
                const uint64_t r_last = *std::max_element(br.begin() + i * kWgs, br.begin() + (i + 1) * kWgs);
                const uint64_t d_last = *std::max_element(bd.begin() + i * kWgs, bd.begin() + (i + 1) * kWgs);
                int            n_early = 0;
                double         dm = 0;
                for (int w = 0; w < kWgs; ++w) {
                    n_early += bs[i * kWgs + w] < a_last;
                    dm += ((double)bd[i * kWgs + w] - (double)a_last) / 100.0;
                }
                early += n_early;
                ready += ((double)r_last - (double)a_last) / 100.0;
                done_last += ((double)d_last - (double)a_last) / 100.0;
                done_mean += dm / kWgs;
            }
            const int n = pairs - 1;
            printf("%-82s: %6.2f us per pair | consumers started before the producer's last end: %5.1f of %d | after the producer's last end: last READY %+5.2f us, "
                   "weights + x on chip: mean workgroup %+5.2f us, last workgroup %+5.2f us | spin timeouts %u\n",
                   names[mode], ms * 1000.f / pairs, early / n, kWgs, ready / n, done_mean / n, done_last / n, to);
        }
    }
    return 0;
}
