// Round-5 probe (no product code): can a dependent kernel START before its predecessor has finished on this stack?
// DESIGN.md 3.7 / 8 name "overlap across kernel boundaries" as the remaining lever of the decode step: the next kernel's weight
// stream does not depend on its predecessor, so its workgroups could prefetch and then wait on a device-side "predecessor done"
// counter.  Two ways to get a successor's workgroups dispatched early: (a) hipExtLaunchKernel(..., hipExtAnyOrderLaunch) -- the
// AQL packet without its barrier bit (hip_ext.h says "not supported on AMD GFX9xx boards" for the module form); (b) a second
// stream.  The probe runs pairs (A: every workgroup streams for a few us, then bumps a counter; B: every workgroup stamps its start,
// then waits -- bounded -- until the counter says A is complete, stamps again) and reports, per mode, the time per pair and how
// many of B's workgroups started before A's last workgroup ended.
//   hipcc --offload-arch=gfx950 -O2 -o any_order_probe any_order_probe.hip && ./any_order_probe
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

constexpr int kWgs = 256, kThreads = 256;

// A: stream `bytes_per_wg` through the workgroup, then one agent-scope increment; stamps its end
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(kThreads) void producer(const u32x4* src, size_t vec_per_wg, unsigned* counter, uint64_t* end_stamp, float* sink)
{
    const u32x4* p   = src + (size_t)blockIdx.x * vec_per_wg;
    u32x4        acc = {0, 0, 0, 0};
    for (size_t i = threadIdx.x; i < vec_per_wg; i += kThreads) {
        acc ^= __builtin_nontemporal_load(p + i);
    }
    if (acc.x == 0x12345678u) {
        sink[threadIdx.x] = (float)acc.y;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        end_stamp[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// B: start stamp, bounded wait for `target` arrivals, ready stamp
__global__ __launch_bounds__(kThreads) void consumer(unsigned* counter, unsigned target, uint64_t* start_stamp, uint64_t* ready_stamp, unsigned* timeouts)
{
    if (threadIdx.x == 0) {
        const uint64_t t0     = __builtin_amdgcn_s_memrealtime();
        start_stamp[blockIdx.x] = t0;
        bool ok = false;
        while (__builtin_amdgcn_s_memrealtime() - t0 < 200000) {  // 2 ms at 100 MHz: a spin must never hang the box
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
    }
    __syncthreads();
}

int main()
{
    hipStream_t st, st2;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
    const size_t bytes_per_wg = 96 << 10;  // ~4-5 us of streaming per workgroup at one workgroup per CU
    const size_t vec_per_wg   = bytes_per_wg / 16;
    const int    pairs        = 32;
    u32x4*       src;
    CK(hipMalloc(&src, (size_t)kWgs * bytes_per_wg * 4));  // 4 distinct regions: cold-ish data per pair
    CK(hipMemset(src, 1, (size_t)kWgs * bytes_per_wg * 4));
    unsigned *counter, *timeouts;
    uint64_t *a_end, *b_start, *b_ready;
    float*    sink;
    CK(hipMalloc(&counter, 4));
    CK(hipMalloc(&timeouts, 4));
    CK(hipMalloc(&a_end, 8 * kWgs * pairs));
    CK(hipMalloc(&b_start, 8 * kWgs * pairs));
    CK(hipMalloc(&b_ready, 8 * kWgs * pairs));
    CK(hipMalloc(&sink, 4096));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const char* names[3] = {"in-order launches, one stream (kernel boundary)", "consumer with hipExtAnyOrderLaunch, one stream",
                            "consumer on a second stream (no dependency edge)"};
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {  // rep 0 warms up
            CK(hipMemset(counter, 0, 4));
            CK(hipMemset(timeouts, 0, 4));
            CK(hipMemset(a_end, 0, 8 * kWgs * pairs));
            CK(hipMemset(b_start, 0, 8 * kWgs * pairs));
            CK(hipMemset(b_ready, 0, 8 * kWgs * pairs));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < pairs; ++i) {
                const u32x4* s_i = src + (size_t)(i & 3) * kWgs * vec_per_wg;
                producer<<<kWgs, kThreads, 0, st>>>(s_i, vec_per_wg, counter, a_end + (size_t)i * kWgs, sink);
                unsigned  target = (unsigned)(i + 1) * kWgs;
                uint64_t* bs     = b_start + (size_t)i * kWgs;
                uint64_t* br     = b_ready + (size_t)i * kWgs;
                if (mode == 1) {
                    void* args[] = {&counter, &target, &bs, &br, &timeouts};
                    CK(hipExtLaunchKernel((const void*)consumer, dim3(kWgs), dim3(kThreads), args, 0, st, nullptr, nullptr, hipExtAnyOrderLaunch));
                }
                else {
                    consumer<<<kWgs, kThreads, 0, mode == 2 ? st2 : st>>>(counter, target, bs, br, timeouts);
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
            std::vector<uint64_t> ae(kWgs * pairs), bs(kWgs * pairs), br(kWgs * pairs);
            unsigned              to = 0;
            CK(hipMemcpy(ae.data(), a_end, 8 * kWgs * pairs, hipMemcpyDeviceToHost));
            CK(hipMemcpy(bs.data(), b_start, 8 * kWgs * pairs, hipMemcpyDeviceToHost));
            CK(hipMemcpy(br.data(), b_ready, 8 * kWgs * pairs, hipMemcpyDeviceToHost));
            CK(hipMemcpy(&to, timeouts, 4, hipMemcpyDeviceToHost));
            double early = 0, gap = 0, wait = 0;
            for (int i = 0; i < pairs; ++i) {
                const uint64_t a_last = *std::max_element(ae.begin() + i * kWgs, ae.begin() + (i + 1) * kWgs);
                const uint64_t b_first = *std::min_element(bs.begin() + i * kWgs, bs.begin() + (i + 1) * kWgs);
                const uint64_t r_last  = *std::max_element(br.begin() + i * kWgs, br.begin() + (i + 1) * kWgs);
                int            n_early = 0;
                for (int w = 0; w < kWgs; ++w) {
                    n_early += bs[i * kWgs + w] < a_last;
                }
                early += n_early;
                gap += ((double)b_first - (double)a_last) / 100.0;
                wait += ((double)r_last - (double)a_last) / 100.0;
            }
            printf("%-55s: %6.2f us per pair | consumer workgroups started before the producer's last end: %5.1f of %d | first consumer start - last "
                   "producer end %+6.2f us | last consumer READY - last producer end %+6.2f us | spin timeouts %u\n",
                   names[mode], ms * 1000.f / pairs, early / pairs, kWgs, gap / pairs, wait / pairs, to);
        }
    }
    return 0;
}
