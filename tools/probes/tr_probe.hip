// Empirical semantics of ds_read_b64_tr_b16 / ds_read_b64_tr_b8 on gfx950.
// LDS is filled with u16 (resp. u8) values equal to their element index; every lane passes address base + lane*8.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void probe(uint32_t* out16, uint32_t* out8)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[4096];
    const int l = threadIdx.x;
    uint16_t* p16 = (uint16_t*)lds;
    for (int i = l; i < 1024; i += 64) p16[i] = (uint16_t)i;
    __syncthreads();
    uint32_t a = (uint32_t)(uintptr_t)(lds) + l * 8;   // LDS byte address
    uint32_t r0, r1;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(*(uint64_t*)&r0) : "v"(a) : "memory");
    // the 64-bit result lands in a register pair; fetch both halves
    uint64_t v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    out16[l * 2]     = (uint32_t)v;
    out16[l * 2 + 1] = (uint32_t)(v >> 32);
    __syncthreads();
    for (int i = l; i < 2048; i += 64) lds[i] = (uint8_t)i;
    __syncthreads();
    asm volatile("ds_read_b64_tr_b8 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    out8[l * 2]     = (uint32_t)v;
    out8[l * 2 + 1] = (uint32_t)(v >> 32);
    (void)r0; (void)r1;
}
int main()
{
    uint32_t *d16, *d8, h16[128], h8[128];
    hipMalloc(&d16, 512); hipMalloc(&d8, 512);
    probe<<<1, 64>>>(d16, d8);
    hipMemcpy(h16, d16, 512, hipMemcpyDeviceToHost);
    hipMemcpy(h8, d8, 512, hipMemcpyDeviceToHost);
    printf("tr_b16 (u16 element indices), address = lane*8 bytes:\n");
    for (int l = 0; l < 64; ++l)
        printf("lane %2d: %4u %4u %4u %4u\n", l, h16[2*l] & 0xffff, h16[2*l] >> 16, h16[2*l+1] & 0xffff, h16[2*l+1] >> 16);
    printf("tr_b8 (u8 element indices mod 256), address = lane*8 bytes:\n");
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int j = 0; j < 8; ++j) printf(" %3u", (j < 4 ? (h8[2*l] >> (8*j)) : (h8[2*l+1] >> (8*(j-4)))) & 0xff);
        printf("\n");
    }
    return 0;
}
