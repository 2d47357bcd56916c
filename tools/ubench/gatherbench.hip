// Experiment (diagnostic): is the vector-memory (TA/TCP) path a usable SECOND gather engine
// next to the LDS?  Each lane chases through a 1 KiB table in global memory (L1-resident
// after the first touch), alone and mixed 1:R with LDS lookups of the replicated-table kind.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32;
typedef unsigned long long u64;
extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
typedef __attribute__((address_space(3))) const u32 lds_cu32;
#define CH 16

// RATIO = LDS lookups per global lookup (0: global only, -1: LDS only)
template <int RATIO>
__global__ __launch_bounds__(1024) void k(u32 iters, const u32 *__restrict__ gtab, u64 *cycles, u32 *sink)
{
    const u32 lane = threadIdx.x & 31u;
    for (u32 i = threadIdx.x; i < 256u * 32u; i += blockDim.x) {
        const u32 x = i >> 5, r = i & 31u;
        *(u32 *)(lds + x * 128u + r * 4u) = ((x * 167u + 13u) & 255u);
    }
    __syncthreads();
    u32 a[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) a[c] = (c * 37u + threadIdx.x * 11u) & 255u;
    const u32 slot = lane * 4u;
    const u64 t0 = __builtin_readcyclecounter();
    for (u32 it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const bool global = RATIO == 0 || (RATIO > 0 && (c % (RATIO + 1)) == RATIO);
            if (global) a[c] = gtab[a[c]];
            else        a[c] = *(lds_cu32 *)(uintptr_t)(a[c] * 128u + slot);
        }
    }
    const u64 t1 = __builtin_readcyclecounter();
    u32 acc = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) acc ^= a[c];
    if (acc == 0x12345678u) sink[0] = acc;
    if ((threadIdx.x & 63u) == 0) cycles[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int RATIO>
static void run(const char *name, const u32 *gtab, u32 iters)
{
    const int wgs = 256;
    u64 *d_cyc; u32 *d_sink;
    (void)hipMalloc(&d_cyc, wgs * 16 * sizeof(u64)); (void)hipMalloc(&d_sink, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<RATIO>, dim3(wgs), dim3(1024), 32768, 0, 16, gtab, d_cyc, d_sink);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<RATIO>, dim3(wgs), dim3(1024), 32768, 0, iters, gtab, d_cyc, d_sink);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    u64 *h = (u64 *)malloc(wgs * 16 * sizeof(u64));
    (void)hipMemcpy(h, d_cyc, wgs * 16 * sizeof(u64), hipMemcpyDeviceToHost);
    u64 mx = 0;
    for (int i = 0; i < wgs * 16; ++i) if (h[i] > mx) mx = h[i];
    const double n = 16.0 * iters * CH;
    printf("%-40s %.3f ms  %.2f clk per lookup wave-instr per CU (all lookups)\n", name, ms, (double)mx / n);
    free(h); (void)hipFree(d_cyc); (void)hipFree(d_sink);
}

int main()
{
    u32 h[256];
    for (int x = 0; x < 256; ++x) h[x] = (x * 167u + 13u) & 255u;
    u32 *gtab; (void)hipMalloc(&gtab, 1024); (void)hipMemcpy(gtab, h, 1024, hipMemcpyHostToDevice);
    run<-1>("LDS only (ds_read_b32)", gtab, 4000);
    run<0>("global only (1 KiB table, L1)", gtab, 1000);
    run<15>("15 LDS : 1 global", gtab, 4000);
    run<7>("7 LDS : 1 global", gtab, 4000);
    run<3>("3 LDS : 1 global", gtab, 2000);
    run<-1>("LDS only again", gtab, 4000);
    return 0;
}
