// VALU / LDS co-issue microbenchmark for gfx950 (diagnostic).
//  mode 0: every wave runs independent v_bitop3_b32 chains (3 VGPR sources)
//  mode 1: every wave runs v_xor_b32 chains (VOP2)
//  mode 2: waves 0..7 of the workgroup run bitop3 chains, waves 8..15 run the
//          conflict-free ds_read_b32 pointer chase (wave-specialised hybrid)
//  mode 3: only the LDS half of mode 2 (VALU waves exit immediately)
//  mode 4: only the VALU half of mode 2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32;
typedef unsigned long long u64;
extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
typedef __attribute__((address_space(3))) const u32 lds_cu32;
#define CH 16

template <int MODE>
__global__ __launch_bounds__(1024) void k(u32 iters, u64 *cycles, u32 *sink)
{
    const u32 lane = threadIdx.x & 31u, wave = threadIdx.x >> 6;
    for (u32 i = threadIdx.x; i < 256u * 32u; i += blockDim.x) {
        const u32 x = i >> 5, r = i & 31u;
        *(u32 *)(lds + x * 256u + r * 4u) = ((x * 167u + 13u) & 255u) * 256u + r * 4u;
    }
    __syncthreads();
    const bool lds_wave = (MODE == 2 || MODE == 3 || MODE == 4) ? (wave >= 8) : false;
    if (MODE == 3 && !lds_wave) return;
    if (MODE == 4 && lds_wave) return;
    u32 a[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) a[c] = (lane * 4u) + ((c * 37u + threadIdx.x) & 255u) * 256u;
    u32 b = threadIdx.x * 2654435761u, d = ~b;
    const u64 t0 = __builtin_readcyclecounter();
    if (lds_wave) {
        for (u32 it = 0; it < iters; ++it) {
#pragma unroll
            for (int c = 0; c < CH; ++c) a[c] = *(lds_cu32 *)(uintptr_t)a[c];
        }
    } else {
        for (u32 it = 0; it < iters; ++it) {
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if (MODE == 1) a[c] = a[c] ^ b;
                else a[c] = __builtin_amdgcn_bitop3_b32(a[c], a[(c + 5) & (CH - 1)], d, 0x6a);
            }
        }
    }
    const u64 t1 = __builtin_readcyclecounter();
    u32 acc = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) acc ^= a[c];
    if (acc == 0x12345678u) sink[0] = acc;
    if ((threadIdx.x & 63u) == 0) cycles[blockIdx.x * 16 + wave] = t1 - t0;
}

template <int MODE>
static void run(const char *name, u32 iters)
{
    const int wgs = 256;
    u64 *d_cyc; u32 *d_sink;
    (void)hipMalloc(&d_cyc, wgs * 16 * sizeof(u64)); (void)hipMalloc(&d_sink, 4);
    (void)hipMemset(d_cyc, 0, wgs * 16 * sizeof(u64));
    (void)hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(1024), 65536, 0, 16, d_cyc, d_sink);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(1024), 65536, 0, iters, d_cyc, d_sink);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    u64 *h = (u64 *)malloc(wgs * 16 * sizeof(u64));
    (void)hipMemcpy(h, d_cyc, wgs * 16 * sizeof(u64), hipMemcpyDeviceToHost);
    u64 mxv = 0, mxl = 0;
    for (int w = 0; w < wgs; ++w) for (int i = 0; i < 16; ++i) {
        u64 v = h[w * 16 + i];
        if (i < 8) { if (v > mxv) mxv = v; } else { if (v > mxl) mxl = v; }
    }
    printf("%-40s %.3f ms | waves0-7 max %llu cyc (%.3f clk per wave-instr per CU if 8 waves, %.3f if 16) | waves8-15 max %llu cyc (%.3f clk per wave-instr per CU if 8 waves)\n",
           name, ms, (unsigned long long)mxv, (double)mxv / (8.0 * iters * CH), (double)mxv / (16.0 * iters * CH),
           (unsigned long long)mxl, (double)mxl / (8.0 * iters * CH));
    free(h); (void)hipFree(d_cyc); (void)hipFree(d_sink);
}

int main()
{
    const u32 it = 20000;
    run<0>("all waves bitop3 (16 waves)", it);
    run<1>("all waves v_xor (16 waves)", it);
    run<4>("8 VALU waves alone", it);
    run<3>("8 LDS waves alone", it);
    run<2>("8 VALU waves + 8 LDS waves", it);
    return 0;
}
