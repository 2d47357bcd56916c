// Experiment (diagnostic): the VALU issue rate a BITSLICED cipher would see -- a stream of
// three-input boolean ops (v_bitop3_b32) over N live registers per lane (128 bit planes +
// temporaries), at the occupancy that register count allows.  Reported: cycles per
// wave-instruction per SIMD, and the blocks/clk/CU an AES-128 of `ops_per_block` such
// operations per block would reach (each lane-op processes 32 blocks' worth of one bit).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32;
typedef unsigned long long u64;

template <int N, int OP, int WPE>
__global__ __launch_bounds__(256, WPE) void k(u32 iters, u64 *cycles, u32 *sink, u32 seed)
{
    u32 a[N];
#pragma unroll
    for (int i = 0; i < N; ++i) a[i] = threadIdx.x * (i + 7) + seed;
    const u64 t0 = __builtin_readcyclecounter();
    for (u32 it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const u32 b = a[(i + 7) % N], c = a[(i + 29) % N];
            if (OP == 0) a[i] = __builtin_amdgcn_bitop3_b32(a[i], b, c, 0x96);       // xor3
            else if (OP == 1) a[i] = __builtin_amdgcn_bitop3_b32(a[i], b, c, 0x6a);  // a ^ (b & c)
            else a[i] = a[i] ^ b;                                                   // v_xor_b32
        }
    }
    const u64 t1 = __builtin_readcyclecounter();
    u32 acc = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) acc ^= a[i];
    if (acc == 0x12345678u) sink[0] = acc;
    if ((threadIdx.x & 63u) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int N, int OP, int WPE>
static void run(const char *name, int wg_per_cu)
{
    const int wgs = 256 * wg_per_cu;
    const u32 iters = 4000;
    u64 *d_cyc; u32 *d_sink;
    (void)hipMalloc(&d_cyc, wgs * 4 * sizeof(u64)); (void)hipMalloc(&d_sink, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<N, OP, WPE>), dim3(wgs), dim3(256), 0, 0, 16, d_cyc, d_sink, 3u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<N, OP, WPE>), dim3(wgs), dim3(256), 0, 0, iters, d_cyc, d_sink, 3u);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    u64 *h = (u64 *)malloc(wgs * 4 * sizeof(u64));
    (void)hipMemcpy(h, d_cyc, wgs * 4 * sizeof(u64), hipMemcpyDeviceToHost);
    u64 mx = 0;
    for (int i = 0; i < wgs * 4; ++i) if (h[i] > mx) mx = h[i];
    hipFuncAttributes fa; (void)hipFuncGetAttributes(&fa, (const void *)k<N, OP, WPE>);
    // per SIMD: wg_per_cu waves (a 256-thread WG puts one wave on each SIMD)
    const double per_simd = (double)wg_per_cu * iters * N;
    const double cyc = (double)mx / per_simd;
    // wall-clock rate: instructions per CU per ns
    const double inst_per_cu = 4.0 * per_simd;
    printf("%-34s waves/SIMD=%d vgpr=%3d  %.3f ms  %.2f cycles per wave-instr per SIMD; %.2f wave-instr/ns/CU;"
           " AES-128 at 420 / 470 / 520 ops per block: %.0f / %.0f / %.0f GiB/s\n",
           name, wg_per_cu, fa.numRegs, ms, cyc, inst_per_cu / (ms * 1e6),
           inst_per_cu / (ms * 1e-3) * 256 * 64 / 420 * 16 / (1 << 30),
           inst_per_cu / (ms * 1e-3) * 256 * 64 / 470 * 16 / (1 << 30),
           inst_per_cu / (ms * 1e-3) * 256 * 64 / 520 * 16 / (1 << 30));
    free(h); (void)hipFree(d_cyc); (void)hipFree(d_sink);
}

int main()
{
    for (int pass = 0; pass < 2; ++pass) {
        run<48, 0, 8>("bitop3 xor3, 48 regs", 8);
        run<112, 0, 4>("bitop3 xor3, 112 regs", 4);
        run<160, 0, 3>("bitop3 xor3, 160 regs", 3);
        run<160, 1, 3>("bitop3 a^(b&c), 160 regs", 3);
        run<224, 0, 2>("bitop3 xor3, 224 regs", 2);
        run<224, 0, 2>("bitop3 xor3, 224 regs, 1 wave", 1);
        run<160, 2, 3>("v_xor_b32, 160 regs", 3);
        run<48, 2, 8>("v_xor_b32, 48 regs", 8);
    }
    return 0;
}
