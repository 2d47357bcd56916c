// VALU instruction issue-rate microbenchmark (diagnostic): clk per wave-instruction
// per CU for the candidate address-building / combining ops, 16 waves per CU,
// 16 independent chains per lane.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32;
typedef unsigned long long u64;
#define CH 16

template <int OP>
__global__ __launch_bounds__(1024) void k(u32 iters, u64 *cycles, u32 *sink, u32 s0, u32 s1)
{
    u32 a[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) a[c] = threadIdx.x * (c + 7) + s0;
    const u32 lc = threadIdx.x * 4u;
    u32 selv = 0x0c020500u + (threadIdx.x >> 10);   // VGPR copy of the selector
    asm volatile("" : "+v"(selv));
    const u64 t0 = __builtin_readcyclecounter();
    for (u32 it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const u32 b = a[(c + 5) & (CH - 1)];
            if (OP == 0) a[c] = a[c] ^ b;                                                   // v_xor_b32
            else if (OP == 1) a[c] = __builtin_amdgcn_bitop3_b32(a[c], b, lc, 0x96);        // v_bitop3_b32
            else if (OP == 2) a[c] = __builtin_amdgcn_perm(a[c], b, 0x0c020500u);           // v_perm_b32 (sel in SGPR)
            else if (OP == 3) asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2" : "+v"(a[c]) : "v"(b));
            else if (OP == 4) asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(a[c]) : "v"(b), "s"(s1), "v"(a[c]));
            else if (OP == 5) asm volatile("v_bfe_u32 %0, %1, 8, 8" : "=v"(a[c]) : "v"(b));
            else if (OP == 6) asm volatile("v_lshl_or_b32 %0, %1, 8, %2" : "=v"(a[c]) : "v"(b), "v"(a[c]));
            else if (OP == 7) a[c] = __builtin_amdgcn_alignbit(a[c], b, 8);                  // v_alignbit_b32
            else if (OP == 8) asm volatile("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(a[c]) : "v"(a[c]), "v"(b));
            else if (OP == 9) asm volatile("v_bfi_b32 %0, %1, %2, %3" : "=v"(a[c]) : "s"(s1), "v"(b), "v"(a[c]));
            else if (OP == 10) asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(a[c]) : "v"(b), "s"(s1), "v"(a[c]));
            else if (OP == 12) a[c] = __builtin_amdgcn_perm(a[c], b, selv);                  // v_perm_b32, selector in a VGPR
            else if (OP == 13) a[c] = __builtin_amdgcn_bitop3_b32(a[c], b, s1, 0x96);        // v_bitop3_b32 with an SGPR operand
            else if (OP == 14) a[c] = a[c] ^ s1 ^ b;                                          // xor with SGPR
            else if (OP == 15) asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(a[c]) : "v"(b), "v"(selv), "v"(a[c]));
            else if (OP == 16) asm volatile("v_bfe_u32 %0, %1, %2, %3" : "=v"(a[c]) : "v"(b), "v"(lc), "v"(selv));
            else if (OP == 17) a[c] = __builtin_amdgcn_alignbit(a[c], b, selv);
            else if (OP == 11) asm volatile("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(a[c]) : "s"(s1), "v"(b));
        }
    }
    const u64 t1 = __builtin_readcyclecounter();
    u32 acc = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) acc ^= a[c];
    if (acc == 0x12345678u) sink[0] = acc;
    if ((threadIdx.x & 63u) == 0) cycles[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP>
static void run(const char *name)
{
    const u32 iters = 20000;
    const int wgs = 256;
    u64 *d_cyc; u32 *d_sink;
    (void)hipMalloc(&d_cyc, wgs * 16 * sizeof(u64)); (void)hipMalloc(&d_sink, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(wgs), dim3(1024), 0, 0, 16, d_cyc, d_sink, 3u, 0xff00u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(wgs), dim3(1024), 0, 0, iters, d_cyc, d_sink, 3u, 0xff00u);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    u64 *h = (u64 *)malloc(wgs * 16 * sizeof(u64));
    (void)hipMemcpy(h, d_cyc, wgs * 16 * sizeof(u64), hipMemcpyDeviceToHost);
    u64 mx = 0;
    for (int i = 0; i < wgs * 16; ++i) if (h[i] > mx) mx = h[i];
    const double n = 16.0 * iters * CH;
    printf("%-28s %.3f ms  %.3f clk/instr/CU  (%.2f cycles per instr per SIMD)  %.3f ns/instr/CU\n", name, ms,
           (double)mx / n, 4.0 * mx / n, ms * 1e6 / n);
    free(h); (void)hipFree(d_cyc); (void)hipFree(d_sink);
}

int main()
{
    run<0>("v_xor_b32"); run<1>("v_bitop3_b32"); run<2>("v_perm_b32"); run<3>("v_mov_b32_sdwa preserve");
    run<4>("v_and_or_b32"); run<5>("v_bfe_u32"); run<6>("v_lshl_or_b32"); run<7>("v_alignbit_b32");
    run<8>("v_or_b32_sdwa BYTE_2"); run<9>("v_bfi_b32"); run<10>("v_mad_u32_u24"); run<11>("v_lshlrev_b32_sdwa");
    run<12>("v_perm_b32 VGPR selector"); run<13>("v_bitop3_b32 SGPR operand"); run<14>("v_xor x2 with SGPR"); run<15>("v_and_or_b32 all VGPR"); run<16>("v_bfe_u32 all VGPR"); run<17>("v_alignbit all VGPR");
    return 0;
}
