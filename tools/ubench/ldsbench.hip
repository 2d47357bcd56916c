// LDS lookup-rate microbenchmark for gfx950 (diagnostic, not part of the product).
// Question: how many ds_read_b32 / b64 / b128 wave-instructions per clock can one
// CU sustain when every lane reads its own bank slot at a data-dependent row --
// the access pattern of the replicated AES T-tables -- and what does interleaved
// VALU (v_perm address building, xors) cost?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef unsigned int u32;
typedef unsigned long long u64;
extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
typedef __attribute__((address_space(3))) const u32 lds_cu32;
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const u32x2 lds_cu64;

#define CH 16

// MODE 0: pure pointer chase, CH independent chains, ds_read_b32
// MODE 1: chase + v_perm address build (value holds next index in byte 1)
// MODE 2: ds_read_b64 (8-byte rows slots: stride 8 per lane -> lanes 0..31 cover 256 B)
// MODE 3: MODE 1 + 0.5 extra xor per lookup (AES-like VALU load)
// MODE 4: pure chase, only lanes 0..31 active
// MODE 5: ds_read_u8 pure chase
template <int MODE>
__global__ __launch_bounds__(1024) void k(u32 iters, u64 *cycles, u32 *sink)
{
    const u32 lane = threadIdx.x & 31u;
    // fill: row x (256 B) slot r holds the chain value
    for (u32 i = threadIdx.x; i < 256u * 32u; i += blockDim.x) {
        const u32 x = i >> 5, r = i & 31u;
        const u32 nx = (x * 167u + 13u) & 255u;
        if (MODE == 2) {
            // 8-byte slots: row stride 256 B, slot r at r*8: only banks 0..63 (b64 has 64 banks)
            *(uint2 *)(lds + x * 256u + r * 8u) = make_uint2(nx * 256u + r * 8u, 0x01010101u * nx);
        } else if (MODE == 5) {
            // byte table: row x at x*128, slot r at r*4 (byte 0 holds next x)
            *(u32 *)(lds + x * 128u + r * 4u) = nx;
        } else {
            *(u32 *)(lds + x * 256u + r * 4u) = (MODE == 1 || MODE == 3) ? (nx << 8) | (nx * 0x01010001u & 0xffff00ffu) : nx * 256u + r * 4u;
        }
    }
    __syncthreads();
    if (MODE == 4 && (threadIdx.x & 32u)) return;
    u32 a[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) a[c] = MODE == 2 ? ((lane * 8u) + ((c * 37u + threadIdx.x) & 255u) * 256u)
                                     : MODE == 5 ? ((c * 37u + threadIdx.x) & 255u)
                                     : ((lane * 4u) + ((c * 37u + threadIdx.x) & 255u) * 256u);
    u32 acc = 0;
    const u32 slot = lane * 4u;
    const u64 t0 = __builtin_readcyclecounter();
    for (u32 it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (MODE == 0 || MODE == 4) {
                a[c] = *(lds_cu32 *)(uintptr_t)a[c];
            } else if (MODE == 1 || MODE == 3) {
                const u32 v = *(lds_cu32 *)(uintptr_t)a[c];
                a[c] = __builtin_amdgcn_perm(v, slot, 0x0c0c0500u);     // (v.b1 << 8) | slot
                if (MODE == 3 && (c & 1)) acc = __builtin_amdgcn_bitop3_b32(acc, v, a[c ^ 1], 0x96);
            } else if (MODE == 2) {
                const u32x2 v = *(lds_cu64 *)(uintptr_t)a[c];
                a[c] = v.x; acc ^= v.y;
            } else if (MODE == 5) {
                const u32 v = *(__attribute__((address_space(3))) const unsigned char *)(uintptr_t)(a[c] * 128u + slot);
                a[c] = v;
            }
        }
    }
    const u64 t1 = __builtin_readcyclecounter();
#pragma unroll
    for (int c = 0; c < CH; ++c) acc ^= a[c];
    if (acc == 0x12345678u) sink[0] = acc;
    if ((threadIdx.x & 63u) == 0) cycles[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
static void run(const char *name, int wgs, int threads, u32 iters)
{
    u64 *d_cyc; u32 *d_sink;
    hipMalloc(&d_cyc, wgs * 16 * sizeof(u64)); hipMalloc(&d_sink, 4);
    hipMemset(d_cyc, 0, wgs * 16 * sizeof(u64));
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(threads), 65536, 0, 16, d_cyc, d_sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(threads), 65536, 0, iters, d_cyc, d_sink);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    u64 *h = (u64 *)malloc(wgs * 16 * sizeof(u64));
    hipMemcpy(h, d_cyc, wgs * 16 * sizeof(u64), hipMemcpyDeviceToHost);
    double avg = 0; int nw = threads / 64; u64 mx = 0;
    for (int w = 0; w < wgs; ++w) for (int i = 0; i < nw; ++i) { avg += h[w * 16 + i]; if (h[w * 16 + i] > mx) mx = h[w * 16 + i]; }
    avg /= (double)wgs * nw;
    const double instr_per_cu = (double)nw * iters * CH;           // wave-instructions per CU (1 WG per CU)
    printf("%-34s wgs=%d thr=%d: %.3f ms  wave-cycles avg %.0f max %llu -> %.2f clk/wave-instr/CU (max-wave basis), "
           "eff clock %.2f GHz (if counter=shader clk)\n",
           name, wgs, threads, ms, avg, (unsigned long long)mx, (double)mx / instr_per_cu, (double)mx / (ms * 1e6));
    free(h); hipFree(d_cyc); hipFree(d_sink);
}

int main()
{
    const u32 it = 20000;
    run<0>("b32 chase", 256, 1024, it);
    run<0>("b32 chase 512thr", 256, 512, it);
    run<0>("b32 chase 256thr", 256, 256, it);
    run<4>("b32 chase lanes0-31 only", 256, 1024, it);
    run<1>("b32 chase + v_perm", 256, 1024, it);
    run<3>("b32 chase + v_perm + 0.5 bitop3", 256, 1024, it);
    run<2>("b64 chase", 256, 1024, it);
    run<5>("u8 chase (+mad addr)", 256, 1024, it);
    run<0>("b32 chase 1 WG only", 1, 1024, it);
    return 0;
}
