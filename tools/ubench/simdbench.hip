// SIMD issue-time model microbenchmark (diagnostic): T(clk per DS wave-instr per CU)
// as a function of K = independent VALU instructions issued per DS instruction by
// the SAME wave, for several DS instruction kinds.  16 pointer-chase chains per lane.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32;
typedef unsigned long long u64;
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
#define AS3 __attribute__((address_space(3)))
#define CH 16

// KIND 0: ds_read_b32  1: ds_read_b64  2: ds_read_b128  3: ds_read_u8  4: ds_read2_b32 (2 lookups / instr)
template <int KIND, int K, int VK>
__global__ __launch_bounds__(1024) void k(u32 iters, u64 *cycles, u32 *sink)
{
    const u32 lane = threadIdx.x & 31u;
    const u32 ssz = KIND == 2 ? 16u : (KIND == 1 || KIND == 4) ? 8u : 4u;   // slot bytes
    const u32 rowb = 32u * ssz;                                            // row bytes
    for (u32 i = threadIdx.x; i < 128u * 32u; i += blockDim.x) {           // 128 rows
        const u32 x = i >> 5, r = i & 31u;
        const u32 nx = (x * 37u + 13u) & 127u;
        u32 *p = (u32 *)(lds + x * rowb + r * ssz);
        p[0] = nx * rowb + r * ssz;
        for (u32 w = 1; w < ssz / 4; ++w) p[w] = ((nx * 53u + w * 7u) & 127u) * rowb + r * ssz;
    }
    __syncthreads();
    u32 a[CH], f[8];
#pragma unroll
    for (int c = 0; c < CH; ++c) a[c] = (lane * ssz) + ((c * 37u + threadIdx.x) & 127u) * rowb;
#pragma unroll
    for (int c = 0; c < 8; ++c) f[c] = threadIdx.x * (c + 3);
    u32 acc = 0;
    const u64 t0 = __builtin_readcyclecounter();
    for (u32 it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (KIND == 0) a[c] = *(AS3 const u32 *)(uintptr_t)a[c];
            else if (KIND == 1) { u32x2 v = *(AS3 const u32x2 *)(uintptr_t)a[c]; a[c] = v.x; acc ^= v.y; }
            else if (KIND == 2) { u32x4 v = *(AS3 const u32x4 *)(uintptr_t)a[c]; a[c] = v.x; acc ^= v.y ^ v.z ^ v.w; }
            else if (KIND == 3) { u32 v = *(AS3 const unsigned char *)(uintptr_t)a[c]; a[c] = (v & 0x7cu) + (a[c] & ~0x7fu); }
            else if (KIND == 4) {
                // two 4-byte lookups from one address register: words 0 and 1 of an 8-byte slot
                u32x2 v = *(AS3 const u32x2 *)(uintptr_t)a[c]; a[c] = v.x; acc ^= v.y;
            }
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int j = (c * K + k) & 7;
                if (VK == 0) f[j] = f[j] ^ (f[(j + 3) & 7] + 0);                     // v_xor (VOP2)
                else if (VK == 1) f[j] = __builtin_amdgcn_bitop3_b32(f[j], f[(j + 3) & 7], f[(j + 5) & 7], 0x96);
                else f[j] = __builtin_amdgcn_perm(f[j], f[(j + 3) & 7], 0x0c020500u);
            }
        }
    }
    const u64 t1 = __builtin_readcyclecounter();
#pragma unroll
    for (int c = 0; c < CH; ++c) acc ^= a[c];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc ^= f[c];
    if (acc == 0x12345678u) sink[0] = acc;
    if ((threadIdx.x & 63u) == 0) cycles[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND, int K, int VK>
static void run(const char *name, int wgs_per_cu)
{
    const u32 iters = 8000;
    const int wgs = 256 * wgs_per_cu;
    u64 *d_cyc; u32 *d_sink;
    (void)hipMalloc(&d_cyc, wgs * 16 * sizeof(u64)); (void)hipMalloc(&d_sink, 4);
    (void)hipFuncSetAttribute((const void *)k<KIND, K, VK>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, K, VK>), dim3(wgs), dim3(1024), 65536, 0, 16, d_cyc, d_sink);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, K, VK>), dim3(wgs), dim3(1024), 65536, 0, iters, d_cyc, d_sink);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    u64 *h = (u64 *)malloc(wgs * 16 * sizeof(u64));
    (void)hipMemcpy(h, d_cyc, wgs * 16 * sizeof(u64), hipMemcpyDeviceToHost);
    u64 mx = 0;
    for (int i = 0; i < wgs * 16; ++i) if (h[i] > mx) mx = h[i];
    const double instr_per_cu = 16.0 * wgs_per_cu * iters * CH;
    printf("%-26s K=%d wg/CU=%d: %.3f ms, %.3f clk per DS instr per CU (cycle counter), %.3f ns per DS instr per CU\n",
           name, K, wgs_per_cu, ms, (double)mx / instr_per_cu, ms * 1e6 / instr_per_cu);
    free(h); (void)hipFree(d_cyc); (void)hipFree(d_sink);
}

int main()
{
    run<0, 0, 0>("b32 + K v_xor", 1); run<0, 1, 0>("b32 + K v_xor", 1); run<0, 2, 0>("b32 + K v_xor", 1);
    run<0, 3, 0>("b32 + K v_xor", 1); run<0, 4, 0>("b32 + K v_xor", 1);
    run<0, 1, 1>("b32 + K bitop3", 1); run<0, 2, 1>("b32 + K bitop3", 1); run<0, 4, 1>("b32 + K bitop3", 1);
    run<0, 2, 2>("b32 + K perm", 1); run<0, 4, 2>("b32 + K perm", 1);
    run<0, 2, 0>("b32 + K v_xor", 2); run<0, 4, 0>("b32 + K v_xor", 2);
    run<1, 0, 0>("b64 + K v_xor", 1); run<1, 2, 0>("b64 + K v_xor", 1); run<1, 4, 0>("b64 + K v_xor", 1);
    run<2, 0, 0>("b128 + K v_xor", 1); run<2, 4, 0>("b128 + K v_xor", 1);
    run<3, 0, 0>("u8 + K v_xor", 1); run<3, 2, 0>("u8 + K v_xor", 1);
    return 0;
}
