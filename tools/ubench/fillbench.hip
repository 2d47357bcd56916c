// Where do the ~4 us go that a one-workgroup kernel spends before its first lookup (128 KiB of replicated tables)?
// Stamps of the 100 MHz counter: kernel entry -> Te0 word arrived from memory -> LDS stores issued -> barrier passed,
// for thread 0 and for the LAST wave of the workgroup (the barrier waits for it).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I micro-aes_amd/csrc tools/ubench/fillbench.hip -o tools/ubench/fillbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "uaes_aes.hip.h"

__global__ __launch_bounds__(1024) void k_fill(const u32 *te0, unsigned long long *out, int variant)
{
    const unsigned long long t0 = wall_clock64();
    if (variant == 0) {                               // rounds 1-2: one ENTRY per lane, eight 16-byte stores each: lanes
        for (u32 i = threadIdx.x; i < 1024u; i += blockDim.x) {   // 256 bytes apart = the same banks (8-way conflict)
            const u32 x = i & 255u, k = i >> 8;
            store_replicas(x * 256u + (k & 1u) * 128u + (k >> 1) * 65536u, rotl32(te0[x], 8u * k));
        }
        __syncthreads();
    } else if (variant == 1) {
        fill_enc_tables(te0);                         // round 3: eight lanes per entry, conflict-free
    } else {
        fill_tables64(te0, 0);                        // the 64 KiB layout, sixteen lanes per row
    }
    const unsigned long long t3 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = t0; out[1] = t3; }
}

int main()
{
    u32 h[256];
    for (int i = 0; i < 256; ++i) h[i] = 0x01010101u * i;
    u32 *te0; unsigned long long *out, r[2];
    (void)hipMalloc(&te0, 1024); (void)hipMalloc(&out, 64);
    (void)hipMemcpy(te0, h, 1024, hipMemcpyHostToDevice);
    (void)hipFuncSetAttribute((const void *)k_fill, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    static const char *names[] = { "128 KiB, one entry per lane (rounds 1-2)", "128 KiB, eight lanes per entry (round 3)", "64 KiB layout, sixteen lanes per row" };
    for (int variant = 0; variant < 3; ++variant)
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(k_fill, dim3(1), dim3(1024), 131072, 0, te0, out, variant);
            (void)hipMemcpy(r, out, 16, hipMemcpyDeviceToHost);
            printf("%-42s kernel entry -> tables ready: %5.2f us\n", names[variant], (double)(r[1] - r[0]) / 100.0);
        }
    return 0;
}
