// What does one round of the serial chain cost, in shader cycles and in nanoseconds?
// One wave runs row_encrypt (sixteen lanes per block, uaes_aes.hip.h) / quad_encrypt (four lanes per block)
// N times back to back on its own output and reads s_memtime (shader clock) and s_memrealtime (100 MHz)
// around the loop: cycles per block, ns per block and the clock the lone wave actually ran at.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I micro-aes_amd/csrc tools/ubench/chainbench.hip -o tools/ubench/chainbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "uaes_aes.hip.h"

static u32 te0_host[256];
static void make_te0()
{
    unsigned char sb[256];
    unsigned p = 1, q = 1;
    do {                                       // S-box from the field arithmetic (3 is a generator)
        p = p ^ ((p << 1) & 0xff) ^ ((p & 0x80) ? 0x1b : 0);
        q ^= q << 1; q ^= q << 2; q ^= q << 4; q &= 0xff; if (q & 0x80) q ^= 0x09;
        unsigned x = q ^ ((q << 1) | (q >> 7)) ^ ((q << 2) | (q >> 6)) ^ ((q << 3) | (q >> 5)) ^ ((q << 4) | (q >> 4));
        sb[p] = (unsigned char)((x ^ 0x63) & 0xff);
    } while (p != 1);
    sb[0] = 0x63;
    for (int i = 0; i < 256; ++i) {
        unsigned s = sb[i], s2 = ((s << 1) ^ ((s & 0x80) ? 0x11b : 0)) & 0xff, s3 = s2 ^ s;
        te0_host[i] = s2 | (s << 8) | (s << 16) | (s3 << 24);
    }
}

template <int MODE>
__global__ __launch_bounds__(64) void k_chain(uaesk_rk rk, const u32 *te0, u32 n, unsigned long long *out)
{
    unsigned long long c0, c1, r0, r1;
    u32 acc;
    if (MODE == 0) {
        row_fill_tables(te0, rk);
        const RowLane<10> L = row_lane<10>();
        u32 m = threadIdx.x;
        c0 = __builtin_readcyclecounter(); r0 = wall_clock64();
        for (u32 i = 0; i < n; ++i) m = row_encrypt<10>(m, L);
        c1 = __builtin_readcyclecounter(); r1 = wall_clock64();
        acc = m;
    } else {
        quad_fill_tables(te0, rk);
        const LaneConst lc = quad_lane_const();
        u32 t[4] = { 1, 2, 3, 4 };
        c0 = __builtin_readcyclecounter(); r0 = wall_clock64();
        for (u32 i = 0; i < n; ++i) quad_encrypt<10>(t, rk, lc);
        c1 = __builtin_readcyclecounter(); r1 = wall_clock64();
        acc = t[0] ^ t[1] ^ t[2] ^ t[3];
    }
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; out[2] = acc; }
}

int main()
{
    make_te0();
    u32 *te0; unsigned long long *out, h[3];
    (void)hipMalloc(&te0, 1024); (void)hipMalloc(&out, 24);
    (void)hipMemcpy(te0, te0_host, 1024, hipMemcpyHostToDevice);
    uaesk_rk rk;
    for (int i = 0; i < 60; ++i) rk.w[i] = 0x01010101u * (unsigned)i;
    (void)hipFuncSetAttribute((const void *)k_chain<0>, hipFuncAttributeMaxDynamicSharedMemorySize, UAES_LDS_ROW);
    (void)hipFuncSetAttribute((const void *)k_chain<1>, hipFuncAttributeMaxDynamicSharedMemorySize, UAES_LDS_QUAD);
    for (int rep = 0; rep < 2; ++rep)
        for (u32 n : { 1000u, 100000u, 1000000u }) {
            hipLaunchKernelGGL(k_chain<0>, dim3(1), dim3(64), UAES_LDS_ROW, 0, rk, te0, n, out);
            (void)hipMemcpy(h, out, 24, hipMemcpyDeviceToHost);
            printf("row_encrypt  n=%7u: %7.1f cycles/block (%5.1f per round)  %6.1f ns/block  -> %4.0f MHz\n", n,
                   (double)h[0] / n, (double)h[0] / n / 10, (double)h[1] * 10.0 / n, (double)h[0] / ((double)h[1] * 10.0) * 1e3);
            hipLaunchKernelGGL(k_chain<1>, dim3(1), dim3(64), UAES_LDS_QUAD, 0, rk, te0, n, out);
            (void)hipMemcpy(h, out, 24, hipMemcpyDeviceToHost);
            printf("quad_encrypt n=%7u: %7.1f cycles/block (%5.1f per round)  %6.1f ns/block  -> %4.0f MHz\n", n,
                   (double)h[0] / n, (double)h[0] / n / 10, (double)h[1] * 10.0 / n, (double)h[0] / ((double)h[1] * 10.0) * 1e3);
        }
    return 0;
}
