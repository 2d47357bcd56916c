// Ablation (diagnostic): the ECB-128 encrypt loop of k_ecb with its global loads and/or
// stores removed, to see how much of the gap to the LDS lookup rate is the vector-memory
// path (TA/TD address + data return) competing with the LDS pipe.
//   MODE bit 0: no load, bit 1: no store, bit 2: nontemporal load, bit 3: nontemporal store.
// Each run also reports the shader clock (s_memtime cycles of one wave / wall time).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../micro-aes_amd/csrc/uaes_aes.hip.h"
#define UAES_U 4
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(UAES_WG) void k(uaesk_rk rk, const u32 *__restrict__ te0,
                                             const uint4 *__restrict__ in, uint4 *__restrict__ out, u64 nfull,
                                             u64 *cyc)
{
    const u64 t0 = __builtin_readcyclecounter();
    fill_enc_tables(te0);
    const LaneConst lc = make_lane_const();
    const u64 stride = (u64)gridDim.x * UAES_WG * UAES_U;
    u32 acc[4] = { 0, 0, 0, 0 };
    for (u64 base = (u64)blockIdx.x * UAES_WG * UAES_U; base < nfull; base += stride) {
        u32 s[UAES_U][4];
        u64 idx[UAES_U];
#pragma unroll
        for (int u = 0; u < UAES_U; ++u) {
            idx[u] = base + (u64)u * UAES_WG + threadIdx.x;
            uint4 d;
            if (MODE & 1) d = make_uint4((u32)idx[u], (u32)(idx[u] >> 32) ^ acc[0], acc[1] + u, acc[2]);
            else if (MODE & 4) {
                const u32x4 v = __builtin_nontemporal_load((const u32x4 *)&in[idx[u]]);
                d = make_uint4(v[0], v[1], v[2], v[3]);
            } else d = in[idx[u]];
            s[u][0] = d.x; s[u][1] = d.y; s[u][2] = d.z; s[u][3] = d.w;
        }
        enc_blocks_skewed<10>(s[0], s[1], rk, lc);
        enc_blocks_skewed<10>(s[2], s[3], rk, lc);
#pragma unroll
        for (int u = 0; u < UAES_U; ++u) {
            if (MODE & 2) { acc[0] ^= s[u][0]; acc[1] ^= s[u][1]; acc[2] ^= s[u][2]; acc[3] ^= s[u][3]; }
            else if (MODE & 8) {
                u32x4 v = { s[u][0], s[u][1], s[u][2], s[u][3] };
                __builtin_nontemporal_store(v, (u32x4 *)&out[idx[u]]);
            } else out[idx[u]] = make_uint4(s[u][0], s[u][1], s[u][2], s[u][3]);
        }
    }
    if ((MODE & 2) && (acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u)
        out[threadIdx.x] = make_uint4(acc[0], acc[1], acc[2], acc[3]);
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = __builtin_readcyclecounter() - t0;
}

__global__ void k_fill(u64 *p, u64 nwords)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (u64)gridDim.x * blockDim.x) {
        u64 z = (i + 1) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        p[i] = z ^ (z >> 31);
    }
}

static int g_reps = 40;
static unsigned char sbox[256];
static void make_sbox()
{
    unsigned char p = 1, q = 1;
    do {
        p = p ^ (unsigned char)(p << 1) ^ ((p & 0x80) ? 0x1B : 0);
        q ^= q << 1; q ^= q << 2; q ^= q << 4; if (q & 0x80) q ^= 0x09;
        unsigned char x = q ^ (unsigned char)((q << 1) | (q >> 7)) ^ (unsigned char)((q << 2) | (q >> 6)) ^
                          (unsigned char)((q << 3) | (q >> 5)) ^ (unsigned char)((q << 4) | (q >> 4));
        sbox[p] = x ^ 0x63;
    } while (p != 1);
    sbox[0] = 0x63;
}

template <int MODE>
static void run(const char *name, const uaesk_rk &rk, const u32 *te0, const uint4 *in, uint4 *out, u64 nblk)
{
    (void)hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, UAES_LDS_ENC);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    u64 *d_cyc, h_cyc = 0; (void)hipMalloc(&d_cyc, 8);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(UAES_WG), UAES_LDS_ENC, 0, rk, te0, in, out, nblk, d_cyc);
    (void)hipEventRecord(e0);
    const int reps = g_reps;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(UAES_WG), UAES_LDS_ENC, 0, rk, te0, in, out, nblk, d_cyc);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    (void)hipMemcpy(&h_cyc, d_cyc, 8, hipMemcpyDeviceToHost);
    printf("%-22s %.4f ms per GiB-launch  %7.1f GiB/s   wave0 %.0f kcycles -> %.0f MHz, %.2f clk per block per CU\n", name, ms,
           (double)nblk * 16 / (ms * 1e-3) / (1 << 30), h_cyc / 1e3, h_cyc / (ms * 1e3),
           (double)h_cyc * 256 / nblk);
}

int main(int argc, char **argv)
{
    make_sbox();
    u32 te0[256];
    for (int x = 0; x < 256; ++x) {
        const unsigned s = sbox[x], s2 = ((s << 1) ^ ((s >> 7) * 0x1b)) & 0xff, s3 = s2 ^ s;
        te0[x] = s2 | (s << 8) | (s << 16) | (s3 << 24);
    }
    uaesk_rk rk; for (int i = 0; i < 60; ++i) rk.w[i] = 0x9e3779b9u * (i + 1);
    const u64 nblk = 1ull << 26;
    u32 *d_te0; uint4 *in, *out;
    (void)hipMalloc(&d_te0, 1024); (void)hipMemcpy(d_te0, te0, 1024, hipMemcpyHostToDevice);
    (void)hipMalloc(&in, nblk * 16); (void)hipMalloc(&out, nblk * 16);
    (void)hipMemset(in, 0x5a, nblk * 16);
    if (argc > 1) {                                  /* random input instead of a constant fill */
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (u64 *)in, nblk * 2);
        (void)hipDeviceSynchronize();
        printf("input: splitmix64 stream\n");
    } else printf("input: constant 0x5a bytes\n");
    if (argc > 3) {                                  /* membench random <mode> <reps>: one long run (clock polling) */
        g_reps = atoi(argv[3]);
        switch (atoi(argv[2])) {
        case 0: run<0>("load + store", rk, d_te0, in, out, nblk); break;
        case 3: run<3>("no load, no store", rk, d_te0, in, out, nblk); break;
        case 1: run<1>("no load", rk, d_te0, in, out, nblk); break;
        default: run<2>("no store", rk, d_te0, in, out, nblk); break;
        }
        return 0;
    }
    for (int pass = 0; pass < 3; ++pass) {           /* the first pass also warms the clocks up */
        printf("pass %d\n", pass);
        run<0>("load + store", rk, d_te0, in, out, nblk);
        run<12>("nt load + nt store", rk, d_te0, in, out, nblk);
        run<1>("no load", rk, d_te0, in, out, nblk);
        run<2>("no store", rk, d_te0, in, out, nblk);
        run<3>("no load, no store", rk, d_te0, in, out, nblk);
        run<4>("nt load + store", rk, d_te0, in, out, nblk);
        run<8>("load + nt store", rk, d_te0, in, out, nblk);
        run<0>("load + store", rk, d_te0, in, out, nblk);
    }
    return 0;
}
