// Experiment (diagnostic): the 64 KiB "split halves" table layout (LaneConst2 of
// uaes_aes.hip.h) against the 128 KiB layout, on the plain ECB-128 encrypt loop.
// Question: do two 16-wave workgroups per CU (32 waves) overlap the LDS pipe and the
// VALU better than one, and what does the per-lane key delta of the odd rounds cost?
// Every variant's output is compared with variant A's (and A was pinned by the suite).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../micro-aes_amd/csrc/uaes_aes.hip.h"

// V: 0 = 128 KiB layout, 4 blocks/lane (two skewed pairs)    [the product k_ecb]
//    1 = 64 KiB layout, 1 block/lane lock-step
//    2 = 64 KiB layout, 2 blocks/lane skewed
//    3 = 64 KiB layout, 4 blocks/lane (two skewed pairs)
//    4 = 64 KiB layout, 2 blocks/lane lock-step
//    5 = 128 KiB layout, decrypt of the output (round trip check only)
//    6 = 64 KiB layout, decrypt, 2 blocks/lane skewed
template <int V, int WPE>
__global__ __launch_bounds__(UAES_WG, WPE) void k(uaesk_rk rk, const u32 *__restrict__ t0,
                                                  const uint4 *__restrict__ in, uint4 *__restrict__ out, u64 nfull)
{
    constexpr int U = (V == 0 || V == 3 || V == 5) ? 4 : (V == 1 ? 1 : 2);
    const u64 stride = (u64)gridDim.x * UAES_WG * U;
    if (V == 0 || V == 5) {
        fill_enc_tables(t0);
        const LaneConst lc = make_lane_const();
        for (u64 base = (u64)blockIdx.x * UAES_WG * U; base < nfull; base += stride) {
            u32 s[U][4];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint4 d = in[base + (u64)u * UAES_WG + threadIdx.x];
                s[u][0] = d.x; s[u][1] = d.y; s[u][2] = d.z; s[u][3] = d.w;
            }
            if (V == 0) {
                enc_blocks_skewed<10>(s[0], s[1], rk, lc);
                enc_blocks_skewed<10>(s[2 % U], s[3 % U], rk, lc);
            } else {
                dec_blocks<10, U>(s, rk, lc);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                out[base + (u64)u * UAES_WG + threadIdx.x] = make_uint4(s[u][0], s[u][1], s[u][2], s[u][3]);
        }
    } else {
        fill_tables64(t0);
        const LaneConst2 lc = make_lane_const2();
        for (u64 base = (u64)blockIdx.x * UAES_WG * U; base < nfull; base += stride) {
            u32 s[U][4];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint4 d = in[base + (u64)u * UAES_WG + threadIdx.x];
                s[u][0] = d.x; s[u][1] = d.y; s[u][2] = d.z; s[u][3] = d.w;
            }
            if (V == 1 || V == 4) enc_blocks<10, U>(s, rk, lc);
            else if (V == 2) enc_blocks_skewed<10>(s[0], s[1 % U], rk, lc);
            else if (V == 3) { enc_blocks_skewed<10>(s[0], s[1 % U], rk, lc); enc_blocks_skewed<10>(s[2 % U], s[3 % U], rk, lc); }
            else dec_blocks<10, U>(s, rk, lc);
#pragma unroll
            for (int u = 0; u < U; ++u)
                out[base + (u64)u * UAES_WG + threadIdx.x] = make_uint4(s[u][0], s[u][1], s[u][2], s[u][3]);
        }
    }
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

__global__ void k_cmp(const uint4 *a, const uint4 *b, u64 n, unsigned long long *bad)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const uint4 x = a[i], y = b[i];
        if (x.x != y.x || x.y != y.y || x.z != y.z || x.w != y.w) atomicAdd(bad, 1ull);
    }
}

static unsigned char sbox[256], isbox[256];
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
    for (int i = 0; i < 256; ++i) isbox[sbox[i]] = (unsigned char)i;
}
static unsigned xt(unsigned a) { return ((a << 1) ^ ((a >> 7) * 0x1b)) & 0xff; }

static unsigned long long compare(const uint4 *a, const uint4 *b, u64 n)
{
    unsigned long long *d_bad, h = 0;
    (void)hipMalloc(&d_bad, 8); (void)hipMemset(d_bad, 0, 8);
    hipLaunchKernelGGL(k_cmp, dim3(4096), dim3(256), 0, 0, a, b, n, d_bad);
    (void)hipMemcpy(&h, d_bad, 8, hipMemcpyDeviceToHost);
    (void)hipFree(d_bad);
    return h;
}

template <int V, int WPE>
static void run(const char *name, int wg_per_cu, const uaesk_rk &rk, const u32 *t0, const uint4 *in, uint4 *out,
                u64 nblk, const uint4 *expect)
{
    const unsigned lds = (V == 0 || V == 5) ? UAES_LDS_ENC : UAES_LDS_T64;
    (void)hipFuncSetAttribute((const void *)k<V, WPE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const unsigned grid = 256 * wg_per_cu;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<V, WPE>), dim3(grid), dim3(UAES_WG), lds, 0, rk, t0, in, out, nblk);
    (void)hipEventRecord(e0);
    const int reps = 40;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<V, WPE>), dim3(grid), dim3(UAES_WG), lds, 0, rk, t0, in, out, nblk);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    hipError_t err = hipGetLastError();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    hipFuncAttributes fa; (void)hipFuncGetAttributes(&fa, (const void *)k<V, WPE>);
    unsigned long long bad = expect ? compare(out, expect, nblk) : 0;
    printf("%-44s wg/CU=%d vgpr=%3d  %.4f ms  %7.1f GiB/s  mismatching blocks: %llu %s\n", name, wg_per_cu, fa.numRegs, ms,
           (double)nblk * 16 / (ms * 1e-3) / (1 << 30), bad, err == hipSuccess ? "" : hipGetErrorString(err));
}

int main()
{
    make_sbox();
    u32 te0[256], td0[256];
    for (int x = 0; x < 256; ++x) {
        const unsigned s = sbox[x], s2 = xt(s), s3 = s2 ^ s;
        te0[x] = s2 | (s << 8) | (s << 16) | (s3 << 24);
        const unsigned v = isbox[x], v2 = xt(v), v4 = xt(v2), v8 = xt(v4);
        td0[x] = (v8 ^ v4 ^ v2) | ((v8 ^ v) << 8) | ((v8 ^ v4 ^ v) << 16) | ((v8 ^ v2 ^ v) << 24);
    }
    // encryption keys: arbitrary words (any 44 words define a valid round structure);
    // decryption keys for the round-trip check: dk[0]=ek[nr], dk[i]=InvMixColumns(ek[nr-i]), dk[nr]=ek[0]
    uaesk_rk rk, dk;
    for (int i = 0; i < 60; ++i) rk.w[i] = 0x9e3779b9u * (i + 1);
    memset(&dk, 0, sizeof dk);
    for (int c = 0; c < 4; ++c) { dk.w[c] = rk.w[40 + c]; dk.w[40 + c] = rk.w[c]; }
    for (int i = 1; i < 10; ++i)
        for (int c = 0; c < 4; ++c) {
            const u32 w = rk.w[4 * (10 - i) + c];
            u32 r = 0;
            for (int b = 0; b < 4; ++b) {
                const u32 t = td0[sbox[(w >> (8 * b)) & 0xff]];
                r ^= (t << (8 * b)) | (b ? t >> (32 - 8 * b) : 0);
            }
            dk.w[4 * i + c] = r;
        }
    const u64 nblk = 1ull << 26;
    u32 *d_te0, *d_td0; uint4 *in, *out, *ref;
    (void)hipMalloc(&d_te0, 1024); (void)hipMemcpy(d_te0, te0, 1024, hipMemcpyHostToDevice);
    (void)hipMalloc(&d_td0, 1024); (void)hipMemcpy(d_td0, td0, 1024, hipMemcpyHostToDevice);
    (void)hipMalloc(&in, nblk * 16); (void)hipMalloc(&out, nblk * 16); (void)hipMalloc(&ref, nblk * 16);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (u64 *)in, nblk * 2);
    (void)hipDeviceSynchronize();
    for (int pass = 0; pass < 3; ++pass) {
        printf("pass %d\n", pass);
        run<0, 4>("A  128K, 4 blk/lane skewed pairs", 1, rk, d_te0, in, ref, nblk, nullptr);
        run<1, 8>("B1 64K, 1 blk/lane", 2, rk, d_te0, in, out, nblk, ref);
        run<4, 8>("B4 64K, 2 blk/lane lock-step", 2, rk, d_te0, in, out, nblk, ref);
        run<2, 8>("B2 64K, 2 blk/lane skewed (<=64 vgpr)", 2, rk, d_te0, in, out, nblk, ref);
        run<2, 4>("B2 64K, 2 blk/lane skewed, 1 WG/CU", 1, rk, d_te0, in, out, nblk, ref);
        run<3, 4>("B3 64K, 4 blk/lane skewed pairs, 1 WG/CU", 1, rk, d_te0, in, out, nblk, ref);
        run<1, 4>("B1 64K, 1 blk/lane, 1 WG/CU", 1, rk, d_te0, in, out, nblk, ref);
        run<0, 4>("A  128K again", 1, rk, d_te0, in, out, nblk, ref);
    }
    // round trips: decrypt the reference ciphertext with both layouts, expect the input back
    run<5, 4>("D0 128K decrypt (round trip)", 1, dk, d_td0, ref, out, nblk, in);
    run<6, 8>("D1 64K decrypt 2 blk/lane skewed", 2, dk, d_td0, ref, out, nblk, in);
    run<6, 4>("D1 64K decrypt, 1 WG/CU", 1, dk, d_td0, ref, out, nblk, in);
    return 0;
}
