// Experiment: a complete BITSLICED AES-128-CTR kernel for gfx950 --
// no table, no LDS: 32 blocks per lane as 128 bit planes, the round as generated 3-input-LUT code
// (bs_aes_gen.h: Boyar-Peralta S-box + MixColumns + AddRoundKey mapped onto v_bitop3_b32),
// counter planes built directly (no input transpose), keystream transposed back in registers.
// Output is compared with the product's table-driven CTR path, then both are timed on 1 GiB.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../micro-aes_amd/csrc/uaes_ctr.hip.h"
#include "bs_aes.h"

// ctr: 56-bit counter v0 must be a multiple of 2048 (the experiment handles whole chunks only)
__global__ __launch_bounds__(256, 2) void k_bs_ctr(uaesk_ctr ctr, const u32 *__restrict__ kp,
                                                   const uint4 *in, uint4 *out, u64 nchunks)
{
    const u32 lane = threadIdx.x & 63u;
    const u64 wave = (u64)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const u64 nwaves = (u64)gridDim.x * (blockDim.x >> 6);
    for (u64 ch = wave; ch < nchunks; ch += nwaves) {
        const u64 v = (ctr.v0 + ch * 2048ull) & 0x00FFFFFFFFFFFFFFull;      /* low 11 bits are zero */
        u32 st[16][8];
        const u32 fixed[3] = { ctr.w0, ctr.w1, ctr.b8 };
#pragma unroll
        for (int b = 0; b < 9; ++b)
#pragma unroll
            for (int i = 0; i < 8; ++i) st[b][i] = 0u - ((fixed[b >> 2] >> (8 * (b & 3) + i)) & 1u);
#pragma unroll
        for (int b = 9; b < 16; ++b)                     /* counter byte b = bits 8(15-b).. of v (big endian) */
#pragma unroll
            for (int i = 0; i < 8; ++i) st[b][i] = 0u - (u32)((v >> (8 * (15 - b) + i)) & 1ull);
        /* the 11 low counter bits: block k*64 + lane of the chunk sits in bit k of every plane */
#pragma unroll
        for (int i = 0; i < 6; ++i) st[15][i] = 0u - ((lane >> i) & 1u);
        st[15][6] = 0xAAAAAAAAu; st[15][7] = 0xCCCCCCCCu;
        st[14][0] = 0xF0F0F0F0u; st[14][1] = 0xFF00FF00u; st[14][2] = 0xFFFF0000u;

        bs_encrypt<10>(st, kp);

        const u64 base = ch * 2048ull + lane;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            u32 a[32];
#pragma unroll
            for (int j = 0; j < 32; ++j) a[j] = st[4 * w + (j >> 3)][j & 7];
            bs_transpose32(a);
#pragma unroll
            for (int j = 0; j < 32; ++j) st[4 * w + (j >> 3)][j & 7] = a[j];       /* now: word w of block j */
        }
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            const uint4 d = in[base + 64ull * k];
            out[base + 64ull * k] = make_uint4(d.x ^ st[0 + (k >> 3)][k & 7], d.y ^ st[4 + (k >> 3)][k & 7],
                                               d.z ^ st[8 + (k >> 3)][k & 7], d.w ^ st[12 + (k >> 3)][k & 7]);
        }
    }
}

// the product's generic table-driven path, as the reference for the comparison
__global__ __launch_bounds__(UAES_WG) void k_ref_ctr(uaesk_rk rk, const u32 *__restrict__ te0, uaesk_ctr ctr,
                                                     const uint4 *in, uint4 *out, u64 nblk)
{
    fill_enc_tables(te0);
    const LaneConst lc = make_lane_const();
    for (u64 i = (u64)blockIdx.x * UAES_WG + threadIdx.x; i < nblk; i += (u64)gridDim.x * UAES_WG)
        (void)ctr_one_block<10>(rk, ctr, in, out, i, lc);
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
static unsigned xt(unsigned a) { return ((a << 1) ^ ((a >> 7) * 0x1b)) & 0xff; }

int main(int argc, char **argv)
{
    make_sbox();
    u32 te0[256];
    for (int x = 0; x < 256; ++x) {
        const unsigned s = sbox[x], s2 = xt(s), s3 = s2 ^ s;
        te0[x] = s2 | (s << 8) | (s << 16) | (s3 << 24);
    }
    // AES-128 key schedule for key 00 01 .. 0f (LE words of the byte stream)
    uaesk_rk rk; memset(&rk, 0, sizeof rk);
    unsigned char kb[176];
    for (int i = 0; i < 16; ++i) kb[i] = (unsigned char)i;
    unsigned rcon = 1;
    for (int i = 16; i < 176; i += 4) {
        unsigned char t[4] = { kb[i - 4], kb[i - 3], kb[i - 2], kb[i - 1] };
        if (i % 16 == 0) {
            const unsigned char u = t[0];
            t[0] = sbox[t[1]] ^ (unsigned char)rcon; t[1] = sbox[t[2]]; t[2] = sbox[t[3]]; t[3] = sbox[u];
            rcon = xt(rcon);
        }
        for (int j = 0; j < 4; ++j) kb[i + j] = kb[i - 16 + j] ^ t[j];
    }
    memcpy(rk.w, kb, 176);
    static u32 kp[11 * 128];
    for (int r = 0; r <= 10; ++r)
        for (int c = 0; c < 4; ++c)
            for (int q = 0; q < 4; ++q)
                for (int i = 0; i < 8; ++i) kp[128 * r + 32 * c + 8 * q + i] = 0u - (u32)((kb[16 * r + 4 * c + q] >> i) & 1);
    uaesk_ctr ctr; memset(&ctr, 0, sizeof ctr);
    ctr.w0 = 0xf3f2f1f0u; ctr.w1 = 0xf7f6f5f4u; ctr.b8 = 0xf8; ctr.v0 = 0x00f9fafb00000800ull;   /* multiple of 2048 */

    const u64 nblk = 1ull << 26, nchunks = nblk / 2048;
    u32 *d_te0, *d_kp; uint4 *in, *out, *ref;
    (void)hipMalloc(&d_te0, 1024); (void)hipMemcpy(d_te0, te0, 1024, hipMemcpyHostToDevice);
    (void)hipMalloc(&d_kp, sizeof kp); (void)hipMemcpy(d_kp, kp, sizeof kp, hipMemcpyHostToDevice);
    (void)hipMalloc(&in, nblk * 16); (void)hipMalloc(&out, nblk * 16); (void)hipMalloc(&ref, nblk * 16);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (u64 *)in, nblk * 2);
    (void)hipFuncSetAttribute((const void *)k_ref_ctr, hipFuncAttributeMaxDynamicSharedMemorySize, UAES_LDS_ENC);
    hipLaunchKernelGGL(k_ref_ctr, dim3(256), dim3(UAES_WG), UAES_LDS_ENC, 0, rk, d_te0, ctr, in, ref, nblk);
    (void)hipDeviceSynchronize();

    hipFuncAttributes fa; (void)hipFuncGetAttributes(&fa, (const void *)k_bs_ctr);
    printf("k_bs_ctr: %d VGPRs, %zu B scratch per lane\n", fa.numRegs, (size_t)fa.localSizeBytes);
    const int grids[] = { 512, 1024, 2048 };
    for (int gi = 0; gi < 3; ++gi) {
        const unsigned grid = grids[gi];
        (void)hipMemset(out, 0, nblk * 16);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_bs_ctr, dim3(grid), dim3(256), 0, 0, ctr, d_kp, in, out, nchunks);
        (void)hipEventRecord(e0);
        const int reps = 20;
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_bs_ctr, dim3(grid), dim3(256), 0, 0, ctr, d_kp, in, out, nchunks);
        (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
        unsigned long long *d_bad, bad = 0;
        (void)hipMalloc(&d_bad, 8); (void)hipMemset(d_bad, 0, 8);
        hipLaunchKernelGGL(k_cmp, dim3(4096), dim3(256), 0, 0, out, ref, nblk, d_bad);
        (void)hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost);
        printf("bitsliced AES-128-CTR, 1 GiB, grid %4u x 256: %.4f ms  %7.1f GiB/s  mismatching blocks vs the table path: %llu  %s\n",
               grid, ms, (double)nblk * 16 / (ms * 1e-3) / (1 << 30), bad, hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
