/* bs_aes.h -- bitsliced AES (experiment, tools/bitslice): 32 blocks per 32-bit word, 128 bit
 * planes st[byte][bit]; one round column = generated 3-input-LUT code (bs_aes_gen.h).
 * Compiles for gfx950 (v_bitop3_b32) and for the host (emulated LUT) from the same source. */
#ifndef BS_AES_H_
#define BS_AES_H_
#include <stdint.h>
typedef uint32_t u32;

#if defined(__HIPCC__)
#define BS_FN __host__ __device__ __forceinline__
#else
#define BS_FN static inline
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define BS_LUT(a, b, c, tt) __builtin_amdgcn_bitop3_b32((a), (b), (c), (tt))
#else
BS_FN u32 bs_lut_emul(u32 a, u32 b, u32 c, unsigned tt)
{
    u32 r = 0;
    for (int i = 0; i < 8; ++i)
        if ((tt >> i) & 1) r |= ((i & 4) ? a : ~a) & ((i & 2) ? b : ~b) & ((i & 1) ? c : ~c);
    return r;
}
#define BS_LUT(a, b, c, tt) bs_lut_emul((a), (b), (c), (tt))
#endif

#include "bs_aes_gen.h"

/* kp: key planes, (NR + 1) x 128 words; plane of round r, column c, row q, bit i at
 * kp[128 r + 32 c + 8 q + i] = all-ones iff that key bit is set                       */
template <int NR>
BS_FN void bs_encrypt(u32 (&st)[16][8], const u32 *kp)
{
#pragma unroll
    for (int b = 0; b < 16; ++b)
#pragma unroll
        for (int i = 0; i < 8; ++i) st[b][i] ^= kp[32 * (b >> 2) + 8 * (b & 3) + i];
#pragma unroll
    for (int r = 1; r <= NR; ++r) {
        u32 nw[16][8];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const u32 *a0 = st[4 * c], *a1 = st[4 * ((c + 1) & 3) + 1], *a2 = st[4 * ((c + 2) & 3) + 2],
                      *a3 = st[4 * ((c + 3) & 3) + 3];
            if (r < NR) bs_col_full(a0, a1, a2, a3, kp + 128 * r + 32 * c, &nw[4 * c][0]);
            else bs_col_last(a0, a1, a2, a3, kp + 128 * r + 32 * c, &nw[4 * c][0]);
        }
#pragma unroll
        for (int b = 0; b < 16; ++b)
#pragma unroll
            for (int i = 0; i < 8; ++i) st[b][i] = nw[b][i];
    }
}

/* 32 x 32 bit transpose in place: on entry a[j] = plane of bit j (bit k = block k), on exit
 * a[k] = the word of block k (bit j = plane j's bit k)                                   */
BS_FN void bs_transpose32(u32 (&a)[32])
{
    u32 m = 0x0000ffffu;
#pragma unroll
    for (int j = 16; j != 0; j >>= 1, m ^= (m << j)) {
#pragma unroll
        for (int k = 0; k < 32; k = (k + j + 1) & ~j) {
            const u32 t = (a[k] ^ (a[k + j] << j)) & ~m;     /* upper part of a[k] <-> lower part of a[k+j] */
            const u32 t2 = ((a[k] >> j) ^ a[k + j]) & m;
            a[k] ^= t;
            a[k + j] ^= t2;
        }
    }
}
#endif
