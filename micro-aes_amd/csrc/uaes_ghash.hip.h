/*
 * uaes_ghash.hip.h -- the GHASH machinery the GCM-family translation units share (uaes_gcm.hip, uaes_gcm_records.hip,
 * uaes_siv.hip): the scratch layout (GS_*), the source descriptor of a GHASH input sequence (GSrc), the table
 * multiplications (64 KiB byte tables through ds_read_b128, 8 KiB nibble tables), the radix-4 tree of the last levels
 * (gh_tree), the wave-cooperative field multiplication, and the nibble tables of a key made inside a kernel.
 * Device helpers and constants only -- no kernels, no host state.  mulGF128: micro_aes.c:476-493; gHash: :1127-1137.
 */
#ifndef UAES_GHASH_HIP_H_
#define UAES_GHASH_HIP_H_

#include <hip/hip_runtime.h>
#include <string.h>
#include <stdlib.h>
#include "uaes_aes.hip.h"
#include "uaes_ctr.hip.h"
#include "uaes_gf.h"
#include "uaes_device.h"
#include "uaes_plan.h"

#define GH_T        1024u           /* threads of the last-levels workgroup   */
#define GH_PT       1024u           /* threads per bulk-level workgroup       */
#define GH_MAXLOG   18u             /* largest bulk stride 2^18 (256 x 1024)  */
#define GH_LOGB     14u             /* second-level stride 2^14 (16 x 1024)   */
#define GH_DIRECT   32768u          /* <= this many blocks: last kernel alone */

/* scratch layout (bytes) */
#define GS_H        0u
#define GS_EJ0      16u
#define GS_POW      64u                         /* 18 x 16                    */
#define GS_TAB4     1024u                       /* six 8 KiB nibble tables: H^1024, H^256, H^64, H^16, H^4, H */
#define GT_NTAB     6u
#define GS_TAB8_A   (GS_TAB4 + GT_NTAB * 8192u) /* H^(2^logA), 64 KiB         */
#define GS_TAB8_B   (GS_TAB8_A + 65536u)        /* H^(2^14), 64 KiB           */
#define GS_ACC1     (GS_TAB8_B + 65536u)        /* 2^17 x 16 = 2 MiB          */
#define GS_ACC2     (GS_ACC1 + (16u << GH_MAXLOG))
#define GS_POW64    (GS_ACC2 + (16u << GH_LOGB))   /* H^(2^k), k = 0..63 (sharded GCM) */
#define GS_PART     (GS_POW64 + 1024u)              /* raw GHASH of a shard             */
#define GS_RUN      (GS_PART + 16u)                 /* running GHASH of a streamed message */
#define GS_SMALL    (GS_PART + 64u)                 /* everything the streamed / sharded paths need */
#define GF_MAXLOG   20u                             /* fused encrypt: lanes of the whole grid, 2048 per workgroup */
#define GS_TAB8_F   GS_SMALL                        /* H^(2048 * workgroups), 64 KiB (fused encrypt) */
#define GS_YLO      (GS_TAB8_F + 65536u)            /* Y^0..Y^15, Y = H^2048 (workgroup weights)     */
#define GS_ZHI      (GS_YLO + 256u)                 /* Z^0..Z^15, Z = Y^16                           */
#define GS_T        (GS_ZHI + 256u)                 /* XOR of the workgroups' weighted partial hashes */
#define GS_YTAB     (GS_T + 64u)                    /* key contexts: the combine kernel's nibble tables of Y^256 .. Y, 40 KiB
                                                       each for Y = H^1024, H^2048 .. H^131072 (k_gcm_ytables)                */
#define GS_YTAB_SET (5u * 8192u)
#define GS_YTAB_SETS 8u                         /* Y = H^1024 .. H^131072 (GMC_MAXLOGSTEPS + 1) */
#define GS_SIV      (GS_YTAB + GS_YTAB_SETS * GS_YTAB_SET)   /* a long GCM-SIV message's per-nonce values, made and used on the device: */
#define GS_SIV_RK   GS_SIV                          /*   the message-encryption key's schedule (uaesk_rk, 240 B) */
#define GS_SIV_HG   (GS_SIV + 256u)                 /*   the POLYVAL key in GHASH form                           */
#define GS_SIV_PV   (GS_SIV + 272u)                 /*   the raw hash                                            */
#define GS_SIV_CTR  (GS_SIV + 288u)                 /*   the keystream's counter description (uaesk_ctr)         */
#define GS_TOTAL    (GS_SIV + 512u)

static inline hipStream_t S(void *s) { return (hipStream_t)s; }

/* the sequence of GHASH input blocks: AAD blocks, CT blocks, length block */
struct GSrc {
    const unsigned char *aad;
    u64 aad_len;
    const unsigned char *ct;      /* 16-byte aligned */
    u64 ct_len;
    u32 has_len;
    u64 len_aad, len_ct;          /* byte lengths written into the length block (totals of the
                                     whole message when this is one shard of it) */
    u32 rev;                      /* POLYVAL: every input block byte-reversed, LE length block */
};

__device__ __forceinline__ uint4 load_bytes_padded(const unsigned char *p, u64 avail)
{
    u32 w[4] = { 0, 0, 0, 0 };
    const u32 n = avail < 16 ? (u32)avail : 16u;
    for (u32 i = 0; i < n; ++i) w[i >> 2] |= (u32)p[i] << (8 * (i & 3));
    return make_uint4(w[0], w[1], w[2], w[3]);
}

__device__ __forceinline__ uint4 rev16(uint4 b)
{
    return make_uint4(bswap32(b.w), bswap32(b.z), bswap32(b.y), bswap32(b.x));
}

__device__ __forceinline__ uint4 load_vblock_fwd(const GSrc &s, u64 v)
{
    const u64 ab = (s.aad_len + 15) >> 4, cb = (s.ct_len + 15) >> 4;
    if (v < ab) {                                   /* whole blocks of 16-byte aligned AAD: one load (GMAC of a bulk text) */
        if ((v + 1) * 16 <= s.aad_len && (((uintptr_t)s.aad) & 15u) == 0) return ((const uint4 *)s.aad)[v];
        return load_bytes_padded(s.aad + v * 16, s.aad_len - v * 16);
    }
    v -= ab;
    if (v < cb) {
        if ((v + 1) * 16 <= s.ct_len) return ((const uint4 *)s.ct)[v];
        return load_bytes_padded(s.ct + v * 16, s.ct_len - v * 16);
    }
    const u64 abits = s.len_aad * 8, cbits = s.len_ct * 8;      /* N6 */
    if (s.rev)                                                  /* POLYVAL: two little-endian 64-bit lengths */
        return make_uint4((u32)abits, (u32)(abits >> 32), (u32)cbits, (u32)(cbits >> 32));
    return make_uint4(bswap32((u32)(abits >> 32)), bswap32((u32)abits),
                      bswap32((u32)(cbits >> 32)), bswap32((u32)cbits));
}

__device__ __forceinline__ uint4 load_vblock(const GSrc &s, u64 v)
{
    const uint4 b = load_vblock_fwd(s, v);
    return s.rev ? rev16(b) : b;
}

__device__ __forceinline__ uint4 x4(uint4 a, uint4 b)
{
    return make_uint4(a.x ^ b.x, a.y ^ b.y, a.z ^ b.z, a.w ^ b.w);
}

/* a * M through the byte-indexed table of M held in LDS, bank-conflict free.
 *
 * Layout: row v (256 B) holds the sixteen 16-byte entries Tab_j[v], j = slot.
 * ds_read_b128 is serviced in four 16-lane groups and a 16-byte slot is
 * (addr/16) mod 16 (MI355X_MICROARCH.md, LDS), so a data-dependent row is free
 * but two lanes of a group must not share a SLOT.  Lane l therefore walks the
 * 16 bytes of its block in its own order -- byte g ^ t at step t, where g =
 * position of the lane inside its service group -- so at every step the 16
 * lanes of a group read 16 different tables = 16 different slots.  (A naive
 * [j][v] layout puts the slot at v mod 16: random, ~2.9x serialisation.)
 *
 * Why g ^ t and not (g + t) mod 16 (rounds 2-4): XOR has no carries.  The WORD a lane needs at step t is word
 * (g >> 2) ^ (t >> 2) -- the accumulator's words permuted once per product by two levels of v_cndmask -- and the BYTE
 * inside it is (g & 3) ^ (t & 3), which the lane's own v_perm selector (a VGPR, one per t & 3) picks: 8 VALU per
 * product where rotating the block by g bytes took 12 (8 selects + 4 v_alignbyte).  1 GiB GCM 1262 -> 1268 GiB/s
 * (profiles/r05_gcm_rotate_ab.log, which also says why the other 8 cannot go).                                    */
struct GhLane {
    u32 so[4];          /* byte m of so[q] = slot offset (g ^ (4q + m)) << 4 */
    u32 sel[4];         /* v_perm selector of the steps with t & 3 = m: byte 1 <- data byte (g & 3) ^ m, byte 0 <- so byte m */
    u32 g;
};

/* position of lane (l & 31) inside its ds_read_b128 service group:
 * groups {0-3,12-15,20-27} and {4-11,16-19,28-31} (and the same +32)      */
__device__ __forceinline__ u32 b128_group_pos(u32 tid = threadIdx.x)
{
    const u32 l = tid & 31u;
    const u64 pack_lo = 0x7654765432103210ull;     /* lanes 0..15  */
    const u64 pack_hi = 0xfedcfedcba98ba98ull;     /* lanes 16..31 */
    return (u32)(((l & 16u) ? pack_hi : pack_lo) >> (4 * (l & 15u))) & 15u;
}

__device__ __forceinline__ GhLane gh_lane_setup()
{
    GhLane gl;
    gl.g = b128_group_pos();
#pragma unroll
    for (u32 q = 0; q < 4; ++q) {
        u32 v = 0;
#pragma unroll
        for (u32 m = 0; m < 4; ++m) v |= ((gl.g ^ (4 * q + m)) << 4) << (8 * m);
        gl.so[q] = v;
    }
#pragma unroll
    for (u32 m = 0; m < 4; ++m) gl.sel[m] = 0x0c0c0000u | ((4u + ((gl.g & 3u) ^ m)) << 8) | m;
    return gl;
}

/* r[q] = word q ^ (g >> 2) of a */
__device__ __forceinline__ void gh_words(const uint4 a, u32 g, u32 (&r)[4])
{
    const bool w1 = (g & 4u) != 0, w2 = (g & 8u) != 0;
    const u32 t0 = w1 ? a.y : a.x, t1 = w1 ? a.x : a.y, t2 = w1 ? a.w : a.z, t3 = w1 ? a.z : a.w;
    r[0] = w2 ? t2 : t0; r[1] = w2 ? t3 : t1; r[2] = w2 ? t0 : t2; r[3] = w2 ? t1 : t3;
}

/* bytes of a rotated left by g: result byte k = a byte (k + g) mod 16 (the nibble-table products below) */
__device__ __forceinline__ void gh_rotate(const uint4 a, u32 g, u32 (&r)[4])
{
    const bool w1 = (g & 4u) != 0, w2 = (g & 8u) != 0;
    const u32 t0 = w1 ? a.y : a.x, t1 = w1 ? a.z : a.y, t2 = w1 ? a.w : a.z, t3 = w1 ? a.x : a.w;
    const u32 u0 = w2 ? t2 : t0, u1 = w2 ? t3 : t1, u2 = w2 ? t0 : t2, u3 = w2 ? t1 : t3;
    const u32 sh = g & 3u;
    r[0] = __builtin_amdgcn_alignbyte(u1, u0, sh);
    r[1] = __builtin_amdgcn_alignbyte(u2, u1, sh);
    r[2] = __builtin_amdgcn_alignbyte(u3, u2, sh);
    r[3] = __builtin_amdgcn_alignbyte(u0, u3, sh);
}

typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const u32x4 lds_cu128;

/* (a * M) ^ x; the table occupies LDS bytes [0, 65536) */
__device__ __forceinline__ uint4 tabmul8_xor(uint4 a, uint4 x, const GhLane &gl)
{
    u32 r[4];
    gh_words(a, gl.g, r);
    u32x4 e[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        /* address = (byte g ^ t of the block) << 8 | slot offset of step t */
        const u32 addr = __builtin_amdgcn_perm(r[t >> 2], gl.so[t >> 2], gl.sel[t & 3]);
        e[t] = *(lds_cu128 *)(uintptr_t)addr;
    }
    /* 17 inputs per dword -> 8 three-input XORs */
    u32 z[4] = { x.x, x.y, x.z, x.w };
#pragma unroll
    for (int w = 0; w < 4; ++w) {
#pragma unroll
        for (int t = 0; t < 16; t += 2) z[w] = xor3(z[w], e[t][w], e[t + 1][w]);
    }
    return make_uint4(z[0], z[1], z[2], z[3]);
}

/* ---- nibble-indexed tables: a * M = the XOR of 32 entries, one per nibble of a --------------------------------
 * Layout of one table (512 entries of 16 B = 8 KiB, anywhere in LDS): ROW v (512 B) holds the entries of nibble
 * VALUE v for the 32 nibble positions -- entry index nib_entry(p, v) = 32 v + slot(p), slot = j for the high
 * nibble of byte j (p = 2j), 16 + j for its low nibble (p = 2j + 1).  As with the byte tables above, the 16-byte
 * slot of an LDS address is (addr / 16) mod 16 and a ds_read_b128 serves 16 lanes at a time: the data-dependent
 * part of the address (the row) is free, the lanes of a service group only must not share a slot, so every lane
 * walks the bytes of its block in an order rotated by its position in the group and the sixteen lanes read
 * sixteen different positions at every step.  (Round 2's [position][value] layout put the VALUE into the slot:
 * random, ~2.9x serialisation of every multiplication.)                                                      */
__device__ __forceinline__ u32 nib_entry(u32 p, u32 v)
{
    return v * 32u + (p >> 1) + ((p & 1u) << 4);
}

/* a * M, every lane its own product */
/* (tid: a kernel that loops hands in an opaque copy of threadIdx.x, or the sixteen per-step slot constants are
 * hoisted out of its loop and spilled) */
template <int PARTS = 2>                                       /* 4: eight lookups in flight (32 registers) -- inside a loop
                                                                * that keeps a cipher's state and keys alive around it */
__device__ __forceinline__ uint4 tabmul4(const uint4 *T, uint4 a, u32 tid = threadIdx.x)
{
    const u32 g = b128_group_pos(tid);
    u32 r[4];
    gh_rotate(a, g, r);                                        /* byte k of r = byte (k + g) mod 16 of a */
    const u32 base = (u32)(uintptr_t)(__attribute__((address_space(3))) const uint4 *)T;
    /* two halves of sixteen lookups, the second not started before the first is folded: 64 instead of 128 registers
     * of entries in flight (a caller with live state around the product would spill it otherwise) */
    u32 z[4] = { 0, 0, 0, 0 };
    constexpr int NQ = 16 / PARTS;                             /* bytes per part */
#pragma unroll
    for (int h = 0; h < PARTS; ++h) {
        u32x4 e[2 * NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int k = NQ * h + q;
            const u32 c = base + (((g + (u32)k) & 15u) << 4);  /* slot of byte (k + g) mod 16 */
            e[2 * q] = *(lds_cu128 *)(uintptr_t)(c + (__builtin_amdgcn_ubfe(r[k >> 2], 8u * (k & 3) + 4u, 4u) << 9));
            e[2 * q + 1] = *(lds_cu128 *)(uintptr_t)(c + 256u + (__builtin_amdgcn_ubfe(r[k >> 2], 8u * (k & 3), 4u) << 9));
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            u32 t = xor3(z[d], e[0][d], e[1][d]);
#pragma unroll
            for (int k = 2; k < 2 * NQ; k += 2) t = xor3(t, e[k][d], e[k + 1][d]);
            z[d] = t;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    return make_uint4(z[0], z[1], z[2], z[3]);
}

/* The same product shared by the four lanes of a quad: in a dependent chain (the last levels)
 * what counts is the latency of one multiplication, and one lane alone issues 32 lookups, their
 * addresses and a 31-term XOR per dword.  Lane c of the quad takes word c of `a` (8 nibbles),
 * the quad XORs its four partial products with two DPP exchanges per dword; every lane returns
 * the full product.  `a` must be the same in the four lanes.  A service group is four whole quads:
 * quad number qd of the group starts its four bytes at byte qd, so that at every step the sixteen lanes
 * are at sixteen different bytes.                                                              */
__device__ __forceinline__ uint4 tabmul4q(const uint4 *T, uint4 a, u32 tid = threadIdx.x)
{
    const u32 c = tid & 3u, qd = b128_group_pos(tid) >> 2;
    const u32 w = c == 0 ? a.x : c == 1 ? a.y : c == 2 ? a.z : a.w;
    const u32 wr = __builtin_amdgcn_alignbit(w, w, 8u * qd);                /* byte s of wr = byte (s + qd) mod 4 of w */
    const u32 base = (u32)(uintptr_t)(__attribute__((address_space(3))) const uint4 *)T + c * 64u;   /* byte j = 4c + kk */
    u32x4 e[8];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const u32 ca = base + ((((u32)s + qd) & 3u) << 4);
        e[2 * s] = *(lds_cu128 *)(uintptr_t)(ca + (__builtin_amdgcn_ubfe(wr, 8u * s + 4u, 4u) << 9));
        e[2 * s + 1] = *(lds_cu128 *)(uintptr_t)(ca + 256u + (__builtin_amdgcn_ubfe(wr, 8u * s, 4u) << 9));
    }
    u32 z[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        u32 t = xor3(xor3(e[0][d], e[1][d], e[2][d]), e[3][d], e[4][d]);
        t = xor3(t, e[5][d], e[6][d]) ^ e[7][d];
        t ^= (u32)__builtin_amdgcn_mov_dpp((int)t, 0xB1, 0xf, 0xf, true);      /* quad_perm [1,0,3,2] */
        t ^= (u32)__builtin_amdgcn_mov_dpp((int)t, 0x4E, 0xf, 0xf, true);      /* quad_perm [2,3,0,1] */
        z[d] = t;
    }
    return make_uint4(z[0], z[1], z[2], z[3]);
}


/* The last levels, shared by k_ghash_final, k_gcm_small and the fused kernel's epilogue: 1024
 * accumulators (one per thread; the last `live` of them are not padding) -> 256 -> 64 -> 16 -> 4 -> 1,
 * radix 4: level l folds rows k = 0..3 of its input (row k = entries [k m, (k+1) m)) with the nibble
 * table of H^m,  out_j = ((in_j H^m ^ in_{m+j}) H^m ^ in_{2m+j}) H^m ^ in_{3m+j}.  These are DEPENDENT
 * multiplications of one wave each (~270 ns: ~80 dependent VALU instructions), so their number is what
 * a short call pays: 3+3+3+3+4 = 16 here against 15+15+4 = 34 with the radix-16 levels of round 1
 * (k_ghash_final 12 -> 7 us).  A quad of lanes shares each product (tabmul4q).  Leading rows that hold
 * only padding are skipped.  GHASH_LAST: the last level multiplies AFTER adding (every block of a GHASH
 * carries at least one factor H); otherwise the result is sum in_q H^(1023-q) (a workgroup's share).
 * T: the six nibble tables (T[0..512) = H^1024 is not used here).  buf: GT_BUF entries.  The result is
 * valid in thread 0.                                                                              */
#define GT_BUF  (1024u + 256u + 64u + 16u + 4u + 4u)

template <bool GHASH_LAST>
__device__ __forceinline__ uint4 gh_tree(uint4 *buf, const uint4 *T, uint4 acc, u32 live)
{
    buf[threadIdx.x] = acc;
    __syncthreads();
    const u32 qi = threadIdx.x >> 2;                           /* accumulator this quad works on */
    u32 n = 1024u, off = 0;
#pragma unroll
    for (u32 lvl = 1; lvl <= 4; ++lvl) {                       /* tables H^256, H^64, H^16, H^4 */
        const u32 m = n >> 2;
        const uint4 *Tl = T + 512u * lvl;
        if (threadIdx.x < n) {
            const u32 k0 = 4u - (live + m - 1u) / m;           /* first row with a live entry */
            acc = buf[off + k0 * m + qi];
            for (u32 k = k0 + 1; k < 4; ++k) acc = x4(tabmul4q(Tl, acc), buf[off + k * m + qi]);
            if ((threadIdx.x & 3u) == 0) buf[off + n + qi] = acc;
        }
        __syncthreads();
        off += n;
        n = m;
        live = live < m ? live : m;
    }
    if (threadIdx.x < 4) {
        const uint4 *TF = T + 512u * 5u;
        if (GHASH_LAST) {
            acc = make_uint4(0, 0, 0, 0);
            for (u32 k = 4u - live; k < 4; ++k) acc = tabmul4q(TF, x4(acc, buf[off + k]));
        } else {
            acc = buf[off + 4u - live];
            for (u32 k = 5u - live; k < 4; ++k) acc = x4(tabmul4q(TF, acc), buf[off + k]);
        }
    }
    return acc;
}

#define GHF_LDS ((GT_NTAB * 512u + GT_BUF) * 16u)

__device__ __forceinline__ u64 shfl_xor64(u64 v, int off)
{
    const u32 lo = __shfl_xor((u32)v, off, 64), hi = __shfl_xor((u32)(v >> 32), off, 64);
    return ((u64)hi << 32) | lo;
}

/* x*y computed by one whole wave; every lane returns the product */
__device__ __forceinline__ Gf wave_gfmul(Gf x, Gf y, u32 lane)
{
    Gf z = gf_mul_slice(x, y, gf_mul_x64(y), lane);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        z.hi ^= shfl_xor64(z.hi, off);
        z.lo ^= shfl_xor64(z.lo, off);
    }
    return z;
}


__device__ __forceinline__ Gf gf_from4(uint4 v) { return gf_from_words(v.x, v.y, v.z, v.w); }

/* The six nibble tables of a key made INSIDE a kernel, from H in buf[GT_BUF - 3] (k_gcm_small, one-shot calls:
 * no k_gcm_setup launch in front).  Same arithmetic as k_gcm_setup: H^(2^k) = F^k H by the constant Frobenius
 * matrices (one wave per power), the 128 generators M x^q of every table, then entry (p, v) = the XOR of the
 * generators the nibble v selects.  buf is scratch here (the reduction buffer is not in use yet).  All 1024
 * threads call it; ends with a barrier.                                                                    */
/* YPOW = false: table t holds H^(2^(10 - 2t)) (the last levels of a sequence of blocks).
 * YPOW = true : table t (t >= 2) holds Y^(4^(5 - t)), Y = H^2048, i.e. H^(2^(21 - 2t)) = H^(2^17), H^(2^15), H^(2^13),
 *               H^(2^11), and table 1 holds Y^256 = H^(2^19): the same radix-4 tree over the partial hashes of
 *               2048-block chunks (k_gcm_combine); table 0 holds H (the finisher's length-block step).                                        */
template <bool YPOW = false>
__device__ __forceinline__ void gcm_build_nibble_tables(uint4 *TC, uint4 *buf, const uint64_t *__restrict__ frob, u32 ylog = 11)
{
    Gf *shPow = (Gf *)buf;                         /* 6 powers                  */
    Gf *shGen = (Gf *)(buf + 16);                  /* 6 x 128 generators, one slot of skew per eight: 13.5 KiB */
    const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    if (wave < GT_NTAB) {
        const u32 k = YPOW ? (wave ? ylog + 10u - 2u * wave : 0u) : 10u - 2u * wave;   /* Y = H^(2^ylog): chunks of 2^ylog positions */
        const Gf h = gf_from4(buf[GT_BUF - 3]);
        Gf pw = h;
        if (k) {
            const uint64_t *rows = frob + (u64)(k - 1) * 256u;
            const u32 b0 = (u32)(__popcll(rows[2 * lane] & h.hi) + __popcll(rows[2 * lane + 1] & h.lo)) & 1u;
            const u32 b1 = (u32)(__popcll(rows[128 + 2 * lane] & h.hi) + __popcll(rows[128 + 2 * lane + 1] & h.lo)) & 1u;
            pw.hi = __ballot(b0);
            pw.lo = __ballot(b1);
        }
        if (lane == 0) shPow[wave] = pw;
    }
    __syncthreads();
    /* (generator g sits at slot g + g / 8: the lanes of the loop below read generators eight apart, 128 bytes, which
     * without the skew are two bank sets for sixteen lanes) */
    for (u32 idx = threadIdx.x; idx < GT_NTAB * 128u; idx += GH_T) shGen[idx + (idx >> 3)] = gf_mul_xq128(shPow[idx >> 7], idx & 127u);
    __syncthreads();
    for (u32 e = threadIdx.x; e < GT_NTAB * 512u; e += GH_T) {
        /* e IS the entry's place in its table (nib_entry: 32 v + slot): neighbouring lanes store neighbouring 16-byte
         * slots.  (Counting e = 16 p + v instead put the eight lanes of a store group 512 bytes apart: an 8-way bank
         * conflict on every one of the 3072 stores, ~2 us of the 4.5 us this function took.) */
        const u32 t = e >> 9, v = (e >> 5) & 15u, sl = e & 31u, p = sl < 16u ? 2u * sl : 2u * (sl - 16u) + 1u;
        const u32 g0 = 128u * t + 4u * p;                     /* four generators, never across a multiple of eight */
        const Gf *gen = shGen + g0 + (g0 >> 3);
        Gf x = { 0, 0 };
#pragma unroll
        for (u32 i = 0; i < 4; ++i) {
            const u64 m = 0 - (u64)((v >> (3 - i)) & 1u);
            x.hi ^= gen[i].hi & m;
            x.lo ^= gen[i].lo & m;
        }
        u32 w[4];
        gf_to_words(x, w);
        TC[e] = make_uint4(w[0], w[1], w[2], w[3]);           /* = TC[512 t + nib_entry(p, v)] */
    }
    __syncthreads();
}

#define GHFB_LDS (GHF_LDS + 1024u)      /* the last-levels layout + an unreplicated Te0 copy (k_gcm_combine) */

#define GSM_MAXNV      2046u           /* two padding positions in front: Enc(J0) and (one-shot calls) H = Enc(0) */
#define GSM_LDS_TAB    65536u
#define GSM_LDS_TOTAL  (GSM_LDS_TAB + (GT_NTAB * 512u + GT_BUF) * 16u)

#define GCM_NR(CALL)                                              \
    switch (nr) {                                                 \
    case 10: { constexpr int NR = 10; rc = CALL; } break;         \
    case 12: { constexpr int NR = 12; rc = CALL; } break;         \
    case 14: { constexpr int NR = 14; rc = CALL; } break;         \
    default: return (int)hipErrorInvalidValue;                    \
    }

#endif /* UAES_GHASH_HIP_H_ */
