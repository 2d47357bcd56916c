/*
 * uaes_aes.hip.h -- the Rijndael round as gfx950 device code.
 *
 * What it reproduces: rijndaelEncrypt / rijndaelDecrypt of the reference
 * (micro_aes.c:242-259, :315-332) -- AddRoundKey, SubBytes, ShiftRows,
 * MixColumns on a 16-byte state whose byte i is column i/4, row i%4.
 *
 * How (MI355X-first, not a translation of the byte loops):
 *  - one 16-byte block per lane, state = 4 little-endian column words;
 *  - SubBytes+ShiftRows+MixColumns fused into 4 table lookups per column
 *    (FIPS-197 sec. 5.1 algebra) served from LDS;
 *  - the tables are replicated 32x so that lane l only ever touches LDS bank
 *    (l mod 32): ds_read_b32 services a wave64 as two 32-lane groups and the
 *    bank of byte address a is (a/4) mod 32 (MI355X_MICROARCH.md, LDS), so a
 *    data-dependent lookup is conflict-free by construction;
 *  - entry x of table k lives at  x*256 + (k&1)*128 + (k>>1)*65536 + 4*(l&31)
 *    which lets ONE v_perm_b32 build the whole LDS address from the state
 *    word (byte select -> bits 8..15) and a per-lane constant (bits 0..7 and
 *    16..23): 1 VALU + 1 ds_read_b32 per S-box application;
 *  - round keys are wave-uniform kernel arguments (SGPRs).
 *
 * LDS budget: 4 tables x 32 KiB = 128 KiB in both directions (Te0..Te3 or
 * Td0..Td3).  The inverse cipher's last round needs plain Si[x], which no Td
 * byte holds -- but 14 ^ 9 ^ 13 ^ 11 = 1 in GF(2^8), so Si[x] is the XOR of the
 * four bytes of Td0[x]: a fold in registers instead of a fifth table.
 */
#ifndef UAES_AES_HIP_H_
#define UAES_AES_HIP_H_

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "uaes_device.h"

typedef uint32_t u32;
typedef uint64_t u64;

#define UAES_WG        1024u              /* threads per workgroup (16 waves)  */
#define UAES_LDS_ENC   (128u * 1024u)
#define UAES_LDS_DEC   (128u * 1024u)

extern __shared__ __attribute__((aligned(16))) unsigned char uaes_lds[];

/* Table reads use ABSOLUTE LDS addresses: the kernels declare no static LDS,
 * so the dynamic segment starts at LDS address 0 (checked by k_selftest).
 * Going through the `uaes_lds` symbol instead makes hipcc emit a
 * `v_add_u32 v, 0, v` (the unresolved symbol offset) in front of every one of
 * the ~160 lookups per block.                                               */
typedef __attribute__((address_space(3))) const u32 lds_cu32;

__device__ __forceinline__ u32 lds_word(u32 byte_addr)
{
    return *(lds_cu32 *)(uintptr_t)byte_addr;
}

/* ---- completion ticket riding on a kernel (uaes_device.h: uaesk_done) ----------------------------------------
 * EVERY thread of EVERY workgroup must call this as its last action: the workgroup meets, thread 0 releases the
 * workgroup's stores to system scope and counts the workgroup in, and the last one to arrive releases the ticket
 * number to the pinned host word (the classic last-block pattern; one workgroup: no counting).               */
uaesk_done uaesk_ticket_take();                     /* the calling host thread's armed ticket (cleared), uaes_kernels.hip */
void uaesk_ticket_unused();
unsigned *uaesk_done_word_take();                    /* the armed zero-between-calls word (cleared), uaes_kernels.hip */
hipError_t uaesk_want_lds(const void *kern, unsigned bytes);   /* dynamic-LDS attribute, set once per (kernel, device) */

/* a multi-launch routine takes the ticket at its entry, so that the single-launch building blocks it calls do not
 * pick it up in the middle of the sequence, and hands it to the one path that is a single launch (use()); if no
 * such path was taken the host layer is told so (uaesk_ticket_disarm() = 1) and sends k_ticket itself          */
struct TicketScope {
    uaesk_done d;
    bool used;
    TicketScope() : d(uaesk_ticket_take()), used(false) {}
    ~TicketScope() { if (d.flag && !used) uaesk_ticket_unused(); }
    const uaesk_done &use() { used = true; return d; }
};

__device__ __forceinline__ void ticket_release(const uaesk_done &d)
{
    if (!d.flag) return;                             /* kernel argument: uniform */
    /* every wave first waits until its own stores have been acknowledged (in the usual, non-tgsplit mode a
     * workgroup-scope release does not have to: the waves of a workgroup share their CU's L1 path, so the barrier
     * alone would let thread 0's fence run while another wave's stores are still on their way to L2 / the fabric) */
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                 /* workgroup-scope release / acquire: the workgroup's stores
                                                      * happen-before what thread 0 does next                      */
    if (threadIdx.x != 0) return;
    __threadfence_system();                          /* ONE system-scope release per workgroup (cumulative), not
                                                      * one per thread: sixteen waves fencing cost more than the
                                                      * second launch this saves                                   */
    if (gridDim.x > 1) {
        const unsigned arrived = __hip_atomic_fetch_add(d.count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (arrived != gridDim.x - 1u) return;
        __hip_atomic_store(d.count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     /* for the next launch */
        __threadfence_system();
    }
    __hip_atomic_store(d.flag, d.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

/* three-input boolean ops in one VALU instruction (v_bitop3_b32, gfx950) */
__device__ __forceinline__ u32 xor3(u32 a, u32 b, u32 c)
{
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
}

__device__ __forceinline__ u32 or_xor(u32 a, u32 b, u32 c)       /* (a | b) ^ c */
{
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x56);
}

__device__ __forceinline__ u32 rotl32(u32 v, u32 n)
{
    return __builtin_amdgcn_alignbit(v, v, (32u - n) & 31u);
}

__device__ __forceinline__ u32 bswap32(u32 v)
{
    return __builtin_amdgcn_perm(0u, v, 0x00010203u);
}

/* per-lane address constants: bits 0..7 = bank slot (+128 for odd tables),
 * bits 16..23 = 64 KiB region of the table pair                            */
struct LaneConst {
    static constexpr int LAY = 0;
    u32 t[4];
    u32 m1;          /* 0x0000ff00 held in a VGPR (all-VGPR bitop3 issues faster) */
};

/* ---- the 64 KiB layout ("split halves") -----------------------------------
 * Every table is replicated 16x instead of 32x: entry x of table k lives at
 *     x*256 + (k&1)*128 + (k>>1)*64 + 4*(l&15)
 * i.e. Te0/Te1 own banks 0..15 and Te2/Te3 banks 16..31.  ds_read_b32 serves
 * lanes {0..31} then {32..63}; inside each group of 32 the lanes come in two
 * TYPES, h = (l>>4)&1, and at every lookup instruction the two types read
 * tables from OPPOSITE bank halves, so the lookup stays conflict-free:
 *   type 0 (natural): instruction slot r takes byte r of its register through Te_r;
 *   type 1 (shifted): slot r takes byte r+2 through Te_{r+2} (mod 4).
 * Both are valid MixColumns terms; a type-1 lane simply produces output column
 * j+2 where a type-0 lane produces column j, so its registers hold the columns
 * rotated by two after every ODD round and aligned again after every even one
 * (NR is even: the result is aligned).  The only cost is the round key of the
 * odd rounds, whose words 0<->2 and 1<->3 swap on type-1 lanes (one extra XOR per
 * column with a per-lane delta).  The last round and the inverse cipher follow the
 * same rule through per-lane v_perm selectors (uaes_aes.hip.h: tlook, combine16).
 * 64 KiB of tables let TWO 1024-thread workgroups share a CU (32 waves), which
 * overlaps the LDS pipe and the VALU far better than 16 waves do.            */
#define UAES_LDS_T64   (64u * 1024u)

struct LaneConst2 {
    static constexpr int LAY = 1;
    u32 t[4];        /* address byte 0 for "table k" (type 1: table k+2)            */
    u32 sel[4];      /* v_perm selector taking "byte b" (type 1: byte b+2) to bits 8..15 */
    u32 selA, selB;  /* last-round merges (swapped on type-1 lanes)                  */
    u32 selD;        /* decrypt last round: order of the two folded halves           */
    u32 hmask;       /* ~0 on type-1 lanes                                           */
};

__device__ __forceinline__ LaneConst make_lane_const()
{
    LaneConst lc;
    const u32 slot = (threadIdx.x & 31u) << 2;
#pragma unroll
    for (u32 k = 0; k < 4; ++k)
        lc.t[k] = slot | ((k & 1u) << 7) | ((k >> 1) << 16);
    lc.m1 = 0x0000ff00u;
    asm volatile("" : "+v"(lc.m1));           /* keep it in a VGPR */
    return lc;
}

/* base: LDS byte address of the 64 KiB of tables, a multiple of 65536 (it rides in byte 2
 * of the per-lane constants, which the v_perm copies into the address)                */
__device__ __forceinline__ LaneConst2 make_lane_const2(u32 base = 0)
{
    LaneConst2 lc;
    const u32 h = (threadIdx.x >> 4) & 1u, slot = (threadIdx.x & 15u) << 2;
#pragma unroll
    for (u32 k = 0; k < 4; ++k) {
        const u32 kk = (k + 2u * h) & 3u;
        lc.t[k] = base | slot | ((kk & 1u) << 7) | ((kk >> 1) << 6);
        lc.sel[k] = 0x0c020000u | ((4u + kk) << 8);
    }
    lc.selA = h ? 0x07020c0cu : 0x0c0c0500u;
    lc.selB = h ? 0x0c0c0500u : 0x07020c0cu;
    lc.selD = h ? 0x02000604u : 0x06040200u;
    lc.hmask = 0u - h;
    return lc;
}

/* lookup table TBL with byte BYTE of w as index */
template <int TBL, int BYTE>
__device__ __forceinline__ u32 tlook(u32 w, const LaneConst2 &lc)
{
    return lds_word(__builtin_amdgcn_perm(w, lc.t[TBL], lc.sel[BYTE]));
}

/* the true table TBL at true byte BYTE whatever the lane type (for rare, non-uniform
 * lookups: may bank-conflict with the other half of the lanes)                   */
template <int TBL, int BYTE>
__device__ __forceinline__ u32 tlook_true(u32 w, const LaneConst2 &lc)
{
    const u32 c = (lc.t[0] & 0xffff003cu) | ((TBL & 1u) << 7) | ((TBL >> 1) << 6);
    return lds_word(__builtin_amdgcn_perm(w, c, 0x0c020000u | ((4u + BYTE) << 8)));
}
template <int TBL, int BYTE>
__device__ __forceinline__ u32 tlook(u32 w, const LaneConst &lc);
template <int TBL, int BYTE>
__device__ __forceinline__ u32 tlook_true(u32 w, const LaneConst &lc) { return tlook<TBL, BYTE>(w, lc); }

template <int TBL, int BYTE>
__device__ __forceinline__ u32 tlook(u32 w, const LaneConst &lc)
{
    u32 addr;
    if (BYTE == 1) {
        /* the index byte already sits at bits 8..15: (w & 0xff00) | lane constant is ONE
         * v_bitop3_b32 on VGPR operands (~2.9 issue cycles vs ~4.5 for v_perm_b32,
         * profiles/ubench/r01_valurate.log)                                          */
        addr = __builtin_amdgcn_bitop3_b32(w, lc.m1, lc.t[TBL], 0xea);   /* (a & b) | c */
    } else {
        /* D.b0 = lc.b0, D.b1 = w.byte[BYTE], D.b2 = lc.b2, D.b3 = 0 */
        addr = __builtin_amdgcn_perm(w, lc.t[TBL], 0x0c020000u | ((4u + BYTE) << 8));
    }
    return lds_word(addr);
}

/* ---- table construction (once per workgroup) --------------------------- */
/* 16-byte stores are serviced eight lanes at a time and those eight must fall into different banks, so the
 * eight lanes of a service group write the eight 16-byte pieces of ONE entry (128 contiguous bytes) instead of
 * one entry each (256 bytes apart = the same four banks, an 8-way conflict: 3.3 us for the 128 KiB instead of
 * 0.7, tools/ubench/fillbench.hip -- and every launch of every kernel pays it before its first lookup).      */
__device__ __forceinline__ void store_replicas(u32 byte_addr, u32 v)
{
    uint4 vv = make_uint4(v, v, v, v);
    uint4 *dst = (uint4 *)(uaes_lds + byte_addr);
#pragma unroll
    for (int r = 0; r < 8; ++r) dst[r] = vv;
}

/* Te_k[x] = rotl(Te0[x], 8k); Te0 bytes = {2S,S,S,3S} */
/* the share j = first, first + step, ... of the 8192 pieces (no barrier: a kernel whose first wave has something
 * else to do meanwhile -- k_xts_small's tweak encryption -- lets the others fill).  Eight pieces at a time: their eight
 * table words are requested before the first is used -- a loop that loads, rotates and stores piece by piece waits for
 * every load where it is issued, one cache round trip per piece (32 in a row for the 256-thread workgroups of the
 * short XTS / OCB launches: ~4 us before the first block) */
__device__ __forceinline__ void fill_enc_tables_share(const u32 *__restrict__ te0, u32 first, u32 step)
{
    for (u32 j0 = first; j0 < 8192u; j0 += 8u * step) {
        u32 t[8];
#pragma unroll
        for (u32 k = 0; k < 8; ++k) {
            const u32 j = j0 + k * step;
            t[k] = te0[((j < 8192u ? j : 0u) >> 3) & 255u];
        }
#pragma unroll
        for (u32 k = 0; k < 8; ++k) {
            const u32 j = j0 + k * step;
            if (j < 8192u) {
                const u32 r = j & 7u, e = j >> 3, x = e & 255u, kk = e >> 8;
                const u32 v = rotl32(t[k], 8u * kk);
                *(uint4 *)(uaes_lds + x * 256u + (kk & 1u) * 128u + (kk >> 1) * 65536u + 16u * r) = make_uint4(v, v, v, v);
            }
        }
    }
}

__device__ __forceinline__ void fill_enc_tables(const u32 *__restrict__ te0)
{
    fill_enc_tables_share(te0, threadIdx.x, blockDim.x);
    __syncthreads();
}

/* 64 KiB layout: 16 replicas (64 B) per entry, the four tables of an x side by side in one 256-byte row
 * (Te0 | Te2 | Te1 | Te3): sixteen lanes write the row's sixteen 16-byte pieces                          */
__device__ __forceinline__ void fill_tables64(const u32 *__restrict__ t0, u32 base = 0)
{
    for (u32 j0 = threadIdx.x; j0 < 4096u; j0 += 4u * blockDim.x) {       /* four pieces at a time, their table words first */
        u32 t[4];
#pragma unroll
        for (u32 k = 0; k < 4; ++k) {
            const u32 j = j0 + k * blockDim.x;
            t[k] = t0[(j < 4096u ? j : 0u) >> 4];
        }
#pragma unroll
        for (u32 k = 0; k < 4; ++k) {
            const u32 j = j0 + k * blockDim.x;
            if (j < 4096u) {
                const u32 p = j & 15u, x = j >> 4, q = p >> 2, kk = (q >> 1) | ((q & 1u) << 1);
                const u32 v = rotl32(t[k], 8u * kk);
                *(uint4 *)(uaes_lds + base + x * 256u + 16u * p) = make_uint4(v, v, v, v);
            }
        }
    }
    __syncthreads();
}

/* Td_k[x] = rotl(Td0[x], 8k); Td0 bytes = {14Si, 9Si, 13Si, 11Si}; same layout as Te */
__device__ __forceinline__ void fill_dec_tables(const u32 *__restrict__ td0)
{
    fill_enc_tables(td0);
}

/* ---- rounds --------------------------------------------------------------- */
/* Round-key word c for the state produced by round `r` (LaneConst2: after an odd
 * round the columns of a type-1 lane sit two places on, so it needs word c^2).    */
template <typename LC>
__device__ __forceinline__ void key_delta(const u32 *rk, const LC &lc, bool odd, u32 (&d)[2])
{
    d[0] = d[1] = 0;
    if (LC::LAY == 1 && odd) {
        d[0] = (rk[0] ^ rk[2]) & ((const LaneConst2 &)lc).hmask;
        d[1] = (rk[1] ^ rk[3]) & ((const LaneConst2 &)lc).hmask;
    }
}

/* issue the 16 lookups of one round of one block; results are consumed later.
 * DEC: the equivalent inverse cipher (FIPS-197 sec. 5.3.5): column c takes row r
 * from column (c - r) mod 4 instead of (c + r); its last round reads Td0 only.
 * Last encryption round: S[x] sits in byte r of Te2 (r=0), Te3 (r=1), Te0 (r=2), Te1 (r=3). */
template <bool LAST, bool DEC, typename LC>
__device__ __forceinline__ void issue16(const u32 (&s)[4], u32 (&t)[16], const LC &lc)
{
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const u32 a = s[c], b = s[(DEC ? c + 3 : c + 1) & 3], cc = s[(c + 2) & 3], d = s[(DEC ? c + 1 : c + 3) & 3];
        if (!LAST) {
            t[4 * c + 0] = tlook<0, 0>(a, lc); t[4 * c + 1] = tlook<1, 1>(b, lc);
            t[4 * c + 2] = tlook<2, 2>(cc, lc); t[4 * c + 3] = tlook<3, 3>(d, lc);
        } else if (!DEC) {
            t[4 * c + 0] = tlook<2, 0>(a, lc); t[4 * c + 1] = tlook<3, 1>(b, lc);
            t[4 * c + 2] = tlook<0, 2>(cc, lc); t[4 * c + 3] = tlook<1, 3>(d, lc);
        } else {
            t[4 * c + 0] = tlook<0, 0>(a, lc); t[4 * c + 1] = tlook<0, 1>(b, lc);
            t[4 * c + 2] = tlook<0, 2>(cc, lc); t[4 * c + 3] = tlook<0, 3>(d, lc);
        }
    }
}

/* Si[x0] | Si[x1] << 16 (bytes 1 and 3 are junk) from the Td0 words of x0 and x1:
 * g = t ^ rot16(t) has b0^b2 | b1^b3 in its low half; pair the two low halves and
 * fold the odd bytes onto the even ones.  (Invariant under rot16 of t: a type-1
 * lane of the 64 KiB layout reads Td2 = rot16(Td0) here.)                      */
__device__ __forceinline__ u32 fold2(u32 t0, u32 t1)
{
    const u32 g0 = t0 ^ __builtin_amdgcn_alignbit(t0, t0, 16), g1 = t1 ^ __builtin_amdgcn_alignbit(t1, t1, 16);
    const u32 p = __builtin_amdgcn_perm(g1, g0, 0x05040100u);       /* g0.b0 g0.b1 g1.b0 g1.b1 */
    return p ^ (p >> 8);
}

/* `odd`: the round number producing this state is odd (matters to LaneConst2 only) */
template <bool LAST, bool DEC, typename LC>
__device__ __forceinline__ void combine16(const u32 (&t)[16], const u32 *rk, u32 (&s)[4], const LC &lc, bool odd)
{
    u32 dl[2];
    key_delta(rk, lc, !LAST && odd, dl);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (!LAST) {
            s[c] = xor3(xor3(t[4 * c], t[4 * c + 1], t[4 * c + 2]), t[4 * c + 3], rk[c]);
            if (LC::LAY == 1 && odd) s[c] ^= dl[c & 1];
        } else if (!DEC) {
            if (LC::LAY == 1) {
                const LaneConst2 &l2 = (const LaneConst2 &)lc;
                s[c] = or_xor(__builtin_amdgcn_perm(t[4 * c + 1], t[4 * c], l2.selA),
                              __builtin_amdgcn_perm(t[4 * c + 3], t[4 * c + 2], l2.selB), rk[c]);
            } else {
                s[c] = or_xor(__builtin_amdgcn_perm(t[4 * c + 1], t[4 * c], 0x0c0c0500u),
                              __builtin_amdgcn_perm(t[4 * c + 3], t[4 * c + 2], 0x07020c0cu), rk[c]);
            }
        } else {                                      /* rows 0,1 | rows 2,3 -> bytes 0..3 */
            const u32 seld = LC::LAY == 1 ? ((const LaneConst2 &)lc).selD : 0x06040200u;
            s[c] = __builtin_amdgcn_perm(fold2(t[4 * c + 2], t[4 * c + 3]), fold2(t[4 * c], t[4 * c + 1]), seld) ^ rk[c];
        }
    }
}

/* rounds FIRST..NR on U blocks in lock step; states hold the input of round FIRST
 * (i.e. after AddRoundKey(FIRST-1)); rk = all round keys                          */
template <int NR, int U, int FIRST, bool DEC, typename LC>
__device__ __forceinline__ void rounds_from(u32 (&s)[U][4], const uaesk_rk &rk, const LC &lc)
{
    u32 t[U][16];
#pragma unroll
    for (int r = FIRST; r < NR; ++r) {
#pragma unroll
        for (int u = 0; u < U; ++u) issue16<false, DEC>(s[u], t[u], lc);
#pragma unroll
        for (int u = 0; u < U; ++u) combine16<false, DEC>(t[u], &rk.w[4 * r], s[u], lc, (r & 1) != 0);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) issue16<true, DEC>(s[u], t[u], lc);
#pragma unroll
    for (int u = 0; u < U; ++u) combine16<true, DEC>(t[u], &rk.w[4 * NR], s[u], lc, false);
}

template <int NR, int U, int FIRST, typename LC>
__device__ __forceinline__ void enc_rounds_from(u32 (&s)[U][4], const uaesk_rk &rk, const LC &lc)
{
    rounds_from<NR, U, FIRST, false>(s, rk, lc);
}

/* full cipher on U blocks (state = plaintext words on entry) */
template <int NR, int U, typename LC>
__device__ __forceinline__ void enc_blocks(u32 (&s)[U][4], const uaesk_rk &rk, const LC &lc)
{
#pragma unroll
    for (int u = 0; u < U; ++u) {
        s[u][0] ^= rk.w[0]; s[u][1] ^= rk.w[1]; s[u][2] ^= rk.w[2]; s[u][3] ^= rk.w[3];
    }
    rounds_from<NR, U, 1, false>(s, rk, lc);
}

/* ---- split-phase rounds (software pipelining across two blocks) ---------- */
/* rounds FIRST..NR of two blocks A and B, half a round out of phase: while the
 * 16 lookups of one block are in flight the wave does the other block's XORs
 * and address building, so its LDS queue never drains (a wave can have at most
 * 15 LDS operations outstanding, and the compiler's default schedule for the
 * lock-step version drains them to zero every ~12 lookups).                 */
/* RKV: round keys FIRST..NR as an array indexed from 0 (may live in VGPRs: a
 * v_bitop3_b32 with an SGPR operand issues ~1.5 cycles slower than all-VGPR) */
template <int NR, int FIRST, typename RKV, bool DEC = false, typename LC = LaneConst, int PHI = 1, int PLO = 0>
__device__ __forceinline__ void enc_rounds_skewed(u32 (&sa)[4], u32 (&sb)[4], const RKV &rkv, const LC &lc)
{
#define rkp(r) (&rkv.w[4 * ((r) - FIRST)])
    /* Wave priority: a wave that is issuing a round's lookups (16 address computations + 16
     * ds_read) runs at priority 1, a wave that is combining results at 0.  Every SIMD holds
     * 4 waves; with equal priorities the oldest wave wins the VALU whatever it is about to do,
     * and the LDS pipe -- the scarcer resource -- idles while combines run.  Measured on 1 GiB,
     * interleaved A/B (profiles/HISTORY.md): CTR +3.0 %, ECB +2.1 %, ECB decrypt
     * +6.5 %, XTS-256 +3.6 %, GCM +1.4 %, neutral for OCB and CBC/CFB decrypt.  Priority 2 or 3
     * instead of 1 does the same for those and costs OCB / CBC / CFB decrypt 3-5 %.
     * PHI = the priority of the issue phase (the fused GCM kernel uses 2 and runs its GHASH
     * lookups at 1, between the two: +0.9 % over 1/0 with the GHASH lookups at 0).          */
#define XPRIO(n) __builtin_amdgcn_s_setprio((n) ? PHI : PLO)
    u32 ta[16], tb[16];
    issue16<false, DEC>(sa, ta, lc);
#pragma unroll
    for (int r = FIRST; r < NR; ++r) {
        __builtin_amdgcn_sched_barrier(0);
        XPRIO(1);
        issue16<false, DEC>(sb, tb, lc);
        XPRIO(0);
        __builtin_amdgcn_sched_barrier(0);
        combine16<false, DEC>(ta, rkp(r), sa, lc, (r & 1) != 0);
        XPRIO(1);
        if (r + 1 < NR) issue16<false, DEC>(sa, ta, lc); else issue16<true, DEC>(sa, ta, lc);
        XPRIO(0);
        __builtin_amdgcn_sched_barrier(0);
        combine16<false, DEC>(tb, rkp(r), sb, lc, (r & 1) != 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    issue16<true, DEC>(sb, tb, lc);
    __builtin_amdgcn_sched_barrier(0);
    combine16<true, DEC>(ta, rkp(NR), sa, lc, false);
    __builtin_amdgcn_sched_barrier(0);
    combine16<true, DEC>(tb, rkp(NR), sb, lc, false);
#undef rkp
#undef XPRIO
}

/* hand-scheduled variants of the same two-block loop (tools/gen_rounds_asm.py), selected at build time */

/* full cipher on two blocks, skewed (state = plaintext words on entry) */
struct RkView {
    const u32 *w;
};

template <int NR, typename LC>
__device__ __forceinline__ void enc_blocks_skewed(u32 (&sa)[4], u32 (&sb)[4], const uaesk_rk &rk, const LC &lc)
{
#pragma unroll
    for (int c = 0; c < 4; ++c) { sa[c] ^= rk.w[c]; sb[c] ^= rk.w[c]; }
    const RkView v = { rk.w + 4 };
    enc_rounds_skewed<NR, 1, RkView, false, LC>(sa, sb, v, lc);
}

/* ---- decryption (equivalent inverse cipher, FIPS-197 sec. 5.3.5) ---------- */
/* two blocks, skewed; dk = equivalent-inverse round keys (uaesk_rk of uaes_device.h) */
template <int NR, typename LC>
__device__ __forceinline__ void dec_blocks_skewed(u32 (&sa)[4], u32 (&sb)[4], const uaesk_rk &dk, const LC &lc)
{
#pragma unroll
    for (int c = 0; c < 4; ++c) { sa[c] ^= dk.w[c]; sb[c] ^= dk.w[c]; }
    const RkView v = { dk.w + 4 };
    enc_rounds_skewed<NR, 1, RkView, true, LC>(sa, sb, v, lc);
}

/* U blocks: pairs run skewed, a single block runs round by round */
template <int NR, int U, typename LC>
__device__ __forceinline__ void dec_blocks(u32 (&s)[U][4], const uaesk_rk &dk, const LC &lc)
{
    if (U % 2 == 0) {
#pragma unroll
        for (int u = 0; u < U; u += 2) dec_blocks_skewed<NR>(s[u], s[(u + 1) % U], dk, lc);
        return;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int c = 0; c < 4; ++c) s[u][c] ^= dk.w[c];
    }
    rounds_from<NR, U, 1, true>(s, dk, lc);
}

/* ---- one block through an UNREPLICATED 1 KiB copy of Te0 (anywhere in LDS) ---------------------------
 * For the odd single block a kernel needs before its bulk work (XTS: Enc_key2(tweak); GCM: H = Enc(0), Enc(J0))
 * when the replicated cipher tables are not there (decrypt direction, GHASH-only kernels): all lanes of a wave
 * that run it read the same entries -- broadcasts, no bank conflicts -- and Te1..Te3 are rotations.          */
template <int NR>
__device__ __forceinline__ void plain_encrypt(const u32 *te0, const uaesk_rk &rk, u32 (&s)[4])
{
#pragma unroll
    for (int c = 0; c < 4; ++c) s[c] ^= rk.w[c];
    for (int r = 1; r < NR; ++r) {
        u32 t[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
            t[c] = te0[s[c] & 0xffu] ^ rotl32(te0[(s[(c + 1) & 3] >> 8) & 0xffu], 8) ^
                   rotl32(te0[(s[(c + 2) & 3] >> 16) & 0xffu], 16) ^ rotl32(te0[s[(c + 3) & 3] >> 24], 24) ^ rk.w[4 * r + c];
#pragma unroll
        for (int c = 0; c < 4; ++c) s[c] = t[c];
    }
    u32 t[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)                     /* S[x] = byte 1 of Te0[x] */
        t[c] = (((te0[s[c] & 0xffu] >> 8) & 0xffu) | (te0[(s[(c + 1) & 3] >> 8) & 0xffu] & 0xff00u) |
                ((te0[(s[(c + 2) & 3] >> 16) & 0xffu] & 0xff00u) << 8) | ((te0[s[(c + 3) & 3] >> 24] & 0xff00u) << 16)) ^
               rk.w[4 * NR + c];
#pragma unroll
    for (int c = 0; c < 4; ++c) s[c] = t[c];
}

/* ---- one block at a time, four lanes per block (the serial chains) ---------------- */
/* CBC/CFB encryption, OFB and the CBC-MACs are chains: block i+1 cannot start before block i
 * is done, so what counts is the LATENCY of one block.  One lane alone issues 16 address
 * computations + 16 lookups + 8 combines per round (~250 cycles); here the four lanes of a
 * quad share the block: lane c computes output column c (4 lookups), then the quad exchanges
 * the four new columns with three DPP quad_perm moves.  Every lane keeps the state as a
 * ROTATED VIEW (w0..w3 = columns c, c+1, c+2, c+3), which is exactly what its four lookups
 * need (row r comes from column c+r) and what the quad_perm rotations deliver, so no
 * per-lane register indexing is needed.  The callers keep a plain replicated block in all
 * lanes: entry = AddRoundKey(0) + 12 selects, exit = 4 quad broadcasts.
 *
 * Tables: the 128 KiB layout with only slots 0..3 filled (4 KiB of stores by one wave instead
 * of 128 KiB by sixteen): lane c reads slot c, so the quad's four lookups never conflict and
 * the other fifteen quads of the wave, which run the same data redundantly, broadcast.
 * Round keys live in LDS after the table region (lane c needs word 4r + c of round r).   */
#define UAES_LDS_QUAD  (UAES_LDS_ENC + 256u)          /* tables + 60 key words */

__device__ __forceinline__ void quad_fill_tables(const u32 *__restrict__ t0, const uaesk_rk &rk,
                                                 u32 key_base = UAES_LDS_ENC)
{
    for (u32 i = threadIdx.x; i < 1024u; i += blockDim.x) {
        const u32 x = i & 255u, k = i >> 8;
        const u32 v = rotl32(t0[x], 8u * k);
        *(uint4 *)(uaes_lds + x * 256u + (k & 1u) * 128u + (k >> 1) * 65536u) = make_uint4(v, v, v, v);
    }
    for (u32 i = threadIdx.x; i < 60u; i += blockDim.x) ((u32 *)(uaes_lds + key_base))[i] = rk.w[i];
    __syncthreads();
}

__device__ __forceinline__ LaneConst quad_lane_const()
{
    LaneConst lc;
    const u32 slot = (threadIdx.x & 3u) << 2;
#pragma unroll
    for (u32 k = 0; k < 4; ++k) lc.t[k] = slot | ((k & 1u) << 7) | ((k >> 1) << 16);
    lc.m1 = 0x0000ff00u;
    asm volatile("" : "+v"(lc.m1));
    return lc;
}

template <int CTRL>
__device__ __forceinline__ u32 quad_perm(u32 v)
{
    return (u32)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true);
}

/* t = the block (replicated in the quad's four lanes) -> its encryption, replicated again */
template <int NR>
__device__ __forceinline__ void quad_encrypt(u32 (&t)[4], const uaesk_rk &rk, const LaneConst &lc,
                                             u32 key_base = UAES_LDS_ENC)
{
    const u32 c = threadIdx.x & 3u;
    const u32 kaddr = key_base + 4u * c;                        /* + 16 r: this lane's word of round key r */
#pragma unroll
    for (int j = 0; j < 4; ++j) t[j] ^= rk.w[j];
    /* rotated view: w[k] = column (c + k) mod 4 */
    u32 w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const u32 a = t[k], b = t[(k + 1) & 3], cc = t[(k + 2) & 3], d = t[(k + 3) & 3];
        w[k] = c == 0 ? a : c == 1 ? b : c == 2 ? cc : d;
    }
#pragma unroll
    for (int r = 1; r < NR; ++r) {
        const u32 key = lds_word(kaddr + 16u * r);
        const u32 n = xor3(xor3(tlook<0, 0>(w[0], lc), tlook<1, 1>(w[1], lc), tlook<2, 2>(w[2], lc)),
                           tlook<3, 3>(w[3], lc), key);
        w[0] = n; w[1] = quad_perm<0x39>(n); w[2] = quad_perm<0x4E>(n); w[3] = quad_perm<0x93>(n);
    }
    const u32 key = lds_word(kaddr + 16u * NR);
    const u32 lo = __builtin_amdgcn_perm(tlook<3, 1>(w[1], lc), tlook<2, 0>(w[0], lc), 0x0c0c0500u);
    const u32 hi = __builtin_amdgcn_perm(tlook<1, 3>(w[3], lc), tlook<0, 2>(w[2], lc), 0x07020c0cu);
    const u32 n = or_xor(lo, hi, key);
    t[0] = quad_perm<0x00>(n); t[1] = quad_perm<0x55>(n); t[2] = quad_perm<0xAA>(n); t[3] = quad_perm<0xFF>(n);
}


/* ---- one block at a time, SIXTEEN lanes per block (the serial chains, round 3) ------------
 * quad_encrypt above issues 14 instructions per round from one wave (key word, 4 addresses, 4 lookups,
 * 2 XORs, 3 DPP moves) and a lone wave issues one instruction every ~8 cycles: 170 cycles per round of which
 * only ~50 are the LDS round trip.  Here lane i of every 16-lane DPP row owns state byte i (column i/4, row
 * i%4): a round is ONE lookup per lane and four VALU instructions,
 *     t = T_r[byte]                             lane (c,r) looks its byte up in the table of its row
 *     y = t ^ row_ror:11(t);  z = y ^ row_ror:6(y)
 *                                               lane 4c' now holds t[4c'] ^ t[4c'+5] ^ t[4c'+10] ^ t[4c'+15] (mod 16):
 *                                               rows 0..3 of columns c', c'+1, c'+2, c'+3 = ShiftRows + MixColumns
 *                                               of output column c' (the other lanes hold sums nobody needs)
 *     w = key[c] ^ quad_perm:[0,0,0,0](z)       the column word, in all four lanes of quad c
 *     address = v_perm(w, lane constant)        byte r of the word -> bits 8..15
 * so the chain is  ds_read -> 3 DPP XORs -> v_perm -> ds_read.  The four rows of the wave run the same block
 * redundantly (identical addresses broadcast).  Between blocks the state stays a COLUMN WORD per lane; the
 * chains XOR the next text column in (one more instruction) without ever assembling a 16-byte block.
 *
 * Tables: entry x at x*256; bytes 0..63 = (Te0,Te1,Te2,Te3)[x] four times (slot i = lane i of the row reads
 * T_(i%4): bank i), bytes 64..127 = (Te2,Te3,Te0,Te1)[x] four times (last round: S[x] sits in byte r of
 * Te_(r+2), bank 16+i).  32 KiB of stores; the round keys follow at 64 KiB.                          */
#define UAES_LDS_ROW   (65536u + 256u)

template <int NR>
struct RowLane {
    u32 tmain, tlast;     /* address byte 0 for the main / last-round table                         */
    u32 sel;              /* v_perm selector: byte r of the column word -> bits 8..15                */
    u32 lsel;             /* v_perm selector keeping byte r of the last-round lookup, zeros elsewhere */
    u32 c;                /* this lane's column                                                      */
    u32 kc[NR + 1];       /* this lane's column of every round key                                   */
};

__device__ __forceinline__ void row_fill_tables(const u32 *__restrict__ te0, const uaesk_rk &rk)
{
    /* eight lanes per entry: the 128 bytes of an entry are eight 16-byte pieces (bank-conflict-free stores).
     * The chain kernels' one wave has 32 pieces per lane, and a loop that loads a table word, rotates and stores it
     * waits for 32 cache round trips in a row (~4 us of a 7 us CMAC call): sixteen words are requested at a time */
    for (u32 j0 = threadIdx.x; j0 < 2048u; j0 += 16u * blockDim.x) {
        u32 t[16];
#pragma unroll
        for (u32 k = 0; k < 16; ++k) {
            const u32 j = j0 + k * blockDim.x;
            t[k] = te0[(j < 2048u ? j : 0u) >> 3];
        }
#pragma unroll
        for (u32 k = 0; k < 16; ++k) {
            const u32 j = j0 + k * blockDim.x;
            if (j < 2048u) {
                const u32 q = j & 7u, x = j >> 3;
                const u32 t0 = t[k], t1 = rotl32(t0, 8), t2 = rotl32(t0, 16), t3 = rotl32(t0, 24);
                *(uint4 *)(uaes_lds + x * 256u + 16u * q) = q < 4u ? make_uint4(t0, t1, t2, t3) : make_uint4(t2, t3, t0, t1);
            }
        }
    }
    for (u32 i = threadIdx.x; i < 60u; i += blockDim.x) ((u32 *)(uaes_lds + 65536u))[i] = rk.w[i];
    __syncthreads();
}

template <int NR>
__device__ __forceinline__ RowLane<NR> row_lane()
{
    RowLane<NR> L;
    const u32 i = threadIdx.x & 15u, r = i & 3u;
    L.c = i >> 2;
    L.tmain = 4u * i;
    L.tlast = 64u + 4u * i;
    L.sel = 0x0c020000u | ((4u + r) << 8);
    L.lsel = (0x0c0c0c0cu & ~(0xffu << (8u * r))) | ((4u + r) << (8u * r));
#pragma unroll
    for (int j = 0; j <= NR; ++j) L.kc[j] = lds_word(65536u + 16u * (u32)j + 4u * L.c);
    return L;
}

/* The same with the FOUR ROWS OF A WAVE ON FOUR DIFFERENT BLOCKS (batches of independent chains, uaes_chain.hip):
 * row q of a wave needs banks of its own, so the entry of x (still at x*256) holds (Te0,Te1,Te2,Te3)[x] sixteen
 * times -- row q reads bytes 64 q .. 64 q + 63, banks 16 q .. 16 q + 15 -- and the last round's
 * (Te2,Te3,Te0,Te1)[x] sits in a second region 64 KiB up (bit 16 of the address comes from byte 2 of the lane
 * constant, which the same v_perm selector already copies); the round keys follow at 128 KiB.                 */
#define UAES_LDS_ROW4  (131072u + 256u)

__device__ __forceinline__ void row4_fill_tables(const u32 *__restrict__ te0, const uaesk_rk &rk)
{
    for (u32 j0 = threadIdx.x; j0 < 8192u; j0 += 8u * blockDim.x) {      /* 256 entries x 32 pieces of 16 bytes */
        u32 t[8];
#pragma unroll
        for (u32 k = 0; k < 8; ++k) {
            const u32 j = j0 + k * blockDim.x;
            t[k] = te0[(j < 8192u ? j : 0u) >> 5];
        }
#pragma unroll
        for (u32 k = 0; k < 8; ++k) {
            const u32 j = j0 + k * blockDim.x;
            if (j < 8192u) {
                const u32 q = j & 31u, x = j >> 5;
                const u32 t0 = t[k], t1 = rotl32(t0, 8), t2 = rotl32(t0, 16), t3 = rotl32(t0, 24);
                *(uint4 *)(uaes_lds + (q >> 4) * 65536u + x * 256u + 16u * (q & 15u)) =
                    q < 16u ? make_uint4(t0, t1, t2, t3) : make_uint4(t2, t3, t0, t1);
            }
        }
    }
    for (u32 i = threadIdx.x; i < 60u; i += blockDim.x) ((u32 *)(uaes_lds + 131072u))[i] = rk.w[i];
    __syncthreads();
}

template <int NR>
__device__ __forceinline__ RowLane<NR> row4_lane()
{
    RowLane<NR> L;
    const u32 i = threadIdx.x & 15u, r = i & 3u, q = (threadIdx.x >> 4) & 3u;
    L.c = i >> 2;
    L.tmain = 64u * q + 4u * i;
    L.tlast = 65536u + 64u * q + 4u * i;
    L.sel = 0x0c020000u | ((4u + r) << 8);
    L.lsel = (0x0c0c0c0cu & ~(0xffu << (8u * r))) | ((4u + r) << (8u * r));
#pragma unroll
    for (int j = 0; j <= NR; ++j) L.kc[j] = lds_word(131072u + 16u * (u32)j + 4u * L.c);
    return L;
}

template <int CTRL>
__device__ __forceinline__ u32 row_dpp(u32 v)
{
    return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}

/* w = this lane's column word of the block (no round key applied) -> the same of its encryption */
template <int NR>
__device__ __forceinline__ u32 row_encrypt(u32 w, const RowLane<NR> &L)
{
    w ^= L.kc[0];
    u32 addr = __builtin_amdgcn_perm(w, L.tmain, L.sel);
#pragma unroll
    for (int r = 1; r <= NR; ++r) {
        u32 t = lds_word(addr);
        if (r == NR) t = __builtin_amdgcn_perm(t, 0u, L.lsel);
        const u32 y = t ^ row_dpp<0x12B>(t);          /* row_ror:11 = take lane i+5  */
        const u32 z = y ^ row_dpp<0x126>(y);          /* row_ror:6  = take lane i+10 */
        w = L.kc[r] ^ row_dpp<0x00>(z);               /* quad_perm:[0,0,0,0]         */
        if (r < NR) addr = __builtin_amdgcn_perm(w, r + 1 < NR ? L.tmain : L.tlast, L.sel);
    }
    return w;
}

/* the column word of lane column c from a block every lane holds / back to such a block */
__device__ __forceinline__ u32 row_pick(const u32 (&b)[4], u32 c)
{
    return c == 0 ? b[0] : c == 1 ? b[1] : c == 2 ? b[2] : b[3];
}

__device__ __forceinline__ void row_spread(u32 w, u32 (&b)[4])
{
    b[0] = (u32)__builtin_amdgcn_readlane((int)w, 0);  b[1] = (u32)__builtin_amdgcn_readlane((int)w, 4);
    b[2] = (u32)__builtin_amdgcn_readlane((int)w, 8);  b[3] = (u32)__builtin_amdgcn_readlane((int)w, 12);
}

/* this lane's column word of the 16 bytes at p, zero padded after `avail` bytes */
__device__ __forceinline__ u32 row_load(const unsigned char *p, u64 avail, u32 c)
{
    if (avail >= 16 && (((uintptr_t)p) & 3u) == 0) return ((const u32 *)p)[c];
    u32 v = 0;
#pragma unroll
    for (u32 k = 0; k < 4; ++k)
        if (4u * c + k < avail) v |= (u32)p[4u * c + k] << (8u * k);
    return v;
}

/* the first nbytes of the block whose column words the lanes hold -> p (lanes 0, 4, 8, 12 store) */
__device__ __forceinline__ void row_store(unsigned char *p, u32 w, u32 nbytes)
{
    if (threadIdx.x >= 16u || (threadIdx.x & 3u)) return;
    const u32 c = threadIdx.x >> 2;
    if (nbytes >= 16 && (((uintptr_t)p) & 3u) == 0) { ((u32 *)p)[c] = w; return; }
#pragma unroll
    for (u32 k = 0; k < 4; ++k)
        if (4u * c + k < nbytes) p[4u * c + k] = (unsigned char)(w >> (8u * k));
}

/* full blocks inside the chain loops: the alignment is decided once per message (A4 = both pointers
 * 4-byte aligned), so that the loop body is straight-line code -- a load behind a branch makes the
 * compiler wait for it (s_waitcnt vmcnt(0)) right where it was issued, and the chain then pays the
 * memory latency of every block.  With A4 every lane stores its column word (the four lanes of a
 * quad and the four rows write the same dword with the same value): no exec masking in the loop.   */
template <bool A4>
__device__ __forceinline__ u32 row_load_full(const unsigned char *p, u32 c)
{
    if (A4) return ((const u32 *)p)[c];
    const unsigned char *q = p + 4u * c;
    return (u32)q[0] | ((u32)q[1] << 8) | ((u32)q[2] << 16) | ((u32)q[3] << 24);
}

template <bool A4>
__device__ __forceinline__ void row_store_full(unsigned char *p, u32 w, u32 c)
{
    if (A4) { ((u32 *)p)[c] = w; return; }
    unsigned char *q = p + 4u * c;
    q[0] = (unsigned char)w; q[1] = (unsigned char)(w >> 8); q[2] = (unsigned char)(w >> 16); q[3] = (unsigned char)(w >> 24);
}

/* f(i, x): the full 16-byte blocks i = 0..nblk-1 at p as column words x.  The text is requested a whole
 * chunk of ROW_CH blocks ahead: the loads of chunk k+1 are issued before the chain walks chunk k and are
 * first touched when it is done (~16 us later), so whatever s_waitcnt vmcnt the compiler places inside a
 * chunk finds them complete -- with a four-block register ring it waited for the newest load in the middle
 * of every fourth block (a register copy the allocator put there) and the chain paid the memory latency.
 * Indices past the end are clamped, not branched around.                                              */
#define ROW_CH 16
template <bool A4, typename F>
__device__ __forceinline__ void row_walk(const unsigned char *p, u64 nblk, u32 c, F f)
{
    if (nblk == 0) return;
    const u64 last = nblk - 1;
    u32 cur[ROW_CH], nxt[ROW_CH];
#pragma unroll
    for (u32 j = 0; j < ROW_CH; ++j) cur[j] = row_load_full<A4>(p + 16u * (j < last ? j : last), c);
    for (u64 i = 0; i < nblk; i += ROW_CH) {
#pragma unroll
        for (u32 j = 0; j < ROW_CH; ++j) {
            const u64 nx = i + ROW_CH + j;
            nxt[j] = row_load_full<A4>(p + 16u * (nx < last ? nx : last), c);
        }
        if (i + ROW_CH <= nblk) {
#pragma unroll
            for (u32 j = 0; j < ROW_CH; ++j) f(i + j, cur[j]);
        } else {
#pragma unroll
            for (u32 j = 0; j < ROW_CH; ++j)
                if (i + j < nblk) f(i + j, cur[j]);
        }
#pragma unroll
        for (u32 j = 0; j < ROW_CH; ++j) cur[j] = nxt[j];
    }
}

#endif
