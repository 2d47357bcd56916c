/*
 * uaes_gcm_records.hip -- many short GCM messages under one key in ONE launch (uaesk_gcm_records; the host layer's
 * uaes_gcm_key_encrypt_records / _decrypt_records).  Each record's result is exactly what AES_GCM_encrypt / _decrypt
 * (micro_aes.c:1164-1212) give for it.  GHASH machinery: uaes_ghash.hip.h.
 */
#include "uaes_ghash.hip.h"

/* ------------------------------------------------------------------------ */
/* many short messages under one key: one launch, 1 / 4 / 16 records per workgroup at a time */
/* ------------------------------------------------------------------------ */
/* The GCM counterpart of the XTS sector call ("necessary extensions": a message per API call is
 * 12 us whatever its size, so a stream of TLS records or packets has to arrive as ONE call to run at the
 * GPU's rate).  Record r has its own 12-byte nonce, its own (or the shared) AAD, `rec_len` bytes of text at
 * in + r * in_stride; its output is text || 16-byte tag at out + r * out_stride (decrypt: input is
 * text || tag, output the text -- untouched when the tag is wrong, N7 -- and verdicts[r] = 0 / 0x1A).
 *
 * A workgroup is k_gcm_small's body in a loop: the AES tables and the key's nibble tables are brought into LDS
 * once per workgroup, not once per record.  One record's hash is a chain of DEPENDENT multiplications (the radix-4
 * tree: ~0.3 us per level whatever the width), so a short record leaves the workgroup idle: the 1024 threads are
 * therefore cut into G = 1, 4 or 16 groups of S = 1024, 256 or 64 threads, a record per group -- S is the smallest
 * of the three with room for the record's GHASH positions + one slot for Enc(J0) at two positions per thread --
 * and the G trees run side by
 * side as ONE tree whose first (1024-wide) level is already the groups' own first level (gh_tree_groups).  */
#define GREC_SLOTS     (GT_BUF)                 /* 16 Enc(J0) slots, 16 verdict slots behind the tree buffer */
#define GREC_LDS_TOTAL (GSM_LDS_TOTAL + 32u * 16u + 128u)   /* ... and sixteen live counts, sixteen lengths */

/* gh_tree for 4^(5-lg) independent groups of S = 4^lg consecutive entries each (lg = 5, 4, 3); every group's last
 * `live` entries are not padding.  Group g's hash comes back in the four threads 4g .. 4g+3.            */
__device__ __forceinline__ uint4 gh_tree_groups(uint4 *buf, const uint4 *T, uint4 acc, u32 live, u32 lg, u32 tid,
                                                const u32 *lives = nullptr)
{
    /* lives != nullptr: records of different lengths -- group g's count of non-padding entries is lives[g] (LDS, written
     * before this call; the first barrier below orders it), and a quad looks up the count of the group ITS accumulator
     * belongs to at every level */
    buf[tid] = acc;
    __syncthreads();
    const u32 o = tid >> 2;                            /* the accumulator this quad makes at every level */
    u32 n = 1024u, off = 0;
#pragma unroll
    for (u32 lvl = 1; lvl <= 4; ++lvl) {                       /* tables H^256, H^64, H^16, H^4: m = 4^(5 - lvl) */
        if (lvl + lg < 6u) continue;                           /* the groups are shorter than 4 m (uniform) */
        const u32 m = 1u << (2u * (5u - lvl));
        const uint4 *Tl = T + 512u * lvl;
        if (tid < n) {
            const u32 g = o >> (2u * (5u - lvl)), ql = o & (m - 1u);
            const uint4 *row = buf + off + g * 4u * m + ql;
            u32 lv = lives ? lives[g] : live;
            lv = lv < 4u * m ? lv : 4u * m;
            const u32 k0 = 4u - (lv + m - 1u) / m;             /* first row with a live entry (4: none, the sum is zero) */
            acc = k0 < 4u ? row[k0 * m] : make_uint4(0, 0, 0, 0);
            for (u32 k = k0 + 1; k < 4; ++k) acc = x4(tabmul4q(Tl, acc, tid), row[k * m]);
            if ((tid & 3u) == 0) buf[off + n + o] = acc;
        }
        __syncthreads();
        off += n;
        n >>= 2;
    }
    if (tid < n) {                                     /* n = 4 G: a quad per group, GHASH's last level */
        const uint4 *TF = T + 512u * 5u;
        const uint4 *row = buf + off + 4u * o;
        u32 lv = lives ? lives[o] : live;
        lv = lv < 4u ? lv : 4u;
        acc = make_uint4(0, 0, 0, 0);
        for (u32 k = 4u - lv; k < 4; ++k) acc = tabmul4q(TF, x4(acc, row[k]), tid);
    }
    return acc;
}

template <int NR, bool DEC>
__global__ __launch_bounds__(GH_T) void k_gcm_records(uaesk_rk rk, uaesk_tables tb, const unsigned char *__restrict__ nonces,
                                                      const unsigned char *aad, u64 aad_len, u64 aad_stride,
                                                      const unsigned char *in, u64 rec_len, u64 in_stride,
                                                      unsigned char *out, u64 out_stride, u64 nrec, u32 lg,
                                                      const u32 *__restrict__ lens,
                                                      const unsigned char *__restrict__ scratch,
                                                      unsigned char *verdicts, int *status)
{
    uint4 *TC = (uint4 *)(uaes_lds + GSM_LDS_TAB);
    uint4 *buf = TC + GT_NTAB * 512u;
    uint4 *slots = buf + GREC_SLOTS;
    u32 *lives = (u32 *)(slots + 32);                         /* records of different lengths: a group's live positions ... */
    u32 *rls = lives + 16;                                    /* ... and its record's length */
    {
        const uint4 *g4 = (const uint4 *)(scratch + GS_TAB4);
        for (u32 i = threadIdx.x; i < GT_NTAB * 512u; i += GH_T) TC[i] = g4[i];
    }
    fill_tables64(tb.te0, 0);                                 /* ends with a barrier */
    const LaneConst2 lc = make_lane_const2(0);
    /* rec_len is the longest record (lens != nullptr: record r has min(lens[r], rec_len) bytes); the arrangement --
     * groups, positions per thread -- is the one the longest record needs, a shorter one only has more padding */
    const u64 ablk = (aad_len + 15) >> 4, nv_max = ablk + ((rec_len + 15) >> 4) + 1;
    const u32 G = GH_T >> (2u * lg);                          /* groups = records per turn */
    const u32 grp = threadIdx.x >> (2u * lg);
    /* the nonce of the group's record in the turn that starts at `base` (three little-endian words); fetched one turn
     * ahead, as are a turn's texts before its block encryptions: nothing else hides a load's latency here */
    const bool nonce_words = (((uintptr_t)nonces) & 3u) == 0;
    auto load_nonce = [&](u64 base, u32 (&nw)[3], u32 &rl) {
        const u64 r = base + grp;
        const u64 rr = r < nrec ? r : 0;
        const unsigned char *np = nonces + 12 * rr;
        rl = (u32)rec_len;
        if (lens) { const u32 l = lens[rr]; rl = l < rl ? l : rl; }
        if (nonce_words) {
            nw[0] = ((const u32 *)np)[0]; nw[1] = ((const u32 *)np)[1]; nw[2] = ((const u32 *)np)[2];
        } else {
#pragma unroll
            for (u32 q = 0; q < 3; ++q)
                nw[q] = (u32)np[4 * q] | (u32)np[4 * q + 1] << 8 | (u32)np[4 * q + 2] << 16 | (u32)np[4 * q + 3] << 24;
        }
    };
    u32 nw[3], rl_next;
    load_nonce((u64)blockIdx.x * G, nw, rl_next);
    for (u64 base = (u64)blockIdx.x * G; base < nrec; base += (u64)gridDim.x * G) {
        /* everything the rounds derive from the lane constants (the odd rounds' per-lane key words above all) is made
         * again in every turn: held across the turn it is forty registers that end up in scratch memory */
        LaneConst2 lcv = lc;
        u32 tid = threadIdx.x, lgv = lg;
        asm volatile("" : "+v"(lcv.hmask), "+v"(lcv.t[0]), "+v"(lcv.t[1]), "+v"(lcv.t[2]), "+v"(lcv.t[3]), "+v"(tid), "+s"(lgv));
        const u32 S = 1u << (2u * lgv), t = tid & (S - 1u);    /* group size, position in the group */
        const u32 steps = nv_max + 1 > S ? 2u : 1u;           /* positions per thread (launch_records: nv_max + 1 <= 2 S) */
        const u64 len_r = rl_next;                            /* this turn's record of the group */
        const u64 cblk = (len_r + 15) >> 4, nv = ablk + cblk + 1;
        const u64 pad = (u64)steps * S - nv;                  /* >= 1: the group's position 0 is Enc(J0)'s */
        const u32 live_n = nv < S ? (u32)nv : S;
        const u64 r = base + grp;
        const bool have = r < nrec;                           /* the last turn may leave groups without a record */
        const u64 rr = have ? r : 0;
        const uint4 *rin = (const uint4 *)(in + rr * in_stride);
        uint4 *rout = (uint4 *)(out + rr * out_stride);
        GSrc rest;                                            /* AAD blocks and the length block */
        rest.aad = aad + rr * aad_stride;
        rest.aad_len = aad_len;
        rest.ct = nullptr;
        rest.ct_len = 0;
        rest.has_len = 1;
        rest.len_aad = aad_len;
        rest.len_ct = len_r;
        rest.rev = 0;
        uint4 xk[2] = { make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0) };   /* the position's GHASH input; first the text read */
        bool is_text[2] = { false, false };
        u32 ti[2] = { 0, 0 }, tn[2] = { 0, 0 };                /* text block index, bytes in it */
#pragma unroll                                                /* constant indices: the per-step arrays stay in registers */
        for (u32 k = 0; k < 2; ++k) {
            if (k >= steps) break;
            const u64 u = (u64)k * S + t;
            const bool live = have && u >= pad;
            const u64 v = live ? u - pad : 0;
            is_text[k] = live && v >= ablk && v < ablk + cblk;
            if (is_text[k]) {
                ti[k] = (u32)(v - ablk);
                const u64 avail = len_r - 16ull * ti[k];
                tn[k] = avail < 16 ? (u32)avail : 16u;
                xk[k] = tn[k] == 16 ? rin[ti[k]] : load_bytes_padded((const unsigned char *)(rin + ti[k]), tn[k]);
            } else if (live) {
                xk[k] = load_vblock_fwd(rest, v < ablk ? v : ablk);
            }
        }
        /* J0 = nonce || 00 00 00 01 (GCMsetup, micro_aes.c:1140-1152); text block i takes J0 + 1 + i (N4) */
        uaesk_ctr ctr;
        ctr.w0 = nw[0];
        ctr.w1 = nw[1];
        ctr.w2 = 0; ctr.w3 = 0;
        ctr.b8 = nw[2] & 0xffu;
        const u32 n911 = bswap32(nw[2]) & 0x00ffffffu;        /* nonce bytes 9, 10, 11: the top of the 56-bit counter */
        ctr.v0 = ((u64)n911 << 32) + 2u;
        ctr.le32 = 0;
        const u32 j0w2 = nw[2];
        load_nonce(base + (u64)gridDim.x * G, nw, rl_next);   /* the next turn's */
        uint4 hold[2];
#pragma unroll                                                /* constant indices: the per-step arrays stay in registers */
        for (u32 k = 0; k < 2; ++k) {
            if (k >= steps) break;
            const bool is_j0 = have && k == 0 && t == 0;      /* the group's position 0 is padding: Enc(J0) rides there */
            u32 s1[1][4];
            ctr_words(ctr, ti[k], s1[0]);
            if (is_j0) {                                      /* words 0-1 are the nonce's already */
                s1[0][2] = j0w2;
                s1[0][3] = 0x01000000u;
            }
            if (__builtin_amdgcn_ballot_w64(is_text[k] || is_j0) != 0) enc_blocks<NR, 1>(s1, rk, lcv);
            if (is_j0) slots[grp] = make_uint4(s1[0][0], s1[0][1], s1[0][2], s1[0][3]);
            if (k == 0 && t == 0) { lives[grp] = have ? live_n : 0u; rls[grp] = (u32)len_r; }
            if (is_text[k]) {
                const uint4 d = xk[k];
                const u32 nb = tn[k];
                u32 o[4] = { d.x ^ s1[0][0], d.y ^ s1[0][1], d.z ^ s1[0][2], d.w ^ s1[0][3] };
                if (nb < 16) {
#pragma unroll
                    for (u32 w = 0; w < 4; ++w) {
                        const u32 keep = nb >= 4 * w + 4 ? 0xffffffffu : nb <= 4 * w ? 0u : (1u << (8 * (nb - 4 * w))) - 1u;
                        o[w] &= keep;
                    }
                }
                const uint4 ov = make_uint4(o[0], o[1], o[2], o[3]);
                if (DEC) {
                    hold[k] = ov;                             /* GHASH takes the ciphertext read (xk[k] stays) */
                } else {
                    xk[k] = ov;
                    if (nb == 16) {
                        rout[ti[k]] = ov;
                    } else {
                        unsigned char *dst = (unsigned char *)(rout + ti[k]);
                        for (u32 b = 0; b < nb; ++b) dst[b] = (unsigned char)(o[b >> 2] >> (8 * (b & 3)));
                    }
                }
            }
        }
        uint4 acc = xk[0];
        if (steps == 2) acc = x4(tabmul4(TC + 512u * (5u - lgv), acc, tid), xk[1]);   /* by H^S: tables H^1024, H^256, H^64 */
        acc = gh_tree_groups(buf, TC, acc, live_n, lgv, tid, lens ? lives : nullptr);       /* group q's hash: threads 4q .. 4q+3 */
        const u32 q = tid >> 2;
        if ((tid & 3u) == 0 && q < G && base + q < nrec) {
            const u64 rq = base + q;
            acc = x4(acc, slots[q]);
            const u32 w[4] = { acc.x, acc.y, acc.z, acc.w };
            if (DEC) {
                const unsigned char *tag = in + rq * in_stride + rls[q];
                u32 diff = 0;
                for (u32 b = 0; b < 16; ++b) diff |= (u32)tag[b] ^ ((w[b >> 2] >> (8 * (b & 3))) & 0xffu);
                if (verdicts) verdicts[rq] = diff ? 0x1A : 0;
                if (diff) atomicOr(status, 0x1A);
                slots[16 + q] = make_uint4(diff, 0, 0, 0);
            } else {
                unsigned char *tag = out + rq * out_stride + rls[q];
                for (u32 b = 0; b < 16; ++b) tag[b] = (unsigned char)(w[b >> 2] >> (8 * (b & 3)));
            }
        }
        if (DEC) {
            __syncthreads();
            if (have && slots[16 + grp].x == 0) {
#pragma unroll
                for (u32 k = 0; k < 2; ++k) {
                    if (!is_text[k]) continue;
                    if (tn[k] == 16) {
                        rout[ti[k]] = hold[k];
                    } else {
                        const u32 o[4] = { hold[k].x, hold[k].y, hold[k].z, hold[k].w };
                        unsigned char *dst = (unsigned char *)(rout + ti[k]);
                        for (u32 b = 0; b < tn[k]; ++b) dst[b] = (unsigned char)(o[b >> 2] >> (8 * (b & 3)));
                    }
                }
            }
        }
        __syncthreads();                                      /* the Enc(J0) and verdict slots are rewritten by the next turn */
    }
}

template <int NR>
static int launch_records(hipStream_t st, const uaesk_tables *tb, const uaesk_rk *ek, int decrypt, const void *nonces,
                          const void *aad, size_t aad_len, size_t aad_stride, const void *in, size_t rec_len,
                          size_t in_stride, void *out, size_t out_stride, size_t nrec, const void *sc,
                          unsigned char *verdicts, int *status, const void *lens)
{
    int cus = 0;
    uaesk_device_info(&cus, nullptr);
    const u64 nv = (aad_len + 15) / 16 + (rec_len + 15) / 16 + 1;
    const u32 lg = nv + 1 <= 128 ? 3u : nv + 1 <= 512 ? 4u : 5u;   /* one or two positions per thread */
    const u64 turns = (nrec + (GH_T >> (2 * lg)) - 1) / (GH_T >> (2 * lg));
    const u64 cap = cus > 0 ? (u64)cus : 256u;
    const unsigned grid = (unsigned)(turns < cap ? turns : cap);
    hipError_t e;
    if (decrypt) {
        e = uaesk_want_lds((const void *)k_gcm_records<NR, true>, (unsigned)GREC_LDS_TOTAL);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL((k_gcm_records<NR, true>), dim3(grid), dim3(GH_T), GREC_LDS_TOTAL, st, *ek, *tb,
                           (const unsigned char *)nonces, (const unsigned char *)aad, (u64)aad_len, (u64)aad_stride,
                           (const unsigned char *)in, (u64)rec_len, (u64)in_stride, (unsigned char *)out, (u64)out_stride,
                           (u64)nrec, lg, (const u32 *)lens, (const unsigned char *)sc, verdicts, status);
    } else {
        e = uaesk_want_lds((const void *)k_gcm_records<NR, false>, (unsigned)GREC_LDS_TOTAL);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL((k_gcm_records<NR, false>), dim3(grid), dim3(GH_T), GREC_LDS_TOTAL, st, *ek, *tb,
                           (const unsigned char *)nonces, (const unsigned char *)aad, (u64)aad_len, (u64)aad_stride,
                           (const unsigned char *)in, (u64)rec_len, (u64)in_stride, (unsigned char *)out, (u64)out_stride,
                           (u64)nrec, lg, (const u32 *)lens, (const unsigned char *)sc, verdicts, status);
    }
    return (int)hipGetLastError();
}

/* records of up to uaesk_gcm_record_max(aad_len) bytes; texts and strides 16-byte aligned; decrypt: *status must be
 * zero when the kernel starts (it ORs 0x1A in), verdicts may be NULL */
extern "C" size_t uaesk_gcm_record_max(size_t aad_len)
{
    const size_t ablk = (aad_len + 15) / 16;
    return ablk + 2 > GSM_MAXNV ? 0 : (GSM_MAXNV - 1 - ablk) * 16;
}

extern "C" int uaesk_gcm_records(void *stream, const uaesk_tables *tb, int nr, const uaesk_rk *ek, int decrypt,
                                 const void *nonces12, const void *aad, size_t aad_len, size_t aad_stride,
                                 const void *in, size_t rec_len, size_t in_stride, void *out, size_t out_stride,
                                 size_t nrec, const void *key_scratch, unsigned char *verdicts, int *status,
                                 const void *lens)
{
    if (!nrec) return 0;
    if (((aad_len + 15) / 16) + ((rec_len + 15) / 16) + 1 > GSM_MAXNV) return (int)hipErrorInvalidValue;
    if ((((uintptr_t)in | (uintptr_t)out | in_stride | out_stride) & 15u) != 0) return (int)hipErrorInvalidValue;
    if (decrypt && !status) return (int)hipErrorInvalidValue;
    if (((uintptr_t)lens) & 3u) return (int)hipErrorInvalidValue;
    switch (nr) {
    case 10: return launch_records<10>(S(stream), tb, ek, decrypt, nonces12, aad, aad_len, aad_stride, in, rec_len, in_stride,
                                       out, out_stride, nrec, key_scratch, verdicts, status, lens);
    case 12: return launch_records<12>(S(stream), tb, ek, decrypt, nonces12, aad, aad_len, aad_stride, in, rec_len, in_stride,
                                       out, out_stride, nrec, key_scratch, verdicts, status, lens);
    case 14: return launch_records<14>(S(stream), tb, ek, decrypt, nonces12, aad, aad_len, aad_stride, in, rec_len, in_stride,
                                       out, out_stride, nrec, key_scratch, verdicts, status, lens);
    default: return (int)hipErrorInvalidValue;
    }
}
