/*
 * uaes_kernels.hip -- gfx950 kernels for the block-parallel AES modes and
 * their C launchers (interface: uaes_device.h).
 *
 *   k_ecb   <- AES_ECB_encrypt/decrypt loops      micro_aes.c:644-651, :671-674
 *   k_ctr   <- CTR_cipher loop + incBlock         micro_aes.c:943-949, :421-427
 *   k_xts*  <- XTS_cipher + doubleLblock          micro_aes.c:1008-1055, :449-458
 *
 * Launch shape: persistent workgroups of 1024 threads (16 waves), one per CU
 * (the replicated tables take 96-128 KiB of the CU's 160 KiB LDS), grid-stride
 * over 16-byte blocks with 128-bit coalesced global loads/stores: lane l of a
 * workgroup touches block base + u*1024 + l, so every wave-level access is one
 * contiguous 1 KiB segment.  U independent blocks per lane give the LDS
 * pipeline enough parallel lookups to hide its latency.
 */
#include <hip/hip_runtime.h>
#include <string.h>
#include "uaes_aes.hip.h"
#include "uaes_gf.h"
#include "uaes_device.h"

#define UAES_U 4            /* blocks per lane per iteration */

static inline hipStream_t S(void *s) { return (hipStream_t)s; }

/* ------------------------------------------------------------------------ */
/* ECB                                                                        */
/* ------------------------------------------------------------------------ */
template <int NR, bool DEC>
__global__ __launch_bounds__(UAES_WG) void k_ecb(uaesk_rk rk, uaesk_tables tb,
                                                 const uint4 *__restrict__ in, uint4 *__restrict__ out,
                                                 u64 nfull, u32 rem)
{
    if (DEC) fill_dec_tables(tb.td0, tb.si4); else fill_enc_tables(tb.te0);
    const LaneConst lc = make_lane_const();
    const u64 stride = (u64)gridDim.x * UAES_WG * UAES_U;

    for (u64 base = (u64)blockIdx.x * UAES_WG * UAES_U; base < nfull; base += stride) {
        u32 s[UAES_U][4];
        u64 idx[UAES_U];
#pragma unroll
        for (int u = 0; u < UAES_U; ++u) {
            idx[u] = base + (u64)u * UAES_WG + threadIdx.x;
            uint4 d = make_uint4(0, 0, 0, 0);
            if (idx[u] < nfull) d = in[idx[u]];
            s[u][0] = d.x; s[u][1] = d.y; s[u][2] = d.z; s[u][3] = d.w;
        }
        if (DEC) dec_blocks<NR, UAES_U>(s, rk, lc); else enc_blocks<NR, UAES_U>(s, rk, lc);
#pragma unroll
        for (int u = 0; u < UAES_U; ++u)
            if (idx[u] < nfull) out[idx[u]] = make_uint4(s[u][0], s[u][1], s[u][2], s[u][3]);
    }

    /* reference N1: a trailing partial block is zero padded and ENCRYPTED into
     * a full output block (micro_aes.c:648-651); decrypt never gets here with
     * rem != 0 (host copies the ragged tail through and reports 0x1D).       */
    if (!DEC && rem && blockIdx.x == 0 && threadIdx.x == 0) {
        const unsigned char *src = (const unsigned char *)(in + nfull);
        unsigned char pad[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) pad[i] = (u32)i < rem ? src[i] : 0;
        u32 s1[1][4];
#pragma unroll
        for (int w = 0; w < 4; ++w)
            s1[0][w] = pad[4 * w] | (pad[4 * w + 1] << 8) | (pad[4 * w + 2] << 16) | ((u32)pad[4 * w + 3] << 24);
        enc_blocks<NR, 1>(s1, rk, lc);
        out[nfull] = make_uint4(s1[0][0], s1[0][1], s1[0][2], s1[0][3]);
    }
}

/* ------------------------------------------------------------------------ */
/* CTR                                                                        */
/* ------------------------------------------------------------------------ */
/* counter block for stream block i: bytes 0..8 fixed, bytes 9..15 = 56-bit
 * big-endian (v0 + i) mod 2^56 (reference N2)                               */
__device__ __forceinline__ void ctr_words(const uaesk_ctr &c, u64 i, u32 (&w)[4])
{
    const u64 v = (c.v0 + i) & 0x00FFFFFFFFFFFFFFull;
    w[0] = c.w0;
    w[1] = c.w1;
    w[2] = bswap32((c.b8 << 24) | (u32)(v >> 32));
    w[3] = bswap32((u32)v);
}

template <int NR>
__global__ __launch_bounds__(UAES_WG) void k_ctr(uaesk_rk rk, uaesk_tables tb, uaesk_ctr ctr,
                                                 const uint4 *__restrict__ in, uint4 *__restrict__ out,
                                                 u64 nfull, u32 rem, const int *__restrict__ gate)
{
    if (gate && *gate != 0) return;            /* GCM decrypt: tag mismatch -> untouched */
    fill_enc_tables(tb.te0);
    const LaneConst lc = make_lane_const();
    const u64 stride = (u64)gridDim.x * UAES_WG * UAES_U;

    for (u64 base = (u64)blockIdx.x * UAES_WG * UAES_U; base < nfull; base += stride) {
        u32 s[UAES_U][4];
        uint4 d[UAES_U];
        u64 idx[UAES_U];
#pragma unroll
        for (int u = 0; u < UAES_U; ++u) {
            idx[u] = base + (u64)u * UAES_WG + threadIdx.x;
            d[u] = make_uint4(0, 0, 0, 0);
            if (idx[u] < nfull) d[u] = in[idx[u]];
            ctr_words(ctr, idx[u], s[u]);
        }
        enc_blocks<NR, UAES_U>(s, rk, lc);
#pragma unroll
        for (int u = 0; u < UAES_U; ++u)
            if (idx[u] < nfull)
                out[idx[u]] = make_uint4(d[u].x ^ s[u][0], d[u].y ^ s[u][1], d[u].z ^ s[u][2], d[u].w ^ s[u][3]);
    }

    /* reference N3: len%16 tail bytes use Enc(ctr_final) (mixThenXor, :949) */
    if (rem && blockIdx.x == 0 && threadIdx.x == 0) {
        u32 s1[1][4];
        ctr_words(ctr, nfull, s1[0]);
        enc_blocks<NR, 1>(s1, rk, lc);
        const unsigned char *src = (const unsigned char *)(in + nfull);
        unsigned char *dst = (unsigned char *)(out + nfull);
        for (u32 i = 0; i < rem; ++i)
            dst[i] = src[i] ^ (unsigned char)(s1[0][i >> 2] >> (8 * (i & 3)));
    }
}

/* ------------------------------------------------------------------------ */
/* XTS                                                                        */
/* ------------------------------------------------------------------------ */
/* tweak arithmetic (T * alpha^k by shifting): uaes_gf.h, struct Tw          */
#define XTS_CHUNK 256u      /* blocks per chunk = one wave x 4 blocks per lane */

/* pre-pass: one thread per data unit.  T0 = Enc_key2(tweak) (:1026-1027),
 * then the tweak at the start of every 256-block chunk of the unit.         */
template <int NR>
__global__ __launch_bounds__(UAES_WG) void k_xts_tweaks(uaesk_rk k2, uaesk_tables tb,
                                                        uint4 raw_tweak, u32 use_raw, u64 first_sector,
                                                        u64 nsectors, u64 chunks_per_sector,
                                                        uint4 *__restrict__ chunk_tw)
{
    fill_enc_tables(tb.te0);
    const LaneConst lc = make_lane_const();
    const u64 stride = (u64)gridDim.x * UAES_WG;
    for (u64 sct = (u64)blockIdx.x * UAES_WG + threadIdx.x; sct < nsectors; sct += stride) {
        u32 s[1][4];
        if (use_raw) {
            s[0][0] = raw_tweak.x; s[0][1] = raw_tweak.y; s[0][2] = raw_tweak.z; s[0][3] = raw_tweak.w;
        } else {
            const u64 id = first_sector + sct;  /* copyLint, micro_aes.c:399-404 */
            s[0][0] = (u32)id; s[0][1] = (u32)(id >> 32); s[0][2] = 0; s[0][3] = 0;
        }
        enc_blocks<NR, 1>(s, k2, lc);
        Tw t;
        t.lo = s[0][0] | ((u64)s[0][1] << 32);
        t.hi = s[0][2] | ((u64)s[0][3] << 32);
        for (u64 c = 0; c < chunks_per_sector; ++c) {
            chunk_tw[sct * chunks_per_sector + c] =
                make_uint4((u32)t.lo, (u32)(t.lo >> 32), (u32)t.hi, (u32)(t.hi >> 32));
            t = tw_mul_pow64(tw_mul_pow64(tw_mul_pow64(tw_mul_pow64(t))));
        }
    }
}

/* main pass: one wave per chunk; lane l handles blocks l, l+64, l+128, l+192
 * of the chunk so each wave-level load/store is a contiguous 1 KiB segment. */
template <int NR, bool DEC>
__global__ __launch_bounds__(UAES_WG) void k_xts(uaesk_rk k1, uaesk_tables tb,
                                                 const uint4 *__restrict__ chunk_tw,
                                                 u64 nsectors, u64 chunks_per_sector,
                                                 u64 main_blocks,      /* whole blocks handled here, per unit */
                                                 u64 sector_bytes,
                                                 const unsigned char *__restrict__ in,
                                                 unsigned char *__restrict__ out)
{
    if (DEC) fill_dec_tables(tb.td0, tb.si4); else fill_enc_tables(tb.te0);
    const LaneConst lc = make_lane_const();
    const u32 lane = threadIdx.x & 63u;
    const u64 wave = (u64)blockIdx.x * (UAES_WG / 64) + (threadIdx.x >> 6);
    const u64 nwaves = (u64)gridDim.x * (UAES_WG / 64);
    const u64 nchunks = nsectors * chunks_per_sector;

    for (u64 ch = wave; ch < nchunks; ch += nwaves) {
        const u64 sct = ch / chunks_per_sector;
        const u64 c = ch - sct * chunks_per_sector;
        const u64 first = c * XTS_CHUNK;                       /* first block of chunk in unit */
        const u64 left = main_blocks - first;
        const u32 cnt = left < XTS_CHUNK ? (u32)left : XTS_CHUNK;
        const uint4 tb4 = chunk_tw[ch];
        Tw t;
        t.lo = tb4.x | ((u64)tb4.y << 32);
        t.hi = tb4.z | ((u64)tb4.w << 32);
        t = tw_mul_pow(t, lane);

        const uint4 *src = (const uint4 *)(in + sct * sector_bytes) + first;
        uint4 *dst = (uint4 *)(out + sct * sector_bytes) + first;
        u32 s[UAES_U][4], tw[UAES_U][4];
#pragma unroll
        for (int u = 0; u < UAES_U; ++u) {
            const u32 j = lane + 64u * u;
            uint4 d = make_uint4(0, 0, 0, 0);
            if (j < cnt) d = src[j];
            tw[u][0] = (u32)t.lo; tw[u][1] = (u32)(t.lo >> 32);
            tw[u][2] = (u32)t.hi; tw[u][3] = (u32)(t.hi >> 32);
            s[u][0] = d.x ^ tw[u][0]; s[u][1] = d.y ^ tw[u][1];
            s[u][2] = d.z ^ tw[u][2]; s[u][3] = d.w ^ tw[u][3];
            t = tw_mul_pow64(t);
        }
        if (DEC) dec_blocks<NR, UAES_U>(s, k1, lc); else enc_blocks<NR, UAES_U>(s, k1, lc);
#pragma unroll
        for (int u = 0; u < UAES_U; ++u) {
            const u32 j = lane + 64u * u;
            if (j < cnt)
                dst[j] = make_uint4(s[u][0] ^ tw[u][0], s[u][1] ^ tw[u][1], s[u][2] ^ tw[u][2], s[u][3] ^ tw[u][3]);
        }
    }
}

/* ciphertext stealing (micro_aes.c:1037-1053): one thread per data unit handles
 * the last whole block m and the r-byte tail.  All inputs are read before any
 * output is written, so in == out is fine.                                  */
template <int NR, bool DEC>
__global__ __launch_bounds__(UAES_WG) void k_xts_cts(uaesk_rk k1, uaesk_tables tb,
                                                     const uint4 *__restrict__ chunk_tw,
                                                     u64 nsectors, u64 chunks_per_sector,
                                                     u64 m, u32 r, u64 sector_bytes,
                                                     const unsigned char *__restrict__ in,
                                                     unsigned char *__restrict__ out)
{
    if (DEC) fill_dec_tables(tb.td0, tb.si4); else fill_enc_tables(tb.te0);
    const LaneConst lc = make_lane_const();
    const u64 stride = (u64)gridDim.x * UAES_WG;
    for (u64 sct = (u64)blockIdx.x * UAES_WG + threadIdx.x; sct < nsectors; sct += stride) {
        /* tweak of block m: chunk base * alpha^(m mod 256) */
        const uint4 tb4 = chunk_tw[sct * chunks_per_sector + m / XTS_CHUNK];
        Tw tm;
        tm.lo = tb4.x | ((u64)tb4.y << 32);
        tm.hi = tb4.z | ((u64)tb4.w << 32);
        const u32 off = (u32)(m % XTS_CHUNK);
        tm = tw_mul_pow(tm, off & 63u);
        for (u32 q = 0; q < (off >> 6); ++q) tm = tw_mul_pow64(tm);
        const Tw tn = tw_mul_pow(tm, 1);
        /* encrypt: block m uses T_m, the stolen block T_{m+1}; decrypt swaps (:1041) */
        const Tw ta = DEC ? tn : tm, tbk = DEC ? tm : tn;

        const unsigned char *src = in + sct * sector_bytes + m * 16;
        unsigned char *dst = out + sct * sector_bytes + m * 16;
        const uint4 d = *(const uint4 *)src;
        unsigned char tail[16];
        for (u32 i = 0; i < r; ++i) tail[i] = src[16 + i];

        u32 s[1][4];
        s[0][0] = d.x ^ (u32)ta.lo; s[0][1] = d.y ^ (u32)(ta.lo >> 32);
        s[0][2] = d.z ^ (u32)ta.hi; s[0][3] = d.w ^ (u32)(ta.hi >> 32);
        if (DEC) dec_blocks<NR, 1>(s, k1, lc); else enc_blocks<NR, 1>(s, k1, lc);
        u32 cc[4] = { s[0][0] ^ (u32)ta.lo, s[0][1] ^ (u32)(ta.lo >> 32),
                      s[0][2] ^ (u32)ta.hi, s[0][3] ^ (u32)(ta.hi >> 32) };
        /* second block = tail bytes followed by cc[r..16) */
        u32 p[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            u32 v = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const u32 i = 4 * w + b;
                const u32 byte = i < r ? tail[i] : (cc[w] >> (8 * b)) & 0xffu;
                v |= byte << (8 * b);
            }
            p[w] = v;
        }
        s[0][0] = p[0] ^ (u32)tbk.lo; s[0][1] = p[1] ^ (u32)(tbk.lo >> 32);
        s[0][2] = p[2] ^ (u32)tbk.hi; s[0][3] = p[3] ^ (u32)(tbk.hi >> 32);
        if (DEC) dec_blocks<NR, 1>(s, k1, lc); else enc_blocks<NR, 1>(s, k1, lc);
        *(uint4 *)dst = make_uint4(s[0][0] ^ (u32)tbk.lo, s[0][1] ^ (u32)(tbk.lo >> 32),
                                   s[0][2] ^ (u32)tbk.hi, s[0][3] ^ (u32)(tbk.hi >> 32));
        for (u32 i = 0; i < r; ++i) dst[16 + i] = (unsigned char)(cc[i >> 2] >> (8 * (i & 3)));
    }
}

/* ------------------------------------------------------------------------ */
/* self test of the primitives (diagnostics for a fresh box)                  */
/* ------------------------------------------------------------------------ */
__global__ __launch_bounds__(UAES_WG) void k_selftest(uaesk_rk ek, uaesk_rk dk, uaesk_tables tb,
                                                      unsigned *result)
{
    unsigned bad = 0;
    /* v_perm_b32 byte numbering: selector n picks byte n of {S0,S1}, S1 low */
    if (__builtin_amdgcn_perm(0x44332211u, 0x88776655u, 0x07040300u) != 0x44118855u) bad |= 1;
    if (__builtin_amdgcn_perm(0x44332211u, 0x88776655u, 0x0c0c0c0cu) != 0u) bad |= 2;
    if (bswap32(0x11223344u) != 0x44332211u) bad |= 4;
    if (rotl32(0x80000001u, 8) != 0x00000180u) bad |= 8;
    /* the table lookups address LDS absolutely: dynamic segment must start at 0 */
    if ((u32)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)uaes_lds != 0u) bad |= 256;
    if (xor3(0xf0f0f0f0u, 0xccccccccu, 0xaaaaaaaau) != 0x96969696u) bad |= 512;
    if (or_xor(0xf0f0f0f0u, 0xccccccccu, 0xaaaaaaaau) != 0x56565656u) bad |= 1024;

    fill_enc_tables(tb.te0);
    const LaneConst lc = make_lane_const();
    /* FIPS-197 appendix C.1: key 00..0f, pt 00112233..ff */
    u32 s[1][4] = { { 0x33221100u, 0x77665544u, 0xbbaa9988u, 0xffeeddccu } };
    enc_blocks<10, 1>(s, ek, lc);
    if (s[0][0] != 0xd8e0c469u || s[0][1] != 0x30047b6au || s[0][2] != 0x80b7cdd8u || s[0][3] != 0x5ac5b470u) bad |= 16;
    __syncthreads();
    fill_dec_tables(tb.td0, tb.si4);
    dec_blocks<10, 1>(s, dk, lc);
    if (s[0][0] != 0x33221100u || s[0][1] != 0x77665544u || s[0][2] != 0xbbaa9988u || s[0][3] != 0xffeeddccu) bad |= 32;
    /* XTS tweak arithmetic: alpha^k by shifting == k doublings */
    Tw t = { 0x8000000000000001ull * (threadIdx.x + 1), 0xC3A5C85C97CB3127ull ^ threadIdx.x };
    Tw a = t;
    const u32 k = threadIdx.x & 63u;
    for (u32 i = 0; i < k; ++i) a = tw_mul_pow(a, 1);
    Tw b = tw_mul_pow(t, k);
    if (a.lo != b.lo || a.hi != b.hi) bad |= 64;
    a = t;
    for (u32 i = 0; i < 64; ++i) a = tw_mul_pow(a, 1);
    b = tw_mul_pow64(t);
    if (a.lo != b.lo || a.hi != b.hi) bad |= 128;
    if (bad) atomicOr(result, bad);
}

/* ------------------------------------------------------------------------ */
/* launchers                                                                  */
/* ------------------------------------------------------------------------ */
static int g_cus = 0;

extern "C" int uaesk_device_info(int *cu_count, int *lds_bytes)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    hipDeviceProp_t p;
    e = hipGetDeviceProperties(&p, dev);
    if (e != hipSuccess) return (int)e;
    g_cus = p.multiProcessorCount;
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (lds_bytes) *lds_bytes = (int)p.maxSharedMemoryPerMultiProcessor;
    return 0;
}

static unsigned grid_for(u64 work_items, u64 per_wg)
{
    if (g_cus <= 0) uaesk_device_info(nullptr, nullptr);
    u64 want = (work_items + per_wg - 1) / per_wg;
    if (want < 1) want = 1;
    const u64 cap = (u64)(g_cus > 0 ? g_cus : 256);
    return (unsigned)(want < cap ? want : cap);
}

template <typename K>
static hipError_t set_lds(K kern, unsigned bytes)
{
    return hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

#define DISPATCH_NR(nr, CALL)                         \
    switch (nr) {                                     \
    case 10: { constexpr int NR = 10; CALL; } break;  \
    case 12: { constexpr int NR = 12; CALL; } break;  \
    case 14: { constexpr int NR = 14; CALL; } break;  \
    default: return (int)hipErrorInvalidValue;        \
    }

template <int NR, bool DEC>
static int launch_ecb(hipStream_t st, const uaesk_tables *tb, const uaesk_rk *keys,
                      const void *in, void *out, size_t nfull, unsigned rem)
{
    const unsigned lds = DEC ? UAES_LDS_DEC : UAES_LDS_ENC;
    hipError_t e = set_lds(k_ecb<NR, DEC>, lds);
    if (e != hipSuccess) return (int)e;
    const unsigned grid = grid_for(nfull + (rem ? 1 : 0), (u64)UAES_WG * UAES_U);
    hipLaunchKernelGGL((k_ecb<NR, DEC>), dim3(grid), dim3(UAES_WG), lds, st, *keys, *tb,
                       (const uint4 *)in, (uint4 *)out, (u64)nfull, (u32)rem);
    return (int)hipGetLastError();
}

extern "C" int uaesk_ecb(void *stream, const uaesk_tables *tb, int nr, const uaesk_rk *keys,
                         int decrypt, const void *in, void *out, size_t nfull, unsigned rem)
{
    if (nfull == 0 && rem == 0) return 0;
    if (decrypt) { DISPATCH_NR(nr, return (launch_ecb<NR, true>(S(stream), tb, keys, in, out, nfull, 0))); }
    else         { DISPATCH_NR(nr, return (launch_ecb<NR, false>(S(stream), tb, keys, in, out, nfull, rem))); }
    return 0;
}

template <int NR>
static int launch_ctr(hipStream_t st, const uaesk_tables *tb, const uaesk_rk *ek, const uaesk_ctr *ctr,
                      const void *in, void *out, size_t len, const int *gate)
{
    hipError_t e = set_lds(k_ctr<NR>, UAES_LDS_ENC);
    if (e != hipSuccess) return (int)e;
    const u64 nfull = len / 16;
    const u32 rem = (u32)(len % 16);
    const unsigned grid = grid_for(nfull + (rem ? 1 : 0), (u64)UAES_WG * UAES_U);
    hipLaunchKernelGGL((k_ctr<NR>), dim3(grid), dim3(UAES_WG), UAES_LDS_ENC, st, *ek, *tb, *ctr,
                       (const uint4 *)in, (uint4 *)out, nfull, rem, gate);
    return (int)hipGetLastError();
}

extern "C" int uaesk_ctr_xcrypt(void *stream, const uaesk_tables *tb, int nr, const uaesk_rk *ek,
                                const uaesk_ctr *ctr, const void *in, void *out, size_t len,
                                const int *gate)
{
    if (len == 0) return 0;
    DISPATCH_NR(nr, return (launch_ctr<NR>(S(stream), tb, ek, ctr, in, out, len, gate)));
    return 0;
}

static void xts_geometry(size_t sector_bytes, u64 *main_blocks, u32 *rem, u64 *cps)
{
    const u32 r = (u32)(sector_bytes % 16);
    const u64 whole = sector_bytes / 16;
    const u64 mb = whole - (r ? 1 : 0);        /* blocks before the stealing pair */
    u64 c = (whole + XTS_CHUNK - 1) / XTS_CHUNK;   /* covers block m as well */
    if (c == 0) c = 1;
    *main_blocks = mb; *rem = r; *cps = c;
}

extern "C" size_t uaesk_xts_scratch_bytes(size_t sector_bytes, size_t nsectors)
{
    u64 mb, cps; u32 r;
    xts_geometry(sector_bytes, &mb, &r, &cps);
    return (size_t)(cps * nsectors * 16);
}

template <int NR, bool DEC>
static int launch_xts(hipStream_t st, const uaesk_tables *tb, const uaesk_rk *k1, const uaesk_rk *k2,
                      const uint8_t *tweak16, u64 first_sector, size_t sector_bytes, size_t nsectors,
                      const void *in, void *out, void *scratch)
{
    u64 mb, cps; u32 r;
    xts_geometry(sector_bytes, &mb, &r, &cps);
    const unsigned lds = DEC ? UAES_LDS_DEC : UAES_LDS_ENC;
    hipError_t e = set_lds(k_xts_tweaks<NR>, UAES_LDS_ENC);
    if (e == hipSuccess) e = set_lds(k_xts<NR, DEC>, lds);
    if (e == hipSuccess) e = set_lds(k_xts_cts<NR, DEC>, lds);
    if (e != hipSuccess) return (int)e;

    uint4 raw = make_uint4(0, 0, 0, 0);
    if (tweak16) memcpy(&raw, tweak16, 16);
    hipLaunchKernelGGL((k_xts_tweaks<NR>), dim3(grid_for(nsectors, UAES_WG)), dim3(UAES_WG), UAES_LDS_ENC, st,
                       *k2, *tb, raw, (u32)(tweak16 != nullptr), first_sector, (u64)nsectors, cps,
                       (uint4 *)scratch);
    if (mb > 0) {
        const u64 nchunks = (u64)nsectors * ((mb + XTS_CHUNK - 1) / XTS_CHUNK);
        hipLaunchKernelGGL((k_xts<NR, DEC>), dim3(grid_for(nchunks, UAES_WG / 64)), dim3(UAES_WG), lds, st,
                           *k1, *tb, (const uint4 *)scratch, (u64)nsectors, cps, mb, (u64)sector_bytes,
                           (const unsigned char *)in, (unsigned char *)out);
    }
    if (r) {
        hipLaunchKernelGGL((k_xts_cts<NR, DEC>), dim3(grid_for(nsectors, UAES_WG)), dim3(UAES_WG), lds, st,
                           *k1, *tb, (const uint4 *)scratch, (u64)nsectors, cps, mb, r, (u64)sector_bytes,
                           (const unsigned char *)in, (unsigned char *)out);
    }
    return (int)hipGetLastError();
}

extern "C" int uaesk_xts(void *stream, const uaesk_tables *tb, int nr,
                         const uaesk_rk *k1, const uaesk_rk *k2_enc, int decrypt,
                         const uint8_t *tweak16, uint64_t first_sector,
                         size_t sector_bytes, size_t nsectors,
                         const void *in, void *out, void *scratch)
{
    if (nsectors == 0) return 0;
    if (sector_bytes < 16) return (int)hipErrorInvalidValue;
    if (decrypt) { DISPATCH_NR(nr, return (launch_xts<NR, true>(S(stream), tb, k1, k2_enc, tweak16, first_sector, sector_bytes, nsectors, in, out, scratch))); }
    else         { DISPATCH_NR(nr, return (launch_xts<NR, false>(S(stream), tb, k1, k2_enc, tweak16, first_sector, sector_bytes, nsectors, in, out, scratch))); }
    return 0;
}

extern "C" int uaesk_selftest(void *stream, const uaesk_tables *tb, const uaesk_rk *ek128,
                              const uaesk_rk *dk128, unsigned *d_result)
{
    hipError_t e = set_lds(k_selftest, UAES_LDS_ENC);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_selftest, dim3(2), dim3(UAES_WG), UAES_LDS_ENC, S(stream), *ek128, *dk128, *tb, d_result);
    return (int)hipGetLastError();
}
