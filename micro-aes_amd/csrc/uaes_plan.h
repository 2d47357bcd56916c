/* uaes_plan.h -- the ONE table of arrangements: which kernels a call runs, by mode, direction and size.
 *
 * Every launcher of the kernel layer asks a planner of this file's family (uaesk_plan_ecb / _ctr / _xts / _gcm /
 * _ocb / _siv, defined next to the kernels they plan for) and switches on the answer; uaes_debug_plan() in the C ABI
 * (include/uaes_hip.h) returns the same answer as data, and tests/test_gpu_plan.py derives its parity cases from it:
 * it walks the sizes, finds every boundary b at which the answer changes and checks b - 16, b, b + 16 in both
 * directions (and a forged tag) against the oracle.  There is no other place where a size threshold decides what
 * runs, and no threshold is read from the environment.
 *
 *   mode  arrangement           kernels (launches)                                         reached when
 *   ----  --------------------  ---------------------------------------------------------  ---------------------------------
 *   ECB   ECB_SINGLE            k_ecb<U=1>                                          (1)    < half the CUs' worth of 4-block tiles
 *         ECB_TILED             k_ecb<U=4>                                          (1)    otherwise
 *   CTR   CTR_SINGLE            k_ctr<U=1>                                          (1)    < half the CUs' worth of tiles
 *         CTR_QUAD              k_ctr<U=4>                                          (1)    < one grid of 8-group stripes, or LE32 counter
 *         CTR_STRIPED           k_ctr_shared2 (rounds 1-2 shared per 256 counters)  (1)    >= one grid of stripes (8 MiB on 256 CUs)
 *   XTS   XTS_SMALL             k_xts_small [+ k_xts_cts]                           (1-2)  one unit <= 8 MiB, or <= 4 MiB of whole-block units
 *         XTS_PACKED            k_xts_tweaks + k_xts<PACKED>                        (2)    units shorter than a chunk, whole blocks
 *         XTS_BULK              k_xts_tweaks [+ k_xts_expand] + k_xts [+ k_xts_cts] (2-4)  everything else (C3: 2^20 sectors of 4 KiB)
 *   GCM   GCM_SMALL             k_gcm_small                                         (1)    <= 2046 GHASH positions (~32 KiB)
 *         GCM_CHUNKS            k_gcm_chunks<FOLD> [+ gated k_ctr*]                 (1-2)  encrypt <= 16 MiB; tag-first decrypt <= 512 MiB
 *                               (k_gcm_chunks + k_gcm_combine without a counter word or for a one-pass decrypt)
 *         GCM_TWOPHASE          k_ctr_shared2, then hash-only k_gcm_chunks<FOLD>    (2)    encrypt / one-pass decrypt, 16 .. 128 MiB
 *         GCM_STRIPED           k_gcm_setup | k_gcm_ej0, k_gcm_fused, k_ghash_final (3)    encrypt / one-pass decrypt beyond (C4: 1 GiB)
 *         GCM_LEVELS            k_gcm_setup, k_ctr*, k_ghash_pass x0-2, k_ghash_final (3-5) whatever is left (tag-only; > 512 MiB tag-first)
 *   OCB   OCB_SMALL             k_ocb_small                                         (1)    <= OCB_SMALL_BLOCKS blocks and short AAD
 *         OCB_RUNS              k_ocb (last workgroup to arrive makes the tag)      (1)    otherwise
 *   SIV   SIV_SMALL             k_siv_small                                         (1)    <= 2046 POLYVAL positions
 *         SIV_CHUNKS            k_siv_prep, hash-only k_gcm_chunks<FOLD>, k_ctr*    (3)    <= 512 MiB
 *         SIV_LEVELS            k_siv_prep, k_ghash_pass.., k_siv_tag, k_ctr*       (4-6)  beyond
 */
#ifndef UAES_PLAN_H
#define UAES_PLAN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum uaes_plan_mode { UAES_PLAN_ECB = 0, UAES_PLAN_CTR = 1, UAES_PLAN_XTS = 2, UAES_PLAN_GCM = 3, UAES_PLAN_OCB = 4, UAES_PLAN_SIV = 5 };

enum uaes_arrangement {
    UAES_ARR_ECB_SINGLE = 0,
    UAES_ARR_ECB_TILED,
    UAES_ARR_CTR_SINGLE,
    UAES_ARR_CTR_QUAD,
    UAES_ARR_CTR_STRIPED,
    UAES_ARR_XTS_SMALL,
    UAES_ARR_XTS_PACKED,
    UAES_ARR_XTS_BULK,
    UAES_ARR_GCM_SMALL,
    UAES_ARR_GCM_CHUNKS,
    UAES_ARR_GCM_TWOPHASE,
    UAES_ARR_GCM_STRIPED,
    UAES_ARR_GCM_LEVELS,
    UAES_ARR_OCB_SMALL,
    UAES_ARR_OCB_RUNS,
    UAES_ARR_SIV_SMALL,
    UAES_ARR_SIV_CHUNKS,
    UAES_ARR_SIV_LEVELS,
    UAES_ARR_COUNT
};

/* directions: 0 = encrypt, 1 = decrypt (GCM: tag first, N7), 2 = GCM decrypt in one pass (uaes_set_gcm_one_pass_decrypt),
 * 3 = GCM tag only (truncated tags: the host layer compares) */

/* a planner's answer */
typedef struct {
    int      arrangement;        /* enum uaes_arrangement */
    int      launches;           /* kernels the call enqueues */
    unsigned grid, steps;        /* workgroups of the main kernel; CHUNKS / TWOPHASE: GHASH positions per thread */
} uaes_plan;

/* Test / measurement hook: arrangements whose bit (1u << id) is set are not chosen where another one can take the
 * call (GCM_LEVELS, XTS_BULK, CTR_QUAD, ECB_TILED, OCB_RUNS and SIV_LEVELS take everything and cannot be switched
 * off).  Process-wide; UAES_PLAN_DISABLE in the environment (a number, e.g. 0x1c00) is its initial value. */
void     uaesk_plan_disable(unsigned mask);
unsigned uaesk_plan_disabled(void);

/* a = bytes of text (XTS: bytes per data unit), b = bytes of associated data (XTS: number of units; ECB / CTR: unused),
 * flags bit 0: GCM with a key context, bit 1: XTS with an explicit 16-byte tweak, bit 2: no counter word armed (the
 * two-launch form of the one-launch arrangements), bit 3: CTR with the little-endian 32-bit counter (GCM-SIV).
 * The number of CUs decides most boundaries; without a device the answer is the one for a 256-CU MI355X.  Returns 0
 * and fills *p, or a HIP error code for arguments that make no sense. */
int uaesk_plan(int mode, int dir, size_t a, size_t b, unsigned flags, uaes_plan *p);

const char *uaesk_arrangement_name(int id);

#ifdef __cplusplus
}
#endif
#endif
