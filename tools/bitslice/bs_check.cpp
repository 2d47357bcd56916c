// CPU check of the generated bitsliced AES against the oracle restatement (tools/bitslice).
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include "bs_aes.h"
extern "C" {
#include "../../oracle/uaes_oracle.h"
}

int main()
{
    uint8_t key[16];
    for (int i = 0; i < 16; ++i) key[i] = (uint8_t)(i * 17 + 3);
    unsigned char ksbuf[4 + 240];
    orc_key *ks = (orc_key *)ksbuf;
    if (orc_setkey(ks, key, 128)) return 2;
    const uint8_t *rkbytes = ksbuf + 4;            /* round keys as the FIPS-197 byte stream (see pyoracle) */
    static u32 kp[11 * 128];
    for (int r = 0; r <= 10; ++r)
        for (int c = 0; c < 4; ++c)
            for (int q = 0; q < 4; ++q)
                for (int i = 0; i < 8; ++i)
                    kp[128 * r + 32 * c + 8 * q + i] = 0u - (u32)((rkbytes[16 * r + 4 * c + q] >> i) & 1);
    uint8_t blocks[32][16], want[32][16];
    srand(7);
    for (int k = 0; k < 32; ++k)
        for (int i = 0; i < 16; ++i) blocks[k][i] = (uint8_t)rand();
    for (int k = 0; k < 32; ++k) orc_encrypt_block(ks, blocks[k], want[k]);
    u32 st[16][8];
    memset(st, 0, sizeof st);
    for (int k = 0; k < 32; ++k)
        for (int b = 0; b < 16; ++b)
            for (int i = 0; i < 8; ++i) st[b][i] |= (u32)((blocks[k][b] >> i) & 1) << k;
    bs_encrypt<10>(st, kp);
    int bad = 0;
    for (int w = 0; w < 4; ++w) {
        u32 a[32];
        for (int j = 0; j < 32; ++j) a[j] = st[4 * w + (j >> 3)][j & 7];
        bs_transpose32(a);
        for (int k = 0; k < 32; ++k) {
            u32 expect;
            memcpy(&expect, &want[k][4 * w], 4);
            if (a[k] != expect) ++bad;
        }
    }
    printf("bitsliced AES-128 vs oracle: %d mismatching words of 128\n", bad);
    return bad != 0;
}
