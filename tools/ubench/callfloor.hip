// Where does the per-call time of a small synchronous call go?  
//   empty kernel + stream sync          = the runtime's floor for ANY synchronous GPU call
//   engine *_dev call + stream sync     = + key schedule, launcher, table fill, the cipher
//   engine host-API call (device ptrs)  = + pointer classification, context lock
//   engine host-API call (host ptrs)    = + pinned bounce copies
// and the device-side duration of the small kernels (events).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include "../../include/uaes_hip.h"

__global__ void k_empty(int *p) { if (p && threadIdx.x == 12345) *p = 1; }

template <typename F>
static double us_per_call(F f, int reps = 2000)
{
    for (int i = 0; i < 50; ++i) f();
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; ++i) f();
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
}

int main()
{
    unsigned char key[32], iv[12], nonce[12];
    for (int i = 0; i < 32; ++i) key[i] = (unsigned char)i;
    memset(iv, 7, 12); memset(nonce, 9, 12);
    unsigned char ctr0[16] = { 0 }; ctr0[15] = 1;
    void *din, *dout; int *dstat;
    (void)hipMalloc(&din, 1 << 20); (void)hipMalloc(&dout, (1 << 20) + 64); (void)hipMalloc(&dstat, 4);
    (void)hipMemset(din, 1, 1 << 20);
    unsigned char *hin = (unsigned char *)malloc(1 << 20), *hout = (unsigned char *)malloc((1 << 20) + 64);
    memset(hin, 1, 1 << 20);
    if (uaes_init()) { fprintf(stderr, "%s\n", uaes_last_error()); return 1; }
    hipStream_t st; (void)hipStreamCreate(&st);
    printf("empty kernel + hipStreamSynchronize(NULL stream):   %6.2f us\n", us_per_call([&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0, nullptr); (void)hipStreamSynchronize(0); }));
    printf("empty kernel + hipStreamSynchronize(own stream):    %6.2f us\n", us_per_call([&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, nullptr); (void)hipStreamSynchronize(st); }));
    printf("empty kernel launch only (async, amortised):        %6.2f us\n", us_per_call([&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, nullptr); }, 20000));
    (void)hipStreamSynchronize(st);
    uaes_gcm_key *gk = nullptr;
    if (uaes_gcm_key_new(&gk, 128, key)) { fprintf(stderr, "%s\n", uaes_last_error()); return 1; }
    for (size_t n : { (size_t)16, (size_t)4096, (size_t)16384, (size_t)65536 }) {
        printf("-- %zu bytes --\n", n);
        printf("uaes_gcm_key_encrypt_dev + sync (key context): %6.2f us\n", us_per_call([&] { uaes_gcm_key_encrypt_dev(gk, nonce, nullptr, 0, din, n, dout, st); (void)hipStreamSynchronize(st); }));
        printf("uaes_gcm_key_decrypt_dev + sync (key context): %6.2f us\n", us_per_call([&] { uaes_gcm_key_decrypt_dev(gk, nonce, nullptr, 0, dout, n, din, dstat, st); (void)hipStreamSynchronize(st); }));
        (void)hipMemset(din, 1, 1 << 20);
        printf("uaes_ecb_dev + sync:            %6.2f us\n", us_per_call([&] { uaes_ecb_dev(128, key, 0, din, n, dout, st); (void)hipStreamSynchronize(st); }));
        printf("uaes_ctr_xcrypt_at_dev + sync:  %6.2f us\n", us_per_call([&] { uaes_ctr_xcrypt_at_dev(128, key, ctr0, 0, din, n, dout, st); (void)hipStreamSynchronize(st); }));
        printf("uaes_gcm_encrypt_dev + sync:    %6.2f us\n", us_per_call([&] { uaes_gcm_encrypt_dev(128, key, nonce, nullptr, 0, din, n, dout, st); (void)hipStreamSynchronize(st); }));
        printf("uaes_ecb_encrypt (device ptrs): %6.2f us\n", us_per_call([&] { uaes_ecb_encrypt(128, key, din, n, dout); }));
        printf("uaes_ecb_encrypt (host ptrs):   %6.2f us\n", us_per_call([&] { uaes_ecb_encrypt(128, key, hin, n, hout); }));
        printf("uaes_ctr_xcrypt (host ptrs):    %6.2f us\n", us_per_call([&] { uaes_ctr_xcrypt(128, key, iv, hin, n, hout); }));
        printf("uaes_gcm_encrypt (host ptrs):   %6.2f us\n", us_per_call([&] { uaes_gcm_encrypt(128, key, nonce, nullptr, 0, hin, n, hout); }));
        printf("uaes_gcm_encrypt (device ptrs): %6.2f us\n", us_per_call([&] { uaes_gcm_encrypt(128, key, nonce, nullptr, 0, din, n, dout); }));
        // device-side duration of one call's kernels
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        float ms;
        (void)hipEventRecord(e0, st);
        for (int i = 0; i < 200; ++i) uaes_ecb_dev(128, key, 0, din, n, dout, st);
        (void)hipEventRecord(e1, st); (void)hipStreamSynchronize(st); (void)hipEventElapsedTime(&ms, e0, e1);
        printf("ECB back-to-back on one stream:  %6.2f us per call (device side incl. launch gaps)\n", ms * 1e3 / 200);
        (void)hipEventRecord(e0, st);
        for (int i = 0; i < 200; ++i) uaes_gcm_encrypt_dev(128, key, nonce, nullptr, 0, din, n, dout, st);
        (void)hipEventRecord(e1, st); (void)hipStreamSynchronize(st); (void)hipEventElapsedTime(&ms, e0, e1);
        printf("GCM back-to-back on one stream:  %6.2f us per call (device side incl. launch gaps)\n", ms * 1e3 / 200);
    }
    uaes_gcm_key_free(gk);
    return 0;
}
