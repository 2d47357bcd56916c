/*
 * threads_rate.c -- do concurrent host threads scale on the drop-in API?
 *
 * N host threads each call AES_CTR_encrypt / AES_GCM_encrypt / AES_ECB_encrypt / AES_XTS_encrypt of
 * include/micro_aes.h on their own 4 KiB HOST buffers (the reference's call shape), as fast as they can.
 * The reference cannot do this at all (one global RoundKey, micro_aes.c:72).  Prints calls per second
 * for 1, 2, 4, 8, 16 threads and the scaling over one thread; every result is checked against the
 * single-threaded result of the same call.
 *
 *   gcc -O2 -I include tools/threads_rate.c -o /tmp/threads_rate -Lmicro-aes_amd/lib -lmicro_aes_hip_128 \
 *       -Wl,-rpath,$PWD/micro-aes_amd/lib -Wl,-rpath,/opt/rocm/lib -lpthread        (tools/threads_rate.py does it)
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "micro_aes.h"

enum { MAXT = 32 };
static size_t g_len = 4096;
static int g_mode, g_calls;
static const uint8_t key[32] = { 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16,
                                 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32 };
static const uint8_t iv[16] = { 0xf0, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa, 0xfb };
static pthread_barrier_t bar;
static uint8_t *want;

typedef struct { int id; uint8_t *in, *out; int bad; } job;

static void one_call(const uint8_t *in, uint8_t *out)
{
    switch (g_mode) {
    case 0: AES_CTR_encrypt(key, iv, in, g_len, out); break;
    case 1: AES_GCM_encrypt(key, iv, NULL, 0, in, g_len, out); break;
    case 2: AES_ECB_encrypt(key, in, g_len, out); break;
    default: AES_XTS_encrypt(key, iv, in, g_len, out); break;
    }
}

static void *worker(void *p)
{
    job *j = (job *)p;
    int i;
    one_call(j->in, j->out);                      /* the thread's lane is made here, untimed */
    pthread_barrier_wait(&bar);
    for (i = 0; i < g_calls; ++i) one_call(j->in, j->out);
    pthread_barrier_wait(&bar);
    j->bad = memcmp(j->out, want, g_len + (g_mode == 1 ? 16 : 0)) != 0;
    return NULL;
}

static double now(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec + 1e-9 * t.tv_nsec;
}

int main(int argc, char **argv)
{
    static const char *names[] = { "AES_CTR_encrypt", "AES_GCM_encrypt", "AES_ECB_encrypt", "AES_XTS_encrypt" };
    static const int counts[] = { 1, 2, 4, 8, 16 };
    size_t i;
    int m, k, t;
    g_calls = argc > 1 ? atoi(argv[1]) : 3000;
    if (argc > 2) g_len = (size_t)atol(argv[2]);
    uint8_t *src = malloc(g_len);
    want = malloc(g_len + 32);
    for (i = 0; i < g_len; ++i) src[i] = (uint8_t)(i * 131 + 7);
    printf("%zu-byte host buffers, %d calls per thread\n", g_len, g_calls);
    for (m = 0; m < 4; ++m) {
        double base = 0;
        g_mode = m;
        one_call(src, want);
        for (k = 0; k < 5; ++k) {
            const int nt = counts[k];
            pthread_t th[MAXT];
            job jobs[MAXT];
            double t0, dt;
            int bad = 0;
            pthread_barrier_init(&bar, NULL, (unsigned)nt + 1);
            for (t = 0; t < nt; ++t) {
                jobs[t].id = t; jobs[t].bad = 0;
                jobs[t].in = malloc(g_len); jobs[t].out = malloc(g_len + 32);
                memcpy(jobs[t].in, src, g_len);
                pthread_create(&th[t], NULL, worker, &jobs[t]);
            }
            pthread_barrier_wait(&bar);
            t0 = now();
            pthread_barrier_wait(&bar);
            dt = now() - t0;
            for (t = 0; t < nt; ++t) { pthread_join(th[t], NULL); bad |= jobs[t].bad; free(jobs[t].in); free(jobs[t].out); }
            pthread_barrier_destroy(&bar);
            if (nt == 1) base = (double)g_calls / dt;
            printf("%-16s %2d threads: %9.0f calls/s  (%6.1f us per call per thread)  %5.2fx over 1 thread  %s\n",
                   names[m], nt, nt * (double)g_calls / dt, dt / g_calls * 1e6, nt * (double)g_calls / dt / base,
                   bad ? "MISMATCH" : "ok");
            if (bad) return 1;
        }
    }
    return 0;
}
