/* tools/first_call.c -- what a process that uses TWO key sizes pays before its first answers: dlopen of the two
 * per-key-size drop-in libraries and the first AES_ECB_encrypt through each (code-object load, context, tables), then a
 * warm call.  Run once against the thin shims of this round (one engine, DT_NEEDED libuaes_hip.so) and once against
 * "fat" libraries linked the round-4 way (every engine object in each), tools/first_call_latency.sh.
 *     first_call <dir with libmicro_aes_hip_{128,256}.so>                                                           */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef void (*ecb_fn)(const uint8_t *key, const void *in, size_t len, void *out);

static double now_ms(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}

int main(int argc, char **argv)
{
    const char *dir = argc > 1 ? argv[1] : ".";
    const int bits[2] = { 128, 256 };
    uint8_t key[32] = { 0 }, in[16] = { 0 }, out[16];
    double t0 = now_ms(), total0 = t0;
    int i;
    for (i = 0; i < 2; ++i) {
        char path[1024];
        void *h;
        ecb_fn f;
        double t_open, t_first, t_warm;
        snprintf(path, sizeof path, "%s/libmicro_aes_hip_%d.so", dir, bits[i]);
        t0 = now_ms();
        h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
        if (!h) { fprintf(stderr, "%s\n", dlerror()); return 1; }
        t_open = now_ms() - t0;
        f = (ecb_fn)dlsym(h, "AES_ECB_encrypt");
        t0 = now_ms(); f(key, in, 16, out); t_first = now_ms() - t0;
        t0 = now_ms(); f(key, in, 16, out); t_warm = now_ms() - t0;
        printf("  AES-%d: dlopen %8.2f ms   first AES_ECB_encrypt %8.2f ms   second %6.3f ms   (out[0] %02x)\n",
               bits[i], t_open, t_first, t_warm, out[0]);
    }
    printf("  both key sizes ready after %8.2f ms\n", now_ms() - total0);
    {
        /* how many copies of the engine (its code objects are ~6 MB of data in the library) does the process map? */
        FILE *m = fopen("/proc/self/maps", "r");
        char line[2048], seen[8][512];
        int nseen = 0, k;
        while (m && fgets(line, sizeof line, m)) {
            char *path = strchr(line, '/');
            unsigned long a, b;
            if (!path || !(strstr(path, "libuaes_hip") || strstr(path, "libmicro_aes_hip"))) continue;
            path[strcspn(path, "\n")] = 0;
            if (sscanf(line, "%lx-%lx", &a, &b) != 2 || b - a < (1ul << 20)) continue;      /* a mapping of >= 1 MiB */
            for (k = 0; k < nseen && strcmp(seen[k], path); ++k) { }
            if (k == nseen && nseen < 8) snprintf(seen[nseen++], sizeof seen[0], "%s", path);
        }
        if (m) fclose(m);
        printf("  libraries of this project mapped with a segment of 1 MiB or more (= engine copies): %d\n", nseen);
    }
    return 0;
}
