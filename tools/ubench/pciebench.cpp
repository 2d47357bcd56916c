// PCIe staging experiment (diagnostic): what a host-pointer call can hope for.
//  a) pageable hipMemcpy H2D / D2H, 1 GiB      b) both directions at once from two threads
//  c) hipHostRegister cost                      d) registered (pinned) async copies, both directions
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main()
{
    const size_t n = 1ull << 30;
    char *h1 = (char *)malloc(n), *h2 = (char *)malloc(n);
    memset(h1, 1, n); memset(h2, 2, n);
    char *d1, *d2;
    CK(hipMalloc(&d1, n)); CK(hipMalloc(&d2, n));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    CK(hipMemcpy(d1, h1, n, hipMemcpyHostToDevice)); CK(hipMemcpy(h2, d2, n, hipMemcpyDeviceToHost));
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now(); CK(hipMemcpy(d1, h1, n, hipMemcpyHostToDevice)); double t1 = now();
        CK(hipMemcpy(h2, d2, n, hipMemcpyDeviceToHost)); double t2 = now();
        printf("pageable: H2D %.1f ms (%.1f GB/s)  D2H %.1f ms (%.1f GB/s)\n", (t1 - t0) * 1e3, n / (t1 - t0) / 1e9,
               (t2 - t1) * 1e3, n / (t2 - t1) / 1e9);
    }
    {
        double t0 = now();
        std::thread a([&] { (void)hipMemcpyAsync(d1, h1, n, hipMemcpyHostToDevice, s1); (void)hipStreamSynchronize(s1); });
        std::thread b([&] { (void)hipMemcpyAsync(h2, d2, n, hipMemcpyDeviceToHost, s2); (void)hipStreamSynchronize(s2); });
        a.join(); b.join();
        double t1 = now();
        printf("pageable, both directions from two threads: %.1f ms (%.1f GB/s each way)\n", (t1 - t0) * 1e3, n / (t1 - t0) / 1e9);
    }
    {
        double t0 = now();
        CK(hipHostRegister(h1, n, hipHostRegisterDefault));
        double t1 = now();
        CK(hipHostRegister(h2, n, hipHostRegisterDefault));
        double t2 = now();
        printf("hipHostRegister 1 GiB: %.1f ms, %.1f ms\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3);
        t0 = now(); CK(hipMemcpyAsync(d1, h1, n, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1)); t1 = now();
        CK(hipMemcpyAsync(h2, d2, n, hipMemcpyDeviceToHost, s2)); CK(hipStreamSynchronize(s2)); t2 = now();
        printf("registered: H2D %.1f ms (%.1f GB/s)  D2H %.1f ms (%.1f GB/s)\n", (t1 - t0) * 1e3, n / (t1 - t0) / 1e9,
               (t2 - t1) * 1e3, n / (t2 - t1) / 1e9);
        t0 = now();
        CK(hipMemcpyAsync(d1, h1, n, hipMemcpyHostToDevice, s1)); CK(hipMemcpyAsync(h2, d2, n, hipMemcpyDeviceToHost, s2));
        CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
        t1 = now();
        printf("registered, both directions: %.1f ms (%.1f GB/s each way)\n", (t1 - t0) * 1e3, n / (t1 - t0) / 1e9);
        t0 = now(); CK(hipHostUnregister(h1)); CK(hipHostUnregister(h2)); t1 = now();
        printf("hipHostUnregister x2: %.1f ms\n", (t1 - t0) * 1e3);
    }
    return 0;
}
