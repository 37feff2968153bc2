// Probe: pageable host <-> device copies, one direction at a time and both at once from two threads (what a pipelined host API could get).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    for (size_t n : {(size_t)8 << 20, (size_t)64 << 20, (size_t)256 << 20}) {
        uint8_t *hin = (uint8_t*)malloc(n), *hout = (uint8_t*)malloc(n), *din, *dout;
        memset(hin, 1, n); memset(hout, 2, n);
        (void)hipMalloc(&din, n); (void)hipMalloc(&dout, n);
        hipStream_t s1, s2; (void)hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
        auto up = [&](size_t off, size_t len) { (void)hipMemcpyAsync(din + off, hin + off, len, hipMemcpyHostToDevice, s1); (void)hipStreamSynchronize(s1); };
        auto down = [&](size_t off, size_t len) { (void)hipMemcpyAsync(hout + off, dout + off, len, hipMemcpyDeviceToHost, s2); (void)hipStreamSynchronize(s2); };
        up(0, n); down(0, n);
        double t0 = now(); for (int i = 0; i < 5; ++i) up(0, n); double t1 = now(); for (int i = 0; i < 5; ++i) down(0, n); double t2 = now();
        double t3 = now();
        for (int i = 0; i < 5; ++i) { std::thread a([&] { up(0, n); }); std::thread b([&] { down(0, n); }); a.join(); b.join(); }
        double t4 = now();
        // sliced: 8 slices each way, two threads
        double t5 = now();
        for (int i = 0; i < 5; ++i) {
            std::thread a([&] { for (int k = 0; k < 8; ++k) up(k * (n / 8), n / 8); });
            std::thread b([&] { for (int k = 0; k < 8; ++k) down(k * (n / 8), n / 8); });
            a.join(); b.join();
        }
        double t6 = now();
        printf("%4zu MiB: H2D %.1f GB/s, D2H %.1f GB/s; both at once (two threads): %.2f ms per pair = %.1f GB/s each way; in 8 slices each: %.2f ms\n", n >> 20,
               5.0 * n / (t1 - t0) / 1e9, 5.0 * n / (t2 - t1) / 1e9, (t4 - t3) / 5 * 1e3, 5.0 * n / (t4 - t3) / 1e9, (t6 - t5) / 5 * 1e3);
        free(hin); free(hout); (void)hipFree(din); (void)hipFree(dout);
    }
    return 0;
}
