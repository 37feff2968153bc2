// Probe: what pinning a caller's pageable buffer in place costs (hipHostRegister / hipHostUnregister) and what copies from / to it then reach,
// one way and both ways at once from ONE thread.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    for (size_t n : {(size_t)8 << 20, (size_t)64 << 20, (size_t)256 << 20}) {
        uint8_t *hin = (uint8_t*)malloc(n), *hout = (uint8_t*)malloc(n), *din, *dout;
        memset(hin, 1, n); memset(hout, 2, n);
        (void)hipMalloc(&din, n); (void)hipMalloc(&dout, n);
        hipStream_t s1, s2; (void)hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
        double reg = 0, unreg = 0, up = 0, down = 0, both = 0;
        for (int i = 0; i < 4; ++i) {
            double t0 = now();
            hipError_t e1 = hipHostRegister(hin, n, hipHostRegisterDefault), e2 = hipHostRegister(hout, n, hipHostRegisterDefault);
            double t1 = now();
            if (e1 != hipSuccess || e2 != hipSuccess) { printf("register failed: %s %s\n", hipGetErrorString(e1), hipGetErrorString(e2)); return 1; }
            (void)hipMemcpyAsync(din, hin, n, hipMemcpyHostToDevice, s1); (void)hipStreamSynchronize(s1);
            double t2 = now();
            (void)hipMemcpyAsync(hout, dout, n, hipMemcpyDeviceToHost, s2); (void)hipStreamSynchronize(s2);
            double t3 = now();
            (void)hipMemcpyAsync(din, hin, n, hipMemcpyHostToDevice, s1); (void)hipMemcpyAsync(hout, dout, n, hipMemcpyDeviceToHost, s2);
            (void)hipStreamSynchronize(s1); (void)hipStreamSynchronize(s2);
            double t4 = now();
            (void)hipHostUnregister(hin); (void)hipHostUnregister(hout);
            double t5 = now();
            if (i) { reg += t1 - t0; up += t2 - t1; down += t3 - t2; both += t4 - t3; unreg += t5 - t4; }
        }
        printf("%4zu MiB: register both buffers %.2f ms, unregister %.2f ms; pinned in place: H2D %.1f GB/s, D2H %.1f GB/s, both at once %.2f ms = %.1f GB/s each way\n",
               n >> 20, reg / 3 * 1e3, unreg / 3 * 1e3, 3.0 * n / up / 1e9, 3.0 * n / down / 1e9, both / 3 * 1e3, 3.0 * n / both / 1e9);
        free(hin); free(hout); (void)hipFree(din); (void)hipFree(dout);
    }
    return 0;
}
