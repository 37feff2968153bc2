// fetch_calib.hip — calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths the codec kernels use
// (MI355X_MICROARCH.md §HBM: FETCH_SIZE reads half the bytes of 16 B/lane streams; other widths uncalibrated).  Each kernel streams a
// known 1 GiB once: read4 (global_load_dword, 256 B per wave instruction — the rotation encoder's input loads), read16 (dwordx4),
// write4 (global_store_dword), write2 (2-byte stores at 2-byte granularity — the encoder's item stores).
//   hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip ; rocprofv3 --pmc FETCH_SIZE --kernel-trace -- ./fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void read4(const uint32_t* p, uint64_t n, uint32_t* out) { uint32_t a = 0; for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) a ^= p[i]; if (a == 0x12345) out[0] = a; }
__global__ void read16(const uint4* p, uint64_t n, uint32_t* out) { uint32_t a = 0; for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) { uint4 v = p[i]; a ^= v.x ^ v.y ^ v.z ^ v.w; } if (a == 0x12345) out[0] = a; }
__global__ void write4(uint32_t* p, uint64_t n) { for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i; }
__global__ void write2(uint16_t* p, uint64_t n) { for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) p[i] = (uint16_t)i; }
int main() {
    const uint64_t bytes = 1ull << 30;
    void *a, *o;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&o, 64) != hipSuccess) return 1;
    (void)hipMemset(a, 1, bytes);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(read4, dim3(4096), dim3(256), 0, 0, (const uint32_t*)a, bytes / 4, (uint32_t*)o);
        hipLaunchKernelGGL(read16, dim3(4096), dim3(256), 0, 0, (const uint4*)a, bytes / 16, (uint32_t*)o);
        hipLaunchKernelGGL(write4, dim3(4096), dim3(256), 0, 0, (uint32_t*)a, bytes / 4);
        hipLaunchKernelGGL(write2, dim3(4096), dim3(256), 0, 0, (uint16_t*)a, bytes / 2);
    }
    (void)hipDeviceSynchronize();
    printf("each kernel moved %llu bytes\n", (unsigned long long)bytes);
    return 0;
}
