// D2H copy micro-benchmark (gfx950 box): which engine moves a few MB from HBM to pinned host memory, and how fast.
//   (1) hipMemcpyAsync(pinned <- device): SDMA or the runtime's blit kernel (`__amd_rocclr_copyBuffer` in a kernel trace)?
//   (2) a hand-written copy kernel with G workgroups storing straight into the mapped pinned buffer: PCIe rate vs G
//       (a copy that needs few CUs leaves the 1-workgroup-per-CU convolution kernels alone).
// build: hipcc --offload-arch=gfx950 -O2 -o d2h_copy d2h_copy.hip ; run: ./d2h_copy [MB]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(_e)); return 1; } } while (0)

__global__ void __launch_bounds__(256) copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
}

int main(int argc, char** argv) {
    const size_t mb = argc > 1 ? (size_t)atoi(argv[1]) : 4;
    const size_t bytes = mb << 20;
    void *dev, *host;
    CK(hipMalloc(&dev, bytes));
    CK(hipHostMalloc(&host, bytes, hipHostMallocDefault));
    CK(hipMemset(dev, 1, bytes));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 20;
    for (int i = 0; i < 3; ++i) CK(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) CK(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, st));
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("hipMemcpyAsync D2H %zu MB: %.1f us per copy, %.1f GB/s\n", mb, 1e3 * ms / reps, bytes * reps / (ms * 1e6));
    void* hdev;
    CK(hipHostGetDevicePointer(&hdev, host, 0));
    {   // (1b) the same copy right behind a kernel on the SAME stream, and (1c) on a second stream that waits for the kernel's event:
        // does the runtime switch to its blit kernel when the stream's previous command was a kernel?
        void* dev2;
        CK(hipMalloc(&dev2, bytes));
        hipStream_t st2;
        CK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
        hipEvent_t ek;
        CK(hipEventCreateWithFlags(&ek, hipEventDisableTiming));
        for (int mode = 0; mode < 2; ++mode) {
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < reps; ++i) {
                hipLaunchKernelGGL(copy_kernel, dim3(256), dim3(256), 0, st, (const uint4*)dev, (uint4*)dev2, bytes / 16);
                if (mode == 0) CK(hipMemcpyAsync(host, dev2, bytes, hipMemcpyDeviceToHost, st));
                else { CK(hipEventRecord(ek, st)); CK(hipStreamWaitEvent(st2, ek, 0)); CK(hipMemcpyAsync(host, dev2, bytes, hipMemcpyDeviceToHost, st2)); }
            }
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st)); CK(hipStreamSynchronize(st2));
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("kernel + hipMemcpyAsync D2H (%s): %.1f us per pair\n", mode == 0 ? "same stream" : "second stream behind an event", 1e3 * ms / reps);
        }
    }
    for (int g : {1, 2, 4, 8, 16, 32, 64, 256, 512}) {
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(copy_kernel, dim3(g), dim3(256), 0, st, (const uint4*)dev, (uint4*)hdev, bytes / 16);
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(copy_kernel, dim3(g), dim3(256), 0, st, (const uint4*)dev, (uint4*)hdev, bytes / 16);
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("copy kernel, %3d workgroups: %.1f us per copy, %.1f GB/s\n", g, 1e3 * ms / reps, bytes * reps / (ms * 1e6));
    }
    return 0;
}
