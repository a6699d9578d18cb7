// Exploration harness (NOT product code): host<->device copy strategies for 0.8 GB of pageable host memory.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t n = 100000000, bytes = n * 8;
    double* h = (double*)malloc(bytes); double* h2 = (double*)malloc(bytes);
    for (size_t i = 0; i < n; ++i) { h[i] = (double)i; h2[i] = 0; }
    double *d, *d2; CK(hipMalloc(&d, bytes)); CK(hipMalloc(&d2, bytes));
    hipStream_t s0, s1; CK(hipStreamCreate(&s0)); CK(hipStreamCreate(&s1));
    for (int rep = 0; rep < 3; ++rep) {
        double t = now(); CK(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice)); double t1 = now();
        CK(hipMemcpy(h2, d, bytes, hipMemcpyDeviceToHost)); double t2 = now();
        printf("pageable: H2D %.1f ms (%.1f GB/s)  D2H %.1f ms (%.1f GB/s)\n", (t1 - t) * 1e3, bytes / (t1 - t) / 1e9, (t2 - t1) * 1e3, bytes / (t2 - t1) / 1e9);
    }
    for (int rep = 0; rep < 3; ++rep) {
        double t = now(); CK(hipHostRegister(h, bytes, hipHostRegisterDefault)); CK(hipHostRegister(h2, bytes, hipHostRegisterDefault)); double t1 = now();
        CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s0)); CK(hipStreamSynchronize(s0)); double t2 = now();
        CK(hipMemcpyAsync(h2, d, bytes, hipMemcpyDeviceToHost, s0)); CK(hipStreamSynchronize(s0)); double t3 = now();
        // both directions at once on two streams
        CK(hipMemcpyAsync(d2, h, bytes, hipMemcpyHostToDevice, s0)); CK(hipMemcpyAsync(h2, d, bytes, hipMemcpyDeviceToHost, s1)); CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1)); double t4 = now();
        CK(hipHostUnregister(h)); CK(hipHostUnregister(h2)); double t5 = now();
        printf("registered: register %.1f ms  H2D %.1f ms (%.1f GB/s)  D2H %.1f ms (%.1f GB/s)  both ways at once %.1f ms  unregister %.1f ms\n", (t1 - t) * 1e3, (t2 - t1) * 1e3, bytes / (t2 - t1) / 1e9, (t3 - t2) * 1e3, bytes / (t3 - t2) / 1e9, (t4 - t3) * 1e3, (t5 - t4) * 1e3);
    }
    return 0;
}
