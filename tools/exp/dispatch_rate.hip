#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void k_empty64(int* p) { if (threadIdx.x == 9999) *p = 1; }
__global__ __launch_bounds__(256) void k_empty256(int* p) { if (threadIdx.x == 9999) *p = 1; }
__global__ __launch_bounds__(64) void k_lds64(int* p) { __shared__ int a[784]; a[threadIdx.x] = threadIdx.x; __syncthreads(); if (a[(threadIdx.x * 7) & 63] == 9999) *p = 1; }
template <typename F> float timeit(F f, int n) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < n; i++) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / n * 1000.f;
}
int main() {
    int* d; hipMalloc(&d, 4);
    printf("empty 64-thread WGs  (1024 x 264): %.1f us\n", timeit([&] { hipLaunchKernelGGL(k_empty64, dim3(1024, 264), dim3(64), 0, 0, d); }, 20));
    printf("empty 64-thread WGs  (1024 x 132): %.1f us\n", timeit([&] { hipLaunchKernelGGL(k_empty64, dim3(1024, 132), dim3(64), 0, 0, d); }, 20));
    printf("empty 256-thread WGs (1024 x 66):  %.1f us\n", timeit([&] { hipLaunchKernelGGL(k_empty256, dim3(1024, 66), dim3(256), 0, 0, d); }, 20));
    printf("lds   64-thread WGs  (1024 x 264): %.1f us\n", timeit([&] { hipLaunchKernelGGL(k_lds64, dim3(1024, 264), dim3(64), 0, 0, d); }, 20));
    return 0;
}
