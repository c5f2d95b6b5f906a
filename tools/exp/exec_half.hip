// development aid: does a wave64 VALU instruction cost less when only lanes 0-31 are active?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void k(int mode, int iters, int* out)
{
    const int lane = threadIdx.x;
    bool active = mode == 0 ? true : mode == 1 ? lane < 32 : mode == 2 ? (lane & 1) == 0 : lane < 16;
    int a0 = lane, a1 = lane + 1, a2 = lane + 2, a3 = lane + 3, a4 = lane + 4, a5 = lane + 5, a6 = lane + 6, a7 = lane + 7;
    if (active) {
        for (int i = 0; i < iters; i++) {
            a0 = a0 * 3 + 1; a1 = a1 * 5 + 2; a2 = a2 * 7 + 3; a3 = a3 * 9 + 4;
            a4 = a4 * 11 + 5; a5 = a5 * 13 + 6; a6 = a6 * 15 + 7; a7 = a7 * 17 + 8;
            asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        }
    }
    if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345678)
        out[0] = 1;
}
int main()
{
    int* d; hipMalloc(&d, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[] = {"all 64 lanes", "lanes 0-31", "even lanes", "lanes 0-15"};
    for (int waves_per_simd : {1, 4}) {
        for (int mode = 0; mode < 4; mode++) {
            int grid = 256 * 4 * waves_per_simd;
            hipLaunchKernelGGL(k, dim3(grid), dim3(64), 0, 0, mode, 1000, d);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(grid), dim3(64), 0, 0, mode, 20000, d);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // 8 mad (v_mad_u32_u24 or mul+add) per iteration
            printf("%d waves/SIMD, %-12s: %.3f ms  -> %.2f ns per loop trip\n", waves_per_simd, names[mode], ms, ms * 1e6 / 20000);
        }
    }
    return 0;
}
