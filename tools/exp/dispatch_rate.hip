// development aid: how fast does the dispatcher start (and retire) single-wave workgroups?  k_recon launches 25600 of them
// per picture index; if starting them takes as long as the launch lasts, the kernel's own work is not what bounds it.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int kLds, int kSpin>
__global__ void k_empty(uint32_t* out)
{
    __shared__ uint32_t lds[kLds / 4 ? kLds / 4 : 1];
    lds[threadIdx.x % (kLds / 4 ? kLds / 4 : 1)] = threadIdx.x;
    if (kSpin) {
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < kSpin * 100)  // 100 MHz clock: kSpin microseconds
            ;
    }
    if (out && lds[0] == 12345)
        out[0] = 1;
}
template <typename F>
static void run(const char* name, F launch)
{
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 5; i++)
        launch();
    hipEventRecord(a);
    for (int i = 0; i < 50; i++)
        launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("%-56s %8.1f us per launch\n", name, ms * 1000 / 50);
}
int main()
{
    uint32_t* d;
    hipMalloc(&d, 64);
    run("25600 x 64 threads, no LDS", [&] { hipLaunchKernelGGL((k_empty<0, 0>), dim3(1024, 25), dim3(64), 0, 0, d); });
    run("25600 x 64 threads, 8.6 KB LDS", [&] { hipLaunchKernelGGL((k_empty<8640, 0>), dim3(1024, 25), dim3(64), 0, 0, d); });
    run("25600 x 64 threads, 4.5 KB LDS", [&] { hipLaunchKernelGGL((k_empty<4544, 0>), dim3(1024, 25), dim3(64), 0, 0, d); });
    run("6400 x 256 threads, 8.6 KB LDS", [&] { hipLaunchKernelGGL((k_empty<8640, 0>), dim3(1024, 25 / 4 + 1), dim3(256), 0, 0, d); });
    run("25600 x 64 threads, 8.6 KB LDS, waves live 10 us", [&] { hipLaunchKernelGGL((k_empty<8640, 10>), dim3(1024, 25), dim3(64), 0, 0, d); });
    run("25600 x 64 threads, 4.5 KB LDS, waves live 10 us", [&] { hipLaunchKernelGGL((k_empty<4544, 10>), dim3(1024, 25), dim3(64), 0, 0, d); });
    run("25600 x 64 threads, no LDS, waves live 10 us", [&] { hipLaunchKernelGGL((k_empty<0, 10>), dim3(1024, 25), dim3(64), 0, 0, d); });
    run("6400 x 256 threads, 8.6 KB LDS, waves live 10 us", [&] { hipLaunchKernelGGL((k_empty<8640, 10>), dim3(1024, 7), dim3(256), 0, 0, d); });
    run("25600 x 64 threads, 8.6 KB LDS, waves live 5 us", [&] { hipLaunchKernelGGL((k_empty<8640, 5>), dim3(1024, 25), dim3(64), 0, 0, d); });
    return 0;
}
