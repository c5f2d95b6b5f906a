// development aid: what the runtime says about resident workgroups per CU for single-wave workgroups
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(int* p) { extern __shared__ int sh[]; sh[threadIdx.x] = 1; if (p) p[0] = sh[0]; }
int main()
{
    hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
    printf("maxThreadsPerMultiProcessor %d sharedMemPerMultiprocessor %zu maxSharedMemoryPerMultiProcessor %zu regsPerMultiprocessor %d maxBlocksPerMultiProcessor %d\n",
           pr.maxThreadsPerMultiProcessor, pr.sharedMemPerMultiprocessor, pr.maxSharedMemoryPerMultiProcessor, pr.regsPerMultiprocessor, pr.maxBlocksPerMultiProcessor);
    for (int bs : {64, 128, 256})
        for (size_t lds : {(size_t)0, (size_t)4096, (size_t)8192, (size_t)8512, (size_t)9728, (size_t)10240}) {
            int n = 0;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, bs, lds);
            printf("block %3d lds %5zu -> %2d blocks per CU = %2d waves\n", bs, lds, n, n * bs / 64);
        }
    return 0;
}
