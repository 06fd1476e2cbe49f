#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ double lds[];
__global__ void __launch_bounds__(256) k(double* p) { lds[threadIdx.x] = p[threadIdx.x]; __syncthreads(); p[threadIdx.x] = lds[255 - threadIdx.x]; }
int main() {
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
    printf("sharedMemPerBlock=%zu sharedMemPerMultiprocessor=%zu maxSharedMemoryPerMultiProcessor=%zu regsPerBlock=%d\n",
           pr.sharedMemPerBlock, pr.sharedMemPerMultiprocessor, pr.maxSharedMemoryPerMultiProcessor, pr.regsPerBlock);
    for (int kb = 48; kb <= 160; kb += 4) {
        int n = -1;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 256, (size_t)kb * 1024);
        printf("%d KB -> %d (%s)\n", kb, n, hipGetErrorString(e));
    }
    return 0;
}
