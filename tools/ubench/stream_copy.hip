// Measured HBM denominators for the roofline: copy (1 read + 1 write), read-only sum
// and write-only fill over a 2 GiB buffer, double2 per lane, grid-stride.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/stream_copy.hip -o tools/ubench/stream_copy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void copy_k(const double2* __restrict__ a, double2* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void read_k(const double2* __restrict__ a, double* out, size_t n) {
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double2 v = a[i];
        s += v.x + v.y;
    }
    if (s == 12345.678) out[0] = s;   // keeps the loads alive
}
__global__ void fill_k(double2* b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = make_double2(1.0, 2.0);
}

int main() {
    const size_t bytes = (size_t)2 << 30, n = bytes / sizeof(double2);
    double2 *a, *b; double* o;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&o, 8));
    CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grids[] = {256 * 8, 256 * 16, 256 * 32};
    for (int g : grids) {
        for (int which = 0; which < 3; ++which) {
            float best = 1e30f;
            for (int rep = 0; rep < 6; ++rep) {
                CK(hipEventRecord(e0));
                if (which == 0) copy_k<<<g, 256>>>(a, b, n);
                else if (which == 1) read_k<<<g, 256>>>(a, o, n);
                else fill_k<<<g, 256>>>(b, n);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0 && ms < best) best = ms;
            }
            const double moved = (which == 0 ? 2.0 : 1.0) * (double)bytes;
            printf("grid %5d  %-5s  %.3f ms  %.2f TB/s\n", g, which == 0 ? "copy" : (which == 1 ? "read" : "fill"), best, moved / best / 1e9);
        }
    }
    return 0;
}
