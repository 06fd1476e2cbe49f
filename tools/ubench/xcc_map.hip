// Which XCD does block b of a 1-D grid run on?  (The wave-per-particle NUTS kernel deals cache lines of particles to
// the XCDs by blockIdx % 8; this prints the hardware's XCC_ID per block to check that assumption.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned* out) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    if (threadIdx.x == 0) out[blockIdx.x] = id;
}
int main() {
    const int nb = 4096;
    unsigned* d;
    hipMalloc(&d, nb * sizeof(unsigned));
    k<<<nb, 256>>>(d);
    std::vector<unsigned> h(nb);
    hipMemcpy(h.data(), d, nb * sizeof(unsigned), hipMemcpyDeviceToHost);
    int match = 0;
    for (int b = 0; b < nb; ++b) match += ((h[b] & 0xF) == (unsigned)(b % 8));
    printf("first 24 blocks -> XCC_ID:");
    for (int b = 0; b < 24; ++b) printf(" %u", h[b] & 0xF);
    printf("\nblocks with XCC_ID == blockIdx %% 8: %d of %d\n", match, nb);
    return 0;
}
