// Where do the waves of a workgroup land?  Prints (xcc, se, cu, simd) of every wave of a few blocks.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned* out) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    long long t = wall_clock64();
    while (wall_clock64() - t < 20000) {}      // keep the block resident so that blocks do not reuse slots
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        out[2 * w] = hw; out[2 * w + 1] = xcc;
    }
}
int main(int argc, char** argv) {
    const int blocks = 64, threads = argc > 1 ? atoi(argv[1]) : 128;
    const int waves = blocks * threads / 64;
    unsigned* d; hipMalloc(&d, waves * 8);
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 0, 0, d);
    std::vector<unsigned> h(2 * waves);
    hipMemcpy(h.data(), d, waves * 8, hipMemcpyDeviceToHost);
    int same = 0;
    for (int b = 0; b < blocks; b++) {
        if (b < 8) printf("block %2d:", b);
        for (int w = 0; w < threads / 64; w++) {
            unsigned hw = h[2 * (b * threads / 64 + w)], xcc = h[2 * (b * threads / 64 + w) + 1];
            if (b < 8) printf("  [xcc %u se %u cu %u simd %u slot %u]", xcc & 15, (hw >> 13) & 7, (hw >> 8) & 15, (hw >> 4) & 3, hw & 15);
        }
        if (b < 8) printf("\n");
        if (threads >= 128) same += ((h[2 * (b * threads / 64)] >> 4) & 3) == ((h[2 * (b * threads / 64 + 1)] >> 4) & 3);
    }
    printf("blocks whose waves 0 and 1 share a SIMD: %d of %d\n", same, blocks);
    return 0;
}
