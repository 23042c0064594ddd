// Is a lone wave limited by instruction fetch?  Same arithmetic with 8-byte (VOP3) and 4-byte (VOP2) encodings,
// straight-line and in a short loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
#define REP256(x) REP16(REP16(x))
__global__ void k(long long* out, float* sink, float seed, int loops) {
    float a = seed + threadIdx.x, b = seed * 0.5f, c = 1.0001f, d = a + 1, e = a + 2, f = a + 3;
    long long t0, t1;
    t0 = clock64();
    REP256(asm volatile("v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %1, %4, %5, %1\n v_fma_f32 %2, %4, %5, %2\n v_fma_f32 %3, %4, %5, %3"
                        : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(c), "v"(b));)
    t1 = clock64();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    t0 = clock64();
    REP256(asm volatile("v_fmac_f32 %0, %4, %5\n v_fmac_f32 %1, %4, %5\n v_fmac_f32 %2, %4, %5\n v_fmac_f32 %3, %4, %5"
                        : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(c), "v"(b));)
    t1 = clock64();
    if (threadIdx.x == 0) out[1] = t1 - t0;
    // short loop: 16 x 4 instructions per iteration, 16 iterations
    t0 = clock64();
    for (int i = 0; i < loops; i++) {
        REP16(asm volatile("v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %1, %4, %5, %1\n v_fma_f32 %2, %4, %5, %2\n v_fma_f32 %3, %4, %5, %3"
                           : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(c), "v"(b));)
    }
    t1 = clock64();
    if (threadIdx.x == 0) out[2] = t1 - t0;
    t0 = clock64();
    for (int i = 0; i < loops; i++) {
        REP16(asm volatile("v_fmac_f32 %0, %4, %5\n v_fmac_f32 %1, %4, %5\n v_fmac_f32 %2, %4, %5\n v_fmac_f32 %3, %4, %5"
                           : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(c), "v"(b));)
    }
    t1 = clock64();
    if (threadIdx.x == 0) out[3] = t1 - t0;
    // dependent chains in both encodings
    t0 = clock64();
    REP256(asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(a) : "v"(c), "v"(b));)
    t1 = clock64();
    if (threadIdx.x == 0) out[4] = t1 - t0;
    t0 = clock64();
    REP256(asm volatile("v_fmac_f32 %0, %1, %0" : "+v"(a) : "v"(c));)
    t1 = clock64();
    if (threadIdx.x == 0) out[5] = t1 - t0;
    long long w0 = wall_clock64();
    t0 = clock64();
    REP256(asm volatile("v_fmac_f32 %0, %4, %5\n v_fmac_f32 %1, %4, %5\n v_fmac_f32 %2, %4, %5\n v_fmac_f32 %3, %4, %5"
                        : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(c), "v"(b));)
    t1 = clock64();
    long long w1 = wall_clock64();
    if (threadIdx.x == 0) { out[6] = t1 - t0; out[7] = w1 - w0; }
    sink[threadIdx.x] = a + d + e + f;
}
int main() {
    long long* d; float* s; hipMalloc(&d, 128); hipMalloc(&s, 1024);
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, s, 1.0f, 16);
    long long h[16]; hipMemcpy(h, d, 128, hipMemcpyDeviceToHost);
    printf("ticks per instruction, lone wave, 4 independent chains:\n");
    printf("  straight line v_fma_f32 (8 B)    %.2f\n", h[0] / 1024.0);
    printf("  straight line v_fmac_f32 (4 B)   %.2f\n", h[1] / 1024.0);
    printf("  64-instruction loop v_fma_f32    %.2f\n", h[2] / 1024.0);
    printf("  64-instruction loop v_fmac_f32   %.2f\n", h[3] / 1024.0);
    printf("dependent chain v_fma_f32 %.2f, v_fmac_f32 %.2f\n", h[4] / 256.0, h[5] / 256.0);
    printf("clock64 ticks per 100 MHz wall tick: %.2f\n", (double)h[6] / (double)h[7]);
    return 0;
}
