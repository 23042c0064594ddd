// Lone-wave issue and dependency latencies on gfx950: cycles per instruction for dependent and
// independent chains of v_fma_f32, v_rsq_f32, and the v_cmp -> s_and -> v_cndmask round trip.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
#define REP256(x) REP16(REP16(x))
__global__ void k(long long* out, float* sink, float seed) {
    float a = seed + threadIdx.x, b = seed * 0.5f, c = 1.0001f, d = a + 1, e = a + 2, f = a + 3;
    long long t0, t1;
    // dependent fma chain
    t0 = clock64();
    REP256(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(c), "v"(b));)
    t1 = clock64();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    // four independent fma chains interleaved
    t0 = clock64();
    REP256(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                        : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(c), "v"(b));)
    t1 = clock64();
    if (threadIdx.x == 0) out[1] = t1 - t0;
    // two independent chains interleaved
    t0 = clock64();
    REP256(asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3" : "+v"(a), "+v"(d) : "v"(c), "v"(b));)
    t1 = clock64();
    if (threadIdx.x == 0) out[2] = t1 - t0;
    // dependent rsq chain
    a = fabsf(a) + 1.0f;
    t0 = clock64();
    REP256(asm volatile("v_rsq_f32 %0, %0" : "+v"(a));)
    t1 = clock64();
    if (threadIdx.x == 0) out[3] = t1 - t0;
    // rsq followed by dependent mul, then independent fma x3
    t0 = clock64();
    REP256(asm volatile("v_rsq_f32 %0, %0\n v_mul_f32 %0, %0, %4\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                        : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(c), "v"(b));)
    t1 = clock64();
    if (threadIdx.x == 0) out[4] = t1 - t0;
    // four independent rsq chains interleaved: the issue cost of a transcendental
    a = fabsf(a) + 1.0f; d = fabsf(d) + 1.0f; e = fabsf(e) + 1.0f; f = fabsf(f) + 1.0f;
    t0 = clock64();
    REP256(asm volatile("v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3" : "+v"(a), "+v"(d), "+v"(e), "+v"(f));)
    t1 = clock64();
    if (threadIdx.x == 0) out[9] = t1 - t0;
    // one rsq and seven independent fmas: does the transcendental hold the issue port?
    t0 = clock64();
    REP256(asm volatile("v_rsq_f32 %0, %0\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n v_fma_f32 %1, %1, %4, %5"
                        : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(c), "v"(b));)
    t1 = clock64();
    if (threadIdx.x == 0) out[10] = t1 - t0;
    // eight independent fmas (same shape without the rsq)
    t0 = clock64();
    REP256(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n v_fma_f32 %1, %1, %4, %5"
                        : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(c), "v"(b));)
    t1 = clock64();
    if (threadIdx.x == 0) out[11] = t1 - t0;
    // v_cmp -> s_and -> v_cndmask round trip, dependent
    t0 = clock64();
    REP256(asm volatile("v_cmp_gt_f32 vcc, %0, %1\n s_and_b64 vcc, vcc, exec\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(a) : "v"(b), "v"(c) : "vcc");)
    t1 = clock64();
    if (threadIdx.x == 0) out[5] = t1 - t0;
    // v_cmp -> v_cndmask (no SALU), dependent
    t0 = clock64();
    REP256(asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(a) : "v"(b), "v"(c) : "vcc");)
    t1 = clock64();
    if (threadIdx.x == 0) out[6] = t1 - t0;
    // dependent fp64 fma chain
    double x = a, y = 1.0000001, z = 0.5;
    t0 = clock64();
    REP256(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));)
    t1 = clock64();
    if (threadIdx.x == 0) out[7] = t1 - t0;
    // dependent v_mul chain + ds_write/ds_read round trip cost: LDS store then load same address
    __shared__ float l[64];
    t0 = clock64();
    REP16(l[threadIdx.x] = a; __syncthreads(); a = l[threadIdx.x ^ 1] + 1.0f; __syncthreads();)
    t1 = clock64();
    if (threadIdx.x == 0) out[8] = t1 - t0;
    sink[threadIdx.x] = a + d + e + f + (float)x;
}
int main() {
    long long* d; float* s; hipMalloc(&d, 128); hipMalloc(&s, 1024);
    for (int waves = 1; waves <= 2; waves++) {
        for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, s, 1.0f);
        long long h[16]; hipMemcpy(h, d, 128, hipMemcpyDeviceToHost);
        printf("cycles per instruction, lone wave (clock64 ticks):\n");
        printf("  dependent v_fma_f32 chain        %.2f\n", h[0] / 256.0);
        printf("  4 interleaved chains, per fma    %.2f\n", h[1] / 1024.0);
        printf("  2 interleaved chains, per fma    %.2f\n", h[2] / 512.0);
        printf("  dependent v_rsq_f32 chain        %.2f\n", h[3] / 256.0);
        printf("  rsq+dep mul+3 indep fma (group)  %.2f\n", h[4] / 256.0);
        printf("  v_cmp->s_and->v_cndmask (group)  %.2f\n", h[5] / 256.0);
        printf("  v_cmp->v_cndmask (group)         %.2f\n", h[6] / 256.0);
        printf("  dependent v_fma_f64 chain        %.2f\n", h[7] / 256.0);
        printf("  LDS write+barrier+read+barrier   %.2f\n", h[8] / 16.0);
        printf("  4 interleaved rsq chains, per rsq %.2f\n", h[9] / 1024.0);
        printf("  rsq + 7 independent fma (group)  %.2f\n", h[10] / 256.0);
        printf("  8 independent fma (group)        %.2f\n", h[11] / 256.0);
        break;
    }
    return 0;
}
