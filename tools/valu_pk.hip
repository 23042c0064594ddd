// Lone-wave cost of packed fp32 arithmetic: cycles per v_pk_fma_f32 against v_fma_f32, independent chains.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
#define REP256(x) REP16(REP16(x))
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k(long long* out, float* sink, float seed) {
    f2 a = {seed + threadIdx.x, seed}, b = {seed * 0.5f, 1.f}, c = {1.0001f, 0.999f}, d = a + 1.f, e = a + 2.f, f = a + 3.f;
    float sa = a.x, sd = d.x, se = e.x, sf = f.x, sb = b.x, sc = c.x;
    long long t0, t1;
    t0 = clock64();
    REP256(asm volatile("v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3"
                        : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(c), "v"(b));)
    t1 = clock64();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    t0 = clock64();
    REP256(asm volatile("v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %1, %4, %5, %1\n v_fma_f32 %2, %4, %5, %2\n v_fma_f32 %3, %4, %5, %3"
                        : "+v"(sa), "+v"(sd), "+v"(se), "+v"(sf) : "v"(sc), "v"(sb));)
    t1 = clock64();
    if (threadIdx.x == 0) out[1] = t1 - t0;
    t0 = clock64();
    REP256(asm volatile("v_pk_mul_f32 %0, %4, %0\n v_pk_mul_f32 %1, %4, %1\n v_pk_mul_f32 %2, %4, %2\n v_pk_mul_f32 %3, %4, %3"
                        : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(c));)
    t1 = clock64();
    if (threadIdx.x == 0) out[2] = t1 - t0;
    t0 = clock64();
    REP256(asm volatile("v_mul_f32 %0, %4, %0\n v_mul_f32 %1, %4, %1\n v_mul_f32 %2, %4, %2\n v_mul_f32 %3, %4, %3"
                        : "+v"(sa), "+v"(sd), "+v"(se), "+v"(sf) : "v"(sc));)
    t1 = clock64();
    if (threadIdx.x == 0) out[3] = t1 - t0;
    sink[threadIdx.x] = a.x + d.y + e.x + f.y + sa + sd + se + sf;
}
int main() {
    long long* d; float* s; hipMalloc(&d, 128); hipMalloc(&s, 1024);
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, s, 1.0f);
    long long h[16]; hipMemcpy(h, d, 128, hipMemcpyDeviceToHost);
    printf("cycles per instruction, lone wave, 4 independent chains:\n");
    printf("  v_pk_fma_f32 %.2f   v_fma_f32 %.2f\n", h[0] / 1024.0, h[1] / 1024.0);
    printf("  v_pk_mul_f32 %.2f   v_mul_f32 %.2f\n", h[2] / 1024.0, h[3] / 1024.0);
    return 0;
}
