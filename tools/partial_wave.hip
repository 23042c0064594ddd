// Does a wave with fewer than 64 active lanes issue vector instructions faster on gfx950?  A lone wave runs chains of
// v_fma_f32 / v_pk_fma_f32 with 64, 32, 16 and 8 lanes launched (block size) and with 64 launched but EXEC narrowed;
// ticks of clock64 and wall-clock time of a long loop are printed per instruction.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
#define REP256(x) REP16(REP16(x))
typedef float float2v __attribute__((ext_vector_type(2)));

__global__ void k(long long* out, float* sink, float seed, int loops, int narrow) {
    float a = seed + threadIdx.x, b = seed * 0.5f, c = 1.0001f, d = a + 1, e = a + 2, f = a + 3;
    float2v p = {a, d}, q = {c, c}, r = {b, b}, p2 = {e, f}, p3 = {f, a}, p4 = {d, e};
    if (narrow && (int)threadIdx.x >= narrow) return;       // EXEC narrowed inside a 64-lane launch
    long long t0 = clock64();
    for (int l = 0; l < loops; l++) {
        REP256(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                            : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(c), "v"(b));)
    }
    long long t1 = clock64();
    for (int l = 0; l < loops; l++) {
        REP256(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(c), "v"(b));)
    }
    long long t2 = clock64();
    for (int l = 0; l < loops; l++) {
        REP256(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                            : "+v"(p), "+v"(p2), "+v"(p3), "+v"(p4) : "v"(q), "v"(r));)
    }
    long long t3 = clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = t2 - t1; out[2] = t3 - t2; }
    sink[threadIdx.x] = a + d + e + f + p.x + p.y + p2.x + p3.y + p4.x;
}

int main() {
    long long* d; float* s;
    if (hipMalloc(&d, 128) != hipSuccess || hipMalloc(&s, 1024) != hipSuccess) return 1;
    const int loops = 2000;
    const int cases[][2] = {{64, 0}, {32, 0}, {16, 0}, {8, 0}, {1, 0}, {64, 32}, {64, 16}, {64, 1}};
    for (auto& cs : cases) {
        hipLaunchKernelGGL(k, dim3(1), dim3(cs[0]), 0, 0, d, s, 1.0f, 10, cs[1]);
        hipDeviceSynchronize();
        auto w0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(k, dim3(1), dim3(cs[0]), 0, 0, d, s, 1.0f, loops, cs[1]);
        if (hipDeviceSynchronize() != hipSuccess) return 2;
        auto w1 = std::chrono::steady_clock::now();
        long long h[4];
        hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
        const double ns = std::chrono::duration<double, std::nano>(w1 - w0).count();
        const double n_instr = (double)loops * 256 * (4 + 1 + 4);
        printf("lanes launched %2d, narrowed to %2d: ticks per instr: 4 chains %.2f, dependent %.2f, 4 packed chains %.2f; wall %.3f ns per instr\n",
               cs[0], cs[1], h[0] / (loops * 1024.0), h[1] / (loops * 256.0), h[2] / (loops * 1024.0), ns / n_instr);
    }
    return 0;
}
