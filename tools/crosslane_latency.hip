// What moving a value between lanes costs a lone wave on gfx950, against the vector instruction it would feed:
// dependent chains of (cross-lane move, v_fma_f32) for DPP quad_perm / row_shr, ds_swizzle, ds_bpermute and
// v_readlane + broadcast, in clock64 ticks per link of the chain (compare tools/valu_latency.hip: a dependent
// v_fma_f32 alone).  Used to price "several lanes per sample" formulations of the rollout kernel.
//   hipcc --offload-arch=gfx950 -O2 -o build/crosslane tools/crosslane_latency.hip && build/crosslane
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
#define REP256(x) REP16(REP16(x))
__global__ void k(long long* out, float* sink, float seed) {
    float a = seed + threadIdx.x * 1e-3f, b = 0.5f, c = 1.0001f;
    int idx = ((threadIdx.x & ~7) | 3) << 2;          // lane 3 of every group of 8, as a byte address for bpermute
    long long t0, t1;
    t0 = clock64();
    REP256(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(c), "v"(b));)
    t1 = clock64();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    // VOP2 multiply-add whose first source comes from another lane of the quad (DPP is an operand modifier)
    t0 = clock64();
    REP256(asm volatile("v_fmac_f32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(c));)
    t1 = clock64();
    if (threadIdx.x == 0) out[1] = t1 - t0;
    // the same across a row of 16 lanes
    t0 = clock64();
    REP256(asm volatile("v_fmac_f32_dpp %0, %0, %1 row_shr:4 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(c));)
    t1 = clock64();
    if (threadIdx.x == 0) out[2] = t1 - t0;
    // broadcast of lane 3 of every group of 8 through the LDS crossbar, then the multiply-add
    t0 = clock64();
    REP256(asm volatile("ds_swizzle_b32 %0, %0 offset:swizzle(BITMASK_PERM, \"00p11\")\n s_waitcnt lgkmcnt(0)\n v_fma_f32 %0, %0, %1, %2"
                        : "+v"(a) : "v"(c), "v"(b));)
    t1 = clock64();
    if (threadIdx.x == 0) out[3] = t1 - t0;
    // arbitrary gather
    t0 = clock64();
    REP256(asm volatile("ds_bpermute_b32 %0, %3, %0\n s_waitcnt lgkmcnt(0)\n v_fma_f32 %0, %0, %1, %2"
                        : "+v"(a) : "v"(c), "v"(b), "v"(idx));)
    t1 = clock64();
    if (threadIdx.x == 0) out[4] = t1 - t0;
    // seven independent swizzle broadcasts in flight, then seven multiply-adds (a column of a 7x7 factor)
    float r0 = a, r1 = a + 1, r2 = a + 2, r3 = a + 3, r4 = a + 4, r5 = a + 5, r6 = a + 6;
    t0 = clock64();
    REP16(asm volatile("ds_swizzle_b32 %0, %0 offset:swizzle(BITMASK_PERM, \"00p11\")\n ds_swizzle_b32 %1, %1 offset:swizzle(BITMASK_PERM, \"00p11\")\n"
                       "ds_swizzle_b32 %2, %2 offset:swizzle(BITMASK_PERM, \"00p11\")\n ds_swizzle_b32 %3, %3 offset:swizzle(BITMASK_PERM, \"00p11\")\n"
                       "ds_swizzle_b32 %4, %4 offset:swizzle(BITMASK_PERM, \"00p11\")\n ds_swizzle_b32 %5, %5 offset:swizzle(BITMASK_PERM, \"00p11\")\n"
                       "ds_swizzle_b32 %6, %6 offset:swizzle(BITMASK_PERM, \"00p11\")\n s_waitcnt lgkmcnt(0)\n"
                       "v_fma_f32 %0, %0, %7, %8\n v_fma_f32 %1, %1, %7, %8\n v_fma_f32 %2, %2, %7, %8\n v_fma_f32 %3, %3, %7, %8\n"
                       "v_fma_f32 %4, %4, %7, %8\n v_fma_f32 %5, %5, %7, %8\n v_fma_f32 %6, %6, %7, %8"
                       : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6) : "v"(c), "v"(b));)
    t1 = clock64();
    if (threadIdx.x == 0) out[5] = t1 - t0;
    // v_readlane to a scalar register and back into a vector multiply-add (one value for the whole wave)
    t0 = clock64();
    REP256(asm volatile("v_readlane_b32 s20, %0, 3\n s_nop 3\n v_fma_f32 %0, s20, %1, %2" : "+v"(a) : "v"(c), "v"(b) : "s20");)
    t1 = clock64();
    if (threadIdx.x == 0) out[6] = t1 - t0;
    sink[threadIdx.x] = a + r0 + r1 + r2 + r3 + r4 + r5 + r6;
}
int main() {
    long long* d; float* s; hipMalloc(&d, 128); hipMalloc(&s, 1024);
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, s, 1.0f);
    long long h[16]; hipMemcpy(h, d, 128, hipMemcpyDeviceToHost);
    printf("clock64 ticks per link of a dependent chain, lone wave:\n");
    printf("  v_fma_f32                                   %.2f\n", h[0] / 256.0);
    printf("  v_fmac_f32 with DPP quad_perm source         %.2f\n", h[1] / 256.0);
    printf("  v_fmac_f32 with DPP row_shr:4 source         %.2f\n", h[2] / 256.0);
    printf("  ds_swizzle broadcast + wait + v_fma_f32     %.2f\n", h[3] / 256.0);
    printf("  ds_bpermute gather + wait + v_fma_f32       %.2f\n", h[4] / 256.0);
    printf("  7 swizzles in flight + wait + 7 v_fma_f32   %.2f  (per group of 14 instructions)\n", h[5] / 16.0);
    printf("  v_readlane + s_nop + v_fma_f32 (scalar src) %.2f\n", h[6] / 256.0);
    return 0;
}
