// Instruction census of the pieces of the per-sample step (compile with -S and count).
#include <hip/hip_runtime.h>
#include "../trajopt_amd/csrc/env_core.hpp"
#include "../trajopt_amd/csrc/models_gen.hpp"
using namespace mppi;
using MD = Reacher7Dof;
using S = Step<float, MD>;
constexpr int NV = MD::NV;

__global__ void __launch_bounds__(64) c_trig(const float* in, float* out) {
    float cs[NV], sn[NV];
    static_for<NV>([&](auto I) { sincos_t(in[I * 64 + threadIdx.x], sn[I], cs[I]); });
    static_for<NV>([&](auto I) { out[I * 64 + threadIdx.x] = cs[I]; out[(NV + I) * 64 + threadIdx.x] = sn[I]; });
}
__global__ void __launch_bounds__(64) c_bias(const float* in, float* out) {
    float cs[NV], sn[NV], qd[NV], b[NV];
    static_for<NV>([&](auto I) { cs[I] = in[I * 64 + threadIdx.x]; sn[I] = in[(NV + I) * 64 + threadIdx.x]; qd[I] = in[(2 * NV + I) * 64 + threadIdx.x]; });
    S::bias_forces(cs, sn, qd, b);
    static_for<NV>([&](auto I) { out[I * 64 + threadIdx.x] = b[I]; });
}
__global__ void __launch_bounds__(64) c_mass(const float* in, float* out) {
    float cs[NV], sn[NV], M[NV][NV];
    static_for<NV>([&](auto I) { cs[I] = in[I * 64 + threadIdx.x]; sn[I] = in[(NV + I) * 64 + threadIdx.x]; });
    S::mass_matrix(cs, sn, M);
    static_for<NV>([&](auto I) { static_for<I + 1>([&](auto J) { out[(I * NV + J) * 64 + threadIdx.x] = M[I][J]; }); });
}
__global__ void __launch_bounds__(64) c_chol(const float* in, float* out) {
    float A[NV][NV], b[NV];
    static_for<NV>([&](auto I) { b[I] = in[(49 + I) * 64 + threadIdx.x]; static_for<I + 1>([&](auto J) { A[I][J] = in[(I * NV + J) * 64 + threadIdx.x]; }); });
    float dg[NV];
    static_for<NV>([&](auto I) { dg[I] = in[(60 + I) * 64 + threadIdx.x]; });
    S::chol_solve<false>(A, dg, 0.f, dg, b);
    static_for<NV>([&](auto I) { out[I * 64 + threadIdx.x] = b[I]; });
}
__global__ void __launch_bounds__(64) c_substep(const float* in, float* out) {
    float q[NV], v[NV], a[NV];
    static_for<NV>([&](auto I) { q[I] = in[I * 64 + threadIdx.x]; v[I] = in[(NV + I) * 64 + threadIdx.x]; a[I] = in[(2 * NV + I) * 64 + threadIdx.x]; });
    typename S::Out o; o.hand = {{0, 0, 0}};
    unsigned am = ~0u; int sw;
    S::substep(q, v, a, in[0] > 0, o, am, sw);
    static_for<NV>([&](auto I) { out[I * 64 + threadIdx.x] = q[I]; out[(NV + I) * 64 + threadIdx.x] = v[I]; });
    out[1000 + threadIdx.x] = o.hand.v[0] + o.hand.v[1] + o.hand.v[2] + am;
}
