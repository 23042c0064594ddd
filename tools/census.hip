// Instruction census of the pieces of the per-sample step: compile with -S and count per kernel
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-signed-zeros -ffinite-math-only -fno-slp-vectorize -S --cuda-device-only -o /tmp/census.s tools/census.hip
//   python tools/isa_count.py /tmp/census.s c_
// (each kernel's count includes its loads and stores of the inputs and outputs)
#include <hip/hip_runtime.h>
#include "../trajopt_amd/csrc/env_core.hpp"
#include "../trajopt_amd/csrc/models_gen.hpp"
using namespace mppi;
using MD = Reacher7Dof;
using S = Step<float, MD>;
constexpr int NV = MD::NV;
#define IN(n) in[(n) * 64 + threadIdx.x]
#define OUT(n) out[(n) * 64 + threadIdx.x]

__global__ void __launch_bounds__(64) c_io_only(const float* in, float* out) {
    static_for<3 * NV>([&](auto I) { OUT(I) = IN(I); });
}
__global__ void __launch_bounds__(64) c_frames(const float* in, float* out) {
    float q[NV];
    typename S::Frames fr;
    static_for<NV>([&](auto I) { q[I] = IN(I); });
    S::frames(q, fr);
    static_for<NV>([&](auto I) { OUT(I) = fr.cs(I); OUT(NV + I) = fr.sn(I); });
}
__global__ void __launch_bounds__(64) c_smooth_force(const float* in, float* out) {
    float q[NV], qd[NV], u[NV], fs[NV];
    typename S::Frames fr;
    static_for<NV>([&](auto I) { q[I] = 0; fr.cs(I) = IN(I); fr.sn(I) = IN(NV + I); qd[I] = IN(2 * NV + I); u[I] = IN(3 * NV + I); });
    S::smooth_force(q, fr, qd, u, fs);
    static_for<NV>([&](auto I) { OUT(I) = fs[I]; });
}
__global__ void __launch_bounds__(64) c_mass_matrix(const float* in, float* out) {
    float q[NV], M[NV][NV];
    typename S::Frames fr;
    static_for<NV>([&](auto I) { q[I] = 0; fr.cs(I) = IN(I); fr.sn(I) = IN(NV + I); });
    S::mass_matrix(q, fr, M);
    static_for<NV>([&](auto I) { static_for<I + 1>([&](auto J) { OUT(I * NV + J) = M[I][J]; }); });
}
__global__ void __launch_bounds__(64) c_build_rows(const float* in, float* out) {
    float q[NV], qd[NV];
    typename S::Frames fr;
    static_for<NV>([&](auto I) { q[I] = IN(3 * NV + I); fr.cs(I) = IN(I); fr.sn(I) = IN(NV + I); qd[I] = IN(2 * NV + I); });
    typename S::Rows r;
    S::build_rows(q, fr, qd, r);
    typename S::ContactProbe pr;
    const bool wave = S::contact_probe(q, fr, pr);
    static_for<NV>([&](auto I) { OUT(I) = r.Dl[I / 2][I % 2]; OUT(NV + I) = r.sal[I / 2][I % 2]; OUT(2 * NV + I) = r.xs[I / 2][I % 2]; });
    OUT(4 * NV) = pr.dist;
    if (wave) {
        typename S::ContactRow cr;
        S::contact_row(fr, qd, pr, cr);
        static_for<NV>([&](auto I) { OUT(3 * NV + I) = cr.Jc[I]; });
        OUT(4 * NV + 1) = cr.Dc + cr.arc;
    }
}
__global__ void __launch_bounds__(64) c_hand_site(const float* in, float* out) {
    float q[NV];
    typename S::Frames fr;
    static_for<NV>([&](auto I) { q[I] = 0; fr.cs(I) = IN(I); fr.sn(I) = IN(NV + I); });
    const Vec3<float> h = S::hand_site(q, fr);
    OUT(0) = h.v[0]; OUT(1) = h.v[1]; OUT(2) = h.v[2];
}
__global__ void __launch_bounds__(64) c_chol_solve(const float* in, float* out) {
    float A[NV][NV], b[NV], dg[NV];
    static_for<NV>([&](auto I) { b[I] = IN(49 + I); dg[I] = IN(60 + I); static_for<I + 1>([&](auto J) { A[I][J] = IN(I * NV + J); }); });
    S::chol_solve<false>(A, dg, 0.f, dg, b);
    static_for<NV>([&](auto I) { OUT(I) = b[I]; });
}
__global__ void __launch_bounds__(64) c_dual_factor(const float* in, float* out) {
    float A[NV][NV], b[NV], dg[NV], d2[NV];
    static_for<NV>([&](auto I) { b[I] = IN(49 + I); dg[I] = IN(60 + I); d2[I] = IN(70 + I); static_for<I + 1>([&](auto J) { A[I][J] = IN(I * NV + J); }); });
    typename S::DampedFactor f;
    S::dual_factor<false, true>(A, dg, 0.f, dg, b, d2, f.D, f.inv);
    static_for<NV>([&](auto I) { OUT(I) = b[I]; OUT(NV + I) = f.inv[I]; static_for<I>([&](auto J) { OUT(20 + I * NV + J) = f.D.a[I][J][1]; }); });
}
__global__ void __launch_bounds__(64) c_integrate(const float* in, float* out) {
    typename S::DampedFactor f;
    float fs[NV], fc[NV], q[NV], qd[NV];
    static_for<NV>([&](auto I) { fs[I] = IN(I); fc[I] = IN(NV + I); q[I] = IN(2 * NV + I); qd[I] = IN(3 * NV + I); f.inv[I] = IN(4 * NV + I);
                                 static_for<I>([&](auto J) { f.D.a[I][J] = half_pair<1>(IN(40 + I * NV + J)); }); });
    float qacc[NV];
    static_for<NV>([&](auto I) { qacc[I] = fs[I] + fc[I]; });
    S::dual_apply(f.D, f.inv, qacc);
    static_for<NV>([&](auto I) { qd[I] = qd[I] + float(MD::H) * qacc[I]; });
    S::advance_pos(q, qd, q);
    static_for<NV>([&](auto I) { OUT(I) = q[I]; OUT(NV + I) = qd[I]; });
}
__global__ void __launch_bounds__(64) c_substep(const float* in, float* out) {
    float q[NV], v[NV], a[NV];
    static_for<NV>([&](auto I) { q[I] = IN(I); v[I] = IN(NV + I); a[I] = IN(2 * NV + I); });
    typename S::Out o; o.hand = {{0, 0, 0}};
    typename S::Active am; am.reset(); int sw;
    S::substep(q, v, a, in[0] > 0, o, am, sw);
    static_for<NV>([&](auto I) { OUT(I) = q[I]; OUT(NV + I) = v[I]; });
    OUT(1000) = o.hand.v[0] + o.hand.v[1] + o.hand.v[2] + am.idle[0][0] + am.idle_contact;
}
