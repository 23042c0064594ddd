// Host study of solves that run into the sweep cap of the active-set iteration: random start poses of reacher_7dof
// (the 1024-planner workload: q uniform in +-0.3 around a bent pose), filtered unit noise, H = 64, in fp32 and fp64.
// Prints the histogram of sweeps per substep and, for the capped ones, the residuals of the rows that keep flipping.
//   g++ -O2 -std=c++17 -o /tmp/sweep_cap tools/sweep_cap_study.cpp && /tmp/sweep_cap
#include <cstdio>
#include <random>
#include "../trajopt_amd/csrc/env_core.hpp"
#include "../trajopt_amd/csrc/models_gen.hpp"
using namespace mppi;
template <class T> void study(int K) {
    using MD = Reacher7Dof;
    using S = Step<T, MD>;
    constexpr int NV = MD::NV, H = 64;
    std::mt19937_64 gen(1);
    std::normal_distribution<double> nd(0.0, 1.0);
    std::uniform_real_distribution<double> ud(-0.3, 0.3);
    long hist[16] = {0}, total = 0;
    for (int k = 0; k < K; k++) {
        double e[H][NV];
        for (int t = 0; t < H; t++) for (int j = 0; j < NV; j++) e[t][j] = nd(gen);
        for (int t = 2; t < H; t++) for (int j = 0; j < NV; j++) e[t][j] = 0.25 * e[t][j] + 0.8 * e[t - 1][j];
        T q[NV], v[NV];
        for (int j = 0; j < NV; j++) { q[j] = T(ud(gen)); v[j] = T(0); }
        q[3] -= T(0.8); q[5] -= T(0.5);
        typename S::Active am; am.reset();
        for (int t = 0; t < H; t++) {
            T a[NV];
            for (int j = 0; j < NV; j++) a[j] = T(e[t][j]);
            for (int s = 0; s < 2; s++) {
                typename S::Out o; int n;
                T q0[NV], v0[NV];
                for (int j = 0; j < NV; j++) { q0[j] = q[j]; v0[j] = v[j]; }
                S::substep(q, v, a, false, o, am, n);
                hist[n < 15 ? n : 15]++; total++;
                if (n > 12 && hist[13] <= 3) {
                    printf("  capped solve: sample %d step %d.%d q =", k, t, s);
                    for (int j = 0; j < NV; j++) printf(" %.6f", double(q0[j]));
                    printf("\n    v =");
                    for (int j = 0; j < NV; j++) printf(" %.4f", double(v0[j]));
                    printf("\n");
                }
            }
        }
    }
    printf("%s: sweeps per substep over %ld substeps:", sizeof(T) == 4 ? "fp32" : "fp64", total);
    for (int i = 0; i < 14; i++) printf(" %d:%ld", i, hist[i]);
    printf("\n");
}
int main() {
    study<float>(20000);
    study<double>(20000);
    return 0;
}
