// Host study of the active-set iteration: how many sweeps a sample needs per substep when the set is carried
// over from the previous substep, and what a wave of 64 samples (which repeats until its slowest lane has
// settled) pays.  Workload: reacher_7dof from the zero pose, zero base actions, filtered unit noise.
//   g++ -O2 -std=c++17 -o /tmp/sweep_stats tools/sweep_stats.cpp && /tmp/sweep_stats
#include <cstdio>
#include <random>
#include <vector>
#include "../trajopt_amd/csrc/env_core.hpp"
#include "../trajopt_amd/csrc/models_gen.hpp"
using namespace mppi;
int main() {
    using MD = Reacher7Dof;
    using S = Step<double, MD>;
    constexpr int NV = MD::NV, K = 4096, H = 64;
    std::mt19937_64 gen(1);
    std::normal_distribution<double> nd(0.0, 1.0);
    std::vector<int> sw((size_t)K * H * 2);
    long hist[16] = {0};
    for (int k = 0; k < K; k++) {
        double e[H][NV];
        for (int t = 0; t < H; t++) for (int j = 0; j < NV; j++) e[t][j] = nd(gen);
        for (int t = 2; t < H; t++) for (int j = 0; j < NV; j++) e[t][j] = 0.25 * e[t][j] + 0.8 * e[t - 1][j];
        double q[NV] = {0}, v[NV] = {0};
        typename S::Active am; am.reset();
        for (int t = 0; t < H; t++)
            for (int s = 0; s < 2; s++) {
                typename S::Out o; int n;
                S::substep(q, v, e[t], false, o, am, n);
                sw[((size_t)k * H + t) * 2 + s] = n;
                hist[n < 15 ? n : 15]++;
            }
    }
    printf("per sample and substep: sweeps histogram (0 = no rows):");
    for (int i = 0; i < 8; i++) printf(" %d:%.4f", i, hist[i] / double(K * H * 2));
    printf("\n");
    double wave_sum = 0, lane_sum = 0; long wh[16] = {0};
    for (int w = 0; w < K / 64; w++)
        for (int ts = 0; ts < H * 2; ts++) {
            int m = 0;
            for (int l = 0; l < 64; l++) { int n = sw[((size_t)(w * 64 + l)) * H * 2 + ts]; m = n > m ? n : m; lane_sum += n; }
            wave_sum += m; wh[m < 15 ? m : 15]++;
        }
    printf("mean sweeps per substep: sample %.3f, wave of 64 %.3f\nwave histogram:", lane_sum / (K * H * 2.0), wave_sum / (K / 64 * H * 2.0));
    for (int i = 0; i < 8; i++) printf(" %d:%.4f", i, wh[i] / double(K / 64 * H * 2));
    printf("\n");
    return 0;
}
