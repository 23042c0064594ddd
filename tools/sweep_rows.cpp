// Host study: which constraint rows change between the sweeps of the active-set iteration, seen by a wave of 64
// samples (the union over its lanes), for the bench workload.  Answers: how often could a re-sweep keep the leading
// columns of the previous factorisation if the rows that rarely change were eliminated first?
//   g++ -O2 -std=c++17 -o /tmp/sweep_rows tools/sweep_rows.cpp && /tmp/sweep_rows
#include <cstdio>
#include <random>
#include <vector>
static thread_local unsigned g_changed[20];       // per sweep: mask of rows that changed
#define MPPI_SWEEP_TRACE(sweep, row, changed) do { if ((changed) && (sweep) < 20) g_changed[(sweep)] |= 1u << (row); } while (0)
#include "../trajopt_amd/csrc/env_core.hpp"
#include "../trajopt_amd/csrc/models_gen.hpp"
using namespace mppi;
int main() {
    using MD = Reacher7Dof;
    using S = Step<float, MD>;
    constexpr int NV = MD::NV, K = 4096, H = 64, MAXS = 8;
    std::mt19937_64 gen(1);
    std::normal_distribution<double> nd(0.0, 1.0);
    // masks[(k, substep, sweep)]
    std::vector<unsigned> masks((size_t)K * H * 2 * MAXS, 0);
    std::vector<int> nsw((size_t)K * H * 2, 0);
    for (int k = 0; k < K; k++) {
        float e[H][NV];
        for (int t = 0; t < H; t++) for (int j = 0; j < NV; j++) e[t][j] = (float)nd(gen);
        for (int t = 2; t < H; t++) for (int j = 0; j < NV; j++) e[t][j] = 0.25f * e[t][j] + 0.8f * e[t - 1][j];
        float q[NV] = {0}, v[NV] = {0};
        typename S::Active am; am.reset();
        for (int t = 0; t < H; t++)
            for (int s = 0; s < 2; s++) {
                typename S::Out o; int n;
                for (auto& m : g_changed) m = 0;
                S::substep(q, v, e[t], false, o, am, n);
                const size_t at = ((size_t)k * H + t) * 2 + s;
                nsw[at] = n;
                for (int w = 0; w < MAXS; w++) masks[at * MAXS + w] = g_changed[w];
            }
    }
    long resweeps = 0, substeps = 0, rowcount[NV + 1] = {0}, lanes_changing = 0;
    long elig[128] = {0};
    for (int w = 0; w < K / 64; w++)
        for (int ts = 0; ts < H * 2; ts++) {
            substeps++;
            int most = 0;
            for (int l = 0; l < 64; l++) { const int n = nsw[((size_t)(w * 64 + l)) * H * 2 + ts]; most = n > most ? n : most; }
            // sweep index s (1-based as counted by the kernel): the settle after sweep s decides whether sweep s+1 runs
            for (int sw = 1; sw < most && sw < MAXS; sw++) {
                unsigned u = 0;
                for (int l = 0; l < 64; l++) {
                    const unsigned m = masks[(((size_t)(w * 64 + l)) * H * 2 + ts) * MAXS + sw];
                    u |= m; lanes_changing += m != 0;
                }
                resweeps++;
                for (int r = 0; r <= NV; r++) rowcount[r] += (u >> r) & 1;
                for (unsigned lead = 0; lead < 128; lead++) elig[lead] += (u & lead) == 0;      // lead: mask of rows eliminated first
            }
        }
    printf("wave-substeps %ld, re-sweeps %ld (%.3f per substep), lanes whose set changed per re-sweep %.2f\n", substeps, resweeps,
           resweeps / (double)substeps, lanes_changing / (double)resweeps);
    printf("share of re-sweeps in which row r changed in some lane:");
    for (int r = 0; r <= NV; r++) printf(" %d:%.3f", r, rowcount[r] / (double)resweeps);
    printf("\nbest leading sets (rows that no lane changed), by size:\n");
    for (int size = 1; size <= 5; size++) {
        unsigned best = 0; long bv = -1;
        for (unsigned lead = 1; lead < 128; lead++) if (__builtin_popcount(lead) == size && elig[lead] > bv) { bv = elig[lead]; best = lead; }
        printf("  %d rows {", size);
        for (int r = 0; r < NV; r++) if ((best >> r) & 1) printf(" %d", r);
        printf(" }: %.3f of the re-sweeps\n", bv / (double)resweeps);
    }
    return 0;
}
