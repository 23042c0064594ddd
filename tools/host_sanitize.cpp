// Host build of the per-sample step under sanitizers (see tools/host_sanitize.sh): rolls a few samples of the arm
// and of the trees through Env::step in fp32 and fp64 and prints the results, so that an uninitialised value that
// reaches an output, a branch or an address is reported by MemorySanitizer, and undefined arithmetic or an
// out-of-bounds access by UBSan / ASan.
#include <cstdio>
#include <cstdlib>
#include "../tests/host_core_harness.cpp"

int main() {
    const int H = 48;
    double act[H * 18], q0[19], v0[18], target[3] = {0.05, 0.0, 0.1};
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) / 16777216.0 - 0.5) * 2.0; };
    for (int model = 0; model < 5; model++) {
        const int nv = model < 2 ? 7 : 18, nq = model < 3 ? nv : 19;
        for (int sample = 0; sample < 6; sample++) {
            for (int i = 0; i < H * nv; i++) act[i] = 1.5 * rnd();
            for (int i = 0; i < nq; i++) q0[i] = 0.0;
            for (int i = 0; i < nv; i++) v0[i] = 0.0;
            if (model < 2) { const double q[7] = {0.3, 1.2, 1.6, -0.1, 1.4, -0.05, 1.45}; for (int i = 0; i < 7; i++) q0[i] = q[i]; }
            if (nq == 19) q0[3] = 1.0;          // unit quaternion
            for (int f64 = 0; f64 < 2; f64++) {
                double q[19], v[18], hand[3], rew[H], R; int sw;
                core_rollout(f64, model, q0, v0, target, act, H, sample % 2, nullptr, 0, 1.0, 0.97, q, v, hand, rew, &R, &sw);
                double chk = R + hand[0] + hand[1] + hand[2];
                for (int i = 0; i < nq; i++) chk += q[i];
                for (int i = 0; i < nv; i++) chk += v[i];
                for (int t = 0; t < H; t++) chk += rew[t];
                printf("model %d sample %d %s: checksum %.12g sweeps %d\n", model, sample, f64 ? "f64" : "f32", chk, sw);
            }
        }
    }
    return 0;
}
