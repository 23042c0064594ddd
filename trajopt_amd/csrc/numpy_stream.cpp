// The legacy numpy normal stream on the host, bit for bit: what `np.random.seed(s)` followed by
// `np.random.normal(size=...)` calls yields (MT19937 seeded by the Knuth recurrence, 53-bit doubles from two outputs,
// polar Box-Muller with the second variate of a pair cached for the next draw).  The reference's rollout workers
// draw their perturbations from exactly this stream (trajopt/utils.py:118-139); the planner's exact-comparison noise
// mode uploads it.  numpy makes the stream one value at a time; here the generator runs ahead in bulk and the
// logarithms and square roots of the accepted pairs are spread over threads.  Same libm, same operation order:
// same bits (tests compare against numpy itself).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <exception>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/mppi_hip.h"

// products and sums must round one by one, as in numpy's generator
#pragma STDC FP_CONTRACT OFF

namespace {

struct MT19937 {
    uint32_t key[624];
    int pos;
    void seed(uint32_t s) {
        for (int i = 0; i < 624; i++) {
            key[i] = s;
            s = 1812433253u * (s ^ (s >> 30)) + (uint32_t)i + 1u;
        }
        pos = 624;
    }
    void refill() {
        constexpr uint32_t UP = 0x80000000u, LO = 0x7fffffffu, MAT = 0x9908b0dfu;
        int i = 0;
        for (; i < 624 - 397; i++) {
            const uint32_t y = (key[i] & UP) | (key[i + 1] & LO);
            key[i] = key[i + 397] ^ (y >> 1) ^ ((y & 1u) ? MAT : 0u);
        }
        for (; i < 623; i++) {
            const uint32_t y = (key[i] & UP) | (key[i + 1] & LO);
            key[i] = key[i + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? MAT : 0u);
        }
        const uint32_t y = (key[623] & UP) | (key[0] & LO);
        key[623] = key[396] ^ (y >> 1) ^ ((y & 1u) ? MAT : 0u);
        pos = 0;
    }
    // tempered outputs in bulk
    void fill(uint32_t* out, size_t n) {
        size_t done = 0;
        while (done < n) {
            if (pos == 624) refill();
            size_t take = 624 - pos;
            if (take > n - done) take = n - done;
            for (size_t i = 0; i < take; i++) {
                uint32_t y = key[pos + i];
                y ^= y >> 11;
                y ^= (y << 7) & 0x9d2c5680u;
                y ^= (y << 15) & 0xefc60000u;
                y ^= y >> 18;
                out[done + i] = y;
            }
            pos += (int)take;
            done += take;
        }
    }
};

inline double to_double(uint32_t a, uint32_t b) {
    return ((a >> 5) * 67108864.0 + (b >> 6)) * (1.0 / 9007199254740992.0);      // a power of two: same bits as the division
}

// scratch that survives between calls (a planner asks for the same amount at every planning step)
struct Scratch {
    std::mutex mu;
    std::vector<uint32_t> raw;
    std::vector<double> x1, x2, r2;
    std::vector<uint32_t> att;          // attempt index of every kept pair
};
Scratch g_scratch;

template <class F> void parallel_for(int nthreads, size_t n, F&& body) {       // body(thread, lo, hi)
    if (nthreads <= 1 || n < 8192) { body(0, (size_t)0, n); return; }
    const size_t per = (n + nthreads - 1) / nthreads;
    std::vector<std::thread> pool;
    for (int t = 1; t < nthreads; t++) {
        const size_t lo = (size_t)t * per, hi = lo + per < n ? lo + per : n;
        if (lo < hi) pool.emplace_back([&body, t, lo, hi] { body(t, lo, hi); });
    }
    body(0, (size_t)0, per < n ? per : n);
    for (auto& th : pool) th.join();
}

}  // namespace

// n values of the stream that follows np.random.seed(seed), into out; nthreads: host threads for the bulk arithmetic.
// state_out (may be null) receives what np.random.get_state() would show afterwards: 624 key words, position,
// has_gauss, and the cached variate as raw bits -- 628 words -- so that a caller can leave the process-global numpy
// generator exactly where the reference's in-process worker leaves it.
extern "C" int mppi_numpy_normals(uint32_t seed, size_t n, double* out, int32_t nthreads, uint32_t* state_out) try {
    if (!out && n) return -1;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 64) nthreads = 64;
    std::lock_guard<std::mutex> lock(g_scratch.mu);
    Scratch& S = g_scratch;
    MT19937 mt;
    mt.seed(seed);
    const size_t pairs = (n + 1) / 2;                 // accepted attempts needed (the last may give one value only)
    size_t have = 0, words_before = 0, words_used = 0;
    double cached = 0.0;
    while (have < pairs) {
        // An attempt takes four generator outputs and is accepted with probability pi / 4.  Attempts are made in
        // bulk, a little more than the expected number; outputs drawn beyond the last attempt numpy would make are
        // simply never looked at (the generator object does not outlive the call: every call re-seeds).
        const size_t attempts = (size_t)((pairs - have) * 1.2733 + 8.0 * std::sqrt((double)(pairs - have))) + 64;
        if (S.raw.size() < attempts * 4) S.raw.resize(attempts * 4);
        if (S.x1.size() < attempts) { S.x1.resize(attempts); S.x2.resize(attempts); S.r2.resize(attempts); S.att.resize(attempts); }
        mt.fill(S.raw.data(), attempts * 4);                                   // sequential: the recurrence
        size_t count[64] = {0}, first[64] = {0};
        // rejection test and in-order compaction, every thread inside its own range of attempts
        parallel_for(nthreads, attempts, [&](int t, size_t lo, size_t hi) {
            size_t k = lo;
            for (size_t a = lo; a < hi; a++) {
                const uint32_t* w = &S.raw[4 * a];
                const double u1 = 2.0 * to_double(w[0], w[1]) - 1.0;
                const double u2 = 2.0 * to_double(w[2], w[3]) - 1.0;
                const double s = u1 * u1 + u2 * u2;
                S.x1[k] = u1; S.x2[k] = u2; S.r2[k] = s; S.att[k] = (uint32_t)a;
                k += (size_t)(!(s >= 1.0) & (s != 0.0));                      // kept only if accepted
            }
            first[t] = lo;
            count[t] = k - lo;
        });
        size_t offset[64], run = have;
        for (int t = 0; t < nthreads; t++) { offset[t] = run; run += count[t]; }
        // f = sqrt(-2 log(r2) / r2); the pair yields f * x2 first, then the cached f * x1
        parallel_for(nthreads, attempts, [&](int t, size_t, size_t) {
            for (size_t i = 0; i < count[t]; i++) {
                const size_t idx = offset[t] + i, k = first[t] + i;
                if (idx >= pairs) break;
                const double f = std::sqrt(-2.0 * std::log(S.r2[k]) / S.r2[k]);
                out[2 * idx] = f * S.x2[k];
                if (2 * idx + 1 < n) out[2 * idx + 1] = f * S.x1[k];
                if (idx == pairs - 1) {            // the last pair numpy would draw
                    words_used = words_before + 4 * ((size_t)S.att[k] + 1);
                    cached = f * S.x1[k];
                }
            }
        });
        have = run;
        words_before += attempts * 4;
    }
    if (state_out) {
        // the generator after exactly words_used outputs: re-run the recurrence (no tempering needed)
        MT19937 end;
        end.seed(seed);
        size_t refills = (words_used + 623) / 624;
        for (size_t r = 0; r < refills; r++) end.refill();
        std::memcpy(state_out, end.key, 624 * 4);
        state_out[624] = words_used ? (uint32_t)(words_used - 624 * (refills - 1)) : 624u;
        const bool has_gauss = (n % 2) == 1;
        state_out[625] = has_gauss ? 1u : 0u;
        const double g = has_gauss ? cached : 0.0;
        std::memcpy(&state_out[626], &g, 8);
    }
    return 0;
} catch (const std::exception&) {
    return -8;          // out of host memory (or no thread could be started)
}
