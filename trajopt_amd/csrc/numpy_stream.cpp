// The legacy numpy normal stream on the host, bit for bit: what `np.random.seed(s)` followed by
// `np.random.normal(size=...)` calls yields (MT19937 seeded by the Knuth recurrence, 53-bit doubles from two outputs,
// polar Box-Muller with the second variate of a pair cached for the next draw).  The reference's rollout workers
// draw their perturbations from exactly this stream (trajopt/utils.py:118-139); the planner's exact-comparison noise
// mode uploads it.  numpy makes the stream one value at a time; here the generator runs ahead in bulk and the
// logarithms and square roots of the accepted pairs are spread over threads.  Same libm, same operation order:
// same bits (tests compare against numpy itself).
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <exception>
#include <mutex>
#include <sched.h>
#include <thread>
#include <vector>

#include "../../include/mppi_hip.h"

// products and sums must round one by one, as in numpy's generator
#pragma STDC FP_CONTRACT OFF

namespace {

typedef uint32_t u32x8 __attribute__((vector_size(32), aligned(4)));
constexpr uint32_t UP = 0x80000000u, LO = 0x7fffffffu, MAT = 0x9908b0dfu;

inline uint32_t twist(uint32_t a, uint32_t b, uint32_t far) {
    const uint32_t y = (a & UP) | (b & LO);
    return far ^ (y >> 1) ^ ((0u - (y & 1u)) & MAT);
}
// One regeneration of the 624-word state, eight words at a time: word i needs the old words i, i+1 and word
// i+397 (mod 624), which is old for i < 227 and otherwise was rewritten 227 positions earlier, so blocks of eight
// never read a word of their own block's making.  `out` receives the tempered words.
#define MT_REFILL_BODY(LOAD, STORE)                                                                            \
    int i = 0;                                                                                                 \
    for (; i + 8 <= 227; i += 8) {                                                                             \
        const u32x8 a = LOAD(key + i), b = LOAD(key + i + 1), f = LOAD(key + i + 397);                         \
        const u32x8 y = (a & UP) | (b & LO);                                                                   \
        STORE(key + i, f ^ (y >> 1) ^ ((0u - (y & 1u)) & MAT));                                                \
    }                                                                                                          \
    for (; i < 227; i++) key[i] = twist(key[i], key[i + 1], key[i + 397]);                                     \
    for (; i + 8 <= 623; i += 8) {                                                                             \
        const u32x8 a = LOAD(key + i), b = LOAD(key + i + 1), f = LOAD(key + i - 227);                         \
        const u32x8 y = (a & UP) | (b & LO);                                                                   \
        STORE(key + i, f ^ (y >> 1) ^ ((0u - (y & 1u)) & MAT));                                                \
    }                                                                                                          \
    for (; i < 623; i++) key[i] = twist(key[i], key[i + 1], key[i - 227]);                                     \
    key[623] = twist(key[623], key[0], key[396]);                                                              \
    for (i = 0; i + 8 <= 624; i += 8) {                                                                        \
        u32x8 y = LOAD(key + i);                                                                               \
        y ^= y >> 11;                                                                                          \
        y ^= (y << 7) & 0x9d2c5680u;                                                                           \
        y ^= (y << 15) & 0xefc60000u;                                                                          \
        y ^= y >> 18;                                                                                          \
        STORE(out + i, y);                                                                                     \
    }

#define load8(p) (*reinterpret_cast<const u32x8*>(p))
#define store8(p, v) (*reinterpret_cast<u32x8*>(p) = (v))

__attribute__((target("avx2"))) void refill_tempered_avx2(uint32_t* key, uint32_t* out) { MT_REFILL_BODY(load8, store8) }
void refill_tempered_base(uint32_t* key, uint32_t* out) { MT_REFILL_BODY(load8, store8) }

struct MT19937 {
    uint32_t key[624];
    uint32_t block[624];        // tempered words of the current state
    int pos;
    bool wide;
    void seed(uint32_t s) {
        for (int i = 0; i < 624; i++) {
            key[i] = s;
            s = 1812433253u * (s ^ (s >> 30)) + (uint32_t)i + 1u;
        }
        pos = 624;
        wide = __builtin_cpu_supports("avx2");
    }
    void refill(uint32_t* out) { if (wide) refill_tempered_avx2(key, out); else refill_tempered_base(key, out); }
    // the next n tempered outputs
    void fill(uint32_t* out, size_t n) {
        size_t done = 0;
        while (done < n) {
            if (pos == 624) {
                if (n - done >= 624) { refill(out + done); done += 624; continue; }      // whole blocks go straight out
                refill(block);
                pos = 0;
            }
            size_t take = 624 - pos;
            if (take > n - done) take = n - done;
            std::memcpy(out + done, block + pos, take * 4);
            pos += (int)take;
            done += take;
        }
    }
};

inline double to_double(uint32_t a, uint32_t b) {
    return ((a >> 5) * 67108864.0 + (b >> 6)) * (1.0 / 9007199254740992.0);      // a power of two: same bits as the division
}

constexpr size_t CHUNK = 8192;          // attempts a thread takes at a time
constexpr size_t RING = 32;             // chunks of generator output in flight (4 MiB: stays in the last-level cache)

struct Scratch {                        // survives between calls (a planner asks for the same amount at every step)
    std::mutex mu;
    std::vector<uint32_t> ring;
};
Scratch g_scratch[64];                  // one per concurrent stream of mppi_numpy_normals_many; [0] for the single call

// wait for `ready()`: a short spin, then give the processor away (threads may outnumber processors)
template <class F> inline void wait_until(F&& ready) {
    for (int spins = 0; !ready(); spins++) {
        if (spins < 256) __builtin_ia32_pause();
        else std::this_thread::yield();
    }
}

int usable_processors() {
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int n = CPU_COUNT(&set); if (n > 0) return n; }
    const unsigned n = std::thread::hardware_concurrency();
    return n ? (int)n : 1;
}

// One stream: n values after np.random.seed(seed) into out, state_out as described at mppi_numpy_normals.
// The calling thread runs the recurrence (sequential by nature) chunk by chunk into a ring; the other threads follow
// behind it: a chunk's attempts are tested and compacted, its slot is handed back, its number of accepted pairs is
// published, and as soon as the chunks before it have published theirs the chunk knows where its values go in `out`
// and writes them there.
void generate(uint32_t seed, size_t n, double* out, int nthreads, uint32_t* state_out, Scratch& S) {
    std::lock_guard<std::mutex> lock(S.mu);
    if (S.ring.size() < RING * CHUNK * 4) S.ring.resize(RING * CHUNK * 4);
    uint32_t* const ring = S.ring.data();
    MT19937 mt;
    mt.seed(seed);
    const size_t pairs = (n + 1) / 2;                 // accepted attempts needed (the last may give one value only)
    size_t have = 0;
    MT19937 at_last = mt;                             // generator at the start of the chunk that holds the last pair
    size_t words_into_last = 0;                       // outputs numpy consumes from there on
    double cached = 0.0;
    while (have < pairs) {
        // An attempt takes four generator outputs and is accepted with probability pi / 4.  Attempts are made in
        // bulk, a little more than the expected number; outputs drawn beyond the last attempt numpy would make are
        // simply never looked at (the generator object does not outlive the call: every call re-seeds).
        const size_t attempts = (size_t)((pairs - have) * 1.2733 + 8.0 * std::sqrt((double)(pairs - have))) + 64;
        const size_t nchunks = (attempts + CHUNK - 1) / CHUNK;
        std::vector<MT19937> snapshot(nchunks);
        std::vector<std::atomic<int64_t>> offset(nchunks + 1);
        std::vector<std::atomic<uint8_t>> released(nchunks);
        for (auto& o : offset) o.store(-1, std::memory_order_relaxed);
        for (auto& r : released) r.store(0, std::memory_order_relaxed);
        offset[0].store((int64_t)have, std::memory_order_release);
        std::atomic<size_t> made{0}, next{0};
        std::atomic<size_t> last_chunk{0}, last_words{0};
        std::atomic<uint64_t> last_cached{0};

        auto produce = [&](size_t c) {
            const size_t lo = c * CHUNK, hi = lo + CHUNK < attempts ? lo + CHUNK : attempts;
            if (c >= RING) wait_until([&] { return released[c - RING].load(std::memory_order_acquire) != 0; });
            snapshot[c] = mt;
            mt.fill(ring + (c % RING) * CHUNK * 4, 4 * (hi - lo));
            made.store(c + 1, std::memory_order_release);
        };
        struct Local { std::vector<double> x1, x2, r2; std::vector<uint32_t> att;
                       Local() : x1(CHUNK), x2(CHUNK), r2(CHUNK), att(CHUNK) {} };
        auto consume = [&](size_t c, Local& L) {
            const size_t lo = c * CHUNK, hi = lo + CHUNK < attempts ? lo + CHUNK : attempts;
            wait_until([&] { return made.load(std::memory_order_acquire) > c; });
            const uint32_t* raw = ring + (c % RING) * CHUNK * 4;
            size_t k = 0;
            for (size_t a = 0; a < hi - lo; a++) {
                const uint32_t* w = &raw[4 * a];
                const double u1 = 2.0 * to_double(w[0], w[1]) - 1.0;
                const double u2 = 2.0 * to_double(w[2], w[3]) - 1.0;
                const double s = u1 * u1 + u2 * u2;
                L.x1[k] = u1; L.x2[k] = u2; L.r2[k] = s; L.att[k] = (uint32_t)a;
                k += (size_t)(!(s >= 1.0) & (s != 0.0));                      // kept only if accepted
            }
            released[c].store(1, std::memory_order_release);
            int64_t at;
            wait_until([&] { return (at = offset[c].load(std::memory_order_acquire)) >= 0; });
            offset[c + 1].store(at + (int64_t)k, std::memory_order_release);
            // f = sqrt(-2 log(r2) / r2); the pair yields f * x2 first, then the cached f * x1
            for (size_t i = 0; i < k; i++) {
                const size_t idx = (size_t)at + i;
                if (idx >= pairs) break;
                const double f = std::sqrt(-2.0 * std::log(L.r2[i]) / L.r2[i]);
                out[2 * idx] = f * L.x2[i];
                if (2 * idx + 1 < n) out[2 * idx + 1] = f * L.x1[i];
                if (idx == pairs - 1) {            // the last pair numpy would draw
                    const double g = f * L.x1[i];
                    uint64_t bits;
                    std::memcpy(&bits, &g, 8);
                    last_cached.store(bits, std::memory_order_relaxed);
                    last_chunk.store(c, std::memory_order_relaxed);
                    last_words.store(4 * ((size_t)L.att[i] + 1), std::memory_order_relaxed);
                }
            }
        };
        auto follow = [&] {
            Local L;
            for (;;) {
                const size_t c = next.fetch_add(1, std::memory_order_relaxed);
                if (c >= nchunks) return;
                consume(c, L);
            }
        };

        if (nthreads > 1 && nchunks >= 4) {
            std::vector<std::thread> pool;
            for (int t = 1; t < nthreads; t++) pool.emplace_back(follow);
            for (size_t c = 0; c < nchunks; c++) produce(c);
            follow();
            for (auto& th : pool) th.join();
        } else {
            Local L;
            for (size_t c = 0; c < nchunks; c++) { produce(c); consume(c, L); }
        }
        const size_t total = (size_t)offset[nchunks].load(std::memory_order_acquire);
        if (total >= pairs) {
            at_last = snapshot[last_chunk.load()];
            words_into_last = last_words.load();
            const uint64_t bits = last_cached.load();
            std::memcpy(&cached, &bits, 8);
        }
        have = total;
    }
    if (state_out) {
        // replay from the start of that chunk up to the last output numpy consumes
        std::vector<uint32_t> sink(4 * CHUNK);
        if (words_into_last) at_last.fill(sink.data(), words_into_last);
        std::memcpy(state_out, at_last.key, 624 * 4);
        state_out[624] = (uint32_t)at_last.pos;
        const bool has_gauss = (n % 2) == 1;
        state_out[625] = has_gauss ? 1u : 0u;
        const double g = has_gauss ? cached : 0.0;
        std::memcpy(&state_out[626], &g, 8);
    }
}

int clamp_threads(int nthreads) {
    const int procs = usable_processors();
    if (nthreads > procs) nthreads = procs;
    if (nthreads < 1) nthreads = 1;
    return nthreads > 64 ? 64 : nthreads;
}

}  // namespace

// n values of the stream that follows np.random.seed(seed), into out; nthreads: host threads for the bulk arithmetic
// (no more than the processors this process may run on are used).
// state_out (may be null) receives what np.random.get_state() would show afterwards: 624 key words, position,
// has_gauss, and the cached variate as raw bits -- 628 words -- so that a caller can leave the process-global numpy
// generator exactly where the reference's in-process worker leaves it.
extern "C" int mppi_numpy_normals(uint32_t seed, size_t n, double* out, int32_t nthreads, uint32_t* state_out) try {
    if (!out && n) return -1;
    generate(seed, n, out, clamp_threads(nthreads), state_out, g_scratch[0]);
    return 0;
} catch (const std::exception&) {
    return -8;          // out of host memory (or no thread could be started)
}

// The streams of several workers at once: worker i draws n_each values after np.random.seed(seeds[i]) into
// out + i * n_each (the reference's pool workers, trajopt/utils.py:151-163, each re-seed their own generator).
// Streams run side by side, sharing the threads between them.
extern "C" int mppi_numpy_normals_many(const uint32_t* seeds, int32_t nstreams, size_t n_each, double* out,
                                       int32_t nthreads) try {
    if (nstreams < 0 || (nstreams && !seeds) || (!out && n_each && nstreams)) return -1;
    nthreads = clamp_threads(nthreads);
    const int lanes = nstreams < nthreads ? nstreams : nthreads;          // streams in flight
    if (lanes <= 1) {
        for (int i = 0; i < nstreams; i++) generate(seeds[i], n_each, out + (size_t)i * n_each, nthreads, nullptr, g_scratch[0]);
        return 0;
    }
    const int per = nthreads / lanes > 0 ? nthreads / lanes : 1;
    std::atomic<int> next{0};
    std::atomic<bool> failed{false};
    auto run = [&](int lane) {
        try {
            for (int i; (i = next.fetch_add(1)) < nstreams;)
                generate(seeds[i], n_each, out + (size_t)i * n_each, per, nullptr, g_scratch[lane]);
        } catch (const std::exception&) { failed.store(true); }
    };
    std::vector<std::thread> pool;
    for (int l = 1; l < lanes; l++) pool.emplace_back(run, l);
    run(0);
    for (auto& th : pool) th.join();
    return failed.load() ? -8 : 0;
} catch (const std::exception&) {
    return -8;
}
