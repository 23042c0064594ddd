// The legacy numpy normal stream on the host, bit for bit: what `np.random.seed(s)` followed by
// `np.random.normal(size=...)` calls yields (MT19937 seeded by the Knuth recurrence, 53-bit doubles from two outputs,
// polar Box-Muller with the second variate of a pair cached for the next draw).  The reference's rollout workers
// draw their perturbations from exactly this stream (trajopt/utils.py:118-139); the planner's exact-comparison noise
// mode uploads it.  numpy makes the stream one value at a time; here one thread runs the generator's recurrence ahead
// in bulk (vector instructions, state blocks written straight into a ring) and the other threads follow behind it
// with the tempering, the rejection test, the logarithms and the square roots.  Same libm, same operation order:
// same bits (tests compare against numpy itself).
#include <array>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <exception>
#include <functional>
#include <mutex>
#include <sched.h>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/mppi_hip.h"
#include "numpy_stream.h"

// products and sums must round one by one, as in numpy's generator
#pragma STDC FP_CONTRACT OFF

namespace {

typedef uint32_t u32x8 __attribute__((vector_size(32), aligned(4)));
typedef uint32_t u32x16 __attribute__((vector_size(64), aligned(4)));
constexpr uint32_t UP = 0x80000000u, LO = 0x7fffffffu, MAT = 0x9908b0dfu;
constexpr int NW = 624;                 // words of generator state = outputs per regeneration

inline uint32_t twist(uint32_t a, uint32_t b, uint32_t far) {
    const uint32_t y = (a & UP) | (b & LO);
    return far ^ (y >> 1) ^ ((0u - (y & 1u)) & MAT);
}

// The state after `prev`, W words at a time: word i needs words i and i+1 of the old state and word i+397 (mod 624),
// which is old for i < 227 and otherwise is word i-227 of the new state, made at least 227 positions earlier, so a
// group of W never reads a word of its own making.
#define MT_NEXT_BODY(VEC, W)                                                                                   \
    int i = 0;                                                                                                 \
    for (; i + W <= 227; i += W) {                                                                             \
        const VEC a = *(const VEC*)(prev + i), b = *(const VEC*)(prev + i + 1), f = *(const VEC*)(prev + i + 397); \
        const VEC y = (a & UP) | (b & LO);                                                                     \
        *(VEC*)(next + i) = f ^ (y >> 1) ^ ((0u - (y & 1u)) & MAT);                                            \
    }                                                                                                          \
    for (; i < 227; i++) next[i] = twist(prev[i], prev[i + 1], prev[i + 397]);                                 \
    for (; i + W <= 623; i += W) {                                                                             \
        const VEC a = *(const VEC*)(prev + i), b = *(const VEC*)(prev + i + 1), f = *(const VEC*)(next + i - 227); \
        const VEC y = (a & UP) | (b & LO);                                                                     \
        *(VEC*)(next + i) = f ^ (y >> 1) ^ ((0u - (y & 1u)) & MAT);                                            \
    }                                                                                                          \
    for (; i < 623; i++) next[i] = twist(prev[i], prev[i + 1], next[i - 227]);                                 \
    next[623] = twist(prev[623], next[0], next[396]);

#define MT_TEMPER_BODY(VEC, W)                                                                                 \
    size_t i = 0;                                                                                              \
    for (; i + W <= n; i += W) {                                                                               \
        VEC y = *(const VEC*)(w + i);                                                                          \
        y ^= y >> 11;                                                                                          \
        y ^= (y << 7) & 0x9d2c5680u;                                                                           \
        y ^= (y << 15) & 0xefc60000u;                                                                          \
        y ^= y >> 18;                                                                                          \
        *(VEC*)(w + i) = y;                                                                                    \
    }                                                                                                          \
    for (; i < n; i++) {                                                                                       \
        uint32_t y = w[i];                                                                                     \
        y ^= y >> 11;                                                                                          \
        y ^= (y << 7) & 0x9d2c5680u;                                                                           \
        y ^= (y << 15) & 0xefc60000u;                                                                          \
        y ^= y >> 18;                                                                                          \
        w[i] = y;                                                                                              \
    }

__attribute__((target("avx512f"))) void next_state_512(const uint32_t* prev, uint32_t* next) { MT_NEXT_BODY(u32x16, 16) }
__attribute__((target("avx2"))) void next_state_256(const uint32_t* prev, uint32_t* next) { MT_NEXT_BODY(u32x8, 8) }
void next_state_base(const uint32_t* prev, uint32_t* next) { MT_NEXT_BODY(u32x8, 8) }
__attribute__((target("avx512f"))) void temper_512(uint32_t* w, size_t n) { MT_TEMPER_BODY(u32x16, 16) }
__attribute__((target("avx2"))) void temper_256(uint32_t* w, size_t n) { MT_TEMPER_BODY(u32x8, 8) }
void temper_base(uint32_t* w, size_t n) { MT_TEMPER_BODY(u32x8, 8) }

struct Cpu {
    void (*next_state)(const uint32_t*, uint32_t*);
    void (*temper)(uint32_t*, size_t);
    Cpu() {
        if (__builtin_cpu_supports("avx512f")) { next_state = next_state_512; temper = temper_512; }
        else if (__builtin_cpu_supports("avx2")) { next_state = next_state_256; temper = temper_256; }
        else { next_state = next_state_base; temper = temper_base; }
    }
};
const Cpu g_cpu;

typedef std::array<uint32_t, NW> State;

void seed_state(uint32_t s, State& key) {           // init_genrand
    for (int i = 0; i < NW; i++) {
        key[i] = s;
        s = 1812433253u * (s ^ (s >> 30)) + (uint32_t)i + 1u;
    }
}

inline double to_double(uint32_t a, uint32_t b) {
    return ((a >> 5) * 67108864.0 + (b >> 6)) * (1.0 / 9007199254740992.0);      // a power of two: same bits as the division
}

constexpr size_t BLOCKS = 53;                       // state blocks a thread takes at a time ...
constexpr size_t CHUNK = BLOCKS * NW / 4;           // ... = 8268 attempts of four outputs each
constexpr size_t SLOT = BLOCKS * NW;                // words
constexpr size_t RING = 32;                         // chunks in flight (4 MiB: stays in the last-level cache)
static_assert(BLOCKS * NW % 4 == 0, "a chunk must hold whole attempts");

struct Local {                                      // a follower's compaction buffers
    std::vector<double> x1, x2, r2;
    std::vector<uint32_t> att;
    void reserve() { if (x1.size() < CHUNK) { x1.resize(CHUNK); x2.resize(CHUNK); r2.resize(CHUNK); att.resize(CHUNK); } }
};
struct Scratch {                        // survives between calls (a planner asks for the same amount at every step)
    std::mutex mu;
    std::vector<uint32_t> ring;
    std::vector<Local> local;
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

// threads that are joined on every way out of the scope that started them
struct Pool {
    std::vector<std::thread> threads;
    ~Pool() { for (auto& t : threads) if (t.joinable()) t.join(); }
};

// test hook: scales the surplus of attempts the first round makes (1: the normal margin; a negative value makes the
// first round fall short, so that the restart below runs)
std::atomic<double> g_margin_scale{1.0};

// One stream: n values after np.random.seed(seed) into out, state_out as described at mppi_numpy_normals.
// The calling thread runs the recurrence (sequential by nature) chunk by chunk into a ring, every state block made
// from the one before it where it lies; the other threads follow behind it: a chunk is tempered in place, its
// attempts are tested and compacted, its slot is handed back, its number of accepted pairs is published, and as soon
// as the chunks before it have published theirs the chunk knows where its values go in `out` and writes them there.
// ready (may be null) is called with the number of leading values of `out` that are final whenever that number has
// grown (from whichever thread finished the chunk, one call at a time), for a consumer that takes them away while
// the rest is still being made.
void generate(uint32_t seed, size_t n, double* out, int nthreads, uint32_t* state_out, Scratch& S,
              const std::function<void(size_t)>* ready = nullptr) {
    std::lock_guard<std::mutex> lock(S.mu);
    if (S.ring.size() < RING * SLOT) S.ring.resize(RING * SLOT);
    if (S.local.size() < (size_t)nthreads) S.local.resize(nthreads);
    for (int t = 0; t < nthreads; t++) S.local[t].reserve();      // every allocation happens here, in the caller
    uint32_t* const ring = S.ring.data();
    const size_t pairs = (n + 1) / 2;                 // accepted attempts needed (the last may give one value only)
    State at_last;                                    // generator at the start of the chunk that holds the last pair
    size_t words_into_last = 0;                       // outputs numpy consumes from there on
    double cached = 0.0;
    size_t reported = 0;                              // (under prefix_mu)
    // An attempt takes four generator outputs and is accepted with probability pi / 4.  Attempts are made in bulk, a
    // little more than the expected number; outputs drawn beyond the last attempt numpy would make are simply never
    // looked at (the generator object does not outlive the call: every call re-seeds).  Should a round end short of
    // the pairs needed (the margin makes that a 1e-40 event), the stream is made again from the seed with twice the
    // margin: every round is numpy's stream from its first word, so a round that suffices is exact.
    for (double margin = 8.0 * g_margin_scale.load();; margin = margin > 0 ? 2.0 * margin : 8.0) {
        State key;                                    // the generator: state after the last block made
        seed_state(seed, key);
        const double surplus = margin * std::sqrt((double)pairs) + (margin > 0 ? 64.0 : 0.0);
        const double want = pairs * 1.2733 + surplus;
        const size_t attempts = want > 4.0 ? (size_t)want : 4;
        const size_t nchunks = (attempts + CHUNK - 1) / CHUNK;
        std::vector<State> snapshot(nchunks);
        std::vector<std::atomic<int64_t>> offset(nchunks + 1);
        std::vector<std::atomic<uint8_t>> released(nchunks), written(nchunks);
        for (auto& o : offset) o.store(-1, std::memory_order_relaxed);
        for (auto& r : released) r.store(0, std::memory_order_relaxed);
        for (auto& w : written) w.store(0, std::memory_order_relaxed);
        offset[0].store(0, std::memory_order_release);
        std::atomic<size_t> made{0}, next{0};
        std::atomic<size_t> last_chunk{0}, last_words{0};
        std::atomic<uint64_t> last_cached{0};
        std::mutex prefix_mu;
        size_t prefix_chunk = 0;

        auto produce = [&](size_t c) {
            const size_t lo = c * CHUNK, hi = lo + CHUNK < attempts ? lo + CHUNK : attempts;
            if (c >= RING) wait_until([&] { return released[c - RING].load(std::memory_order_acquire) != 0; });
            snapshot[c] = key;
            uint32_t* slot = ring + (c % RING) * SLOT;
            const size_t blocks = (4 * (hi - lo) + NW - 1) / NW;
            const uint32_t* prev = key.data();
            for (size_t b = 0; b < blocks; b++) {
                g_cpu.next_state(prev, slot + b * NW);
                prev = slot + b * NW;
            }
            std::memcpy(key.data(), prev, NW * 4);
            made.store(c + 1, std::memory_order_release);
        };
        auto consume = [&](size_t c, Local& L) {
            const size_t lo = c * CHUNK, hi = lo + CHUNK < attempts ? lo + CHUNK : attempts;
            wait_until([&] { return made.load(std::memory_order_acquire) > c; });
            uint32_t* raw = ring + (c % RING) * SLOT;           // this thread's until it is released
            g_cpu.temper(raw, 4 * (hi - lo));
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
            written[c].store(1, std::memory_order_release);
            if (ready) {
                // leading values that are final: everything before the first chunk still being written
                std::lock_guard<std::mutex> g(prefix_mu);
                while (prefix_chunk < nchunks && written[prefix_chunk].load(std::memory_order_acquire)) prefix_chunk++;
                size_t vals = n;
                if (prefix_chunk < nchunks) {
                    const int64_t first_open = offset[prefix_chunk].load(std::memory_order_acquire);     // set: its predecessor wrote
                    vals = 2 * (size_t)first_open < n ? 2 * (size_t)first_open : n;
                } else if ((size_t)offset[nchunks].load(std::memory_order_acquire) < pairs) {
                    vals = 2 * (size_t)offset[nchunks].load(std::memory_order_acquire);                 // a short round
                }
                if (vals > reported) { reported = vals; (*ready)(vals); }
            }
        };
        auto follow = [&](int t) {
            Local& L = S.local[t];
            for (;;) {
                const size_t c = next.fetch_add(1, std::memory_order_relaxed);
                if (c >= nchunks) return;
                consume(c, L);
            }
        };

        bool piped = false;
        if (nthreads > 1 && nchunks >= 4) {
            Pool pool;
            try {
                for (int t = 1; t < nthreads; t++) pool.threads.emplace_back(follow, t);
            } catch (const std::system_error&) {
                // fewer followers than asked for: the pipeline works with any number of them, but not with none
            }
            if (!pool.threads.empty()) {
                piped = true;
                for (size_t c = 0; c < nchunks; c++) produce(c);
                follow(0);
            }
        }
        if (!piped)
            for (size_t c = 0; c < nchunks; c++) { produce(c); consume(c, S.local[0]); }
        const size_t total = (size_t)offset[nchunks].load(std::memory_order_acquire);
        if (total >= pairs) {
            at_last = snapshot[last_chunk.load()];
            words_into_last = last_words.load();
            const uint64_t bits = last_cached.load();
            std::memcpy(&cached, &bits, 8);
            break;
        }
        // (a short round: what it wrote is the stream's true beginning and is written again, identically)
    }
    if (ready && reported < n) (*ready)(n);
    if (state_out) {
        // replay from the start of that chunk up to the block of the last output numpy consumes
        const size_t regenerations = (words_into_last + NW - 1) / NW;
        State other;
        State* cur = &at_last;
        State* nxt = &other;
        for (size_t r = 0; r < regenerations; r++) {
            g_cpu.next_state(cur->data(), nxt->data());
            std::swap(cur, nxt);
        }
        std::memcpy(state_out, cur->data(), NW * 4);
        state_out[624] = regenerations ? (uint32_t)(words_into_last - NW * (regenerations - 1)) : (uint32_t)NW;
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
// Streams run side by side, sharing the threads between them.  ready (may be null): called with (stream, number of
// leading values of that stream that are final) whenever that number has grown; calls for one stream come one at a
// time, calls for different streams may overlap.  state_out (may be null; one stream only): as for mppi_numpy_normals.
int mppi_numpy_normals_streams(const uint32_t* seeds, int32_t nstreams, size_t n_each, double* out, int32_t nthreads,
                               uint32_t* state_out, const std::function<void(int, size_t)>* ready) {
    if (nstreams < 0 || (nstreams && !seeds) || (!out && n_each && nstreams)) return -1;
    if (state_out && nstreams != 1) return -1;
    nthreads = clamp_threads(nthreads);
    const int lanes = nstreams < nthreads ? nstreams : nthreads;          // streams in flight
    if (lanes <= 1) {
        for (int i = 0; i < nstreams; i++) {
            const std::function<void(size_t)> one = [&](size_t have) { (*ready)(i, have); };
            generate(seeds[i], n_each, out + (size_t)i * n_each, nthreads, state_out, g_scratch[0], ready ? &one : nullptr);
        }
        return 0;
    }
    const int per = nthreads / lanes > 0 ? nthreads / lanes : 1;
    std::atomic<int> next{0};
    std::atomic<bool> failed{false};
    auto run = [&](int lane) {
        try {
            for (int i; (i = next.fetch_add(1)) < nstreams;) {
                const std::function<void(size_t)> one = [&](size_t have) { (*ready)(i, have); };
                generate(seeds[i], n_each, out + (size_t)i * n_each, per, nullptr, g_scratch[lane], ready ? &one : nullptr);
            }
        } catch (const std::exception&) { failed.store(true); }
    };
    {
        Pool pool;
        try {
            for (int l = 1; l < lanes; l++) pool.threads.emplace_back(run, l);
        } catch (const std::system_error&) {
            // the streams not taken by a started thread are made by this one
        }
        run(0);
    }
    return failed.load() ? -8 : 0;
}

extern "C" int mppi_numpy_normals_many(const uint32_t* seeds, int32_t nstreams, size_t n_each, double* out,
                                       int32_t nthreads) try {
    return mppi_numpy_normals_streams(seeds, nstreams, n_each, out, nthreads, nullptr, nullptr);
} catch (const std::exception&) {
    return -8;
}

// test hook (see g_margin_scale)
extern "C" void mppi_numpy_normals_test_margin(double scale) { g_margin_scale.store(scale); }
