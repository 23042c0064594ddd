// Kernels and per-model launchers of the device-resident MPPI planning step (gfx950).  Included by one
// translation unit per compiled model (model_ops.hip, which instantiates everything for that model) and by
// mppi_hip.hip (the C ABI, which reaches the kernels only through the ModelOps table).
#pragma once
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <climits>
#include <cstdint>
#include <string>

#include "../../include/mppi_hip.h"
#include "env_core.hpp"
#include "models_gen.hpp"

using namespace mppi;

// ------------------------------------------------------------------------------------------
// device-side state
// ------------------------------------------------------------------------------------------
struct DevState {
    double q[MPPI_MAXV];
    double v[MPPI_MAXV];
    double target[3];
    double alpha;
    int timestep;
    int ref_len;
};

// ------------------------------------------------------------------------------------------
// Philox-4x32-10
// ------------------------------------------------------------------------------------------
MPPI_HD void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
MPPI_HD void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

// two uniforms -> two N(0,1) (Box-Muller) on the hardware transcendental units: log2, square root, and
// sine / cosine of an angle given in revolutions (about 1e-6 absolute, far below what a sampler needs)
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& n0, float& n1) {
    const float u1 = (float)((a >> 8) + 1u) * (1.0f / 16777216.0f);   // (0, 1]
    const float u2 = (float)(b >> 8) * (1.0f / 16777216.0f);          // [0, 1)
    const float r = __builtin_amdgcn_sqrtf(-1.38629436111989061883f * __builtin_amdgcn_logf(u1));   // sqrt(-2 ln u1)
    n0 = r * __builtin_amdgcn_cosf(u2);
    n1 = r * __builtin_amdgcn_sinf(u2);
}

// ------------------------------------------------------------------------------------------
// noise kernels: the recursion along t of one (sample, action dimension) in registers
// eps[t] = b0*sigma*z[t] + b1*eps[t-1] + b2*eps[t-2] for t >= 2 (rows 0, 1 unfiltered); separate
// roundings for every product and sum so the fp64 path reproduces the host arithmetic bit for bit
// ------------------------------------------------------------------------------------------
struct FilterCoefs {
    double sigma[MPPI_MAXV];
    double b0, b1, b2;
};

// A block takes 64 samples.  The host's array is sample-major ([K][H][NV] fp64, a sample's draws contiguous), the
// rollouts want step-major, sample-minor: the block reads TC steps of its 64 samples with consecutive threads on
// consecutive addresses, turns them through LDS, and lane k of wave w runs the recursion of (sample k, dimensions w,
// w + 4, ...) along those steps, 64 consecutive samples per store.  (One lane per sample reading its own row took
// 60 us for 4096 x 64 x 7 draws -- 64 waves, every load a cache line per lane; this shape takes a quarter of that.)
template <class T, int NV>
__global__ void __launch_bounds__(256) noise_upload_kernel(const double* __restrict__ raw, int K, int H, FilterCoefs fc,
                                                           int filter, T* __restrict__ eps) {
    // numpy rounds every product and every sum of the recursion by itself; the compiler must not fuse them (it did:
    // with the reference's usual taps (0.25, 0.8, 0) a fused multiply-add gives the same bits -- 0.25 e is exact, 0 e2
    // vanishes -- so only other coefficients showed the last-bit difference)
#pragma clang fp contract(off)
    constexpr int TC = 112 / NV < 16 ? 112 / NV : 16;       // steps per pass through LDS
    constexpr int ROWW = TC * NV + 1;                        // (+1: lanes of a wave read a column, one bank apart)
    constexpr int JW = (NV + 3) / 4;                         // dimensions per wave
    __shared__ double tile[64 * ROWW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k0 = blockIdx.x * 64, k = k0 + lane;
    raw += (size_t)blockIdx.y * K * H * NV;           // planner blockIdx.y of a batched handle
    eps += (size_t)blockIdx.y * H * NV * K;
    double e1[JW], e2[JW], sg[JW];
    // (sigma of this wave's dimensions once, into registers: indexed by the wave inside the loop it is a load from the
    // argument block at every step, and the wait for it is a wait for every store and prefetch in flight as well)
#pragma unroll
    for (int jj = 0; jj < JW; jj++) {
        e1[jj] = e2[jj] = 0.0;
        sg[jj] = fc.sigma[wave + 4 * jj < NV ? wave + 4 * jj : NV - 1];
    }
    // wave w fetches the rows of samples w, w + 4, ...; the loads of the next pass are in flight while this one is
    // worked through
    constexpr int PER = (TC * NV + 63) / 64;                 // values of one row per lane
    double got[16][PER];
    auto fetch = [&](int tc) {
        // (every load unconditional, from an address that exists: a load behind a branch is waited for before the next
        // one is issued.  What lies beyond the sample count or the horizon is fetched from the last sample / step and
        // never used.)
        if (tc >= H) tc = 0;
        const int span = (H - tc < TC ? H - tc : TC) * NV;
#pragma unroll
        for (int r = 0; r < 16; r++)
#pragma unroll
            for (int c = 0; c < PER; c++) {
                const int smp = wave + 4 * r, off = lane + 64 * c;
                const int sk = k0 + smp < K ? k0 + smp : K - 1, so = off < span ? off : span - 1;
                got[r][c] = raw[((size_t)sk * H + tc) * NV + so];
            }
    };
    fetch(0);
    for (int tc = 0; tc < H; tc += TC) {
        const int nt = H - tc < TC ? H - tc : TC;
#pragma unroll
        for (int r = 0; r < 16; r++)
#pragma unroll
            for (int c = 0; c < PER; c++) {
                const int smp = wave + 4 * r, off = lane + 64 * c;
                if (off < TC * NV) tile[smp * ROWW + off] = got[r][c];
            }
        __syncthreads();
        fetch(tc + TC);
        if (k < K) {
            for (int u = 0; u < nt; u++) {
                const int t = tc + u;
#pragma unroll
                for (int jj = 0; jj < JW; jj++) {
                    const int j = wave + 4 * jj;
                    if (j < NV) {
                        double e = tile[lane * ROWW + u * NV + j];
                        if (filter) {
                            // (plain operators: the rounding intrinsics are inline functions compiled with contraction
                            // allowed, which the pragma above cannot reach)
                            e = e * sg[jj];
                            if (t >= 2) e = (fc.b0 * e + fc.b1 * e1[jj]) + fc.b2 * e2[jj];
                        }
                        e2[jj] = e1[jj];
                        e1[jj] = e;
                        eps[((size_t)t * NV + j) * K + k] = T(e);
                    }
                }
            }
        }
        __syncthreads();
    }
}

// The filtered perturbation of one (sample, action dimension) along the horizon, from the counter-based stream:
// counter = (global sample index, dimension, block c, planner), key = the stream key; block c holds the draws of steps
// 4c .. 4c+3.  emit(t, e) receives every step's value in fp64.  The noise kernels store it; the update that draws its
// noise again instead of reading it back calls the same function, so both see the same bits (every product and sum
// below is spelled out as the instruction it is to be, so that the two kernels cannot be contracted differently).
// one block of four steps (see philox_sequence)
template <class F>
__device__ __forceinline__ void philox_sequence_block(int c, uint32_t kglobal, int j, int pl, uint32_t key_lo, uint32_t key_hi,
                                                      int H, double sigma, const FilterCoefs& fc, double& e1, double& e2, F&& emit) {
    uint32_t ctr[4] = {kglobal, (uint32_t)j, (uint32_t)c, (uint32_t)pl};
    philox4x32_10(ctr, key_lo, key_hi);
    float z[4];
    box_muller(ctr[0], ctr[1], z[0], z[1]);
    box_muller(ctr[2], ctr[3], z[2], z[3]);
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int t = 4 * c + u;
        if (t < H) {
            double e = __dmul_rn((double)z[u], sigma);
            if (t >= 2) e = fma(fc.b0, e, fma(fc.b1, e1, __dmul_rn(fc.b2, e2)));
            e2 = e1;
            e1 = e;
            emit(t, e);
        }
    }
}
// HT = 0: any horizon (a loop); HT > 0: horizons up to HT, fully unrolled, so that t is a constant in every call of emit
template <int HT, class F>
__device__ __forceinline__ void philox_sequence(uint32_t kglobal, int j, int pl, uint32_t key_lo, uint32_t key_hi, int H,
                                                const FilterCoefs& fc, F&& emit) {
    const double sigma = fc.sigma[j];
    double e1 = 0.0, e2 = 0.0;
    if constexpr (HT == 0) {
        for (int c = 0; 4 * c < H; c++) philox_sequence_block(c, kglobal, j, pl, key_lo, key_hi, H, sigma, fc, e1, e2, emit);
    } else {
#pragma unroll
        for (int c = 0; c < (HT + 3) / 4; c++)
            if (4 * c < H) philox_sequence_block(c, kglobal, j, pl, key_lo, key_hi, H, sigma, fc, e1, e2, emit);
    }
}

// one lane per (sample k, action dimension j).  With seed_from_state the stream key is seed + the planner's current
// timestep (the reference re-seeds every planning step with seed + t, trajopt/algos/mppi.py:130-132).  The key used
// is left in key_out[2 * planner ..] for kernels that draw the same noise again.
template <class T, int NV>
__device__ __forceinline__ void noise_philox_block(int block, int pl, int K, int H, int k_offset, uint32_t seed_lo, uint32_t seed_hi,
                                                   int step, int seed_from_state, const FilterCoefs& fc, T* __restrict__ eps,
                                                   uint32_t* __restrict__ key_out) {
    const int idx = block * 256 + threadIdx.x;
    if (idx >= K * NV) return;
    const int j = idx / K, k = idx - j * K;
    eps += (size_t)pl * H * NV * K;                   // planner of a batched handle: its own counter plane
    if (seed_from_state) {
        const uint64_t s = (((uint64_t)seed_hi << 32) | seed_lo) + (uint64_t)(int64_t)step;
        seed_lo = (uint32_t)s;
        seed_hi = (uint32_t)(s >> 32);
    }
    if (idx == 0) { key_out[2 * pl] = seed_lo; key_out[2 * pl + 1] = seed_hi; }
    philox_sequence<0>((uint32_t)(k_offset + k), j, pl, seed_lo, seed_hi, H, fc,
                       [&](int t, double e) { eps[((size_t)t * NV + j) * K + k] = T(e); });
}
template <class T, int NV>
__global__ void __launch_bounds__(256) noise_philox_kernel(int K, int H, int k_offset, uint32_t seed_lo, uint32_t seed_hi,
                                                           const DevState* __restrict__ st, int seed_from_state,
                                                           FilterCoefs fc, T* __restrict__ eps, uint32_t* __restrict__ key_out) {
    noise_philox_block<T, NV>(blockIdx.x, blockIdx.y, K, H, k_offset, seed_lo, seed_hi, st[blockIdx.y].timestep, seed_from_state, fc, eps, key_out);
}

// ------------------------------------------------------------------------------------------
// rollout
// ------------------------------------------------------------------------------------------
// The library is built with -ffinite-math-only: the compiler may assume that no floating-point value is NaN or
// infinity and fold a test for them away, a test on the bit pattern included (it recognises the idiom).  The
// bits are therefore passed through an empty asm statement as an integer, after which nothing is known about
// them.
__device__ __forceinline__ bool nonfinite(float x) {
    unsigned int u = __float_as_uint(x);
    asm volatile("" : "+v"(u));
    return (u & 0x7f800000u) == 0x7f800000u;
}
__device__ __forceinline__ bool nonfinite(double x) {
    unsigned int hi = (unsigned int)((unsigned long long)__double_as_longlong(x) >> 32);
    asm volatile("" : "+v"(hi));
    return (hi & 0x7ff00000u) == 0x7ff00000u;
}

// LEAN: the planning configuration (dense reward, noise-relative samples, nothing dumped, no reference, no
// critic) with those choices fixed at compile time: a lone wave pays about ten vector instructions' time for
// every taken branch, and the general kernel spends eight per env step on options that are off.
template <class T, class MD, bool LEAN>
__global__ void __launch_bounds__(256)       // launched with 64 or 256 threads (launch_rollout)
rollout_kernel(const DevState* __restrict__ st, const double* __restrict__ base, const T* __restrict__ eps,
               const double* __restrict__ ref, int K, int H, int P, int reward_type_, double gamma, int explicit_act_,
               T* __restrict__ R, T* __restrict__ obs_, T* __restrict__ rew_, int* __restrict__ status,
               T* __restrict__ fobs_, int* __restrict__ next_step, int sweep_cap) {
    constexpr int NV = MD::NV, NQ = MD::NQ;
    const int reward_type = LEAN ? int(REWARD_DENSE) : reward_type_;
    const int explicit_act = LEAN ? 0 : explicit_act_;
    T* __restrict__ obs = LEAN ? nullptr : obs_;
    T* __restrict__ rew = LEAN ? nullptr : rew_;
    T* __restrict__ fobs = LEAN ? nullptr : fobs_;
    constexpr int OD = NQ + NV + 6;
    // The grid is a flat list of waves, planner after planner (64 consecutive samples of one planner each), so that
    // blocks of four waves are full whatever K is.
    const int wpp = (K + 63) >> 6;                                   // waves per planner
    // (the wave's index within the block through readfirstlane: the compiler then knows that the planner, and every
    // address formed from it, is the same in all lanes and keeps them in scalar registers and scalar loads)
    const int wv = blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int planner = wv / wpp;
    if (planner >= P) return;
    const int k = (wv - planner * wpp) * 64 + (threadIdx.x & 63);
    if (k >= K) return;
    // index of the planning step after this one, for a noise generator that runs while the state advances
    if (k == 0) next_step[planner] = st[planner].timestep + 1;
    {   // one planner of a batched handle
        const size_t pl = planner;
        st += pl; base += pl * H * NV; eps += pl * H * NV * K; R += pl * K;
        if (obs) obs += pl * K * H * OD;
        if (rew) rew += pl * K * H;
        if (fobs) fobs += pl * K * OD;
    }
    T q[NQ], v[NV];
    static_for<NQ>([&](auto Ic) { q[Ic] = T(st->q[Ic]); });
    static_for<NV>([&](auto Ic) { v[Ic] = T(st->v[Ic]); });
    RewardParams<T> rp;
    rp.type = reward_type;
    rp.alpha = T(st->alpha);
    rp.target[0] = T(st->target[0]); rp.target[1] = T(st->target[1]); rp.target[2] = T(st->target[2]);
    const int t0 = st->timestep, ref_len = LEAN ? 0 : st->ref_len;
    // The reference indexes its trajectory with the time step after every env step (trajopt/envs/reacher_env.py:61,75)
    // and raises IndexError past the end; here the index is held at the last row and the status word says so.
    if constexpr (!LEAN) {
        if (k == 0 && ref_len > 0 && t0 + H >= ref_len && (reward_type == REWARD_TRACKING || reward_type == REWARD_COOLING))
            atomicOr(status, MPPI_STATUS_REF_END);
    }
    const T gm = T(gamma);
    T ret = T(0), g = T(1);
    typename Step<T, MD>::Active act_rows;
    act_rows.reset();
    act_rows.cap = sweep_cap;
    int most_sweeps = 0;
    int first_capped_step = -1;       // (general kernel variant only) where a capped solve first happened in this wave
    // the next step's noise and base action are fetched one step ahead: they do not depend on the
    // state, and a lone wave has nobody to hide a load's latency behind
    T a[NV];
    static_for<NV>([&](auto Jc) { a[Jc] = explicit_act ? eps[(size_t)Jc * K + k] : T(base[Jc]) + eps[(size_t)Jc * K + k]; });
    for (int t = 0; t < H; t++) {
        // issued here, first touched after the step: a use at the top of the next iteration would make the
        // wave wait for the scalar loads right behind their issue (the wait for the pre-loop loads sits there)
        const int tn = (t + 1 < H) ? t + 1 : t;
        T e_nx[NV];
        double b_nx[NV];
        static_for<NV>([&](auto Jc) {
            constexpr int j = Jc;
            e_nx[j] = eps[((size_t)tn * NV + j) * K + k];
            b_nx[j] = base[tn * NV + j];
        });
        T rr[3] = {T(0), T(0), T(0)};
        if (ref_len > 0) {
            int idx = t0 + t + 1;
            idx = idx < ref_len ? idx : ref_len - 1;
            rr[0] = T(ref[3 * idx]); rr[1] = T(ref[3 * idx + 1]); rr[2] = T(ref[3 * idx + 2]);
        }
        Vec3<T> hand;
        const T r = Env<T, MD>::step(q, v, a, rp, rr, hand, act_rows, most_sweeps);
        if constexpr (!LEAN) {
            if (first_capped_step < 0 && most_sweeps > sweep_cap) first_capped_step = t;
        }
        // with a terminal-value critic the last reward is replaced by the critic's value (added by critic_kernel)
        if (!(fobs && t == H - 1)) ret += g * r;
        g *= gm;
        static_for<NV>([&](auto Jc) { a[Jc] = explicit_act ? e_nx[Jc] : T(b_nx[Jc]) + e_nx[Jc]; });
        if (fobs && t == H - 1) {
            T* o = fobs + (size_t)k * OD;
            static_for<NQ>([&](auto Ic) { o[Ic] = q[Ic]; });
            static_for<NV>([&](auto Ic) { o[NQ + Ic] = v[Ic]; });
            o[NQ + NV + 0] = hand.v[0]; o[NQ + NV + 1] = hand.v[1]; o[NQ + NV + 2] = hand.v[2];
            o[NQ + NV + 3] = rp.target[0]; o[NQ + NV + 4] = rp.target[1]; o[NQ + NV + 5] = rp.target[2];
        }
        if (rew) rew[(size_t)k * H + t] = r;
        if (obs) {
            T* o = obs + ((size_t)k * H + t) * OD;
            static_for<NQ>([&](auto Ic) { o[Ic] = q[Ic]; });
            static_for<NV>([&](auto Ic) { o[NQ + Ic] = v[Ic]; });
            o[NQ + NV + 0] = hand.v[0]; o[NQ + NV + 1] = hand.v[1]; o[NQ + NV + 2] = hand.v[2];
            o[NQ + NV + 3] = rp.target[0]; o[NQ + NV + 4] = rp.target[1]; o[NQ + NV + 5] = rp.target[2];
        }
    }
    R[k] = ret;
    if (nonfinite(ret)) atomicOr(status, MPPI_STATUS_NONFINITE);
    if (most_sweeps > sweep_cap) {
        const int before = atomicOr(status, MPPI_STATUS_SWEEP_CAP);
        if constexpr (!LEAN) {
            // the first lane to report leaves where it happened: [1] first sample of the wave, [2] env step, [3] planner
            if (!(before & MPPI_STATUS_SWEEP_CAP) && atomicCAS(status + 1, 0, (k & ~63) + 1) == 0) {
                status[2] = first_capped_step;
                status[3] = planner;
            }
        }
    }
}

// one fp64 env.step of the real environment; io: in [q(nq) v(nv) target(3) act(nv) alpha refrow(3)],
// out [q(nq) v(nv) obs(nq+nv+6) reward]
template <class MD>
__global__ void __launch_bounds__(64) env_step_kernel(const double* __restrict__ in, int reward_type, int has_ref, int forward_only,
                                double* __restrict__ out) {
    constexpr int NV = MD::NV, NQ = MD::NQ, S = NQ + NV;
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double q[NQ], v[NV], a[NV];
    static_for<NQ>([&](auto Ic) { q[Ic] = in[Ic]; });
    static_for<NV>([&](auto Ic) { v[Ic] = in[NQ + Ic]; a[Ic] = in[S + 3 + Ic]; });
    RewardParams<double> rp;
    rp.type = reward_type;
    rp.target[0] = in[S]; rp.target[1] = in[S + 1]; rp.target[2] = in[S + 2];
    rp.alpha = in[S + NV + 3];
    const double* rr = in + S + NV + 4;
    Vec3<double> hand;
    double r = 0.0;
    if (forward_only) {
        // sim.forward(): kinematics only; the substep leaves q, v untouched copies
        double q2[NQ], v2[NV], z[NV];
        static_for<NQ>([&](auto Ic) { q2[Ic] = q[Ic]; });
        static_for<NV>([&](auto Ic) { v2[Ic] = v[Ic]; z[Ic] = 0.0; });
        typename Step<double, MD>::Out o;
        int sw;
        typename Step<double, MD>::Active am;
        am.reset();
        Step<double, MD>::substep(q2, v2, z, true, o, am, sw);
        hand = o.hand;
    } else {
        int ms = 0;
        typename Step<double, MD>::Active am;
        am.reset();
        r = Env<double, MD>::step(q, v, a, rp, has_ref ? rr : nullptr, hand, am, ms);
    }
    static_for<NQ>([&](auto Ic) { out[Ic] = q[Ic]; out[S + Ic] = q[Ic]; });
    static_for<NV>([&](auto Ic) { out[NQ + Ic] = v[Ic]; out[S + NQ + Ic] = v[Ic]; });
    out[2 * S + 0] = hand.v[0]; out[2 * S + 1] = hand.v[1]; out[2 * S + 2] = hand.v[2];
    out[2 * S + 3] = rp.target[0]; out[2 * S + 4] = rp.target[1]; out[2 * S + 5] = rp.target[2];
    out[2 * S + 6] = r;
}

// Device-resident `advance_time` (trajopt/algos/mppi.py:67-89): lane 0 steps the real environment in fp64 with
// act_sequence[0] from the planner's own state, appends (q', v', hand, target, reward, action) to the log ring and
// stores the new state; then the whole block shifts the horizon and fills the tail.
// BT: threads of the block (H * NV <= BT * 16)
template <class MD, int BT = 256>
__device__ __forceinline__ void advance_block(int pl, DevState* __restrict__ st, double* __restrict__ base,
                                              const double* __restrict__ ref, int reward_type, int H, int tail_mode,
                                              const double* __restrict__ mean, double* __restrict__ log, int log_cap,
                                              int* __restrict__ log_count, double* __restrict__ last_row,
                                              int* __restrict__ status, double* __restrict__ mirror) {
    constexpr int NV = MD::NV, NQ = MD::NQ;
    constexpr int ROW = NQ + 2 * NV + 7;
    constexpr int PER = 16;                       // H * NV <= BT * PER
    const int tid = threadIdx.x;
    double keep[PER];
    const int HM = H * NV;
    int steps_logged = 0;                                        // (thread 0 only)
    st += pl; base += (size_t)pl * HM;                           // one block per planner
    log += (size_t)pl * log_cap * ROW; log_count += pl;
#pragma unroll
    for (int u = 0; u < PER; u++) {
        const int i = tid + BT * u;
        double v = 0.0;
        if (i < HM) {
            const int t = i / NV, j = i - t * NV;
            if (t < H - 1) v = base[(t + 1) * NV + j];
            else v = (tail_mode == MPPI_TAIL_REPEAT) ? base[(H - 1) * NV + j] : mean[j];
        }
        keep[u] = v;
    }
    if (tid == 0) {
        double q[NQ], v[NV], a[NV];
        static_for<NQ>([&](auto Ic) { q[Ic] = st->q[Ic]; });
        static_for<NV>([&](auto Ic) { v[Ic] = st->v[Ic]; a[Ic] = base[Ic]; });
        RewardParams<double> rp;
        rp.type = reward_type;
        rp.alpha = st->alpha;
        rp.target[0] = st->target[0]; rp.target[1] = st->target[1]; rp.target[2] = st->target[2];
        double rr[3] = {0.0, 0.0, 0.0};
        const int t0 = st->timestep, ref_len = st->ref_len;
        if (ref_len > 0) {
            int idx = t0 + 1;
            if (idx >= ref_len && (reward_type == REWARD_TRACKING || reward_type == REWARD_COOLING))
                atomicOr(status, MPPI_STATUS_REF_END);      // IndexError in the reference (trajopt/envs/reacher_env.py:61,75)
            idx = idx < ref_len ? idx : ref_len - 1;
            rr[0] = ref[3 * idx]; rr[1] = ref[3 * idx + 1]; rr[2] = ref[3 * idx + 2];
        }
        Vec3<double> hand;
        int ms = 0;
        typename Step<double, MD>::Active am;
        am.reset();
        const double r = Env<double, MD>::step(q, v, a, rp, rr, hand, am, ms);
        const int n = *log_count;
        double* row = log + (size_t)(n % log_cap) * ROW;
        double* last = last_row + (size_t)pl * ROW;         // the newest row of every planner, side by side
        static_for<NQ>([&](auto Ic) { row[Ic] = q[Ic]; last[Ic] = q[Ic]; st->q[Ic] = q[Ic]; });
        static_for<NV>([&](auto Ic) {
            row[NQ + Ic] = v[Ic]; row[NQ + NV + 7 + Ic] = a[Ic];
            last[NQ + Ic] = v[Ic]; last[NQ + NV + 7 + Ic] = a[Ic];
            st->v[Ic] = v[Ic];
        });
        row[NQ + NV + 0] = hand.v[0]; row[NQ + NV + 1] = hand.v[1]; row[NQ + NV + 2] = hand.v[2];
        row[NQ + NV + 3] = rp.target[0]; row[NQ + NV + 4] = rp.target[1]; row[NQ + NV + 5] = rp.target[2];
        row[NQ + NV + 6] = r;
        static_for<7>([&](auto Ic) { last[NQ + NV + Ic] = row[NQ + NV + Ic]; });
        st->timestep = t0 + 1;
        *log_count = n + 1;
        steps_logged = n + 1;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < PER; u++) {
        const int i = tid + BT * u;
        if (i < HM) base[i] = keep[u];
    }
    // The same outcome -- newest log row, shifted act_sequence, then the number of steps logged -- written to
    // page-locked host memory for a host that reads every step back (mppi_read_step): it sees the count change as
    // soon as the data are there, without a copy or a wait on the stream.  [ROW | H * NV | count]
    if (mirror) {
        if (tid < ROW) mirror[tid] = last_row[(size_t)pl * ROW + tid];       // (written by thread 0 before the barrier)
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const int i = tid + BT * u;
            if (i < HM) mirror[ROW + i] = keep[u];
        }
        __threadfence_system();
        __syncthreads();
        if (tid == 0)
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(mirror + ROW + HM), (unsigned long long)steps_logged,
                               __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
template <class MD, int BT>
__global__ void __launch_bounds__(BT) advance_kernel(DevState* __restrict__ st, double* __restrict__ base,
                                                     const double* __restrict__ ref, int reward_type, int H, int tail_mode,
                                                     const double* __restrict__ mean, double* __restrict__ log, int log_cap,
                                                     int* __restrict__ log_count, double* __restrict__ last_row,
                                                     int* __restrict__ status, double* __restrict__ mirror) {
    advance_block<MD, BT>(blockIdx.x, st, base, ref, reward_type, H, tail_mode, mean, log, log_cap, log_count, last_row, status, mirror);
}
// advance_kernel and, in the blocks behind the first of every planner, the Philox noise of the NEXT planning step
// (stream key seed + next_step[planner], published by the rollouts of the current step): the noise does not depend
// on the new state, so it is drawn on otherwise idle CUs while one lane steps the real environment.
template <class T, class MD>
__global__ void __launch_bounds__(256) advance_noise_kernel(DevState* __restrict__ st, double* __restrict__ base,
                                                            const double* __restrict__ ref, int reward_type, int H, int tail_mode,
                                                            const double* __restrict__ mean, double* __restrict__ log, int log_cap,
                                                            int* __restrict__ log_count, double* __restrict__ last_row,
                                                            int* __restrict__ status, int K, int k_offset, uint32_t seed_lo,
                                                            uint32_t seed_hi, const int* __restrict__ next_step, FilterCoefs fc,
                                                            T* __restrict__ eps, uint32_t* __restrict__ key_out,
                                                            double* __restrict__ mirror) {
    if (blockIdx.x == 0) advance_block<MD>(blockIdx.y, st, base, ref, reward_type, H, tail_mode, mean, log, log_cap, log_count, last_row, status, mirror);
    else noise_philox_block<T, MD::NV>(blockIdx.x - 1, blockIdx.y, K, H, k_offset, seed_lo, seed_hi, next_step[blockIdx.y], 1, fc, eps, key_out);
}

// ------------------------------------------------------------------------------------------
// terminal-value critic: tanh MLP in -> 128 -> 128 -> 1 on the final observation of every sample
// (trajopt/models/critic_nets.py:47-60; input slices trajopt/algos/mppi.py:138-154), fp32 like the
// reference's torch module.  R[k] += gamma^(H-1) * V(x_k): the rollout left the last reward out
// (trajopt/algos/mppi.py:97-101).
//
// The two hidden layers are matrix products and run on the matrix cores (v_mfma_f32_32x32x2_f32: fp32 operands,
// fp32 accumulation).  A block of four waves serves 32 samples.  Everything is computed transposed,
// hidden[unit][sample] = W[unit][k] x[k][sample], so that a layer's output tile -- 16 registers per lane: column =
// sample lane % 32, rows = units 8 (v / 4) + 4 (lane / 32) + v % 4 -- is, register by register, already the
// B operand (k = that unit, column = sample) of the next layer's products: no value changes lane between the
// layers.  The weights are stored in the order the A operand wants them (pack_critic_weights), one contiguous
// 256-byte read per product.  Every wave computes all of layer 1 (36 products) and one 32-unit tile of layer 2
// (64 products); the last layer's dot product is reduced over the block in a fixed order.
// ------------------------------------------------------------------------------------------
constexpr int CRITIC_INNER = 128;
constexpr int CRITIC_IN = 17;          // widest input (qpos, qvel, goal); narrower inputs are zero padded
constexpr int CRITIC_K1 = 18;          // layer-1 contraction length: CRITIC_IN padded to a whole number of k-pairs
constexpr int CRITIC_TILES = CRITIC_INNER / 32;
// packed block: [W1p (4 tiles x 9 k-pairs x 64 lanes) | b1 (128) | W2p (4 x 4 tiles x 16 registers x 64 lanes) | b2 (128) | W3 (128) | b3]
constexpr size_t CRITIC_W1P = CRITIC_TILES * (CRITIC_K1 / 2) * 64;
constexpr size_t CRITIC_W2P = (size_t)CRITIC_TILES * CRITIC_TILES * 16 * 64;
constexpr size_t CRITIC_W_FLOATS = CRITIC_W1P + CRITIC_INNER + CRITIC_W2P + CRITIC_INNER + CRITIC_INNER + 1;

// unit (row of the 32-row output tile) that register v of a lane in half-wave `half` holds
MPPI_HD int critic_tile_row(int v, int half) { return 8 * (v / 4) + 4 * half + (v % 4); }

// torch's row-major weights (W1 [128][dim], W2 [128][128], W3 [128]) -> the packed block
inline void pack_critic_weights(int dim, const float* W1, const float* b1, const float* W2, const float* b2, const float* W3,
                                float b3, float* out) {
    float* o = out;
    for (int t = 0; t < CRITIC_TILES; t++)
        for (int kk = 0; kk < CRITIC_K1 / 2; kk++)
            for (int lane = 0; lane < 64; lane++) {
                const int unit = 32 * t + lane % 32, k = 2 * kk + lane / 32;
                *o++ = k < dim ? W1[unit * dim + k] : 0.f;
            }
    for (int j = 0; j < CRITIC_INNER; j++) *o++ = b1[j];
    for (int t2 = 0; t2 < CRITIC_TILES; t2++)
        for (int t = 0; t < CRITIC_TILES; t++)
            for (int v = 0; v < 16; v++)
                for (int lane = 0; lane < 64; lane++)
                    *o++ = W2[(32 * t2 + lane % 32) * CRITIC_INNER + 32 * t + critic_tile_row(v, lane / 32)];
    for (int j = 0; j < CRITIC_INNER; j++) *o++ = b2[j];
    for (int j = 0; j < CRITIC_INNER; j++) *o++ = W3[j];
    *o++ = b3;
}

typedef float critic_acc_t __attribute__((ext_vector_type(16)));

// tanh(x) = 1 - 2 / (1 + e^(2x)) on the hardware exponential and reciprocal: absolute error about 2e-7 (one fp32
// rounding of a value near 1), a sixth of the instructions of the library routine; saturates correctly for large |x|
__device__ __forceinline__ float critic_tanh(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float e = __builtin_amdgcn_exp2f(x * 2.885390081777927f);      // e^(2x) = 2^(2x log2 e)
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + e);
#else
    return tanhf(x);
#endif
}

template <class T, int NV>
__global__ void __launch_bounds__(256) critic_kernel(const T* __restrict__ fobs, int K, int dim, const float* __restrict__ w,
                                                     double g_last, T* __restrict__ R) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int OD = 2 * NV + 6;
    __shared__ float s_part[CRITIC_TILES][32];
    const float* __restrict__ W1p = w;
    const float* __restrict__ b1 = W1p + CRITIC_W1P;
    const float* __restrict__ W2p = b1 + CRITIC_INNER;
    const float* __restrict__ b2 = W2p + CRITIC_W2P;
    const float* __restrict__ W3 = b2 + CRITIC_INNER;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 31, half = lane >> 5;
    const int k = blockIdx.x * 32 + col;                  // this lane's sample (lanes col and col + 32 share it)
    fobs += (size_t)blockIdx.y * K * OD;
    R += (size_t)blockIdx.y * K;
    float x[CRITIC_K1];
#pragma unroll
    for (int i = 0; i < CRITIC_K1; i++) x[i] = 0.f;
    if (k < K) {
        const T* o = fobs + (size_t)k * OD;
        if (dim == 14 || dim == 15 || dim == 17) {
#pragma unroll
            for (int i = 0; i < 14; i++) x[i] = (float)o[i];
            if (dim == 15) {
                const float dx = (float)o[OD - 6] - (float)o[OD - 3], dy = (float)o[OD - 5] - (float)o[OD - 2], dz = (float)o[OD - 4] - (float)o[OD - 1];
                x[14] = sqrtf(dx * dx + dy * dy + dz * dz);
            } else if (dim == 17) {
                x[14] = (float)o[OD - 3]; x[15] = (float)o[OD - 2]; x[16] = (float)o[OD - 1];
            }
        } else if (dim == 3) {
            x[0] = (float)o[OD - 6]; x[1] = (float)o[OD - 5]; x[2] = (float)o[OD - 4];
        } else {  // 6
#pragma unroll
            for (int i = 0; i < 6; i++) x[i] = (float)o[OD - 6 + i];
        }
    }
    // layer 1, all four tiles: B operand of k-pair kk is x[2 kk + half] of the lane's sample
    critic_acc_t h1[CRITIC_TILES];
#pragma unroll
    for (int t = 0; t < CRITIC_TILES; t++) {
        critic_acc_t acc;
#pragma unroll
        for (int v = 0; v < 16; v++) acc[v] = 0.f;
#pragma unroll
        for (int kk = 0; kk < CRITIC_K1 / 2; kk++) {
            const float a = W1p[(size_t)(t * (CRITIC_K1 / 2) + kk) * 64 + lane];
            const float b = half ? x[2 * kk + 1] : x[2 * kk];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
#pragma unroll
        for (int v = 0; v < 16; v++) h1[t][v] = critic_tanh(acc[v] + b1[32 * t + critic_tile_row(v, half)]);
    }
    // layer 2, tile `wave`: register v of layer-1 tile t is the B operand whose k is the unit that register holds
    critic_acc_t acc;
#pragma unroll
    for (int v = 0; v < 16; v++) acc[v] = 0.f;
#pragma unroll
    for (int t = 0; t < CRITIC_TILES; t++) {
#pragma unroll
        for (int v = 0; v < 16; v++) {
            const float a = W2p[((size_t)(wave * CRITIC_TILES + t) * 16 + v) * 64 + lane];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, h1[t][v], acc, 0, 0, 0);
        }
    }
    // layer 3: this wave's 32 units, 16 of them in each half-wave
    float part = 0.f;
#pragma unroll
    for (int v = 0; v < 16; v++) {
        const int unit = 32 * wave + critic_tile_row(v, half);
        part = fmaf(W3[unit], critic_tanh(acc[v] + b2[unit]), part);
    }
    part += __shfl_xor(part, 32, 64);
    if (half == 0) s_part[wave][col] = part;
    __syncthreads();
    if (wave == 0 && half == 0 && k < K) {
        const float out = W3[CRITIC_INNER] + (((s_part[0][col] + s_part[1][col]) + s_part[2][col]) + s_part[3][col]);
        R[k] = R[k] + T(g_last) * T(out);
    }
#endif
}

// ------------------------------------------------------------------------------------------
// update kernels (fp64 accumulation, fixed reduction order -> bitwise reproducible)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void argmax_combine(double& v, int& i, double ov, int oi) {
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
}
__device__ __forceinline__ double shfl_xor_d(double x, int m) { return __shfl_xor(x, m, 64); }
// A rollout that diverged (a start state deep inside the table plane can do that to an explicit integrator) returns
// NaN or infinity and sets MPPI_STATUS_NONFINITE.  Such a sample takes no part in the update: it cannot be the maximum
// and its weight is zero, where exp(kappa (NaN - Rmax)) would turn the whole action sequence into NaN.  (The
// reference's simulator resets itself when an acceleration stops being finite, after rewards that have long weighed
// the sample out.)
template <class T> __device__ __forceinline__ double return_for_max(T r) { return nonfinite(r) ? -DBL_MAX : (double)r; }
template <class T> __device__ __forceinline__ double sample_weight(double kappa, T r, double rmax) {
    const double w = exp(kappa * ((double)r - rmax));
    return nonfinite(r) ? 0.0 : w;
}

// Update in one launch (write_base = 1), or one shard's statistics (Rmax, arg-max, sum S, sum S eps) for the
// exchange between ranks (write_base = 0: act_sequence is left to the merge).  A block owns ROWS consecutive (step, action dimension) rows of one planner:
// it finds Rmax / first arg-max itself (every block gets the same values, in the same fixed order), evaluates
// every weight S_k = exp(kappa (R_k - Rmax)) once and accumulates sum S and its ROWS weighted sums, then writes
// its entries of act_sequence.  ROWS = NV (one block per step) when there are many planners to fill the GPU,
// ROWS = 1, 2 or 4 for few planners: as many blocks as there are CUs, no more (every block evaluates all K
// weights, which is most of its time; a second block on a CU doubles the kernel's).  The last block of a
// planner may own fewer rows.  Any block size that is a multiple of 64 up to 1024.
//
// Large K (weights != nullptr): the weights were evaluated once by update_weights_kernel -- [K] values, then Rmax and the
// local arg-max -- and the blocks only stream them against their rows (otherwise every one of the blocks would evaluate
// all K exponentials again).  Same values, same order of summation, same bits.
template <class T, int NV, int ROWS>
__global__ void __launch_bounds__(1024) update_rows_kernel(const T* __restrict__ R, const T* __restrict__ eps, int K, double kappa,
                                                           int k_offset, int HM, int write_base, double* __restrict__ base,
                                                           double* __restrict__ part, double* __restrict__ result,
                                                           const double* __restrict__ weights) {
    __shared__ double s_v[16][ROWS + 1];
    __shared__ int s_i[16];
    const int row0 = blockIdx.x * ROWS, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nthreads = blockDim.x, nwaves = nthreads >> 6;
    // rows this block owns (HM is a multiple of NV, so the one-block-per-step shape always owns all of its rows)
    const int nr = (ROWS == NV || HM - row0 >= ROWS) ? ROWS : HM - row0;
    R += (size_t)blockIdx.y * K; eps += ((size_t)blockIdx.y * HM + row0) * K;
    base += (size_t)blockIdx.y * HM + row0; part += (size_t)blockIdx.y * (3 + HM); result += 2 * blockIdx.y;
    if (weights) weights += (size_t)blockIdx.y * (K + 2);
    // Up to HOLD samples per thread stay in registers: the returns are read once, and the noise rows are
    // requested before the reduction that finds the maximum, so their latency is spent there.
    constexpr int HOLD = 4;
    // (not for the one-block-per-step shape of large batches: that one streams, and lives on occupancy)
    const bool held = ROWS <= 4 && K <= HOLD * nthreads;
    T rv[HOLD], ev[ROWS][HOLD];
    if (held) {
#pragma unroll
        for (int e = 0; e < HOLD; e++) {
            // (every load unconditional, from an address that exists -- a load behind a bounds test is waited for
            // before the next one is issued, twelve round trips instead of one; what a sample or row beyond the end
            // fetches is never used)
            const int k = tid + e * nthreads, kc = k < K ? k : K - 1;
            rv[e] = R[kc];
#pragma unroll
            for (int j = 0; j < ROWS; j++) ev[j][e] = eps[(size_t)(j < nr ? j : 0) * K + kc];
        }
    }
    double bv = -DBL_MAX;
    int bi = INT_MAX;
    if (weights) {
        bv = weights[K]; bi = (int)weights[K + 1];
    } else {
        if (held) {
    #pragma unroll
            for (int e = 0; e < HOLD; e++) {
                const int k = tid + e * nthreads;
                if (k < K) argmax_combine(bv, bi, return_for_max(rv[e]), k);
            }
        } else {
            for (int k = tid; k < K; k += nthreads) argmax_combine(bv, bi, return_for_max(R[k]), k);
        }
    #pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const double ov = shfl_xor_d(bv, m);
            const int oi = __shfl_xor(bi, m, 64);
            argmax_combine(bv, bi, ov, oi);
        }
        if (lane == 0) { s_v[wave][0] = bv; s_i[wave] = bi; }
        __syncthreads();
        bv = s_v[0][0]; bi = s_i[0];
        for (int w = 1; w < nwaves; w++) argmax_combine(bv, bi, s_v[w][0], s_i[w]);
        __syncthreads();
    }
    double acc[ROWS + 1];            // [ROWS]: sum of the weights
#pragma unroll
    for (int j = 0; j <= ROWS; j++) acc[j] = 0.0;
    if (held) {
#pragma unroll
        for (int e = 0; e < HOLD; e++) {
            const int k = tid + e * nthreads;
            if (k < K) {
                const double sk = sample_weight(kappa, rv[e], bv);
#pragma unroll
                for (int j = 0; j < ROWS; j++) if (j < nr) acc[j] = fma(sk, (double)ev[j][e], acc[j]);
                acc[ROWS] += sk;
            }
        }
    } else {
        // streaming: U samples' loads are in flight before the first is used; the sums run in the order of k
        constexpr int U = 4;
        int k = tid;
        for (; k + (U - 1) * nthreads < K; k += U * nthreads) {
            double sk[U];
            T ru[U];
            T e[ROWS][U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (weights) sk[u] = weights[k + u * nthreads];
                else ru[u] = R[k + u * nthreads];
#pragma unroll
                for (int j = 0; j < ROWS; j++) e[j][u] = eps[(size_t)(j < nr ? j : 0) * K + k + u * nthreads];
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (!weights) sk[u] = sample_weight(kappa, ru[u], bv);
#pragma unroll
                for (int j = 0; j < ROWS; j++) if (j < nr) acc[j] = fma(sk[u], (double)e[j][u], acc[j]);
                acc[ROWS] += sk[u];
            }
        }
        for (; k < K; k += nthreads) {
            const double sk = weights ? weights[k] : sample_weight(kappa, R[k], bv);
#pragma unroll
            for (int j = 0; j < ROWS; j++) if (j < nr) acc[j] = fma(sk, (double)eps[(size_t)j * K + k], acc[j]);
            acc[ROWS] += sk;
        }
    }
#pragma unroll
    for (int j = 0; j <= ROWS; j++) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) acc[j] += shfl_xor_d(acc[j], m);
        if (lane == 0) s_v[wave][j] = acc[j];
    }
    __syncthreads();
    if (tid <= ROWS) {
        double tot = 0.0;
        for (int w = 0; w < nwaves; w++) tot += s_v[w][tid];
        s_v[0][tid] = tot;                 // only thread tid reads column tid of the other rows
    }
    __syncthreads();
    const double d = s_v[0][ROWS];
    if (tid < nr) {
        const double n = s_v[0][tid];
        if (write_base) base[tid] = (base[tid] * d + n) / (d + 1e-6);
        part[3 + row0 + tid] = n;
    }
    if (blockIdx.x == 0 && tid == 0) {
        const int best = bi == INT_MAX ? 0 : bi;          // (no sample with a finite return: none is the best)
        part[0] = bv; part[1] = (double)(k_offset + best); part[2] = d;
        if (write_base) { result[0] = bv; result[1] = (double)(k_offset + best); }
    }
}

// The update of batched planners without reading the noise back: a block owns one action dimension j of one planner
// and ALL steps; every thread draws the perturbations of its samples again (philox_sequence with the key the noise
// kernel left behind: the same bits the rollouts read) and keeps one running sum per step.  Each sum is formed in the
// order update_steps_kernel uses -- launched with 64 threads: a lane's samples k = lane, lane + 64, ... in turn, then
// the lanes by the exclusive-or tree (more threads: the waves' sums in turn after that) -- so act_sequence comes out bit
// for bit as from the stored noise; what changes is the memory traffic (no read of eps: a third of the bytes of a batched planning step).
// HT: compile-time bound on the horizon (one fp64 running sum per step in registers).
template <class T, int NV, int HT>
__global__ void __launch_bounds__(256) update_regen_kernel(const T* __restrict__ R, int K, int H, double kappa, int k_offset,
                                                           int write_base, double* __restrict__ base, double* __restrict__ part,
                                                           double* __restrict__ result, const uint32_t* __restrict__ key,
                                                           FilterCoefs fc) {
    __shared__ double s_v[4][HT + 1];
    __shared__ int s_i[4];
    const int j = blockIdx.x, pl = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nthreads = blockDim.x, nwaves = nthreads >> 6;          // 64 to 256 threads
    const int HM = H * NV;
    R += (size_t)pl * K;
    base += (size_t)pl * HM; part += (size_t)pl * (3 + HM); result += 2 * pl;
    const uint32_t key_lo = key[2 * pl], key_hi = key[2 * pl + 1];
    // Rmax and the first arg-max, as every block of update_rows_kernel finds them
    double bv = -DBL_MAX;
    int bi = INT_MAX;
    for (int k = tid; k < K; k += nthreads) argmax_combine(bv, bi, return_for_max(R[k]), k);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const double ov = shfl_xor_d(bv, m);
        const int oi = __shfl_xor(bi, m, 64);
        argmax_combine(bv, bi, ov, oi);
    }
    if (lane == 0) { s_v[wave][0] = bv; s_i[wave] = bi; }
    __syncthreads();
    bv = s_v[0][0]; bi = s_i[0];
    for (int w = 1; w < nwaves; w++) argmax_combine(bv, bi, s_v[w][0], s_i[w]);
    __syncthreads();
    double acc[HT + 1];              // [HT]: sum of the weights
#pragma unroll
    for (int t = 0; t <= HT; t++) acc[t] = 0.0;
    for (int k = tid; k < K; k += nthreads) {
        const double sk = sample_weight(kappa, R[k], bv);
        // (fully unrolled: t is a constant in every call, the running sums stay in registers)
        philox_sequence<HT>((uint32_t)(k_offset + k), j, pl, key_lo, key_hi, H, fc,
                            [&](int t, double e) { acc[t] = fma(sk, (double)(T)e, acc[t]); });
        acc[HT] += sk;
    }
#pragma unroll
    for (int t = 0; t <= HT; t++) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) acc[t] += shfl_xor_d(acc[t], m);
        if (lane == 0) s_v[wave][t] = acc[t];
    }
    __syncthreads();
    for (int t = tid; t <= HT; t += nthreads) {
        double tot = 0.0;
        for (int w = 0; w < nwaves; w++) tot += s_v[w][t];
        s_v[0][t] = tot;                   // only this thread reads column t of the other rows
    }
    __syncthreads();
    const double d = s_v[0][HT];
    for (int t = tid; t < H; t += nthreads) {
        const double n = s_v[0][t];
        if (write_base) base[t * NV + j] = (base[t * NV + j] * d + n) / (d + 1e-6);
        part[3 + t * NV + j] = n;
    }
    if (j == 0 && tid == 0) {
        const int best = bi == INT_MAX ? 0 : bi;          // (no sample with a finite return: none is the best)
        part[0] = bv; part[1] = (double)(k_offset + best); part[2] = d;
        if (write_base) { result[0] = bv; result[1] = (double)(k_offset + best); }
    }
}

// The weights of a planner with many samples, evaluated once: every block finds Rmax / the first arg-max itself (the
// reduction of update_rows_kernel, so the same values), then the blocks share the K exponentials.  out: [K] weights,
// Rmax, local arg-max, per planner.
template <class T>
__global__ void __launch_bounds__(1024) update_weights_kernel(const T* __restrict__ R, int K, double kappa, double* __restrict__ out) {
    __shared__ double s_v[16];
    __shared__ int s_i[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthreads = blockDim.x, nwaves = nthreads >> 6;
    R += (size_t)blockIdx.y * K; out += (size_t)blockIdx.y * (K + 2);
    double bv = -DBL_MAX;
    int bi = INT_MAX;
    for (int k = tid; k < K; k += nthreads) argmax_combine(bv, bi, return_for_max(R[k]), k);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const double ov = shfl_xor_d(bv, m);
        const int oi = __shfl_xor(bi, m, 64);
        argmax_combine(bv, bi, ov, oi);
    }
    if (lane == 0) { s_v[wave] = bv; s_i[wave] = bi; }
    __syncthreads();
    bv = s_v[0]; bi = s_i[0];
    for (int w = 1; w < nwaves; w++) argmax_combine(bv, bi, s_v[w], s_i[w]);
    const int per = (K + gridDim.x - 1) / gridDim.x, lo = blockIdx.x * per, hi = lo + per < K ? lo + per : K;
    for (int k = lo + tid; k < hi; k += nthreads) out[k] = sample_weight(kappa, R[k], bv);
    if (blockIdx.x == 0 && tid == 0) { out[K] = bv; out[K + 1] = (double)bi; }
}

// ---- batched planners, at most 4096 samples each, weights evaluated beforehand (update_weights_kernel) ----
// With few samples per planner the launch is bound by the number of waves the dispatcher has to start (about a
// nanosecond each), not by HBM: one block of four waves per step starts 262 144 waves for 1024 planners of 64 samples,
// three in four of them without a sample (270 us).  Here every wave owns a step (four steps to a block): lane l sums
// its samples k = l, l + 64, ... in turn, the lanes are added by the exclusive-or tree, and lanes 0 .. NV - 1 write the
// step's entries of act_sequence -- no shared memory, no barrier.  update_regen_kernel launched with 64 threads forms
// its sums in the same order.  Measured against sub-blocks of two and four waves per step (whole step, same box):
// 1024 planners x 512 samples 2.47 -> 2.36 ms, 256 x 256 0.43 -> 0.38, 128 x 512 0.39 -> 0.38, 66 x 1000 unchanged.
// (Several steps per wave, to read a weight once for all of them, were measured too: two steps no faster, four slower
// at every shape -- the registers halve the occupancy.)
template <class T, int NV>
__global__ void __launch_bounds__(256) update_steps_kernel(const T* __restrict__ eps, int K, int H, int k_offset, int write_base,
                                                           double* __restrict__ base, double* __restrict__ part,
                                                           double* __restrict__ result, const double* __restrict__ weights) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int HM = H * NV, pl = blockIdx.y;
    const int t = blockIdx.x * 4 + wave;                // this wave's step
    if (t >= H) return;
    eps += ((size_t)pl * HM + (size_t)t * NV) * K;
    base += (size_t)pl * HM; part += (size_t)pl * (3 + HM); result += 2 * pl; weights += (size_t)pl * (K + 2);
    double acc[NV + 1];              // [NV]: sum of the weights
#pragma unroll
    for (int j = 0; j <= NV; j++) acc[j] = 0.0;
    for (int k = lane; k < K; k += 64) {
        const double sk = weights[k];
        T e[NV];
#pragma unroll
        for (int j = 0; j < NV; j++) e[j] = eps[(size_t)j * K + k];
#pragma unroll
        for (int j = 0; j < NV; j++) acc[j] = fma(sk, (double)e[j], acc[j]);
        acc[NV] += sk;
    }
#pragma unroll
    for (int j = 0; j <= NV; j++) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) acc[j] += shfl_xor_d(acc[j], m);
    }
    const double d = acc[NV];
    double n = 0.0;                   // lane j takes sum j (NV < 64)
#pragma unroll
    for (int j = 0; j < NV; j++) n = lane == j ? acc[j] : n;
    if (lane < NV) {
        const int hm = t * NV + lane;
        if (write_base) base[hm] = (base[hm] * d + n) / (d + 1e-6);
        part[3 + hm] = n;
    }
    if (t == 0 && lane == 0) {
        const double bv = weights[K];
        const int bi = (int)weights[K + 1], best = bi == INT_MAX ? 0 : bi;      // (no finite return: none is the best)
        part[0] = bv; part[1] = (double)(k_offset + best); part[2] = d;
        if (write_base) { result[0] = bv; result[1] = (double)(k_offset + best); }
    }
}

// ---- many samples, few planners (update_in_chunks): the samples are cut into chunks as well as the rows ----
// Four small launches.  (1) chunk_max: 64 blocks per planner each find the maximum / first arg-max of their stretch
// of the returns.  (2) chunk_weights: every wave folds those 64 pairs (same pairs, same order: same values in every
// wave), then the blocks write S_k = exp(kappa (R_k - Rmax)) for their stretch, once.  (3) chunk_sums: block
// (step t, chunk c) streams chunk c's weights against the NV rows of step t: the weights are read once per NV rows
// and stay in L2 between the H blocks that share a chunk.  (4) chunk_finish: the chunks' sums are added in the
// order of the chunks -- whatever order the blocks ran in, the same bits -- and act_sequence is written.
constexpr int UPDATE_MAX_BLOCKS = 64;
template <class T>
__global__ void __launch_bounds__(1024) update_chunk_max_kernel(const T* __restrict__ R, int K, double* __restrict__ maxes) {
    __shared__ double s_v[16];
    __shared__ int s_i[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    R += (size_t)blockIdx.y * K; maxes += (size_t)blockIdx.y * 2 * UPDATE_MAX_BLOCKS;
    const int per = (K + UPDATE_MAX_BLOCKS - 1) / UPDATE_MAX_BLOCKS, lo = blockIdx.x * per, hi = lo + per < K ? lo + per : K;
    double bv = -DBL_MAX;
    int bi = INT_MAX;
    for (int k = lo + tid; k < hi; k += 1024) argmax_combine(bv, bi, return_for_max(R[k]), k);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const double ov = shfl_xor_d(bv, m);
        const int oi = __shfl_xor(bi, m, 64);
        argmax_combine(bv, bi, ov, oi);
    }
    if (lane == 0) { s_v[wave] = bv; s_i[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; w++) argmax_combine(bv, bi, s_v[w], s_i[w]);
        maxes[2 * blockIdx.x] = bv; maxes[2 * blockIdx.x + 1] = (double)bi;     // (INT_MAX is exact in a double)
    }
}
template <class T>
__global__ void __launch_bounds__(1024) update_chunk_weights_kernel(const T* __restrict__ R, int K, double kappa,
                                                                    const double* __restrict__ maxes, double* __restrict__ out) {
    const int tid = threadIdx.x, lane = tid & 63;
    R += (size_t)blockIdx.y * K; out += (size_t)blockIdx.y * (K + 2); maxes += (size_t)blockIdx.y * 2 * UPDATE_MAX_BLOCKS;
    double bv = maxes[2 * lane];
    int bi = (int)maxes[2 * lane + 1];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const double ov = shfl_xor_d(bv, m);
        const int oi = __shfl_xor(bi, m, 64);
        argmax_combine(bv, bi, ov, oi);
    }
    const int per = (K + gridDim.x - 1) / gridDim.x, lo = blockIdx.x * per, hi = lo + per < K ? lo + per : K;
    for (int k = lo + tid; k < hi; k += 1024) out[k] = sample_weight(kappa, R[k], bv);
    if (blockIdx.x == 0 && tid == 0) { out[K] = bv; out[K + 1] = (double)bi; }
}
// sums: [P][chunks][HM + 1] -- per chunk the HM weighted sums, then the sum of the chunk's weights
template <class T, int NV>
__global__ void __launch_bounds__(256) update_chunk_sums_kernel(const T* __restrict__ eps, const double* __restrict__ weights, int K,
                                                                 int HM, int per, double* __restrict__ sums) {
    constexpr int BT = 256;
    __shared__ double s_v[BT / 64][NV + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, row0 = blockIdx.x * NV, c = blockIdx.y, pl = blockIdx.z;
    eps += ((size_t)pl * HM + row0) * K; weights += (size_t)pl * (K + 2);
    sums += ((size_t)pl * gridDim.y + c) * (HM + 1);
    const int lo = c * per, hi = lo + per < K ? lo + per : K;
    double acc[NV + 1];
#pragma unroll
    for (int j = 0; j <= NV; j++) acc[j] = 0.0;
    constexpr int U = 4;
    int k = lo + tid;
    for (; k + (U - 1) * BT < hi; k += U * BT) {
        double sk[U];
        T e[NV][U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            sk[u] = weights[k + u * BT];
#pragma unroll
            for (int j = 0; j < NV; j++) e[j][u] = eps[(size_t)j * K + k + u * BT];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
#pragma unroll
            for (int j = 0; j < NV; j++) acc[j] = fma(sk[u], (double)e[j][u], acc[j]);
            acc[NV] += sk[u];
        }
    }
    for (; k < hi; k += BT) {
        const double sk = weights[k];
#pragma unroll
        for (int j = 0; j < NV; j++) acc[j] = fma(sk, (double)eps[(size_t)j * K + k], acc[j]);
        acc[NV] += sk;
    }
#pragma unroll
    for (int j = 0; j <= NV; j++) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) acc[j] += shfl_xor_d(acc[j], m);
        if (lane == 0) s_v[wave][j] = acc[j];
    }
    __syncthreads();
    if (tid <= NV) {
        double tot = 0.0;
        for (int w = 0; w < BT / 64; w++) tot += s_v[w][tid];
        if (tid < NV) sums[row0 + tid] = tot;
        else if (blockIdx.x == 0) sums[HM] = tot;
    }
}
static __global__ void __launch_bounds__(256) update_chunk_finish_kernel(const double* __restrict__ sums, const double* __restrict__ weights,
                                                                  int K, int HM, int chunks, int k_offset, int write_base,
                                                                  double* __restrict__ base, double* __restrict__ part,
                                                                  double* __restrict__ result) {
    const int hm = blockIdx.x * 256 + threadIdx.x, pl = blockIdx.y;
    sums += (size_t)pl * chunks * (HM + 1); weights += (size_t)pl * (K + 2);
    base += (size_t)pl * HM; part += (size_t)pl * (3 + HM); result += 2 * pl;
    double d = 0.0, n = 0.0;
    for (int c = 0; c < chunks; c++) d += sums[(size_t)c * (HM + 1) + HM];
    if (hm < HM) {
        for (int c = 0; c < chunks; c++) n += sums[(size_t)c * (HM + 1) + hm];
        if (write_base) base[hm] = (base[hm] * d + n) / (d + 1e-6);
        part[3 + hm] = n;
    }
    if (hm == 0) {
        const double bv = weights[K];
        const int bi = (int)weights[K + 1], best = bi == INT_MAX ? 0 : bi;
        part[0] = bv; part[1] = (double)(k_offset + best); part[2] = d;
        if (write_base) { result[0] = bv; result[1] = (double)(k_offset + best); }
    }
}

// ------------------------------------------------------------------------------------------
// per-model dispatch
// ------------------------------------------------------------------------------------------
struct RolloutArgs {
    const DevState* st; const double* base; const void* eps; const double* ref;
    int K, H, P, reward_type; double gamma; int explicit_act;
    bool lean;      // dense reward, noise-relative samples, no dumps, no reference, no critic
    void* R; void* obs; void* rew; int* status; void* fobs; int* next_step;
    int sweep_cap;  // most sweeps of the active-set iteration per substep
};

struct ModelOps {
    const char* name;
    int nv, nq;
    bool (*matches)(const mppi_model_desc*, std::string&);
    hipError_t (*rollout)(int dtype, const RolloutArgs&, hipStream_t);
    hipError_t (*noise_upload)(int dtype, const double* raw, int K, int H, int P, const FilterCoefs&, int filter, void* eps, hipStream_t);
    hipError_t (*noise_philox)(int dtype, int K, int H, int P, int k_offset, uint64_t seed, const DevState* st, int seed_from_state,
                               const FilterCoefs&, void* eps, uint32_t* key_out, hipStream_t);
    hipError_t (*advance)(DevState* st, double* base, const double* ref, int reward_type, int H, int P, int tail_mode,
                          const double* mean, double* log, int log_cap, int* log_count, double* last_row, int* status,
                          double* mirror, hipStream_t);
    hipError_t (*env_step)(const double* in, int reward_type, int has_ref, int forward_only, double* out, hipStream_t);
    hipError_t (*critic)(int dtype, const void* fobs, int K, int P, int dim, const float* w, double g_last, void* R, hipStream_t);
    hipError_t (*advance_noise)(int dtype, DevState* st, double* base, const double* ref, int reward_type, int H, int P, int tail_mode,
                                const double* mean, double* log, int log_cap, int* log_count, double* last_row, int* status,
                                int K, int k_offset, uint64_t seed, const int* next_step, const FilterCoefs& fc, void* eps,
                                uint32_t* key_out, double* mirror, hipStream_t);
    // the update of batched planners from noise drawn again (update_regen_kernel); false: no kernel for this shape
    bool (*update_regen_fits)(int K, int H, int P);
    hipError_t (*update_regen)(int dtype, const void* R, int K, int H, int P, double kappa, int k_offset, int write_base, double* base,
                               double* part, double* result, const uint32_t* key, const FilterCoefs& fc, hipStream_t);
    hipError_t (*update_rows)(int dtype, const void* R, const void* eps, int K, int H, int P, double kappa, int k_offset, int write_base, double* weights, double* base,
                              double* part, double* result, hipStream_t);
};

static bool near(double a, double b) { return std::fabs(a - b) <= 1e-12 * (1.0 + std::fabs(b)); }

template <class MD> static bool model_matches(const mppi_model_desc* d, std::string& why) {
    constexpr int NV = MD::NV;
#define CHK(cond, what)                                                                     \
    if (!(cond)) { why = std::string("model field differs from the compiled kernel: ") + what; return false; }
    CHK(d->nv == NV, "nv");
    CHK(d->nq == MD::NQ, "nq");
    CHK(d->integrator == MD::INTEGRATOR, "integrator");
    CHK(d->frame_skip == MD::FRAME_SKIP, "frame_skip");
    CHK(near(d->h, MD::H), "timestep");
    CHK(d->contact_link == MD::contact_link, "contact_link");
    CHK(d->hand_link == MD::hand_link, "hand_link");
    for (int i = 0; i < NV; i++) {
        CHK(d->parent[i] == MD::parent[i], "parent");
        CHK(d->jtype[i] == MD::jtype[i], "joint type");
        CHK(d->qadr[i] == MD::qadr[i], "qpos address");
        CHK((d->limited[i] != 0) == (MD::limited[i] != 0), "limited");
        const double m = d->mass[i];
        for (int c = 0; c < 3; c++) {
            CHK(near(d->pos[i][c], MD::pos[i][c]), "pos");
            CHK(near(d->axis[i][c], (c == MD::axis_id[i]) ? (double)MD::axis_sign[i] : 0.0), "axis");
            CHK(near(m * d->com[i][c], MD::mc[i][c]), "com");
        }
        // rotational inertia about the link origin
        const double* I = d->inertia[i];
        const double* c = d->com[i];
        const double cc = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
        const double IO[6] = {I[0] + m * (cc - c[0] * c[0]), I[1] + m * (cc - c[1] * c[1]), I[2] + m * (cc - c[2] * c[2]),
                              I[3] - m * c[0] * c[1],        I[4] - m * c[0] * c[2],        I[5] - m * c[1] * c[2]};
        for (int e = 0; e < 6; e++) CHK(near(IO[e], MD::IO[i][e]), "inertia");
        CHK(near(m, MD::mass[i]), "mass");
        CHK(near(d->armature[i], MD::armature[i]), "armature");
        CHK(near(d->damping[i], MD::damping[i]), "damping");
        CHK(near(d->gear[i], MD::gear[i]), "gear");
        CHK(near(d->ctrl_lo[i], MD::ctrl_lo[i]) && near(d->ctrl_hi[i], MD::ctrl_hi[i]), "ctrlrange");
        if (MD::limited[i]) {
            CHK(near(d->range_lo[i], MD::range_lo[i]) && near(d->range_hi[i], MD::range_hi[i]), "range");
            CHK(near(d->dof_invweight[i], MD::dof_invweight[i]), "dof_invweight");
        }
    }
    if (MD::contact_link >= 0) {
        for (int c = 0; c < 3; c++) CHK(near(d->contact_pos[c], MD::contact_pos[c]), "contact_pos");
        CHK(near(d->contact_radius, MD::contact_radius), "contact_radius");
        CHK(near(d->contact_plane_z, MD::contact_plane_z), "contact_plane_z");
        CHK(near(d->contact_margin, MD::contact_margin), "contact_margin");
        CHK(near(d->contact_invweight, MD::contact_invweight), "contact_invweight");
    }
    for (int c = 0; c < 3; c++) CHK(near(d->hand_pos[c], MD::hand_pos[c]), "hand_pos");
    for (int c = 0; c < 3; c++) CHK(near(d->gravity[c], MD::gravity[c]), "gravity");
    CHK(near(d->solref_k, MD::solref_k) && near(d->solref_b, MD::solref_b), "solref");
    for (int c = 0; c < 5; c++) CHK(near(d->solimp[c], MD::solimp[c]), "solimp");
#undef CHK
    return true;
}

template <class MD> static hipError_t launch_rollout(int dtype, const RolloutArgs& a, hipStream_t s) {
    // A wave is 64 consecutive samples of one planner whatever the block size.  The waves of a block go to SIMDs 0, 1,
    // ... of one CU: blocks of four put one wave on every SIMD of a CU, blocks of two one on SIMDs 0 and 1.  Blocks of
    // one wave are placed SIMD by SIMD by the dispatcher, which after a launch whose grid took several rounds (the
    // batched update) packs them two to a SIMD while other SIMDs idle: 128 planners x 512 samples, 1024 waves for 1024
    // SIMDs, 0.61 -> 0.39 ms per step with blocks of four; 64 planners x 200, 256 waves, 0.42 -> 0.34 ms with blocks of
    // two or four.  Up to two waves per CU blocks of two keep the waves spread over all CUs, which the variants that
    // write their trajectories need (18-DoF tree, 512 waves, observations dumped: 0.71 ms against 0.95 with blocks of
    // four); beyond that blocks of two would leave SIMDs 2 and 3 idle.  Up to 64 waves -- the single planner of 4096
    // samples -- every shape takes the same time, and single waves stay.
    const int waves = ((a.K + 63) / 64) * a.P;
    const int bt = waves <= 64 ? 64 : (waves <= 512 ? 128 : 256);
    const dim3 grid((waves + bt / 64 - 1) / (bt / 64)), block(bt);
    if (dtype == MPPI_F32 && a.lean)
        hipLaunchKernelGGL((rollout_kernel<float, MD, true>), grid, block, 0, s, a.st, a.base, (const float*)a.eps, a.ref, a.K,
                           a.H, a.P, a.reward_type, a.gamma, a.explicit_act, (float*)a.R, (float*)a.obs, (float*)a.rew, a.status, (float*)a.fobs, a.next_step, a.sweep_cap);
    else if (dtype == MPPI_F32)
        hipLaunchKernelGGL((rollout_kernel<float, MD, false>), grid, block, 0, s, a.st, a.base, (const float*)a.eps, a.ref, a.K,
                           a.H, a.P, a.reward_type, a.gamma, a.explicit_act, (float*)a.R, (float*)a.obs, (float*)a.rew, a.status, (float*)a.fobs, a.next_step, a.sweep_cap);
    else
        hipLaunchKernelGGL((rollout_kernel<double, MD, false>), grid, block, 0, s, a.st, a.base, (const double*)a.eps, a.ref, a.K,
                           a.H, a.P, a.reward_type, a.gamma, a.explicit_act, (double*)a.R, (double*)a.obs, (double*)a.rew, a.status, (double*)a.fobs, a.next_step, a.sweep_cap);
    return hipGetLastError();
}
template <class MD>
static hipError_t launch_noise_upload(int dtype, const double* raw, int K, int H, int P, const FilterCoefs& fc, int filter, void* eps,
                                      hipStream_t s) {
    const dim3 grid((K + 63) / 64, P), block(256);
    if (dtype == MPPI_F32)
        hipLaunchKernelGGL((noise_upload_kernel<float, MD::NV>), grid, block, 0, s, raw, K, H, fc, filter, (float*)eps);
    else
        hipLaunchKernelGGL((noise_upload_kernel<double, MD::NV>), grid, block, 0, s, raw, K, H, fc, filter, (double*)eps);
    return hipGetLastError();
}
template <class MD>
static hipError_t launch_noise_philox(int dtype, int K, int H, int P, int k_offset, uint64_t seed, const DevState* st,
                                      int seed_from_state, const FilterCoefs& fc, void* eps, uint32_t* key_out, hipStream_t s) {
    const dim3 grid((K * MD::NV + 255) / 256, P), block(256);
    const uint32_t lo = (uint32_t)seed, hi = (uint32_t)(seed >> 32);
    if (dtype == MPPI_F32)
        hipLaunchKernelGGL((noise_philox_kernel<float, MD::NV>), grid, block, 0, s, K, H, k_offset, lo, hi, st, seed_from_state, fc, (float*)eps, key_out);
    else
        hipLaunchKernelGGL((noise_philox_kernel<double, MD::NV>), grid, block, 0, s, K, H, k_offset, lo, hi, st, seed_from_state, fc, (double*)eps, key_out);
    return hipGetLastError();
}
// the shape update_rows_kernel serves with one block of 256 threads per step: many planners, at most 1024 samples each
inline bool update_regen_fits(int K, int H, int P) { return (long)H * P >= 1024 && K <= 1024 && H <= 128; }
template <class MD>
static hipError_t launch_update_regen(int dtype, const void* R, int K, int H, int P, double kappa, int k_offset, int write_base,
                                      double* base, double* part, double* result, const uint32_t* key, const FilterCoefs& fc,
                                      hipStream_t s) {
    constexpr int NV = MD::NV;
    const dim3 grid(NV, P), block(64);           // one wave: the summation order of update_steps_kernel
#define MPPI_LAUNCH_REGEN(T_, HT_)                                                                                            \
    hipLaunchKernelGGL((update_regen_kernel<T_, NV, HT_>), grid, block, 0, s, (const T_*)R, K, H, kappa, k_offset, write_base, \
                       base, part, result, key, fc)
    if (dtype == MPPI_F32) { if (H <= 64) MPPI_LAUNCH_REGEN(float, 64); else MPPI_LAUNCH_REGEN(float, 128); }
    else { if (H <= 64) MPPI_LAUNCH_REGEN(double, 64); else MPPI_LAUNCH_REGEN(double, 128); }
#undef MPPI_LAUNCH_REGEN
    return hipGetLastError();
}
template <class MD>
static hipError_t launch_advance(DevState* st, double* base, const double* ref, int reward_type, int H, int P, int tail_mode,
                                 const double* mean, double* log, int log_cap, int* log_count, double* last_row, int* status,
                                 double* mirror, hipStream_t s) {
    // The fp64 step's registers let a CU hold two blocks of 256 threads: 512 planners a round, each as long as the one
    // lane that steps the environment takes.  From 512 planners on one wave per planner (1024 planners: 52 -> 33 us,
    // 512: 28 -> 22 us; four such waves in one block, a wave on every SIMD of a CU, take 31 us at any count: the lanes
    // that step an environment slow each other down when they share a CU).
    if (P >= 512 && H * MD::NV <= 64 * 16)
        hipLaunchKernelGGL((advance_kernel<MD, 64>), dim3(P), dim3(64), 0, s, st, base, ref, reward_type, H, tail_mode, mean, log,
                           log_cap, log_count, last_row, status, nullptr);
    else
        hipLaunchKernelGGL((advance_kernel<MD, 256>), dim3(P), dim3(256), 0, s, st, base, ref, reward_type, H, tail_mode, mean, log,
                           log_cap, log_count, last_row, status, P == 1 ? mirror : nullptr);
    return hipGetLastError();
}
template <class MD>
static hipError_t launch_advance_noise(int dtype, DevState* st, double* base, const double* ref, int reward_type, int H, int P,
                                       int tail_mode, const double* mean, double* log, int log_cap, int* log_count,
                                       double* last_row, int* status, int K, int k_offset, uint64_t seed, const int* next_step,
                                       const FilterCoefs& fc, void* eps, uint32_t* key_out, double* mirror, hipStream_t s) {
    const dim3 grid(1 + (K * MD::NV + 255) / 256, P), block(256);
    if (P != 1) mirror = nullptr;          // the host mirror serves single planners
    const uint32_t lo = (uint32_t)seed, hi = (uint32_t)(seed >> 32);
    if (dtype == MPPI_F32)
        hipLaunchKernelGGL((advance_noise_kernel<float, MD>), grid, block, 0, s, st, base, ref, reward_type, H, tail_mode, mean, log,
                           log_cap, log_count, last_row, status, K, k_offset, lo, hi, next_step, fc, (float*)eps, key_out, mirror);
    else
        hipLaunchKernelGGL((advance_noise_kernel<double, MD>), grid, block, 0, s, st, base, ref, reward_type, H, tail_mode, mean, log,
                           log_cap, log_count, last_row, status, K, k_offset, lo, hi, next_step, fc, (double*)eps, key_out, mirror);
    return hipGetLastError();
}
template <class MD>
static hipError_t launch_env_step(const double* in, int reward_type, int has_ref, int forward_only, double* out, hipStream_t s) {
    hipLaunchKernelGGL((env_step_kernel<MD>), dim3(1), dim3(64), 0, s, in, reward_type, has_ref, forward_only, out);
    return hipGetLastError();
}
template <class MD>
static hipError_t launch_critic(int dtype, const void* fobs, int K, int P, int dim, const float* w, double g_last, void* R, hipStream_t s) {
    const dim3 grid((K + 31) / 32, P), block(256);
    if (dtype == MPPI_F32)
        hipLaunchKernelGGL((critic_kernel<float, MD::NV>), grid, block, 0, s, (const float*)fobs, K, dim, w, g_last, (float*)R);
    else
        hipLaunchKernelGGL((critic_kernel<double, MD::NV>), grid, block, 0, s, (const double*)fobs, K, dim, w, g_last, (double*)R);
    return hipGetLastError();
}
// more samples than a block holds in registers, few planners: the weights are worth evaluating once
// ... and so are they for batched planners, whose update has one block per step: H blocks would evaluate the same K
// exponentials each
inline bool update_wants_weights(int K, int H, int P) { return (K > 4096 && (long)H * P < 1024) || ((long)H * P >= 1024 && K >= 64); }
// many samples, few planners: rows x chunks of samples (update_chunk_*_kernel)
inline bool update_in_chunks(int K, int H, int P) { return K > 4096 && (long)H * P < 1024; }
inline int update_chunks(int K, int H, int P) {
    // One block per CU, four waves each: measured best from 8192 to 524288 samples.  (More and larger blocks finish
    // this kernel no sooner, and a grid of 1024-thread blocks that takes several rounds leaves the next rollout of
    // one wave per SIMD half as slow again: 65536 samples, 285 -> 470 us.)
    long c = (256 + (long)H * P - 1) / ((long)H * P);
    if (c > K / 256) c = K / 256;                           // a chunk gives every thread of its block a sample
    return (int)(c < 1 ? 1 : (c > 64 ? 64 : c));
}
// the update's scratch, in doubles: [P][K + 2] weights, Rmax, arg-max; chunked: then [P][64][2] the stretches' maxima
// and [P][chunks][HM + 1] the chunks' sums
inline size_t update_scratch_doubles(int K, int H, int P, int NV) {
    size_t n = (size_t)P * ((size_t)K + 2);
    if (update_in_chunks(K, H, P)) n += (size_t)P * 2 * UPDATE_MAX_BLOCKS + (size_t)P * update_chunks(K, H, P) * ((size_t)H * NV + 1);
    return n;
}
// weights: scratch of update_scratch_doubles(K, H, P, NV) doubles, or null (every block evaluates the weights itself)
template <class MD>
static hipError_t launch_update_rows(int dtype, const void* R, const void* eps, int K, int H, int P, double kappa, int k_offset,
                                     int write_base, double* weights, double* base, double* part, double* result, hipStream_t s) {
    constexpr int NV = MD::NV;
    const int HM = H * NV;
    if (weights && update_in_chunks(K, H, P)) {
        const int chunks = update_chunks(K, H, P), per = (K + chunks - 1) / chunks;
        double* maxes = weights + (size_t)P * ((size_t)K + 2);
        double* sums = maxes + (size_t)P * 2 * UPDATE_MAX_BLOCKS;
        const dim3 mgrid(UPDATE_MAX_BLOCKS, P), sgrid(H, chunks, P), fgrid((HM + 255) / 256, P);
        if (dtype == MPPI_F32) {
            hipLaunchKernelGGL((update_chunk_max_kernel<float>), mgrid, dim3(1024), 0, s, (const float*)R, K, maxes);
            hipLaunchKernelGGL((update_chunk_weights_kernel<float>), mgrid, dim3(1024), 0, s, (const float*)R, K, kappa, maxes, weights);
            hipLaunchKernelGGL((update_chunk_sums_kernel<float, NV>), sgrid, dim3(256), 0, s, (const float*)eps, weights, K, HM, per, sums);
        } else {
            hipLaunchKernelGGL((update_chunk_max_kernel<double>), mgrid, dim3(1024), 0, s, (const double*)R, K, maxes);
            hipLaunchKernelGGL((update_chunk_weights_kernel<double>), mgrid, dim3(1024), 0, s, (const double*)R, K, kappa, maxes, weights);
            hipLaunchKernelGGL((update_chunk_sums_kernel<double, NV>), sgrid, dim3(256), 0, s, (const double*)eps, weights, K, HM, per, sums);
        }
        hipLaunchKernelGGL(update_chunk_finish_kernel, fgrid, dim3(256), 0, s, sums, weights, K, HM, chunks, k_offset, write_base, base,
                           part, result);
        return hipGetLastError();
    }
    const dim3 block(K > 1024 ? 1024 : 256);
    // enough blocks to spread over the GPU: one per step once there are many planners, else one per entry
#define MPPI_LAUNCH_UPDATE(ROWS_)                                                                                               \
    do {                                                                                                                        \
        const dim3 grid((HM + (ROWS_) - 1) / (ROWS_), P);                                                                       \
        if (dtype == MPPI_F32)                                                                                                  \
            hipLaunchKernelGGL((update_rows_kernel<float, NV, ROWS_>), grid, block, 0, s, (const float*)R, (const float*)eps,  \
                               K, kappa, k_offset, HM, write_base, base, part, result, w);                                     \
        else                                                                                                                    \
            hipLaunchKernelGGL((update_rows_kernel<double, NV, ROWS_>), grid, block, 0, s, (const double*)R, (const double*)eps, \
                               K, kappa, k_offset, HM, write_base, base, part, result, w);                                     \
    } while (0)
    const int cus = 256;
    const double* w = nullptr;
    if (weights && update_wants_weights(K, H, P)) {
        const dim3 wgrid(K >= 262144 ? 64 : (K > 4096 ? 32 : 1), P);
        if (dtype == MPPI_F32) hipLaunchKernelGGL((update_weights_kernel<float>), wgrid, dim3(1024), 0, s, (const float*)R, K, kappa, weights);
        else hipLaunchKernelGGL((update_weights_kernel<double>), wgrid, dim3(1024), 0, s, (const double*)R, K, kappa, weights);
        w = weights;
    }
    // many planners: a wave per step, up to 4096 samples per planner (16 planners x 4096: whole step 386 -> 376 us, 32 x
    // 2048: 405 -> 369; equal at 8192, and beyond that one wave per step has too few loads in flight: 16 x 65 536 4.6 -> 5.0 ms)
    if ((long)H * P >= 1024 && w && K <= 4096) {
        const dim3 grid((H + 3) / 4, P);
        if (dtype == MPPI_F32)
            hipLaunchKernelGGL((update_steps_kernel<float, NV>), grid, dim3(256), 0, s, (const float*)eps, K, H, k_offset, write_base,
                               base, part, result, w);
        else
            hipLaunchKernelGGL((update_steps_kernel<double, NV>), grid, dim3(256), 0, s, (const double*)eps, K, H, k_offset,
                               write_base, base, part, result, w);
    }
    else if ((long)H * P >= 1024) MPPI_LAUNCH_UPDATE(NV);                   // ... of more than 4096 samples: one block per step
    else if ((long)HM * P <= cus) MPPI_LAUNCH_UPDATE(1);
    else if ((long)((HM + 1) / 2) * P <= cus) MPPI_LAUNCH_UPDATE(2);
    else MPPI_LAUNCH_UPDATE(4);
#undef MPPI_LAUNCH_UPDATE
    return hipGetLastError();
}
template <class MD> static ModelOps make_ops() {
    return {MD::name, MD::NV, MD::NQ, &model_matches<MD>, &launch_rollout<MD>, &launch_noise_upload<MD>, &launch_noise_philox<MD>,
            &launch_advance<MD>, &launch_env_step<MD>, &launch_critic<MD>, &launch_advance_noise<MD>, &update_regen_fits,
            &launch_update_regen<MD>, &launch_update_rows<MD>};
}
