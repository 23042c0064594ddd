// Device-resident MPPI planning step for MI355X (gfx950): kernels and C ABI.
//
// Kernels (one lane = one sampled action sequence; wave = 64 samples):
//   noise_upload_kernel / noise_philox_kernel   sigma scaling + recursive 3-tap filter, writes eps[t][j][k]
//   rollout_kernel<T, MD, LEAN>                 H env steps x FRAME_SKIP substeps, fused reward + discounted return
//   update_rows_kernel<T, NV, ROWS>             max / arg-max over K, S_k = exp(kappa (R_k - Rmax)), sum S, sum S eps:
//                                               the whole update of a single rank, or a shard's statistics
//   merge_kernel                                rank-ordered merge of shard statistics, writes act_sequence
//   advance_kernel / advance_noise_kernel       fp64 step of the real environment, log, receding-horizon shift
//                                               (+ the next step's Philox noise on the other blocks)
//   shift_kernel                                receding-horizon shift + tail fill
//   env_step_kernel<MD>                         one fp64 env.step for the real environment
//   critic_kernel                               terminal value of the 128-wide tanh MLP
//
// Layout in HBM: eps is stored step-major, sample-minor ([H][M][K]) so that the 64 lanes of
// a wave read 64 consecutive values per (t, j); returns are [K]; act_sequence is [H][M] fp64.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <climits>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mppi_hip.h"
#include "env_core.hpp"
#include "models_gen.hpp"

using namespace mppi;

// ------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
#define HIP_OK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return fail(-2, std::string(#expr) + ": " + hipGetErrorString(e_));                        \
    } while (0)

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
extern "C" void mppi_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
    philox4x32_10(c, key[0], key[1]);
    for (int i = 0; i < 4; i++) out[i] = c[i];
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
// noise kernels: one lane per sample, recursion along t in registers
// eps[t] = b0*sigma*z[t] + b1*eps[t-1] + b2*eps[t-2] for t >= 2 (rows 0, 1 unfiltered); separate
// roundings for every product and sum so the fp64 path reproduces the host arithmetic bit for bit
// ------------------------------------------------------------------------------------------
struct FilterCoefs {
    double sigma[MPPI_MAXV];
    double b0, b1, b2;
};

template <class T, int NV>
__global__ void __launch_bounds__(64) noise_upload_kernel(const double* __restrict__ raw, int K, int H, FilterCoefs fc,
                                                          int filter, T* __restrict__ eps) {
    const int k = blockIdx.x * 64 + threadIdx.x;
    if (k >= K) return;
    raw += (size_t)blockIdx.y * K * H * NV;           // planner blockIdx.y of a batched handle
    eps += (size_t)blockIdx.y * H * NV * K;
    double e1[NV], e2[NV];
#pragma unroll
    for (int j = 0; j < NV; j++) e1[j] = e2[j] = 0.0;
    for (int t = 0; t < H; t++) {
#pragma unroll
        for (int j = 0; j < NV; j++) {
            double e = raw[((size_t)k * H + t) * NV + j];
            if (filter) {
                e = __dmul_rn(e, fc.sigma[j]);
                if (t >= 2) e = __dadd_rn(__dadd_rn(__dmul_rn(fc.b0, e), __dmul_rn(fc.b1, e1[j])), __dmul_rn(fc.b2, e2[j]));
            }
            e2[j] = e1[j];
            e1[j] = e;
            eps[((size_t)t * NV + j) * K + k] = T(e);
        }
    }
}

// one lane per (sample k, action dimension j): counter = (global k, j, block c, 0), block c holds the
// draws of steps 4c .. 4c+3.  With seed_from_state the stream key is seed + the planner's current
// timestep (the reference re-seeds every planning step with seed + t, trajopt/algos/mppi.py:130-132)
template <class T, int NV>
__device__ __forceinline__ void noise_philox_block(int block, int pl, int K, int H, int k_offset, uint32_t seed_lo, uint32_t seed_hi,
                                                   int step, int seed_from_state, const FilterCoefs& fc, T* __restrict__ eps) {
    const int idx = block * 256 + threadIdx.x;
    if (idx >= K * NV) return;
    const int j = idx / K, k = idx - j * K;
    eps += (size_t)pl * H * NV * K;                   // planner of a batched handle: its own counter plane
    if (seed_from_state) {
        const uint64_t s = (((uint64_t)seed_hi << 32) | seed_lo) + (uint64_t)(int64_t)step;
        seed_lo = (uint32_t)s;
        seed_hi = (uint32_t)(s >> 32);
    }
    const double sigma = fc.sigma[j];
    double e1 = 0.0, e2 = 0.0;
    for (int c = 0; 4 * c < H; c++) {
        uint32_t ctr[4] = {(uint32_t)(k_offset + k), (uint32_t)j, (uint32_t)c, (uint32_t)pl};
        philox4x32_10(ctr, seed_lo, seed_hi);
        float z[4];
        box_muller(ctr[0], ctr[1], z[0], z[1]);
        box_muller(ctr[2], ctr[3], z[2], z[3]);
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int t = 4 * c + u;
            if (t < H) {
                double e = (double)z[u] * sigma;
                if (t >= 2) e = fc.b0 * e + fc.b1 * e1 + fc.b2 * e2;
                e2 = e1;
                e1 = e;
                eps[((size_t)t * NV + j) * K + k] = T(e);
            }
        }
    }
}
template <class T, int NV>
__global__ void __launch_bounds__(256) noise_philox_kernel(int K, int H, int k_offset, uint32_t seed_lo, uint32_t seed_hi,
                                                           const DevState* __restrict__ st, int seed_from_state,
                                                           FilterCoefs fc, T* __restrict__ eps) {
    noise_philox_block<T, NV>(blockIdx.x, blockIdx.y, K, H, k_offset, seed_lo, seed_hi, st[blockIdx.y].timestep, seed_from_state, fc, eps);
}

// ------------------------------------------------------------------------------------------
// rollout
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ bool nonfinite(float x) { return (__float_as_uint(x) & 0x7f800000u) == 0x7f800000u; }
__device__ __forceinline__ bool nonfinite(double x) {
    return ((unsigned long long)__double_as_longlong(x) & 0x7ff0000000000000ull) == 0x7ff0000000000000ull;
}

// LEAN: the planning configuration (dense reward, noise-relative samples, nothing dumped, no reference, no
// critic) with those choices fixed at compile time: a lone wave pays about ten vector instructions' time for
// every taken branch, and the general kernel spends eight per env step on options that are off.
template <class T, class MD, bool LEAN>
__global__ void __launch_bounds__(64)
rollout_kernel(const DevState* __restrict__ st, const double* __restrict__ base, const T* __restrict__ eps,
               const double* __restrict__ ref, int K, int H, int reward_type_, double gamma, int explicit_act_,
               T* __restrict__ R, T* __restrict__ obs_, T* __restrict__ rew_, int* __restrict__ status,
               T* __restrict__ fobs_, int* __restrict__ next_step) {
    constexpr int NV = MD::NV, NQ = MD::NQ;
    const int reward_type = LEAN ? int(REWARD_DENSE) : reward_type_;
    const int explicit_act = LEAN ? 0 : explicit_act_;
    T* __restrict__ obs = LEAN ? nullptr : obs_;
    T* __restrict__ rew = LEAN ? nullptr : rew_;
    T* __restrict__ fobs = LEAN ? nullptr : fobs_;
    constexpr int OD = NQ + NV + 6;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    // index of the planning step after this one, for a noise generator that runs while the state advances
    if (k == 0) next_step[blockIdx.y] = st[blockIdx.y].timestep + 1;
    {   // planner blockIdx.y of a batched handle
        const size_t pl = blockIdx.y;
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
    const T gm = T(gamma);
    T ret = T(0), g = T(1);
    typename Step<T, MD>::Active act_rows;
    act_rows.reset();
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
        int ms = 0;
        const T r = Env<T, MD>::step(q, v, a, rp, rr, hand, act_rows, ms);
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
    if (nonfinite(ret)) atomicOr(status, 1);
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
template <class MD>
__device__ __forceinline__ void advance_block(int pl, DevState* __restrict__ st, double* __restrict__ base,
                                              const double* __restrict__ ref, int reward_type, int H, int tail_mode,
                                              const double* __restrict__ mean, double* __restrict__ log, int log_cap,
                                              int* __restrict__ log_count) {
    constexpr int NV = MD::NV, NQ = MD::NQ;
    constexpr int ROW = NQ + 2 * NV + 7;
    constexpr int PER = 16;                       // H * NV <= 256 * PER
    const int tid = threadIdx.x;
    double keep[PER];
    const int HM = H * NV;
    st += pl; base += (size_t)pl * HM;                           // one block per planner
    log += (size_t)pl * log_cap * ROW; log_count += pl;
#pragma unroll
    for (int u = 0; u < PER; u++) {
        const int i = tid + 256 * u;
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
        static_for<NQ>([&](auto Ic) { row[Ic] = q[Ic]; st->q[Ic] = q[Ic]; });
        static_for<NV>([&](auto Ic) { row[NQ + Ic] = v[Ic]; row[NQ + NV + 7 + Ic] = a[Ic]; st->v[Ic] = v[Ic]; });
        row[NQ + NV + 0] = hand.v[0]; row[NQ + NV + 1] = hand.v[1]; row[NQ + NV + 2] = hand.v[2];
        row[NQ + NV + 3] = rp.target[0]; row[NQ + NV + 4] = rp.target[1]; row[NQ + NV + 5] = rp.target[2];
        row[NQ + NV + 6] = r;
        st->timestep = t0 + 1;
        *log_count = n + 1;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < PER; u++) {
        const int i = tid + 256 * u;
        if (i < HM) base[i] = keep[u];
    }
}
template <class MD>
__global__ void __launch_bounds__(256) advance_kernel(DevState* __restrict__ st, double* __restrict__ base,
                                                      const double* __restrict__ ref, int reward_type, int H, int tail_mode,
                                                      const double* __restrict__ mean, double* __restrict__ log, int log_cap,
                                                      int* __restrict__ log_count) {
    advance_block<MD>(blockIdx.x, st, base, ref, reward_type, H, tail_mode, mean, log, log_cap, log_count);
}
// advance_kernel and, in the blocks behind the first of every planner, the Philox noise of the NEXT planning step
// (stream key seed + next_step[planner], published by the rollouts of the current step): the noise does not depend
// on the new state, so it is drawn on otherwise idle CUs while one lane steps the real environment.
template <class T, class MD>
__global__ void __launch_bounds__(256) advance_noise_kernel(DevState* __restrict__ st, double* __restrict__ base,
                                                            const double* __restrict__ ref, int reward_type, int H, int tail_mode,
                                                            const double* __restrict__ mean, double* __restrict__ log, int log_cap,
                                                            int* __restrict__ log_count, int K, int k_offset, uint32_t seed_lo,
                                                            uint32_t seed_hi, const int* __restrict__ next_step, FilterCoefs fc,
                                                            T* __restrict__ eps) {
    if (blockIdx.x == 0) advance_block<MD>(blockIdx.y, st, base, ref, reward_type, H, tail_mode, mean, log, log_cap, log_count);
    else noise_philox_block<T, MD::NV>(blockIdx.x - 1, blockIdx.y, K, H, k_offset, seed_lo, seed_hi, next_step[blockIdx.y], 1, fc, eps);
}

// ------------------------------------------------------------------------------------------
// terminal-value critic: tanh MLP in -> 128 -> 128 -> 1 on the final observation of every sample
// (trajopt/models/critic_nets.py:47-60; input slices trajopt/algos/mppi.py:138-154), fp32 like the
// reference's torch module.  One lane per sample; weights are wave-uniform (scalar loads).
// R[k] += gamma^(H-1) * V(x_k): the rollout left the last reward out (trajopt/algos/mppi.py:97-101).
// ------------------------------------------------------------------------------------------
constexpr int CRITIC_INNER = 128;
constexpr int CRITIC_IN = 17;          // widest input (qpos, qvel, goal); narrower inputs are zero padded

template <class T, int NV>
__global__ void __launch_bounds__(64) critic_kernel(const T* __restrict__ fobs, int K, int dim, const float* __restrict__ W1,
                                                    const float* __restrict__ b1, const float* __restrict__ W2,
                                                    const float* __restrict__ b2, const float* __restrict__ W3,
                                                    double g_last, T* __restrict__ R) {
    constexpr int OD = 2 * NV + 6;
    const int k = blockIdx.x * 64 + threadIdx.x;
    if (k >= K) return;
    fobs += (size_t)blockIdx.y * K * OD;
    R += (size_t)blockIdx.y * K;
    const T* o = fobs + (size_t)k * OD;
    float x[CRITIC_IN];
#pragma unroll
    for (int i = 0; i < CRITIC_IN; i++) x[i] = 0.f;
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
    float h1[CRITIC_INNER];
#pragma unroll
    for (int j = 0; j < CRITIC_INNER; j++) {
        float a = b1[j];
#pragma unroll
        for (int i = 0; i < CRITIC_IN; i++) a = fmaf(W1[j * CRITIC_IN + i], x[i], a);
        h1[j] = tanhf(a);
    }
    float out = W3[CRITIC_INNER];          // b3 follows W3 in the packed weight block
    for (int j = 0; j < CRITIC_INNER; j++) {
        float a = b2[j];
#pragma unroll
        for (int i = 0; i < CRITIC_INNER; i++) a = fmaf(W2[j * CRITIC_INNER + i], h1[i], a);
        out = fmaf(W3[j], tanhf(a), out);
    }
    R[k] = R[k] + T(g_last) * T(out);
}

// ------------------------------------------------------------------------------------------
// update kernels (fp64 accumulation, fixed reduction order -> bitwise reproducible)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void argmax_combine(double& v, int& i, double ov, int oi) {
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
}
__device__ __forceinline__ double shfl_xor_d(double x, int m) { return __shfl_xor(x, m, 64); }

// Update in one launch (write_base = 1), or one shard's statistics (Rmax, arg-max, sum S, sum S eps) for the
// exchange between ranks (write_base = 0: act_sequence is left to the merge).  A block owns ROWS consecutive (step, action dimension) rows of one planner:
// it finds Rmax / first arg-max itself (every block gets the same values, in the same fixed order), evaluates
// every weight S_k = exp(kappa (R_k - Rmax)) once and accumulates sum S and its ROWS weighted sums, then writes
// its entries of act_sequence.  ROWS = NV (one block per step) when there are many planners to fill the GPU,
// ROWS = 1 (one block per entry) for a single planner.  Any block size that is a multiple of 64 up to 1024.
template <class T, int NV, int ROWS>
__global__ void __launch_bounds__(1024) update_rows_kernel(const T* __restrict__ R, const T* __restrict__ eps, int K, double kappa,
                                                           int k_offset, int HM, int write_base, double* __restrict__ base,
                                                           double* __restrict__ part, double* __restrict__ result) {
    __shared__ double s_v[16][ROWS + 1];
    __shared__ int s_i[16];
    const int row0 = blockIdx.x * ROWS, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nthreads = blockDim.x, nwaves = nthreads >> 6;
    R += (size_t)blockIdx.y * K; eps += ((size_t)blockIdx.y * HM + row0) * K;
    base += (size_t)blockIdx.y * HM + row0; part += (size_t)blockIdx.y * (3 + HM); result += 2 * blockIdx.y;
    double bv = -DBL_MAX;
    int bi = INT_MAX;
    for (int k = tid; k < K; k += nthreads) argmax_combine(bv, bi, (double)R[k], k);
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
    double acc[ROWS + 1];            // [ROWS]: sum of the weights
#pragma unroll
    for (int j = 0; j <= ROWS; j++) acc[j] = 0.0;
    for (int k = tid; k < K; k += nthreads) {
        const double sk = exp(kappa * ((double)R[k] - bv));
#pragma unroll
        for (int j = 0; j < ROWS; j++) acc[j] += sk * (double)eps[(size_t)j * K + k];
        acc[ROWS] += sk;
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
    if (tid < ROWS) {
        const double n = s_v[0][tid];
        if (write_base) base[tid] = (base[tid] * d + n) / (d + 1e-6);
        part[3 + row0 + tid] = n;
    }
    if (blockIdx.x == 0 && tid == 0) {
        part[0] = bv; part[1] = (double)(k_offset + bi); part[2] = d;
        if (write_base) { result[0] = bv; result[1] = (double)(k_offset + bi); }
    }
}

// act_sequence = sum_k S_k (base + eps_k) / (sum S + 1e-6) from nranks shard statistics, rank order
__global__ void merge_kernel(const double* __restrict__ gathered, int nranks, int HM, double kappa,
                             double* __restrict__ base, double* __restrict__ result) {
    const int stride = (3 + HM) * gridDim.y;        // a rank's row holds its gridDim.y planners back to back
    gathered += (size_t)blockIdx.y * (3 + HM);
    base += (size_t)blockIdx.y * HM;
    result += 2 * blockIdx.y;
    double gmax = -DBL_MAX, gidx = 0.0;
    for (int r = 0; r < nranks; r++) {
        const double m = gathered[(size_t)r * stride];
        if (m > gmax) { gmax = m; gidx = gathered[(size_t)r * stride + 1]; }
    }
    double den = 0.0;
    for (int r = 0; r < nranks; r++) den += exp(kappa * (gathered[(size_t)r * stride] - gmax)) * gathered[(size_t)r * stride + 2];
    for (int hm = blockIdx.x * blockDim.x + threadIdx.x; hm < HM; hm += gridDim.x * blockDim.x) {
        double num = 0.0;
        for (int r = 0; r < nranks; r++)
            num += exp(kappa * (gathered[(size_t)r * stride] - gmax)) * gathered[(size_t)r * stride + 3 + hm];
        base[hm] = (base[hm] * den + num) / (den + 1e-6);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { result[0] = gmax; result[1] = gidx; }
}

// tmp[t] = src[t+1], tail = previous last row (repeat) or mean
__global__ void shift_kernel(const double* __restrict__ src, int H, int M, int tail_mode, const double* __restrict__ mean,
                             double* __restrict__ dst) {
    src += (size_t)blockIdx.y * H * M;
    dst += (size_t)blockIdx.y * H * M;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < H * M; i += gridDim.x * blockDim.x) {
        const int t = i / M, j = i - t * M;
        double v;
        if (t < H - 1) v = src[(t + 1) * M + j];
        else v = (tail_mode == MPPI_TAIL_REPEAT) ? src[(H - 1) * M + j] : mean[j];
        dst[i] = v;
    }
}

// `MPPI.update` for caller-supplied (actions, returns): block hm
__global__ void __launch_bounds__(256) update_direct_kernel(const double* __restrict__ act, const double* __restrict__ R,
                                                            int K, int HM, double kappa, double* __restrict__ out) {
    __shared__ double s_v[4];
    const int hm = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double bv = -DBL_MAX;
    for (int k = tid; k < K; k += 256) bv = R[k] > bv ? R[k] : bv;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { const double o = shfl_xor_d(bv, m); bv = o > bv ? o : bv; }
    if (lane == 0) s_v[wave] = bv;
    __syncthreads();
    bv = s_v[0];
    for (int w = 1; w < 4; w++) bv = s_v[w] > bv ? s_v[w] : bv;
    __syncthreads();
    double num = 0.0, den = 0.0;
    for (int k = tid; k < K; k += 256) {
        const double s = exp(kappa * (R[k] - bv));
        num += s * act[(size_t)k * HM + hm];
        den += s;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { num += shfl_xor_d(num, m); den += shfl_xor_d(den, m); }
    if (lane == 0) s_v[wave] = num;
    __syncthreads();
    const double n = ((s_v[0] + s_v[1]) + s_v[2]) + s_v[3];
    __syncthreads();
    if (lane == 0) s_v[wave] = den;
    __syncthreads();
    if (tid == 0) out[hm] = n / ((((s_v[0] + s_v[1]) + s_v[2]) + s_v[3]) + 1e-6);
}

// ------------------------------------------------------------------------------------------
// per-model dispatch
// ------------------------------------------------------------------------------------------
struct RolloutArgs {
    const DevState* st; const double* base; const void* eps; const double* ref;
    int K, H, P, reward_type; double gamma; int explicit_act;
    bool lean;      // dense reward, noise-relative samples, no dumps, no reference, no critic
    void* R; void* obs; void* rew; int* status; void* fobs; int* next_step;
};

struct ModelOps {
    const char* name;
    int nv, nq;
    bool (*matches)(const mppi_model_desc*, std::string&);
    hipError_t (*rollout)(int dtype, const RolloutArgs&, hipStream_t);
    hipError_t (*noise_upload)(int dtype, const double* raw, int K, int H, int P, const FilterCoefs&, int filter, void* eps, hipStream_t);
    hipError_t (*noise_philox)(int dtype, int K, int H, int P, int k_offset, uint64_t seed, const DevState* st, int seed_from_state,
                               const FilterCoefs&, void* eps, hipStream_t);
    hipError_t (*advance)(DevState* st, double* base, const double* ref, int reward_type, int H, int P, int tail_mode,
                          const double* mean, double* log, int log_cap, int* log_count, hipStream_t);
    hipError_t (*env_step)(const double* in, int reward_type, int has_ref, int forward_only, double* out, hipStream_t);
    hipError_t (*critic)(int dtype, const void* fobs, int K, int P, int dim, const float* w, double g_last, void* R, hipStream_t);
    hipError_t (*advance_noise)(int dtype, DevState* st, double* base, const double* ref, int reward_type, int H, int P, int tail_mode,
                                const double* mean, double* log, int log_cap, int* log_count, int K, int k_offset, uint64_t seed,
                                const int* next_step, const FilterCoefs& fc, void* eps, hipStream_t);
    hipError_t (*update_rows)(int dtype, const void* R, const void* eps, int K, int H, int P, double kappa, int k_offset, int write_base, double* base,
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
    const dim3 grid((a.K + 63) / 64, a.P), block(64);      // one wave per block: waves spread over CUs and XCDs
    if (dtype == MPPI_F32 && a.lean)
        hipLaunchKernelGGL((rollout_kernel<float, MD, true>), grid, block, 0, s, a.st, a.base, (const float*)a.eps, a.ref, a.K,
                           a.H, a.reward_type, a.gamma, a.explicit_act, (float*)a.R, (float*)a.obs, (float*)a.rew, a.status, (float*)a.fobs, a.next_step);
    else if (dtype == MPPI_F32)
        hipLaunchKernelGGL((rollout_kernel<float, MD, false>), grid, block, 0, s, a.st, a.base, (const float*)a.eps, a.ref, a.K,
                           a.H, a.reward_type, a.gamma, a.explicit_act, (float*)a.R, (float*)a.obs, (float*)a.rew, a.status, (float*)a.fobs, a.next_step);
    else
        hipLaunchKernelGGL((rollout_kernel<double, MD, false>), grid, block, 0, s, a.st, a.base, (const double*)a.eps, a.ref, a.K,
                           a.H, a.reward_type, a.gamma, a.explicit_act, (double*)a.R, (double*)a.obs, (double*)a.rew, a.status, (double*)a.fobs, a.next_step);
    return hipGetLastError();
}
template <class MD>
static hipError_t launch_noise_upload(int dtype, const double* raw, int K, int H, int P, const FilterCoefs& fc, int filter, void* eps,
                                      hipStream_t s) {
    const dim3 grid((K + 63) / 64, P), block(64);
    if (dtype == MPPI_F32)
        hipLaunchKernelGGL((noise_upload_kernel<float, MD::NV>), grid, block, 0, s, raw, K, H, fc, filter, (float*)eps);
    else
        hipLaunchKernelGGL((noise_upload_kernel<double, MD::NV>), grid, block, 0, s, raw, K, H, fc, filter, (double*)eps);
    return hipGetLastError();
}
template <class MD>
static hipError_t launch_noise_philox(int dtype, int K, int H, int P, int k_offset, uint64_t seed, const DevState* st,
                                      int seed_from_state, const FilterCoefs& fc, void* eps, hipStream_t s) {
    const dim3 grid((K * MD::NV + 255) / 256, P), block(256);
    const uint32_t lo = (uint32_t)seed, hi = (uint32_t)(seed >> 32);
    if (dtype == MPPI_F32)
        hipLaunchKernelGGL((noise_philox_kernel<float, MD::NV>), grid, block, 0, s, K, H, k_offset, lo, hi, st, seed_from_state, fc, (float*)eps);
    else
        hipLaunchKernelGGL((noise_philox_kernel<double, MD::NV>), grid, block, 0, s, K, H, k_offset, lo, hi, st, seed_from_state, fc, (double*)eps);
    return hipGetLastError();
}
template <class MD>
static hipError_t launch_advance(DevState* st, double* base, const double* ref, int reward_type, int H, int P, int tail_mode,
                                 const double* mean, double* log, int log_cap, int* log_count, hipStream_t s) {
    hipLaunchKernelGGL((advance_kernel<MD>), dim3(P), dim3(256), 0, s, st, base, ref, reward_type, H, tail_mode, mean, log, log_cap, log_count);
    return hipGetLastError();
}
template <class MD>
static hipError_t launch_advance_noise(int dtype, DevState* st, double* base, const double* ref, int reward_type, int H, int P,
                                       int tail_mode, const double* mean, double* log, int log_cap, int* log_count, int K,
                                       int k_offset, uint64_t seed, const int* next_step, const FilterCoefs& fc, void* eps,
                                       hipStream_t s) {
    const dim3 grid(1 + (K * MD::NV + 255) / 256, P), block(256);
    const uint32_t lo = (uint32_t)seed, hi = (uint32_t)(seed >> 32);
    if (dtype == MPPI_F32)
        hipLaunchKernelGGL((advance_noise_kernel<float, MD>), grid, block, 0, s, st, base, ref, reward_type, H, tail_mode, mean, log,
                           log_cap, log_count, K, k_offset, lo, hi, next_step, fc, (float*)eps);
    else
        hipLaunchKernelGGL((advance_noise_kernel<double, MD>), grid, block, 0, s, st, base, ref, reward_type, H, tail_mode, mean, log,
                           log_cap, log_count, K, k_offset, lo, hi, next_step, fc, (double*)eps);
    return hipGetLastError();
}
template <class MD>
static hipError_t launch_env_step(const double* in, int reward_type, int has_ref, int forward_only, double* out, hipStream_t s) {
    hipLaunchKernelGGL((env_step_kernel<MD>), dim3(1), dim3(64), 0, s, in, reward_type, has_ref, forward_only, out);
    return hipGetLastError();
}
// w: [W1 (128 x 17, zero padded) | b1 (128) | W2 (128 x 128) | b2 (128) | W3 (128) | b3 (1)]
constexpr size_t CRITIC_W_FLOATS = CRITIC_INNER * CRITIC_IN + CRITIC_INNER + CRITIC_INNER * CRITIC_INNER + CRITIC_INNER + CRITIC_INNER + 1;
template <class MD>
static hipError_t launch_critic(int dtype, const void* fobs, int K, int P, int dim, const float* w, double g_last, void* R, hipStream_t s) {
    const float* W1 = w;
    const float* b1 = W1 + CRITIC_INNER * CRITIC_IN;
    const float* W2 = b1 + CRITIC_INNER;
    const float* b2 = W2 + CRITIC_INNER * CRITIC_INNER;
    const float* W3 = b2 + CRITIC_INNER;
    const dim3 grid((K + 63) / 64, P), block(64);
    if (dtype == MPPI_F32)
        hipLaunchKernelGGL((critic_kernel<float, MD::NV>), grid, block, 0, s, (const float*)fobs, K, dim, W1, b1, W2, b2, W3, g_last, (float*)R);
    else
        hipLaunchKernelGGL((critic_kernel<double, MD::NV>), grid, block, 0, s, (const double*)fobs, K, dim, W1, b1, W2, b2, W3, g_last, (double*)R);
    return hipGetLastError();
}
template <class MD>
static hipError_t launch_update_rows(int dtype, const void* R, const void* eps, int K, int H, int P, double kappa, int k_offset,
                                     int write_base, double* base, double* part, double* result, hipStream_t s) {
    constexpr int NV = MD::NV;
    const int HM = H * NV;
    const dim3 block(K > 1024 ? 1024 : 256);
    // enough blocks to spread over the GPU: one per step once there are many planners, else one per entry
    if ((long)H * P >= 1024) {
        if (dtype == MPPI_F32)
            hipLaunchKernelGGL((update_rows_kernel<float, NV, NV>), dim3(H, P), block, 0, s, (const float*)R, (const float*)eps, K, kappa, k_offset, HM, write_base, base, part, result);
        else
            hipLaunchKernelGGL((update_rows_kernel<double, NV, NV>), dim3(H, P), block, 0, s, (const double*)R, (const double*)eps, K, kappa, k_offset, HM, write_base, base, part, result);
    } else {
        if (dtype == MPPI_F32)
            hipLaunchKernelGGL((update_rows_kernel<float, NV, 1>), dim3(HM, P), block, 0, s, (const float*)R, (const float*)eps, K, kappa, k_offset, HM, write_base, base, part, result);
        else
            hipLaunchKernelGGL((update_rows_kernel<double, NV, 1>), dim3(HM, P), block, 0, s, (const double*)R, (const double*)eps, K, kappa, k_offset, HM, write_base, base, part, result);
    }
    return hipGetLastError();
}
template <class MD> static ModelOps make_ops() {
    return {MD::name, MD::NV, MD::NQ, &model_matches<MD>, &launch_rollout<MD>, &launch_noise_upload<MD>, &launch_noise_philox<MD>,
            &launch_advance<MD>, &launch_env_step<MD>, &launch_critic<MD>, &launch_advance_noise<MD>, &launch_update_rows<MD>};
}
#if defined(MPPI_ONLY_REACHER)      // quick builds for kernel experiments (make quick / make quickasm)
static const ModelOps g_ops[] = {make_ops<Reacher7Dof>()};
#else
static const ModelOps g_ops[] = {make_ops<Reacher7Dof>(), make_ops<Reacher7DofFree>(), make_ops<Tree18Dof>(), make_ops<Tree18DofQuat>(),
                                 make_ops<Tree18DofRk4>()};
#endif

// ------------------------------------------------------------------------------------------
// handles
// ------------------------------------------------------------------------------------------
struct mppi_model {
    const ModelOps* ops;
    mppi_model_desc desc;
    // per-device scratch of the single-step entry points (lazily allocated)
    double* d_io[16];
};

struct mppi_planner {
    const mppi_model* model;
    mppi_planner_cfg cfg;
    int nv, nq, HM, P;         // P = planners batched in this handle (arrays below carry a leading [P])
    size_t esz;            // sizeof(T)
    hipStream_t stream;
    DevState* d_state;
    double* d_base;        // [H][M] act_sequence (fp64 master)
    double* d_base_tmp;
    double* d_mean;
    void* d_eps;           // T [H][M][K]
    double* d_raw;         // staging for host uploads [K][H][M]
    double* d_ref;
    int ref_cap;
    void* d_R;             // T [K]
    double* d_part;        // [3 + HM]
    double* d_result;      // [2] Rmax, argmax (after merge)
    void* d_obs;           // T [K][H][2M+6]
    void* d_rew;           // T [K][H]
    int* d_status;
    void* d_fobs;          // T [K][2M+6] final observation of every sample (critic input)
    float* d_critic;       // packed critic weights (CRITIC_W_FLOATS) or null
    int critic_dim;        // 0: no critic
    double* d_log;         // [LOG_CAP][3M+7] real-env log ring of mppi_advance
    int* d_log_count;
    int* d_next_step;      // [P] index of the planning step after the one being rolled out
    bool noise_ahead;      // eps already holds the Philox noise of the coming planning step ...
    uint64_t noise_ahead_seed;   // ... drawn with this seed
    int explicit_act;
    FilterCoefs fc;
    hipGraphExec_t graph;
    bool graph_valid;
    hipGraphExec_t step_graph;      // whole planning step: noise -> niter x (rollout, update) -> advance
    int step_graph_niter;
    uint64_t step_graph_seed;
    bool profiling;
    std::vector<hipEvent_t>* prof_events;   // start/stop pairs around every rollout launch
    std::vector<DevState>* h_state;         // host mirror of the fields the host sets (alpha, ref_len, start state)
};

static hipStream_t pick(mppi_planner* p, void* s) { return s ? (hipStream_t)s : p->stream; }

extern "C" {

int mppi_version(void) { return MPPI_ABI_VERSION; }
const char* mppi_last_error(void) { return g_err.c_str(); }

int mppi_model_create(const mppi_model_desc* desc, mppi_model** out) {
    if (!desc || !out) return fail(-1, "null argument");
    std::string why = "unknown model name";
    for (const ModelOps& o : g_ops) {
        if (std::strncmp(desc->name, o.name, sizeof(desc->name)) != 0) continue;
        if (!o.matches(desc, why)) return fail(-3, std::string("model '") + o.name + "': " + why);
        mppi_model* m = new mppi_model();
        m->ops = &o;
        m->desc = *desc;
        for (auto& x : m->d_io) x = nullptr;
        *out = m;
        return 0;
    }
    return fail(-3, std::string("no kernel compiled for model '") + std::string(desc->name, strnlen(desc->name, 64)) + "'");
}

void mppi_model_destroy(mppi_model* m) {
    if (!m) return;
    for (int d = 0; d < 16; d++)
        if (m->d_io[d]) { (void)hipSetDevice(d); (void)hipFree(m->d_io[d]); }
    delete m;
}

int mppi_planner_create(const mppi_model* m, const mppi_planner_cfg* cfg, mppi_planner** out) {
    if (!m || !cfg || !out) return fail(-1, "null argument");
    if (cfg->K < 1 || cfg->H < 1) return fail(-1, "K and H must be positive");
    if (cfg->dtype != MPPI_F32 && cfg->dtype != MPPI_F64) return fail(-1, "dtype must be MPPI_F32 or MPPI_F64");
    HIP_OK(hipSetDevice(cfg->device));
    mppi_planner* p = new mppi_planner();
    std::memset(p, 0, sizeof(*p));
    p->model = m;
    p->cfg = *cfg;
    p->nv = m->ops->nv;
    p->nq = m->ops->nq;
    p->HM = cfg->H * p->nv;
    p->P = cfg->P > 1 ? cfg->P : 1;
    p->cfg.P = p->P;
    p->esz = cfg->dtype == MPPI_F32 ? 4 : 8;
    const size_t P = p->P;
    const size_t K = cfg->K * P, HM = p->HM * P, OD = p->nq + p->nv + 6;   // sizes over all planners
    HIP_OK(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
    HIP_OK(hipMalloc(&p->d_state, sizeof(DevState) * P));
    HIP_OK(hipMemset(p->d_state, 0, sizeof(DevState) * P));
    HIP_OK(hipMalloc(&p->d_base, HM * 8));
    HIP_OK(hipMalloc(&p->d_base_tmp, HM * 8));
    HIP_OK(hipMalloc(&p->d_mean, MPPI_MAXV * 8));
    HIP_OK(hipMemcpy(p->d_mean, cfg->mean, MPPI_MAXV * 8, hipMemcpyHostToDevice));
    HIP_OK(hipMalloc(&p->d_eps, (size_t)p->HM * K * p->esz));
    HIP_OK(hipMemset(p->d_eps, 0, (size_t)p->HM * K * p->esz));
    HIP_OK(hipMalloc(&p->d_R, K * p->esz));
    HIP_OK(hipMemset(p->d_R, 0, K * p->esz));
    HIP_OK(hipMalloc(&p->d_part, (3 * P + HM) * 8));
    HIP_OK(hipMemset(p->d_part, 0, (3 * P + HM) * 8));
    HIP_OK(hipMalloc(&p->d_result, 2 * 8 * P));
    HIP_OK(hipMemset(p->d_result, 0, 2 * 8 * P));
    HIP_OK(hipMalloc(&p->d_status, 4));
    HIP_OK(hipMemset(p->d_status, 0, 4));
    HIP_OK(hipMalloc(&p->d_log, P * MPPI_LOG_CAP * (p->nq + 2 * p->nv + 7) * 8));
    HIP_OK(hipMalloc(&p->d_log_count, 4 * P));
    HIP_OK(hipMemset(p->d_log_count, 0, 4 * P));
    HIP_OK(hipMalloc(&p->d_next_step, 4 * P));
    HIP_OK(hipMemset(p->d_next_step, 0, 4 * P));
    if (cfg->dump) {
        HIP_OK(hipMalloc(&p->d_obs, K * cfg->H * OD * p->esz));
        HIP_OK(hipMalloc(&p->d_rew, K * cfg->H * p->esz));
    }
    // act_sequence = ones((H, m)) * mean
    std::vector<double> base(HM);
    for (size_t t = 0; t < (size_t)cfg->H * P; t++)
        for (int j = 0; j < p->nv; j++) base[t * p->nv + j] = cfg->mean[j];
    HIP_OK(hipMemcpy(p->d_base, base.data(), HM * 8, hipMemcpyHostToDevice));
    for (int j = 0; j < MPPI_MAXV; j++) p->fc.sigma[j] = cfg->sigma[j];
    p->fc.b0 = cfg->beta[0]; p->fc.b1 = cfg->beta[1]; p->fc.b2 = cfg->beta[2];
    p->prof_events = new std::vector<hipEvent_t>();
    p->h_state = new std::vector<DevState>(P);
    std::memset(p->h_state->data(), 0, sizeof(DevState) * P);
    *out = p;
    return 0;
}

void mppi_planner_destroy(mppi_planner* p) {
    if (!p) return;
    (void)hipSetDevice(p->cfg.device);
    (void)hipStreamSynchronize(p->stream);
    for (hipEvent_t e : *p->prof_events) (void)hipEventDestroy(e);
    delete p->prof_events;
    delete p->h_state;
    if (p->graph) (void)hipGraphExecDestroy(p->graph);
    if (p->step_graph) (void)hipGraphExecDestroy(p->step_graph);
    void* bufs[] = {p->d_state, p->d_base, p->d_base_tmp, p->d_mean, p->d_eps, p->d_raw, p->d_ref, p->d_R,
                    p->d_part, p->d_result, p->d_obs, p->d_rew, p->d_status, p->d_log, p->d_log_count, p->d_fobs, p->d_critic, p->d_next_step};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    (void)hipStreamDestroy(p->stream);
    delete p;
}

int mppi_set_start_state(mppi_planner* p, const double* qpos, const double* qvel, const double* target3,
                         const int32_t* timestep) {
    if (!p || !qpos || !qvel || !target3 || !timestep) return fail(-1, "null argument");
    HIP_OK(hipSetDevice(p->cfg.device));
    // alpha / ref_len come from the host mirror; q, v, target, timestep are replaced
    std::vector<DevState>& s = *p->h_state;
    for (int b = 0; b < p->P; b++) {
        for (int i = 0; i < p->nq; i++) s[b].q[i] = qpos[b * p->nq + i];
        for (int i = 0; i < p->nv; i++) s[b].v[i] = qvel[b * p->nv + i];
        for (int c = 0; c < 3; c++) s[b].target[c] = target3[3 * b + c];
        s[b].timestep = timestep[b];
    }
    HIP_OK(hipMemcpyAsync(p->d_state, s.data(), sizeof(DevState) * p->P, hipMemcpyHostToDevice, p->stream));
    HIP_OK(hipStreamSynchronize(p->stream));
    p->noise_ahead = false;          // the stream key follows the timestep
    return 0;
}

int mppi_set_base_actions(mppi_planner* p, const double* act) {
    if (!p || !act) return fail(-1, "null argument");
    HIP_OK(hipSetDevice(p->cfg.device));
    HIP_OK(hipMemcpyAsync(p->d_base, act, (size_t)p->HM * p->P * 8, hipMemcpyHostToDevice, p->stream));
    HIP_OK(hipStreamSynchronize(p->stream));
    return 0;
}

int mppi_set_reference(mppi_planner* p, const double* ref, int32_t T, double alpha) {
    if (!p) return fail(-1, "null argument");
    if (T < 0 || (T > 0 && !ref)) return fail(-1, "bad reference");
    HIP_OK(hipSetDevice(p->cfg.device));
    HIP_OK(hipStreamSynchronize(p->stream));
    if (T > p->ref_cap) {
        if (p->d_ref) HIP_OK(hipFree(p->d_ref));
        HIP_OK(hipMalloc(&p->d_ref, (size_t)T * 3 * 8));
        p->ref_cap = T;
        p->graph_valid = false;
    }
    if (T > 0) HIP_OK(hipMemcpy(p->d_ref, ref, (size_t)T * 3 * 8, hipMemcpyHostToDevice));
    // the device copy of q / v / timestep may be ahead of the mirror (mppi_advance): patch only alpha and ref_len
    std::vector<DevState> s(p->P);
    HIP_OK(hipMemcpy(s.data(), p->d_state, sizeof(DevState) * p->P, hipMemcpyDeviceToHost));
    for (int b = 0; b < p->P; b++) {
        s[b].alpha = alpha; s[b].ref_len = T;
        (*p->h_state)[b].alpha = alpha; (*p->h_state)[b].ref_len = T;
    }
    p->graph_valid = false;       // a captured rollout may be the kernel variant without a reference
    HIP_OK(hipMemcpy(p->d_state, s.data(), sizeof(DevState) * p->P, hipMemcpyHostToDevice));
    return 0;
}

int mppi_set_critic(mppi_planner* p, int32_t dim, int32_t inner, const float* W1, const float* b1, const float* W2,
                    const float* b2, const float* W3, float b3) {
    if (!p) return fail(-1, "null argument");
    HIP_OK(hipSetDevice(p->cfg.device));
    HIP_OK(hipStreamSynchronize(p->stream));
    p->graph_valid = false;
    if (dim == 0) { p->critic_dim = 0; return 0; }
    if (dim != 14 && dim != 15 && dim != 17 && dim != 3 && dim != 6) return fail(-1, "critic input dim must be 14, 15, 17, 3 or 6");
    if (p->nv != 7) return fail(-1, "critic input slices are defined for the 20-dim reacher observation");
    if (inner != CRITIC_INNER) return fail(-1, "only the 128-wide critic has a compiled kernel");
    if (!W1 || !b1 || !W2 || !b2 || !W3) return fail(-1, "null critic weights");
    std::vector<float> w(CRITIC_W_FLOATS, 0.f);
    float* o = w.data();
    for (int j = 0; j < CRITIC_INNER; j++)
        for (int i = 0; i < dim; i++) o[j * CRITIC_IN + i] = W1[j * dim + i];
    o += CRITIC_INNER * CRITIC_IN;
    std::memcpy(o, b1, CRITIC_INNER * 4); o += CRITIC_INNER;
    std::memcpy(o, W2, CRITIC_INNER * CRITIC_INNER * 4); o += CRITIC_INNER * CRITIC_INNER;
    std::memcpy(o, b2, CRITIC_INNER * 4); o += CRITIC_INNER;
    std::memcpy(o, W3, CRITIC_INNER * 4); o += CRITIC_INNER;
    *o = b3;
    if (!p->d_critic) HIP_OK(hipMalloc(&p->d_critic, CRITIC_W_FLOATS * 4));
    if (!p->d_fobs) HIP_OK(hipMalloc(&p->d_fobs, (size_t)p->cfg.K * p->P * (p->nq + p->nv + 6) * p->esz));
    HIP_OK(hipMemcpy(p->d_critic, w.data(), CRITIC_W_FLOATS * 4, hipMemcpyHostToDevice));
    p->critic_dim = dim;
    return 0;
}

int mppi_set_params(mppi_planner* p, double kappa, double gamma, int32_t reward_type) {
    if (!p) return fail(-1, "null argument");
    if (reward_type < 0 || reward_type > 3) return fail(-1, "bad reward_type");
    p->cfg.kappa = kappa; p->cfg.gamma = gamma; p->cfg.reward_type = reward_type;
    p->graph_valid = false;
    return 0;
}

int mppi_upload_noise(mppi_planner* p, const double* raw, void* stream) {
    if (!p || !raw) return fail(-1, "null argument");
    HIP_OK(hipSetDevice(p->cfg.device));
    hipStream_t s = pick(p, stream);
    if (!p->d_raw) HIP_OK(hipMalloc(&p->d_raw, (size_t)p->HM * p->cfg.K * p->P * 8));      // host-upload staging, on first use
    HIP_OK(hipMemcpyAsync(p->d_raw, raw, (size_t)p->HM * p->cfg.K * p->P * 8, hipMemcpyHostToDevice, s));
    HIP_OK(p->model->ops->noise_upload(p->cfg.dtype, p->d_raw, p->cfg.K, p->cfg.H, p->P, p->fc, 1, p->d_eps, s));
    HIP_OK(hipStreamSynchronize(s));     // the host buffer may be reused by the caller
    p->explicit_act = 0;
    p->noise_ahead = false;
    return 0;
}

static int generate_noise(mppi_planner* p, uint64_t seed, int seed_from_state, void* stream) {
    if (!p) return fail(-1, "null argument");
    HIP_OK(hipSetDevice(p->cfg.device));
    HIP_OK(p->model->ops->noise_philox(p->cfg.dtype, p->cfg.K, p->cfg.H, p->P, p->cfg.k_offset, seed, p->d_state, seed_from_state,
                                       p->fc, p->d_eps, pick(p, stream)));
    p->explicit_act = 0;
    p->noise_ahead = false;
    return 0;
}
int mppi_generate_noise(mppi_planner* p, uint64_t seed, void* stream) { return generate_noise(p, seed, 0, stream); }
int mppi_generate_noise_step(mppi_planner* p, uint64_t seed, void* stream) { return generate_noise(p, seed, 1, stream); }

int mppi_advance(mppi_planner* p, void* stream) {
    if (!p) return fail(-1, "null argument");
    if (p->HM > 256 * 16) return fail(-1, "H*M too large for mppi_advance");
    HIP_OK(hipSetDevice(p->cfg.device));
    HIP_OK(p->model->ops->advance(p->d_state, p->d_base, p->d_ref, p->cfg.reward_type, p->cfg.H, p->P, p->cfg.default_act, p->d_mean,
                                  p->d_log, MPPI_LOG_CAP, p->d_log_count, pick(p, stream)));
    return 0;
}

int mppi_advance_noise(mppi_planner* p, uint64_t seed, void* stream) {
    if (!p) return fail(-1, "null argument");
    if (p->HM > 256 * 16) return fail(-1, "H*M too large for mppi_advance_noise");
    HIP_OK(hipSetDevice(p->cfg.device));
    HIP_OK(p->model->ops->advance_noise(p->cfg.dtype, p->d_state, p->d_base, p->d_ref, p->cfg.reward_type, p->cfg.H, p->P,
                                        p->cfg.default_act, p->d_mean, p->d_log, MPPI_LOG_CAP, p->d_log_count, p->cfg.K,
                                        p->cfg.k_offset, seed, p->d_next_step, p->fc, p->d_eps, pick(p, stream)));
    p->explicit_act = 0;
    p->noise_ahead = true;
    p->noise_ahead_seed = seed;
    return 0;
}

int mppi_read_log(mppi_planner* p, int32_t planner, double* rows, int32_t max_rows, int32_t* n_rows, int32_t* total) {
    if (!p || !rows || !n_rows || max_rows < 0 || planner < 0 || planner >= p->P) return fail(-1, "bad argument");
    HIP_OK(hipSetDevice(p->cfg.device));
    HIP_OK(hipStreamSynchronize(p->stream));
    HIP_OK(hipDeviceSynchronize());
    int count = 0;
    HIP_OK(hipMemcpy(&count, p->d_log_count + planner, 4, hipMemcpyDeviceToHost));
    const int row = p->nq + 2 * p->nv + 7;
    const double* ring = p->d_log + (size_t)planner * MPPI_LOG_CAP * row;
    int n = count < MPPI_LOG_CAP ? count : MPPI_LOG_CAP;
    if (n > max_rows) n = max_rows;
    // the n most recent rows, oldest first
    for (int i = 0; i < n; i++) {
        const int src = (count - n + i) % MPPI_LOG_CAP;
        HIP_OK(hipMemcpy(rows + (size_t)i * row, ring + (size_t)src * row, (size_t)row * 8, hipMemcpyDeviceToHost));
    }
    *n_rows = n;
    if (total) *total = count;
    return 0;
}

int mppi_upload_actions(mppi_planner* p, const double* act, void* stream) {
    if (!p || !act) return fail(-1, "null argument");
    HIP_OK(hipSetDevice(p->cfg.device));
    hipStream_t s = pick(p, stream);
    if (!p->d_raw) HIP_OK(hipMalloc(&p->d_raw, (size_t)p->HM * p->cfg.K * p->P * 8));
    HIP_OK(hipMemcpyAsync(p->d_raw, act, (size_t)p->HM * p->cfg.K * p->P * 8, hipMemcpyHostToDevice, s));
    HIP_OK(p->model->ops->noise_upload(p->cfg.dtype, p->d_raw, p->cfg.K, p->cfg.H, p->P, p->fc, 0, p->d_eps, s));
    HIP_OK(hipStreamSynchronize(s));
    p->explicit_act = 1;
    p->noise_ahead = false;
    return 0;
}

static int launch_rollout_only(mppi_planner* p, hipStream_t s) {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (p->profiling) {
        HIP_OK(hipEventCreate(&e0));
        HIP_OK(hipEventCreate(&e1));
        HIP_OK(hipEventRecord(e0, s));
    }
    RolloutArgs a;
    a.st = p->d_state; a.base = p->d_base; a.eps = p->d_eps; a.ref = p->d_ref;
    a.K = p->cfg.K; a.H = p->cfg.H; a.P = p->P; a.reward_type = p->cfg.reward_type; a.gamma = p->cfg.gamma;
    a.explicit_act = p->explicit_act;
    a.R = p->d_R; a.obs = p->d_obs; a.rew = p->d_rew; a.status = p->d_status;
    a.fobs = p->critic_dim ? p->d_fobs : nullptr;
    a.next_step = p->d_next_step;
    bool has_ref = false;
    for (int b = 0; b < p->P; b++) has_ref |= (*p->h_state)[b].ref_len > 0;
    a.lean = a.reward_type == REWARD_DENSE && !a.explicit_act && !a.obs && !a.rew && !a.fobs && !has_ref;
    HIP_OK(p->model->ops->rollout(p->cfg.dtype, a, s));
    if (p->profiling) {
        HIP_OK(hipEventRecord(e1, s));
        p->prof_events->push_back(e0);
        p->prof_events->push_back(e1);
    }
    if (p->critic_dim)
        HIP_OK(p->model->ops->critic(p->cfg.dtype, p->d_fobs, p->cfg.K, p->P, p->critic_dim, p->d_critic,
                                     std::pow(p->cfg.gamma, (double)(p->cfg.H - 1)), p->d_R, s));
    return 0;
}

static int launch_partials(mppi_planner* p, hipStream_t s) {
    HIP_OK(p->model->ops->update_rows(p->cfg.dtype, p->d_R, p->d_eps, p->cfg.K, p->cfg.H, p->P, p->cfg.kappa, p->cfg.k_offset, 0,
                                      p->d_base, p->d_part, p->d_result, s));
    return 0;
}

static int launch_update_fused(mppi_planner* p, hipStream_t s) {
    HIP_OK(p->model->ops->update_rows(p->cfg.dtype, p->d_R, p->d_eps, p->cfg.K, p->cfg.H, p->P, p->cfg.kappa, p->cfg.k_offset, 1,
                                      p->d_base, p->d_part, p->d_result, s));
    return 0;
}

static int launch_merge(mppi_planner* p, const double* gathered, int nranks, hipStream_t s) {
    hipLaunchKernelGGL(merge_kernel, dim3((p->HM + 255) / 256, p->P), dim3(256), 0, s, gathered, nranks, p->HM, p->cfg.kappa,
                       p->d_base, p->d_result);
    HIP_OK(hipGetLastError());
    return 0;
}

int mppi_rollout(mppi_planner* p, void* stream) {
    if (!p) return fail(-1, "null argument");
    HIP_OK(hipSetDevice(p->cfg.device));
    return launch_rollout_only(p, pick(p, stream));
}

int mppi_local_partials(mppi_planner* p, void* stream) {
    if (!p) return fail(-1, "null argument");
    if (p->explicit_act) return fail(-4, "update needs noise-relative samples; call mppi_upload_noise or mppi_generate_noise");
    HIP_OK(hipSetDevice(p->cfg.device));
    hipStream_t s = pick(p, stream);
    int rc = launch_rollout_only(p, s);
    if (rc) return rc;
    return launch_partials(p, s);
}

int mppi_partials_ptr(mppi_planner* p, void** ptr, size_t* bytes) {
    if (!p || !ptr || !bytes) return fail(-1, "null argument");
    *ptr = p->d_part;
    *bytes = (size_t)(3 + p->HM) * p->P * 8;
    return 0;
}

int mppi_merge(mppi_planner* p, const void* gathered, int32_t nranks, void* stream) {
    if (!p || !gathered || nranks < 1) return fail(-1, "bad argument");
    HIP_OK(hipSetDevice(p->cfg.device));
    return launch_merge(p, (const double*)gathered, nranks, pick(p, stream));
}

static int build_graph(mppi_planner* p) {
    if (p->graph) { HIP_OK(hipGraphExecDestroy(p->graph)); p->graph = nullptr; }
    hipGraph_t g;
    HIP_OK(hipStreamBeginCapture(p->stream, hipStreamCaptureModeThreadLocal));
    int rc = launch_rollout_only(p, p->stream);
    if (!rc) rc = launch_update_fused(p, p->stream);
    hipError_t e = hipStreamEndCapture(p->stream, &g);
    if (rc) return rc;
    HIP_OK(e);
    HIP_OK(hipGraphInstantiate(&p->graph, g, nullptr, nullptr, 0));
    HIP_OK(hipGraphDestroy(g));
    p->graph_valid = true;
    return 0;
}

int mppi_iterate(mppi_planner* p, int32_t niter, void* stream) {
    if (!p || niter < 0) return fail(-1, "bad argument");
    if (p->explicit_act) return fail(-4, "update needs noise-relative samples; call mppi_upload_noise or mppi_generate_noise");
    HIP_OK(hipSetDevice(p->cfg.device));
    hipStream_t s = pick(p, stream);
    static const bool use_graph = [] { const char* e = getenv("MPPI_GRAPH"); return !(e && e[0] == '0'); }();
    if (use_graph && !p->cfg.dump && !p->profiling) {
        if (!p->graph_valid) { int rc = build_graph(p); if (rc) return rc; }
        for (int i = 0; i < niter; i++) HIP_OK(hipGraphLaunch(p->graph, s));
        return 0;
    }
    for (int i = 0; i < niter; i++) {
        int rc = launch_rollout_only(p, s);
        if (!rc) rc = launch_update_fused(p, s);
        if (rc) return rc;
    }
    return 0;
}

// one whole planning step enqueued on s (no host synchronisation)
// The noise of the coming step is drawn in the advance launch of the previous one when the handle is small enough
// for that launch to have idle CUs (the fp64 step next to it sets the kernel's register budget, which would slow a
// GPU-filling noise pass down).
static bool plan_draws_ahead(const mppi_planner* p) { return (long)p->cfg.K * p->P <= 65536; }

// niter x (rollouts, update), then the real-env step; the noise of this step must be in place
static int enqueue_plan_body(mppi_planner* p, uint64_t seed, int niter, hipStream_t s) {
    for (int i = 0; i < niter; i++) {
        int rc = launch_rollout_only(p, s);
        if (!rc) rc = launch_update_fused(p, s);
        if (rc) return rc;
    }
    if (plan_draws_ahead(p))
        HIP_OK(p->model->ops->advance_noise(p->cfg.dtype, p->d_state, p->d_base, p->d_ref, p->cfg.reward_type, p->cfg.H, p->P,
                                            p->cfg.default_act, p->d_mean, p->d_log, MPPI_LOG_CAP, p->d_log_count, p->cfg.K,
                                            p->cfg.k_offset, seed, p->d_next_step, p->fc, p->d_eps, s));
    else
        HIP_OK(p->model->ops->advance(p->d_state, p->d_base, p->d_ref, p->cfg.reward_type, p->cfg.H, p->P, p->cfg.default_act,
                                      p->d_mean, p->d_log, MPPI_LOG_CAP, p->d_log_count, s));
    return 0;
}
static int enqueue_plan_noise(mppi_planner* p, uint64_t seed, hipStream_t s) {
    HIP_OK(p->model->ops->noise_philox(p->cfg.dtype, p->cfg.K, p->cfg.H, p->P, p->cfg.k_offset, seed, p->d_state, 1, p->fc,
                                       p->d_eps, s));
    return 0;
}

int mppi_plan(mppi_planner* p, uint64_t seed, int32_t niter, int32_t nsteps, void* stream) {
    if (!p || niter < 1 || nsteps < 0) return fail(-1, "bad argument");
    if (p->HM > 256 * 16) return fail(-1, "H*M too large for mppi_plan");
    if (nsteps == 0) return 0;
    HIP_OK(hipSetDevice(p->cfg.device));
    hipStream_t s = pick(p, stream);
    p->explicit_act = 0;
    const bool ahead = plan_draws_ahead(p);
    // with noise drawn ahead only the first step of a run needs a launch of its own
    const bool have_noise = ahead && p->noise_ahead && p->noise_ahead_seed == seed;
    static const bool use_graph = [] { const char* e = getenv("MPPI_GRAPH"); return !(e && e[0] == '0'); }();
    if (!use_graph || p->profiling) {
        for (int i = 0; i < nsteps; i++) {
            int rc = (ahead && (i > 0 || have_noise)) ? 0 : enqueue_plan_noise(p, seed, s);
            if (!rc) rc = enqueue_plan_body(p, seed, niter, s);
            if (rc) return rc;
        }
    } else {
        if (!p->step_graph || !p->graph_valid || p->step_graph_niter != niter || p->step_graph_seed != seed) {
            if (p->step_graph) { HIP_OK(hipGraphExecDestroy(p->step_graph)); p->step_graph = nullptr; }
            hipGraph_t g;
            HIP_OK(hipStreamBeginCapture(p->stream, hipStreamCaptureModeThreadLocal));
            int rc = ahead ? 0 : enqueue_plan_noise(p, seed, p->stream);
            if (!rc) rc = enqueue_plan_body(p, seed, niter, p->stream);
            const hipError_t e = hipStreamEndCapture(p->stream, &g);
            if (rc) return rc;
            HIP_OK(e);
            HIP_OK(hipGraphInstantiate(&p->step_graph, g, nullptr, nullptr, 0));
            HIP_OK(hipGraphDestroy(g));
            p->step_graph_niter = niter;
            p->step_graph_seed = seed;
            if (!p->graph_valid) { const int rc2 = build_graph(p); if (rc2) return rc2; }    // shares the validity flag
        }
        if (ahead && !have_noise) { const int rc = enqueue_plan_noise(p, seed, s); if (rc) return rc; }
        for (int i = 0; i < nsteps; i++) HIP_OK(hipGraphLaunch(p->step_graph, s));
    }
    p->noise_ahead = ahead;
    p->noise_ahead_seed = seed;
    return 0;
}

int mppi_shift(mppi_planner* p, const double* override_seq, void* stream) {
    if (!p) return fail(-1, "null argument");
    HIP_OK(hipSetDevice(p->cfg.device));
    hipStream_t s = pick(p, stream);
    if (override_seq) {
        HIP_OK(hipMemcpyAsync(p->d_base, override_seq, (size_t)p->HM * p->P * 8, hipMemcpyHostToDevice, s));
        HIP_OK(hipStreamSynchronize(s));
    }
    hipLaunchKernelGGL(shift_kernel, dim3((p->HM + 255) / 256, p->P), dim3(256), 0, s, p->d_base, p->cfg.H, p->nv, p->cfg.default_act,
                       p->d_mean, p->d_base_tmp);
    HIP_OK(hipGetLastError());
    HIP_OK(hipMemcpyAsync(p->d_base, p->d_base_tmp, (size_t)p->HM * p->P * 8, hipMemcpyDeviceToDevice, s));
    return 0;
}

int mppi_profile(mppi_planner* p, int32_t enable) {
    if (!p) return fail(-1, "null argument");
    p->profiling = enable != 0;
    return 0;
}

int mppi_profile_read(mppi_planner* p, double* rollout_ms_total, int32_t* launches) {
    if (!p || !rollout_ms_total || !launches) return fail(-1, "null argument");
    HIP_OK(hipSetDevice(p->cfg.device));
    double total = 0.0;
    int n = 0;
    std::vector<hipEvent_t>& ev = *p->prof_events;
    for (size_t i = 0; i + 1 < ev.size(); i += 2) {
        HIP_OK(hipEventSynchronize(ev[i + 1]));
        float ms = 0.f;
        HIP_OK(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
        total += ms;
        n++;
    }
    for (hipEvent_t e : ev) (void)hipEventDestroy(e);
    ev.clear();
    *rollout_ms_total = total;
    *launches = n;
    return 0;
}

int mppi_synchronize(mppi_planner* p) {
    if (!p) return fail(-1, "null argument");
    HIP_OK(hipSetDevice(p->cfg.device));
    HIP_OK(hipStreamSynchronize(p->stream));
    return 0;
}

static int read_as_double(mppi_planner* p, const void* dsrc, size_t n, double* dst) {
    if (p->cfg.dtype == MPPI_F64) {
        HIP_OK(hipMemcpy(dst, dsrc, n * 8, hipMemcpyDeviceToHost));
    } else {
        std::vector<float> tmp(n);
        HIP_OK(hipMemcpy(tmp.data(), dsrc, n * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; i++) dst[i] = (double)tmp[i];
    }
    return 0;
}

int mppi_read(mppi_planner* p, int32_t what, void* dst, size_t bytes) {
    if (!p || !dst) return fail(-1, "null argument");
    HIP_OK(hipSetDevice(p->cfg.device));
    HIP_OK(hipStreamSynchronize(p->stream));
    // every selector returns the P planners of the handle back to back
    const size_t P = p->P, K = p->cfg.K, H = p->cfg.H, HM = p->HM, OD = p->nq + p->nv + 6;
    auto need = [&](size_t n) { return bytes == n; };
    switch (what) {
    case MPPI_READ_RETURNS:
        if (!need(P * K * 8)) return fail(-1, "RETURNS needs P*K doubles");
        return read_as_double(p, p->d_R, P * K, (double*)dst);
    case MPPI_READ_ACT_SEQ:
        if (!need(P * HM * 8)) return fail(-1, "ACT_SEQ needs P*H*M doubles");
        HIP_OK(hipMemcpy(dst, p->d_base, P * HM * 8, hipMemcpyDeviceToHost));
        return 0;
    case MPPI_READ_ARGMAX:
    case MPPI_READ_RMAX: {
        if (!need(P * 8)) return fail(-1, "ARGMAX / RMAX need P values");
        std::vector<double> r(2 * P);
        HIP_OK(hipMemcpy(r.data(), p->d_result, 16 * P, hipMemcpyDeviceToHost));
        for (size_t b = 0; b < P; b++) {
            if (what == MPPI_READ_ARGMAX) ((int64_t*)dst)[b] = (int64_t)r[2 * b + 1];
            else ((double*)dst)[b] = r[2 * b];
        }
        return 0;
    }
    case MPPI_READ_ACTIONS: {
        if (!need(P * K * HM * 8)) return fail(-1, "ACTIONS needs P*K*H*M doubles");
        std::vector<double> e(P * HM * K), base(P * HM);
        int rc = read_as_double(p, p->d_eps, P * HM * K, e.data());
        if (rc) return rc;
        HIP_OK(hipMemcpy(base.data(), p->d_base, P * HM * 8, hipMemcpyDeviceToHost));
        double* o = (double*)dst;
        for (size_t b = 0; b < P; b++)
            for (size_t k = 0; k < K; k++)
                for (size_t hm = 0; hm < HM; hm++) {
                    // same arithmetic as the kernel: the sum is formed in the rollout dtype
                    const double bb = base[b * HM + hm], x = e[(b * HM + hm) * K + k];
                    double a;
                    if (p->explicit_act) a = x;
                    else if (p->cfg.dtype == MPPI_F32) a = (double)((float)bb + (float)x);
                    else a = bb + x;
                    o[(b * K + k) * HM + hm] = a;
                }
        return 0;
    }
    case MPPI_READ_NEXT_OBS:
        if (!p->d_obs) return fail(-4, "planner was created without cfg.dump");
        if (!need(P * K * H * OD * 8)) return fail(-1, "NEXT_OBS needs P*K*H*(nq+nv+6) doubles");
        return read_as_double(p, p->d_obs, P * K * H * OD, (double*)dst);
    case MPPI_READ_REWARDS:
        if (!p->d_rew) return fail(-4, "planner was created without cfg.dump");
        if (!need(P * K * H * 8)) return fail(-1, "REWARDS needs P*K*H doubles");
        return read_as_double(p, p->d_rew, P * K * H, (double*)dst);
    case MPPI_READ_STATUS:
        if (!need(4)) return fail(-1, "STATUS needs one int32");
        HIP_OK(hipMemcpy(dst, p->d_status, 4, hipMemcpyDeviceToHost));
        return 0;
    case MPPI_READ_PARTIALS:
        if (!need(P * (3 + HM) * 8)) return fail(-1, "PARTIALS needs P*(3+H*M) doubles");
        HIP_OK(hipMemcpy(dst, p->d_part, P * (3 + HM) * 8, hipMemcpyDeviceToHost));
        return 0;
    }
    return fail(-1, "unknown read selector");
}

static int env_io(const mppi_model* m, int device, double** io) {
    if (device < 0 || device >= 16) return fail(-1, "device ordinal out of range");
    HIP_OK(hipSetDevice(device));
    mppi_model* mm = const_cast<mppi_model*>(m);
    if (!mm->d_io[device]) HIP_OK(hipMalloc(&mm->d_io[device], 512 * 8));
    *io = mm->d_io[device];
    return 0;
}

int mppi_env_step(const mppi_model* m, int32_t device, const double* qpos, const double* qvel, const double* target3,
                  const double* act, int32_t timestep, int32_t reward_type, const double* ref, int32_t T, double alpha,
                  double* qpos_out, double* qvel_out, double* obs_out, double* reward_out) {
    if (!m || !qpos || !qvel || !target3 || !act) return fail(-1, "null argument");
    if ((reward_type == MPPI_REWARD_TRACKING || reward_type == MPPI_REWARD_COOLING) && (!ref || T < 1))
        return fail(-1, "tracking / cooling reward needs a reference");
    const int nv = m->ops->nv, nq = m->ops->nq, ns = nq + nv;
    double* io;
    int rc = env_io(m, device, &io);
    if (rc) return rc;
    double in[4 * MPPI_MAXV + 16];
    for (int i = 0; i < nq; i++) in[i] = qpos[i];
    for (int i = 0; i < nv; i++) { in[nq + i] = qvel[i]; in[ns + 3 + i] = act[i]; }
    for (int c = 0; c < 3; c++) in[ns + c] = target3[c];
    in[ns + nv + 3] = alpha;
    const int has_ref = ref && T > 0;
    if (has_ref) {
        if (timestep + 1 >= T) return fail(-5, "reference index out of range");   // IndexError in the reference
        for (int c = 0; c < 3; c++) in[ns + nv + 4 + c] = ref[3 * (timestep + 1) + c];
    }
    const int n_in = ns + nv + 7, n_out = 2 * ns + 7;
    HIP_OK(hipMemcpy(io, in, n_in * 8, hipMemcpyHostToDevice));
    HIP_OK(m->ops->env_step(io, reward_type, has_ref, 0, io + 256, 0));
    double out[4 * MPPI_MAXV + 16];
    HIP_OK(hipMemcpy(out, io + 256, n_out * 8, hipMemcpyDeviceToHost));
    if (qpos_out) for (int i = 0; i < nq; i++) qpos_out[i] = out[i];
    if (qvel_out) for (int i = 0; i < nv; i++) qvel_out[i] = out[nq + i];
    if (obs_out) for (int i = 0; i < ns + 6; i++) obs_out[i] = out[ns + i];
    if (reward_out) *reward_out = out[2 * ns + 6];
    return 0;
}

int mppi_env_forward(const mppi_model* m, int32_t device, const double* qpos, double* hand3_out) {
    if (!m || !qpos || !hand3_out) return fail(-1, "null argument");
    const int nv = m->ops->nv, nq = m->ops->nq, ns = nq + nv;
    double* io;
    int rc = env_io(m, device, &io);
    if (rc) return rc;
    double in[4 * MPPI_MAXV + 16] = {0};
    for (int i = 0; i < nq; i++) in[i] = qpos[i];
    HIP_OK(hipMemcpy(io, in, (ns + nv + 7) * 8, hipMemcpyHostToDevice));
    HIP_OK(m->ops->env_step(io, 0, 0, 1, io + 256, 0));
    double out[4 * MPPI_MAXV + 16];
    HIP_OK(hipMemcpy(out, io + 256, (2 * ns + 7) * 8, hipMemcpyDeviceToHost));
    for (int c = 0; c < 3; c++) hand3_out[c] = out[2 * ns + c];
    return 0;
}

int mppi_update_from_returns(int32_t device, const double* act, const double* R, int32_t K, int32_t HM, double kappa,
                             double* act_seq_out) {
    if (!act || !R || !act_seq_out || K < 1 || HM < 1) return fail(-1, "bad argument");
    HIP_OK(hipSetDevice(device));
    double *d_act = nullptr, *d_R = nullptr, *d_out = nullptr;
    HIP_OK(hipMalloc(&d_act, (size_t)K * HM * 8));
    HIP_OK(hipMalloc(&d_R, (size_t)K * 8));
    HIP_OK(hipMalloc(&d_out, (size_t)HM * 8));
    HIP_OK(hipMemcpy(d_act, act, (size_t)K * HM * 8, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_R, R, (size_t)K * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(update_direct_kernel, dim3(HM), dim3(256), 0, 0, d_act, d_R, K, (int)HM, kappa, d_out);
    HIP_OK(hipGetLastError());
    HIP_OK(hipMemcpy(act_seq_out, d_out, (size_t)HM * 8, hipMemcpyDeviceToHost));
    HIP_OK(hipFree(d_act)); HIP_OK(hipFree(d_R)); HIP_OK(hipFree(d_out));
    return 0;
}

}  // extern "C"
