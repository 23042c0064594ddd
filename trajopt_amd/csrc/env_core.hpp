// One env.step of the reacher family on top of the substep: FRAME_SKIP physics
// substeps with the action held, then the reward of
// `Reacher7DOFEnv._step` (trajopt/envs/reacher_env.py:43-85).
//
// The end-effector position that enters the reward and the observation is the
// one evaluated at the configuration *before* the last substep's integration:
// the reference reads the simulator's derived site position right after
// `sim.step()`, which still belongs to the pre-integration configuration.
#pragma once
#include "dyn_core.hpp"

namespace mppi {

enum RewardType : int { REWARD_DENSE = 0, REWARD_SPARSE = 1, REWARD_TRACKING = 2, REWARD_COOLING = 3 };

template <class T> struct RewardParams {
    int type;          // RewardType
    T alpha;           // cooling blend (reacher_env.py:79)
    T target[3];
};

#if defined(__HIP_DEVICE_COMPILE__)
// fp64 substep of the wide models as a function of its own.  Inlined into the rollout loop, the fp64 kernel of an
// 18-DoF model needs 256 VGPRs + 256 AGPRs + about 700 bytes of scratch + 90 spilled SGPRs, and ROCm 7.2's hipcc
// then emits wrong code for it: every IR-level pass leaves it right (-opt-bisect-limit), the result flips inside
// that one kernel's machine passes, back and forth with which of them run; the host build of the same source is
// clean under MemorySanitizer / UBSan / ASan (tools/host_sanitize.sh, tools/f64_*.sh).  Out of line the substep
// has a frame of its own and matches the fp64 oracle to rounding.  The 7-DoF arm fits and stays inline (right
// under every flag combination tried, and its real-env step is a quarter faster that way); fp32 is always inline.
template <class MD>
__device__ __noinline__ void substep_outofline(double (&q)[MD::NQ], double (&qd)[MD::NV], const double (&act)[MD::NV],
                                               bool want_hand, typename Step<double, MD>::Out& o,
                                               typename Step<double, MD>::Active& rows, int& sw) {
    Step<double, MD>::substep(q, qd, act, want_hand, o, rows, sw);
}
#endif

template <class T, class MD> struct Env {
    static constexpr int NV = MD::NV;
    using S = Step<T, MD>;

    // ref_row: reference[env_timestep_after_step] (3 values) or nullptr
    static MPPI_HD T reward(const RewardParams<T>& rp, const Vec3<T>& hand, const T (&qd)[NV], const T* ref_row) {
        T vv = T(0);
        static_for<NV>([&](auto Ic) { vv += qd[Ic] * qd[Ic]; });
        const T vn = sqrt_t(vv);
        const T dx = hand.v[0] - rp.target[0], dy = hand.v[1] - rp.target[1], dz = hand.v[2] - rp.target[2];
        const T dist = sqrt_t(dx * dx + dy * dy + dz * dz);
        const T sparse = sel(dist < T(0.025), T(100), sel(dist > T(0.8), T(-100), T(-10)));
        if (rp.type == REWARD_SPARSE) return sparse;
        if (rp.type == REWARD_TRACKING || rp.type == REWARD_COOLING) {
            const T ex = hand.v[0] - ref_row[0], ey = hand.v[1] - ref_row[1], ez = hand.v[2] - ref_row[2];
            const T track = T(-10) * sqrt_t(ex * ex + ey * ey + ez * ez) - T(0.25) * vn;
            if (rp.type == REWARD_TRACKING) return track;
            return rp.alpha * track + (T(1) - rp.alpha) * sparse;
        }
        return T(-10) * dist - T(0.25) * vn;
    }

    static MPPI_HD T step(T (&q)[MD::NQ], T (&qd)[NV], const T (&act)[NV], const RewardParams<T>& rp, const T* ref_row,
                          Vec3<T>& hand, typename S::Active& rows, int& max_sweeps) {
        typename S::Out o;
        o.hand = {{T(0), T(0), T(0)}};
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (std::is_same<T, double>::value && (MD::NV > 8)) {
#pragma nounroll
            for (int s = 0; s < MD::FRAME_SKIP; s++) {
                int sw;
                substep_outofline<MD>(q, qd, act, s == MD::FRAME_SKIP - 1, o, rows, sw);
                max_sweeps = sw > max_sweeps ? sw : max_sweeps;
            }
        } else {
            // the substeps are laid out one after the other (no loop branch, the end-effector site only in the last one)
            static_for<MD::FRAME_SKIP>([&](auto Sc) {
                int sw;
                S::substep(q, qd, act, Sc == MD::FRAME_SKIP - 1, o, rows, sw);
                max_sweeps = sw > max_sweeps ? sw : max_sweeps;
            });
        }
#else
        for (int s = 0; s < MD::FRAME_SKIP; s++) {
            int sw;
            S::substep(q, qd, act, s == MD::FRAME_SKIP - 1, o, rows, sw);
            max_sweeps = sw > max_sweeps ? sw : max_sweeps;
        }
#endif
        hand = o.hand;
        return reward(rp, hand, qd, ref_row);
    }
};

}  // namespace mppi
