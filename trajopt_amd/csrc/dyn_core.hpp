// Per-sample articulated-body step, specialised at compile time on the model.
//
// One call advances one sample by one physics substep; every quantity lives in
// registers (the caller is one GPU lane).  The formulation is in link-local
// coordinates -- velocities, forces and composite inertias are expressed in the
// frame of the link they belong to -- so that the small wrist inertias are never
// formed as differences of large world-frame terms; that is what keeps fp32
// rollouts close to an fp64 simulator.  All model data are `static constexpr`
// members of MD (see models_gen.hpp): loops are unrolled with compile-time
// indices, so structural zeros vanish from the instruction stream.
//
// Replaces, for the planner's rollouts, the simulator call made by the
// reference's `MujocoEnv.do_simulation` (trajopt/envs/mujoco_env.py:134-138).
//
// The same header compiles for the host (tests build it with g++ to compare
// against the CPU oracle without a GPU).
#pragma once
#include <cmath>
#include <type_traits>
#include <utility>

#if defined(__HIPCC__)
#define MPPI_HD __host__ __device__ __forceinline__
#else
#define MPPI_HD inline __attribute__((always_inline))
#endif

// host studies of the active-set iteration define this to see which row changed in which sweep
#ifndef MPPI_SWEEP_TRACE
#define MPPI_SWEEP_TRACE(sweep, row, changed) ((void)0)
#endif

namespace mppi {

template <class F, int... Is>
MPPI_HD void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
MPPI_HD void static_for(F&& f) {
    static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

MPPI_HD void sincos_t(float x, float& s, float& c) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MPPI_PRECISE_TRIG)
    s = __sinf(x); c = __cosf(x);
#elif defined(__HIP_DEVICE_COMPILE__)
    sincosf(x, &s, &c);
#else
    s = std::sin(x); c = std::cos(x);
#endif
}
MPPI_HD void sincos_t(double x, double& s, double& c) {
#if defined(__HIP_DEVICE_COMPILE__)
    sincos(x, &s, &c);
#else
    s = std::sin(x); c = std::cos(x);
#endif
}
MPPI_HD double sqrt_t(double x) { return sqrt(x); }
// fp32 on the device uses the hardware square root / reciprocal / reciprocal square root / sine / cosine
// (about 1 ulp; the angle argument is bounded by the joint ranges); fp64 keeps libm sine / cosine
MPPI_HD float sqrt_t(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sqrtf(x);
#else
    return sqrtf(x);
#endif
}
MPPI_HD float rsqrt_t(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __frsqrt_rn(x);
#else
    return 1.0f / std::sqrt(x);
#endif
}
// fp64 on the device: the hardware estimate (about 26 bits) and two Newton steps, a dozen instructions; sqrt() followed
// by a division is four times that, most of it special cases that a positive pivot cannot be
MPPI_HD double rsqrt_t(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    y = fma(y, fma(-hx * y, y, 0.5), y);
    y = fma(y, fma(-hx * y, y, 0.5), y);
    return y;
#else
    return 1.0 / sqrt(x);
#endif
}
MPPI_HD float rcp_t(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);            // v_rcp_f32, 1 ulp (a correctly rounded quotient costs ten instructions)
#else
    return 1.0f / x;
#endif
}
MPPI_HD double rcp_t(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rcp(x);
    y = fma(y, fma(-x, y, 1.0), y);
    y = fma(y, fma(-x, y, 1.0), y);
    return y;
#else
    return 1.0 / x;
#endif
}
MPPI_HD float min_t(float a, float b) { return fminf(a, b); }
MPPI_HD double min_t(double a, double b) { return fmin(a, b); }
MPPI_HD float max_t(float a, float b) { return fmaxf(a, b); }
MPPI_HD double max_t(double a, double b) { return fmax(a, b); }
MPPI_HD float abs_t(float x) { return fabsf(x); }
MPPI_HD double abs_t(double x) { return fabs(x); }
MPPI_HD float clamp_t(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
MPPI_HD double clamp_t(double x, double lo, double hi) { return fmin(fmax(x, lo), hi); }

// Conditions on a sample are consumed with sel() (a select, never a branch): the per-sample code stays
// one straight instruction stream whatever the lanes of a wave decide.
template <class T> using Mask = bool;
MPPI_HD float sel(bool c, float x, float y) { return c ? x : y; }
MPPI_HD double sel(bool c, double x, double y) { return c ? x : y; }
MPPI_HD bool any_of(bool c) { return c; }

// true if the predicate holds on any active lane of the wave (host: the lane itself)
MPPI_HD bool wave_any(bool x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ballot_w64(x) != 0ull;
#else
    return x;
#endif
}

// Layout hints for wave-uniform branches: the expected side becomes the fall-through path.  A taken branch
// costs a wave that owns its SIMD roughly ten vector instructions' time; one that falls through, next to none.
#define MPPI_LIKELY(x) __builtin_expect(!!(x), 1)
#define MPPI_UNLIKELY(x) __builtin_expect(!!(x), 0)

constexpr int sym_idx(int i, int j) { return i == j ? i : i + j + 2; }   // xx yy zz xy xz yz

template <class T> struct Vec3 { T v[3]; };

// Two values that go through the same arithmetic side by side (angular and linear velocity, force and
// moment, ...).  On gfx950 arithmetic on such a pair is one packed instruction (v_pk_fma_f32 and friends)
// with scalars broadcast for free, which halves the issue slots of everything written in pairs; fp64
// pairs are simply two instructions.  [0] and [1] address the halves.
template <class T> struct PairOf;
template <> struct PairOf<float> { typedef float type __attribute__((vector_size(8))); };
template <> struct PairOf<double> { typedef double type __attribute__((vector_size(16))); };
template <class T> using Pair = typename PairOf<T>::type;
template <class T> MPPI_HD Pair<T> make_pair_of(T a, T b) { Pair<T> p = {a, b}; return p; }
// the operations that have no packed instruction, half by half
MPPI_HD Pair<float> abs_t(Pair<float> x) { return make_pair_of(abs_t(x[0]), abs_t(x[1])); }
MPPI_HD Pair<double> abs_t(Pair<double> x) { return make_pair_of(abs_t(x[0]), abs_t(x[1])); }
MPPI_HD Pair<float> min_t(Pair<float> x, float b) { return make_pair_of(min_t(x[0], b), min_t(x[1], b)); }
MPPI_HD Pair<double> min_t(Pair<double> x, double b) { return make_pair_of(min_t(x[0], b), min_t(x[1], b)); }
MPPI_HD Pair<float> rcp_t(Pair<float> x) { return make_pair_of(rcp_t(x[0]), rcp_t(x[1])); }
MPPI_HD Pair<double> rcp_t(Pair<double> x) { return make_pair_of(rcp_t(x[0]), rcp_t(x[1])); }
// a pair of which only half H is ever looked at (the other half costs no instruction on the device)
template <int H, class T> MPPI_HD Pair<T> half_pair(T a) {
#if defined(__HIP_DEVICE_COMPILE__) || defined(MPPI_UNDEF_HALVES)      // (the macro: host builds under MemorySanitizer)
    Pair<T> p;
    p[H] = a;
    return p;
#else
    Pair<T> p = {a, a};
    return p;
#endif
}
// both halves set to half H of a pair
template <int H, class P> MPPI_HD P splat_half(const P& a) {
#if defined(__clang__)
    return __builtin_shufflevector(a, a, H, H);
#else
    P r = {a[H], a[H]};
    return r;
#endif
}
template <class T> MPPI_HD Vec3<T> half(const Vec3<Pair<T>>& a, int h) { return {{a.v[0][h], a.v[1][h], a.v[2][h]}}; }

// a x b for any mix of scalar and paired operands
template <class EA, class EB> MPPI_HD auto cross(const Vec3<EA>& a, const Vec3<EB>& b) -> Vec3<decltype(a.v[0] * b.v[0])> {
    return {{a.v[1] * b.v[2] - a.v[2] * b.v[1], a.v[2] * b.v[0] - a.v[0] * b.v[2], a.v[0] * b.v[1] - a.v[1] * b.v[0]}};
}
template <class E> MPPI_HD Vec3<E> add(const Vec3<E>& a, const Vec3<E>& b) {
    return {{a.v[0] + b.v[0], a.v[1] + b.v[1], a.v[2] + b.v[2]}};
}
template <class T, class E> MPPI_HD Vec3<E> scale(T s, const Vec3<E>& a) { return {{s * a.v[0], s * a.v[1], s * a.v[2]}}; }
// dst[1] += r x src[0] for a vector of pairs (the moment of a force about a displaced origin added to the moment half):
// written as packed multiply-adds of {0, r_k} with the force half broadcast, so that the pair is updated as a pair
template <class T> MPPI_HD void add_cross_to_second(Vec3<Pair<T>>& dst, const Vec3<T>& r, const Vec3<Pair<T>>& src) {
    const Pair<T> r0 = make_pair_of(T(0), r.v[0]), r1 = make_pair_of(T(0), r.v[1]), r2 = make_pair_of(T(0), r.v[2]);
    const Pair<T> s0 = splat_half<0>(src.v[0]), s1 = splat_half<0>(src.v[1]), s2 = splat_half<0>(src.v[2]);
    dst.v[0] += r1 * s2 - r2 * s1;
    dst.v[1] += r2 * s0 - r0 * s2;
    dst.v[2] += r0 * s1 - r1 * s0;
}

// rotation about coordinate axis A by angle (c, s): parent coordinates from child coordinates
template <int A, class T, class E> MPPI_HD Vec3<E> to_parent(T c, T s, const Vec3<E>& x) {
    constexpr int U = (A + 1) % 3, W = (A + 2) % 3;
    Vec3<E> y;
    y.v[A] = x.v[A];
    y.v[U] = c * x.v[U] - s * x.v[W];
    y.v[W] = s * x.v[U] + c * x.v[W];
    return y;
}
template <int A, class T, class E> MPPI_HD Vec3<E> to_child(T c, T s, const Vec3<E>& x) {
    constexpr int U = (A + 1) % 3, W = (A + 2) % 3;
    Vec3<E> y;
    y.v[A] = x.v[A];
    y.v[U] = c * x.v[U] + s * x.v[W];
    y.v[W] = c * x.v[W] - s * x.v[U];
    return y;
}

// spatial inertia about a frame origin: mass, first moment h = m*c, rotational inertia I (sym 6)
template <class T> struct SpInertia { T m; Vec3<T> h; T I[6]; };

template <class T, class E> MPPI_HD Vec3<E> sym_mul(const T (&I)[6], const Vec3<E>& w) {
    return {{I[0] * w.v[0] + I[3] * w.v[1] + I[4] * w.v[2],
             I[3] * w.v[0] + I[1] * w.v[1] + I[5] * w.v[2],
             I[4] * w.v[0] + I[5] * w.v[1] + I[2] * w.v[2]}};
}

template <class T, class MD> struct Step {
    static constexpr int NV = MD::NV;
    static constexpr int NQ = MD::NQ;            // qpos entries: NV + one per ball joint (unit quaternion)
    static constexpr bool is_ball(int i) { return MD::jtype[i] >= 2; }
    static constexpr bool any_ball() {
        for (int i = 0; i < NV; i++) if (MD::jtype[i] >= 2) return true;
        return false;
    }

    // Orientation of every link frame in its parent's frame: an angle about a coordinate axis for a hinge,
    // a rotation matrix (from the unit quaternion) for the first link of a ball joint, identity otherwise.
    template <bool BALLS, class Dummy = void> struct FramesRb { T rb[NV][9]; };     // row-major
    template <class Dummy> struct FramesRb<false, Dummy> {};
    struct Frames : FramesRb<any_ball()> {
        alignas(8) T c_s[NV][2];     // cosine and sine of every hinge angle, side by side
        MPPI_HD T& cs(int i) { return c_s[i][0]; }
        MPPI_HD T& sn(int i) { return c_s[i][1]; }
        MPPI_HD const T& cs(int i) const { return c_s[i][0]; }
        MPPI_HD const T& sn(int i) const { return c_s[i][1]; }
    };
    // the rotation matrix of link I (a template so that ball-free models never name the member)
    template <int I, class Fr> static MPPI_HD auto& rb_of(Fr& fr) { return fr.rb[I]; }
    // parent coordinates -> link-I coordinates and back, for single vectors and pairs alike
    template <int I, class E> static MPPI_HD Vec3<E> down(const Frames& fr, const Vec3<E>& x) {
        if constexpr (MD::jtype[I] == 0) {
            return to_child<MD::axis_id[I]>(fr.cs(I), fr.sn(I), x);
        } else if constexpr (MD::jtype[I] == 2) {
            const T (&R)[9] = rb_of<I>(fr);
            return {{R[0] * x.v[0] + R[3] * x.v[1] + R[6] * x.v[2], R[1] * x.v[0] + R[4] * x.v[1] + R[7] * x.v[2],
                     R[2] * x.v[0] + R[5] * x.v[1] + R[8] * x.v[2]}};
        } else {
            return x;
        }
    }
    template <int I, class E> static MPPI_HD Vec3<E> up(const Frames& fr, const Vec3<E>& x) {
        if constexpr (MD::jtype[I] == 0) {
            return to_parent<MD::axis_id[I]>(fr.cs(I), fr.sn(I), x);
        } else if constexpr (MD::jtype[I] == 2) {
            const T (&R)[9] = rb_of<I>(fr);
            return {{R[0] * x.v[0] + R[1] * x.v[1] + R[2] * x.v[2], R[3] * x.v[0] + R[4] * x.v[1] + R[5] * x.v[2],
                     R[6] * x.v[0] + R[7] * x.v[1] + R[8] * x.v[2]}};
        } else {
            return x;
        }
    }

    struct Out {
        Vec3<T> hand;        // end-effector site in world coordinates at the input q
    };

    template <int I> static MPPI_HD Vec3<T> pos_() { return {{T(MD::pos[I][0]), T(MD::pos[I][1]), T(MD::pos[I][2])}}; }
    // origin of link I in its parent's frame: constant for a hinge, displaced along the axis by q for a slide
    template <int I> static MPPI_HD Vec3<T> off_(const T (&q)[NQ]) {
        Vec3<T> r = pos_<I>();
        if constexpr (MD::jtype[I] == 1) r.v[MD::axis_id[I]] += T(MD::axis_sign[I]) * q[MD::qadr[I]];
        return r;
    }
    template <int I> static MPPI_HD Vec3<T> mc_() { return {{T(MD::mc[I][0]), T(MD::mc[I][1]), T(MD::mc[I][2])}}; }

    // true if link a is link l or one of its ancestors
    static constexpr bool on_path(int a, int l) {
        while (l >= 0) {
            if (l == a) return true;
            l = MD::parent[l];
        }
        return false;
    }

    // M[i][j] can be nonzero only between a link and its ancestors
    static constexpr bool coupled(int i, int j) { return i >= j ? on_path(j, i) : on_path(i, j); }
    static constexpr bool branching() {
        for (int i = 0; i < NV; i++) if (MD::parent[i] != i - 1) return true;
        return false;
    }
    static constexpr bool has_rows() {
        if (MD::contact_link >= 0) return true;
        for (int i = 0; i < NV; i++) if (MD::limited[i] != 0) return true;
        return false;
    }
    // Branching models without constraint rows factor M + h B from the leaves towards the root (below): no
    // fill-in between branches.  Models with rows keep the root-first order their constraint solve shares.
    static constexpr bool LEAF_FIRST = branching() && !has_rows();

    // coordinates of a point fixed in link I, expressed in every ancestor frame on the way to the
    // world: x[i] = point in link-i coordinates, returns world coordinates
    template <int I>
    static MPPI_HD Vec3<T> point_to_world(const T (&q)[NQ], const Frames& fr, const Vec3<T>& x,
                                          Vec3<T> (&xs)[NV]) {
        xs[I] = x;
        const Vec3<T> xp = add(off_<I>(q), up<I>(fr, x));
        if constexpr (MD::parent[I] >= 0) return point_to_world<MD::parent[I]>(q, fr, xp, xs);
        else return xp;
    }

    // soft constraint rows of one substep: joint limits (one possible row per joint) and the contact.
    // A limit row is described in joint coordinates, without the sign of its Jacobian: xs = q - clamp(q, lo, hi) is
    // the signed violation (negative below the lower limit, positive above the upper one, zero inside the range),
    // sal = J * aref the reference acceleration of the joint itself.  With t = qacc - sal the row's residual
    // J qacc - aref is J t and J = -sign(xs): the row is active (residual < 0) iff t * xs > 0, and its force on the
    // joint is -D t.
    // The per-joint quantities are held as pairs of neighbouring joints {2p, 2p+1}: the elementwise arithmetic of the
    // solve (right-hand sides, residuals, forces) is one packed instruction per pair.
    static constexpr int NP = (NV + 1) / 2;
    struct Rows {
        Pair<T> Dl[NP], sal[NP], xs[NP];   // limit rows: weight (0 = no row), reference acceleration of the joint, signed violation
    };
    // The contact row is made in two steps: every substep measures the distance of the sphere from the plane
    // (ContactProbe); the row itself -- Jacobian, weight, reference acceleration -- is only made in a substep in which
    // some sample of the wave is within the margin, inside the variant of the solve that has a contact row.
    struct ContactProbe {
        Vec3<T> xc[NV];                    // the sphere's centre in the coordinates of every link on the way to the world
        T dist;                            // distance of the sphere from the plane
        Mask<T> near;                      // within the margin: this sample has the row
    };
    struct ContactRow { T Jc[NV], Dc, arc; };      // Jacobian, weight (0 = no row), reference acceleration
    // Which rows ended the previous substep inactive: the
    // weight such a row was not given, > 0 only for a row that existed and was inactive.  Carried through a
    // rollout as the starting guess of the active-set solve; every other row, the ones that did not exist
    // included, starts active when it next appears.
    struct Active {
        Pair<T> idle[NP];       // limit rows, joints {2p, 2p+1}
        T idle_contact;
        // most sweeps the iteration may take (uniform over a wave); a solve that stops there with the set
        // still moving reports cap + 1 sweeps.  reset() keeps it.
        int cap = 16;
        MPPI_HD void reset() {
            static_for<NP>([&](auto Pc) { idle[Pc] = make_pair_of(T(0), T(0)); });
            idle_contact = T(0);
        }
    };

    static MPPI_HD void frames(const T (&q)[NQ], Frames& fr) {
        static_for<NV>([&](auto Ic) {
            constexpr int i = Ic;
            constexpr int a = MD::qadr[i];
            if constexpr (MD::jtype[i] == 0) {
                T s, c;
                sincos_t(q[a], s, c);
                fr.cs(i) = c;
                fr.sn(i) = (MD::axis_sign[i] > 0) ? s : -s;
            } else {
                fr.cs(i) = T(1); fr.sn(i) = T(0);    // slides and the later links of a ball joint do not turn their frame
                if constexpr (MD::jtype[i] == 2) {
                    const T w = q[a], x = q[a + 1], y = q[a + 2], z = q[a + 3];
                    T (&R)[9] = rb_of<decltype(Ic)::value>(fr);
                    R[0] = T(1) - T(2) * (y * y + z * z); R[1] = T(2) * (x * y - w * z); R[2] = T(2) * (x * z + w * y);
                    R[3] = T(2) * (x * y + w * z); R[4] = T(1) - T(2) * (x * x + z * z); R[5] = T(2) * (y * z - w * x);
                    R[6] = T(2) * (x * z - w * y); R[7] = T(2) * (y * z + w * x); R[8] = T(1) - T(2) * (x * x + y * y);
                }
            }
        });
    }

    // end-effector site in world coordinates
    static MPPI_HD Vec3<T> hand_site(const T (&q)[NQ], const Frames& fr) {
        if constexpr (MD::hand_link >= 0) {
            Vec3<T> tmp[NV];
            return point_to_world<MD::hand_link>(q, fr, {{T(MD::hand_pos[0]), T(MD::hand_pos[1]), T(MD::hand_pos[2])}}, tmp);
        } else {
            return {{T(0), T(0), T(0)}};
        }
    }

    // actuation (clamped to the control range), joint damping and bias forces
    static MPPI_HD void smooth_force(const T (&q)[NQ], const Frames& fr, const T (&qd)[NV],
                                     const T (&ctrl)[NV], T (&fs)[NV]) {
        T bias[NV];
        bias_forces(q, fr, qd, bias);
        static_for<NV>([&](auto Ic) {
            constexpr int i = Ic;
            T u = ctrl[i];
            u = sel(u < T(MD::ctrl_lo[i]), T(MD::ctrl_lo[i]), u);
            u = sel(u > T(MD::ctrl_hi[i]), T(MD::ctrl_hi[i]), u);
            fs[i] = T(MD::gear[i]) * u - T(MD::damping[i]) * qd[i] - bias[i];
        });
    }

    static MPPI_HD void build_rows(const T (&q)[NQ], const Frames& fr, const T (&qd)[NV], Rows& r) {
        // joint limits: at most one side can be violated; evaluated without branches.  (A row deeper than the
        // impedance ramp has constant weight, and a version that skipped the ramp polynomial unless some lane of
        // the wave was inside the ramp existed: among 64 samples one almost always is, so the test and its
        // branch only cost.)
        // Neighbouring joints go through the row arithmetic as a pair (one packed instruction for both wherever
        // the instruction set has one).
        auto clamped = [&](auto Ic) -> T {
            constexpr int i = Ic;
            return clamp_t(q[MD::qadr[i]], T(MD::range_lo[i]), T(MD::range_hi[i]));
        };
        static_for<NP>([&](auto Pc) {
            constexpr int i0 = 2 * Pc, i1 = (2 * Pc + 1 < NV) ? 2 * Pc + 1 : i0;
            static_assert(MD::limited[i0] == 0 || !is_ball(i0), "ball joints have no limit rows");
            static_assert(MD::limited[i1] == 0 || !is_ball(i1), "ball joints have no limit rows");
            if constexpr (RAMP_ONE_EXPRESSION && i1 != i0 && MD::limited[i0] != 0 && MD::limited[i1] != 0) {
                // (positions and velocities stay single values: pairing them here would tie their registers down
                // for the whole step)
                const Pair<T> xs = make_pair_of(q[MD::qadr[i0]] - clamped(std::integral_constant<int, i0>{}),
                                                q[MD::qadr[i1]] - clamped(std::integral_constant<int, i1>{}));
                Pair<T> D, sa;
                row_from_ramp(make_pair_of(ramp_product(xs[0]), ramp_product(xs[1])), xs,
                              make_pair_of(T(MD::solref_b) * qd[i0], T(MD::solref_b) * qd[i1]),
                              make_pair_of(T(MD::dof_invweight[i0]), T(MD::dof_invweight[i1])), D, sa);
                r.Dl[Pc] = make_pair_of(sel(xs[0] != T(0), D[0], T(0)), sel(xs[1] != T(0), D[1], T(0)));
                r.sal[Pc] = sa;      // (meaningless without the row, and never used then: a row of weight 0 stays out)
                r.xs[Pc] = xs;
            } else {
                r.Dl[Pc] = make_pair_of(T(0), T(0)); r.sal[Pc] = r.Dl[Pc]; r.xs[Pc] = r.Dl[Pc];
                static_for<(i1 != i0) ? 2 : 1>([&](auto Hc) {
                    constexpr int h = Hc, i = i0 + h;
                    if constexpr (MD::limited[i] != 0) {
                        const T xs = q[MD::qadr[i]] - clamped(std::integral_constant<int, i>{});
                        T D, sa;
                        row_constants(xs, T(MD::solref_b) * qd[i], T(MD::dof_invweight[i]), D, sa);
                        r.Dl[Pc][h] = sel(xs != T(0), D, T(0));
                        r.sal[Pc][h] = sa;
                        r.xs[Pc][h] = xs;
                    }
                });
            }
        });
    }

    // distance of the contact sphere from the plane; true if some sample of the wave is within the margin
    static MPPI_HD bool contact_probe(const T (&q)[NQ], const Frames& fr, ContactProbe& c) {
        if constexpr (MD::contact_link >= 0) {
            constexpr int L = MD::contact_link;
            const Vec3<T> cw = point_to_world<L>(q, fr, {{T(MD::contact_pos[0]), T(MD::contact_pos[1]), T(MD::contact_pos[2])}}, c.xc);
            c.dist = cw.v[2] - T(MD::contact_radius) - T(MD::contact_plane_z);
            c.near = c.dist < T(MD::contact_margin);
            return wave_any(c.near);       // uniform over the wave
        } else {
            return false;
        }
    }
    // the contact row of the samples within the margin; the others get zeros
    static MPPI_HD void contact_row(const Frames& fr, const T (&qd)[NV], const ContactProbe& c, ContactRow& r) {
        r.Dc = T(0); r.arc = T(0);
        static_for<NV>([&](auto Ic) { r.Jc[Ic] = T(0); });
        if constexpr (MD::contact_link >= 0) {
            constexpr int L = MD::contact_link;
            // d(z of the sphere centre)/dq_i = z_i . (axis_i x x_i): world z axis and the point, both in link-i coordinates
            Vec3<T> zl[NV];
            T vel = T(0);
            static_for<NV>([&](auto Ic) {
                constexpr int i = Ic;
                if constexpr (on_path(i, L)) {
                    static_assert(MD::jtype[i] == 0, "the contact row is written for hinge chains");
                    constexpr int p = MD::parent[i];
                    constexpr int A = MD::axis_id[i], U = (A + 1) % 3, W = (A + 2) % 3;
                    Vec3<T> zp;
                    if constexpr (p < 0) zp = {{T(0), T(0), T(1)}};
                    else zp = zl[p];
                    zl[i] = down<i>(fr, zp);
                    // (e_A x x)[U] = -x[W], (e_A x x)[W] = x[U]
                    r.Jc[i] = sel(c.near, T(MD::axis_sign[i]) * (zl[i].v[W] * c.xc[i].v[U] - zl[i].v[U] * c.xc[i].v[W]), T(0));
                    vel += r.Jc[i] * qd[i];
                }
            });
            T Dc, arc;
            row_constants(c.dist - T(MD::contact_margin), T(MD::solref_b) * vel, T(MD::contact_invweight), Dc, arc);
            r.Dc = sel(c.near, Dc, T(0));
            r.arc = sel(c.near, arc, T(0));
        }
    }

    // semi-implicit Euler with implicit joint damping: (M + h B) a = fs + fc, v += h a, q += h v.
    // The factor of M + h B does not depend on the forces and may be computed ahead of them.
    // a lower-triangular matrix held as pairs of rows (see the dense solver below)
    struct PackedSystem { Pair<T> col[NV][NP]; };      // col[j][p] = rows {2p, 2p+1} of column j, 2p+1 >= j
    // two lower-triangular matrices held entry by entry side by side (see the dense solver below)
    struct DualSystem { Pair<T> a[NV][NV]; };
    // Models with constraint rows factor M + h B next to the first sweep's system, entry by entry in pairs.
    static constexpr bool DUAL = has_rows() && !LEAF_FIRST;
    struct DampedFactor {                                // inv[j] = 1 / D_jj (root-first), 1 / L_jj (leaf-first)
        DualSystem D;        // half [1]: root-first factor, made next to the constraint system's
        PackedSystem P;      // root-first factor (row pairs)
        T L[NV][NV];         // leaf-first factor of a branching model
        T inv[NV];
    };
    static MPPI_HD void factor_damped(const T (&M)[NV][NV], DampedFactor& f) {
        T hb[NV];
        static_for<NV>([&](auto Ic) { hb[Ic] = T(MD::INTEGRATOR == 0 ? MD::H * MD::damping[Ic] : 0.0); });
        if constexpr (LEAF_FIRST) leaf_first_factor(M, hb, f.L, f.inv);
        else if constexpr (DUAL) {
            // no row in the whole wave (rare for a model that has rows): the same routine, its first system idle
            T none[NV], unused[NV];
            static_for<NV>([&](auto Ic) { none[Ic] = T(0); unused[Ic] = T(0); });
            dual_factor<false, false>(M, none, T(0), none, unused, hb, f.D, f.inv);
        } else chol_factor<false>(M, hb, T(0), hb, f.P, f.inv);
    }

    // exact active-set solve of the soft constraint rows: ftot = fs + J^T f with f >= 0 on the active rows.
    //   act      : active rows carried from the previous substep; reset() at the start of a rollout.  The
    //              iteration converges to the same unique solution from any start; starting from the
    //              previous set usually needs one sweep.
    //   damped   : receives the factor of M + h B, which the step needs afterwards; it is computed
    //              inside the first sweep, entry by entry next to the constraint system's own factor
    //              (one packed instruction serves both).
    // The loop is uniform over the wave: lanes whose set has settled repeat the same solve (same result)
    // until every lane has settled, so the only cross-lane operation is one vote per sweep.  A model with rows
    // always runs the first sweep, also in the rare substep in which no lane of the wave has a row (all weights
    // zero then: the sweep is the unconstrained solve): a test for that case is a branch on every substep.
    //   contact_wave / probe : whether some sample of the wave is within the contact margin in this substep, and the
    //              measurements its row is made from (contact_probe)
    static MPPI_HD void constraint_forces(const T (&M)[NV][NV], const T (&fs)[NV], const Rows& r, Active& act,
                                          T (&ftot)[NV], int& sweeps_out, DampedFactor& damped, const Frames& fr,
                                          const T (&qd)[NV], bool contact_wave, const ContactProbe& probe) {
        if constexpr (!has_rows()) {
            static_for<NV>([&](auto Ic) { ftot[Ic] = fs[Ic]; });
            factor_damped(M, damped);
            sweeps_out = 0;
        } else {
            int sweeps = 0;
            // active weight of every row: its D while active, 0 while inactive or absent
            Pair<T> dgp[NP];
            T idle_contact = T(0);      // weight withheld from the contact row at the end of this substep
            static_for<NP>([&](auto Pc) {
                dgp[Pc] = make_pair_of(sel(act.idle[Pc][0] > T(0), T(0), r.Dl[Pc][0]), sel(act.idle[Pc][1] > T(0), T(0), r.Dl[Pc][1]));
            });
            Pair<T> fsp[NP], tlp[NP];
            static_for<NP>([&](auto Pc) {
                constexpr int i1 = (2 * Pc + 1 < NV) ? 2 * Pc + 1 : 2 * Pc;
                fsp[Pc] = make_pair_of(fs[2 * Pc], fs[i1]);
                tlp[Pc] = make_pair_of(T(0), T(0));
            });
            T acc[NV], dg[NV], resc = T(0);
            T hb[NV];
            static_for<NV>([&](auto Ic) { hb[Ic] = T(MD::INTEGRATOR == 0 ? MD::H * MD::damping[Ic] : 0.0); });
            // the sweeps exist twice, with and without the contact row, chosen once per substep: a taken
            // branch costs a lone wave about as much as a dozen vector instructions
            auto sweep_all = [&](auto Wc) {
                constexpr bool WC = decltype(Wc)::value;
                // the contact row exists in this variant only (without it: fields nothing reads)
                ContactRow cr;
                T dc = T(0);
                if constexpr (WC) {
                    contact_row(fr, qd, probe, cr);
                    dc = sel(act.idle_contact > T(0), T(0), cr.Dc);
                }
                // residual signs decide the next active set; true if this lane's set changed.  Grow: rows may join the
                // set but not leave it (the late sweeps of a solve that keeps changing, see below).
                auto settle = [&](auto Grow) -> Mask<T> {
                    constexpr bool GROW = decltype(Grow)::value;
                    Mask<T> changed{};
                    static_for<NP>([&](auto Pc) {
                        constexpr int i1 = (2 * Pc + 1 < NV) ? 2 * Pc + 1 : 2 * Pc;
                        tlp[Pc] = make_pair_of(acc[2 * Pc], acc[i1]) - r.sal[Pc];
                        const Pair<T> side = tlp[Pc] * r.xs[Pc];
                        static_for<(i1 != 2 * Pc) ? 2 : 1>([&](auto Hc) {
                            constexpr int h = Hc, i = 2 * Pc + h;
                            if constexpr (MD::limited[i] != 0) {
                                const T nd = sel(side[h] > T(0), r.Dl[Pc][h], GROW ? dgp[Pc][h] : T(0));
                                MPPI_SWEEP_TRACE(sweeps, i, nd != dgp[Pc][h]);
                                changed |= (nd != dgp[Pc][h]);
                                dgp[Pc][h] = nd;
                            }
                        });
                    });
                    if constexpr (WC) {
                        T ja = T(0);
                        static_for<NV>([&](auto Ic) { ja += cr.Jc[Ic] * acc[Ic]; });
                        resc = ja - cr.arc;
                        const T nd = sel(resc < T(0), cr.Dc, GROW ? dc : T(0));
                        MPPI_SWEEP_TRACE(sweeps, NV, nd != dc);
                        changed |= (nd != dc);
                        dc = nd;
                    }
                    return changed;
                };
                // right-hand side and diagonal of the current set, handed to the solver as single values
                auto right_hand_side = [&]() {
                    static_for<NP>([&](auto Pc) {
                        const Pair<T> b = fsp[Pc] + dgp[Pc] * r.sal[Pc];
                        acc[2 * Pc] = b[0]; dg[2 * Pc] = dgp[Pc][0];
                        if constexpr (2 * Pc + 1 < NV) { acc[2 * Pc + 1] = b[1]; dg[2 * Pc + 1] = dgp[Pc][1]; }
                    });
                    if constexpr (WC) static_for<NV>([&](auto Ic) { acc[Ic] += dc * cr.arc * cr.Jc[Ic]; });
                };
                sweeps = 1;
                right_hand_side();
                dual_factor<WC, true>(M, dg, WC ? dc : T(0), cr.Jc, acc, hb, damped.D, damped.inv);
                Mask<T> changed = settle(std::false_type{});
                // Among the 64 samples of a wave one nearly always moves a row (a limit row released by a new torque,
                // a joint arriving at its limit), so the second sweep is laid out as the fall-through path.
                if (MPPI_LIKELY(wave_any(changed))) {
                    sweeps = 2;
                    right_hand_side();
                    chol_solve<WC>(M, dg, WC ? dc : T(0), cr.Jc, acc);
                    changed = settle(std::false_type{});
                    // Rows join and leave the set freely for the first FREE sweeps, which settles all but a few
                    // solves in a million.  The rest are rows whose residual is zero to rounding and whose sign
                    // differs between the system with the row and the system without it: such a row would enter and
                    // leave for ever.  From then on rows may only join (at most one sweep per row more); a row kept
                    // that way carries a force of rounding size.
                    constexpr int FREE = 6;
                    const int free_end = act.cap < FREE ? act.cap : FREE;
                    if (MPPI_UNLIKELY(wave_any(changed))) {
                        while (sweeps < free_end) {
                            sweeps++;
                            right_hand_side();
                            chol_solve<WC>(M, dg, WC ? dc : T(0), cr.Jc, acc);
                            changed = settle(std::false_type{});
                            if (!wave_any(changed)) break;
                        }
                        if (MPPI_UNLIKELY(wave_any(changed))) {
                            while (sweeps < act.cap) {
                                sweeps++;
                                right_hand_side();
                                chol_solve<WC>(M, dg, WC ? dc : T(0), cr.Jc, acc);
                                changed = settle(std::true_type{});
                                if (!wave_any(changed)) break;
                            }
                            if (MPPI_UNLIKELY(wave_any(changed))) sweeps = act.cap + 1;    // gave up: the forces belong to the last set tried
                        }
                    }
                }
                // smooth force plus the force of every active row on its joint
                static_for<NP>([&](auto Pc) {
                    const Pair<T> f = fsp[Pc] - dgp[Pc] * tlp[Pc];
                    ftot[2 * Pc] = f[0];
                    if constexpr (2 * Pc + 1 < NV) ftot[2 * Pc + 1] = f[1];
                });
                if constexpr (WC) {
                    const T f = -dc * resc;
                    static_for<NV>([&](auto Ic) { ftot[Ic] += cr.Jc[Ic] * f; });
                    idle_contact = cr.Dc - dc;
                }
            };
            if constexpr (MD::contact_link >= 0) {
                if (MPPI_UNLIKELY(contact_wave)) sweep_all(std::true_type{});
                else sweep_all(std::false_type{});
            } else {
                sweep_all(std::false_type{});
            }
            // the weight withheld from every row: positive for the inactive ones, zero for active and absent rows
            static_for<NP>([&](auto Pc) { act.idle[Pc] = r.Dl[Pc] - dgp[Pc]; });
            act.idle_contact = idle_contact;
            sweeps_out = sweeps;
        }
    }

    // qpos advanced from q0 by h along `vel`: hinge and slide coordinates move linearly, a ball joint's quaternion
    // is turned by the body-frame rotation vector h w and renormalised.  q may be q0.
    static MPPI_HD void advance_pos(const T (&q0)[NQ], const T (&vel)[NV], T (&q)[NQ]) {
        static_for<NV>([&](auto Ic) {
            constexpr int i = Ic;
            constexpr int a = MD::qadr[i];
            if constexpr (MD::jtype[i] == 4) {               // a ball is integrated at its last link, all three rates known
                constexpr int i0 = i - 2;
                const T wx = vel[i0], wy = vel[i0 + 1], wz = vel[i0 + 2];
                const T n2 = wx * wx + wy * wy + wz * wz;
                const T n = sqrt_t(n2);
                const T half_angle = T(0.5 * MD::H) * n;
                T sh, ch;
                sincos_t(half_angle, sh, ch);
                const T g = sel(n2 > T(1e-24), sh * rcp_t(sel(n2 > T(1e-24), n, T(1))), T(0.5 * MD::H));      // sin(h n / 2) / n
                const T dx = g * wx, dy = g * wy, dz = g * wz;
                const T qw = q0[a], qx = q0[a + 1], qy = q0[a + 2], qz = q0[a + 3];
                const T r0 = qw * ch - qx * dx - qy * dy - qz * dz;
                const T r1 = qw * dx + qx * ch + qy * dz - qz * dy;
                const T r2 = qw * dy - qx * dz + qy * ch + qz * dx;
                const T r3 = qw * dz + qx * dy - qy * dx + qz * ch;
                const T inv = rsqrt_t(r0 * r0 + r1 * r1 + r2 * r2 + r3 * r3);
                q[a] = r0 * inv; q[a + 1] = r1 * inv; q[a + 2] = r2 * inv; q[a + 3] = r3 * inv;
            } else if constexpr (!is_ball(i)) {
                q[a] = q0[a] + T(MD::H) * vel[i];
            }
        });
    }

    // joint acceleration at (q, v, ctrl): the factor's matrix is M + h B for the Euler step (joint damping
    // implicit) and M alone for Runge-Kutta stages
    static MPPI_HD void acceleration(const T (&q)[NQ], const T (&qd)[NV], const T (&ctrl)[NV], bool want_hand, Out& out,
                                     Active& act, int& sweeps_out, T (&qacc)[NV]) {
        Frames fr;
        frames(q, fr);
        if constexpr (MD::hand_link >= 0) {
            if (want_hand) out.hand = hand_site(q, fr);
        }
        T fs[NV];
        smooth_force(q, fr, qd, ctrl, fs);
        T M[NV][NV];     // lower triangle used: M[i][j], i >= j
        mass_matrix(q, fr, M);
        Rows rows;
        build_rows(q, fr, qd, rows);
        ContactProbe probe;
        const bool contact_wave = contact_probe(q, fr, probe);
        DampedFactor f;
        constraint_forces(M, fs, rows, act, qacc, sweeps_out, f, fr, qd, contact_wave, probe);
        if constexpr (LEAF_FIRST) leaf_first_apply(f.L, f.inv, qacc);
        else if constexpr (DUAL) dual_apply(f.D, f.inv, qacc);
        else chol_apply(f.P, f.inv, qacc);
    }

    // (q, v, ctrl) -> (q', v') in place.
    //   want_hand : also evaluate the end-effector site at the configuration the simulator's derived quantities
    //               hold after the step: the last one its dynamics were evaluated at (Euler: the input q;
    //               Runge-Kutta: the fourth stage)
    //   act       : see constraint_forces
    static MPPI_HD void substep(T (&q)[NQ], T (&qd)[NV], const T (&ctrl)[NV], bool want_hand, Out& out,
                                Active& act, int& sweeps_out) {
        if constexpr (MD::INTEGRATOR == 0) {
            // semi-implicit Euler: v += h a with (M + h B) a = fs + fc, then q advanced along the new v
            T qacc[NV];
            acceleration(q, qd, ctrl, want_hand, out, act, sweeps_out, qacc);
            static_for<NV>([&](auto Ic) { qd[Ic] = qd[Ic] + T(MD::H) * qacc[Ic]; });
            advance_pos(q, qd, q);
        } else {
            // classical fourth-order Runge-Kutta over (qpos, qvel): stage states X_s = X_0 + h a_s (v_{s-1}, F_{s-1})
            // with a = (1/2, 1/2, 1), positions advanced on the manifold; result X_0 + h sum_s b_s (v_s, F_s),
            // b = (1, 2, 2, 1) / 6.  The four stages run through one copy of the dynamics code.
            T q0[NQ], v0[NV], dvel[NV], dacc[NV];
            static_for<NQ>([&](auto Ic) { q0[Ic] = q[Ic]; });
            static_for<NV>([&](auto Ic) { v0[Ic] = qd[Ic]; dvel[Ic] = T(0); dacc[Ic] = T(0); });
            int sweeps = 0;
#if defined(__clang__)
#pragma nounroll
#endif
            for (int s = 0; s < 4; s++) {
                T F[NV];
                int sw;
                acceleration(q, qd, ctrl, want_hand && s == 3, out, act, sw, F);
                sweeps = sw > sweeps ? sw : sweeps;
                const T b = (s == 0 || s == 3) ? T(1.0 / 6.0) : T(1.0 / 3.0);
                const T a = (s == 2) ? T(1) : T(0.5);          // coefficient of the stage that follows
                T va[NV];
                static_for<NV>([&](auto Ic) {
                    dvel[Ic] += b * qd[Ic];
                    dacc[Ic] += b * F[Ic];
                    va[Ic] = a * qd[Ic];
                });
                if (s < 3) {
                    advance_pos(q0, va, q);
                    static_for<NV>([&](auto Ic) { qd[Ic] = v0[Ic] + T(MD::H) * a * F[Ic]; });
                }
            }
            static_for<NV>([&](auto Ic) { qd[Ic] = v0[Ic] + T(MD::H) * dacc[Ic]; });
            advance_pos(q0, dvel, q);
            sweeps_out = sweeps;
        }
    }

    // bias forces: recursive Newton-Euler with zero joint acceleration, link-local coordinates.
    // Velocity and velocity-product acceleration travel as pairs, angular {omega, alpha} and linear {v, a}
    // (same transforms, same inertia products); force and moment travel back as a pair too.
    static MPPI_HD void bias_forces(const T (&q)[NQ], const Frames& fr, const T (&qd)[NV], T (&bias)[NV]) {
        using P = Pair<T>;
        Vec3<P> ang[NV], lin[NV];       // [0]: velocity, [1]: acceleration
        Vec3<P> fm[NV];                 // [0]: force, [1]: moment about the link origin
        Vec3<T> wc[NV], vc[NV];         // velocity that carries the joint axis (see below)
        static_for<NV>([&](auto Ic) {
            constexpr int i = Ic;
            constexpr int p = MD::parent[i];
            constexpr int A = MD::axis_id[i];
            const Vec3<T> r = off_<i>(q);
            Vec3<P> a, l;
            if constexpr (p < 0) {
                const P z = make_pair_of(T(0), T(0));
                a = {{z, z, z}};
                l = a;
                // gravity enters as an upward acceleration of the world
                if constexpr (MD::has_gravity) {
                    const Vec3<T> g = down<i>(fr, Vec3<T>{{T(-MD::gravity[0]), T(-MD::gravity[1]), T(-MD::gravity[2])}});
                    static_for<3>([&](auto Cn) { l.v[Cn][1] = g.v[Cn]; });
                }
            } else {
                a = down<i>(fr, ang[p]);
                l = down<i>(fr, add(lin[p], cross(ang[p], r)));
            }
            // joint rate along / about axis A; velocity-product acceleration V x (S qd), V the velocity that carries
            // the axis: the parent's for a hinge or slide.  The three axes of a ball joint are fixed in the body;
            // summed over the joint that is the same as carrying all three with the velocity in front of the ball
            // (the joint's own rate crossed with itself vanishes), which the first link of the ball hands on.
            Vec3<T> sq = {{T(0), T(0), T(0)}};
            sq.v[A] = T(MD::axis_sign[i]) * qd[i];
            const Vec3<T> wcar = [&]() -> Vec3<T> { if constexpr (MD::jtype[i] >= 3) return wc[p]; else return half<T>(a, 0); }();
            const Vec3<T> vcar = [&]() -> Vec3<T> { if constexpr (MD::jtype[i] >= 3) return vc[p]; else return half<T>(l, 0); }();
            if constexpr (MD::jtype[i] >= 2) { wc[i] = wcar; vc[i] = vcar; }      // handed on inside a ball joint only
            if constexpr (MD::jtype[i] == 1) {
                const Vec3<T> da = cross(wcar, sq);
                static_for<3>([&](auto Cn) { l.v[Cn][1] += da.v[Cn]; });
                l.v[A][0] += sq.v[A];
            } else if constexpr (MD::jtype[i] == 0) {
                // hinge: V x (s e_A) has the components +s V_W (at U) and -s V_U (at W); it goes into the acceleration
                // halves as one packed multiply-add per component, {0, s} times the velocity half of the other
                // component (the velocity halves receive + 0)
                constexpr int U = (A + 1) % 3, W = (A + 2) % 3;
                // ({0, s} by one packed multiply of a constant pair saves the zero's register copy and costs a lone
                // wave more than it saves: one more link in the dependent chain of every joint)
                const P sp = make_pair_of(T(0), sq.v[A]);
                const P aU = a.v[U], lU = l.v[U];
                a.v[U] = aU + sp * splat_half<0>(a.v[W]);
                a.v[W] = a.v[W] - sp * splat_half<0>(aU);
                l.v[U] = lU + sp * splat_half<0>(l.v[W]);
                l.v[W] = l.v[W] - sp * splat_half<0>(lU);
                a.v[A][0] += sq.v[A];
            } else {
                const Vec3<T> dw = cross(wcar, sq), dv = cross(vcar, sq);
                static_for<3>([&](auto Cn) { a.v[Cn][1] += dw.v[Cn]; l.v[Cn][1] += dv.v[Cn]; });
                a.v[A][0] += sq.v[A];
            }
            ang[i] = a; lin[i] = l;
            const T m = T(MD::mass[i]);
            const Vec3<T> h = mc_<i>();
            const T IO[6] = {T(MD::IO[i][0]), T(MD::IO[i][1]), T(MD::IO[i][2]), T(MD::IO[i][3]), T(MD::IO[i][4]), T(MD::IO[i][5])};
            // momentum {of the velocity, of the acceleration}: linear and angular about the link origin
            const Vec3<P> lm = add(scale(m, l), cross(a, h));
            const Vec3<P> am = add(sym_mul(IO, a), cross(h, l));
            const Vec3<T> w = half<T>(a, 0), v = half<T>(l, 0), lV = half<T>(lm, 0), nV = half<T>(am, 0);
            const Vec3<T> f = add(half<T>(lm, 1), cross(w, lV));
            const Vec3<T> n = add(half<T>(am, 1), add(cross(w, nV), cross(v, lV)));
            static_for<3>([&](auto Cn) { fm[i].v[Cn] = make_pair_of(f.v[Cn], n.v[Cn]); });
        });
        static_for<NV>([&](auto Ic) {
            constexpr int i = NV - 1 - Ic;
            constexpr int p = MD::parent[i];
            constexpr int A = MD::axis_id[i];
            bias[i] = T(MD::axis_sign[i]) * fm[i].v[A][MD::jtype[i] == 1 ? 0 : 1];
            if constexpr (p >= 0) {
                const Vec3<T> r = off_<i>(q);
                const Vec3<P> par = up<i>(fr, fm[i]);
                const Vec3<T> rf = cross(r, half<T>(par, 0));
                static_for<3>([&](auto Cn) {
                    fm[p].v[Cn] += par.v[Cn];
                    fm[p].v[Cn][1] += rf.v[Cn];
                });
            }
        });
    }

    // joint-space inertia (lower triangle, armature included): composite rigid bodies in link coordinates
    static MPPI_HD void mass_matrix(const T (&q)[NQ], const Frames& fr, T (&M)[NV][NV]) {
        SpInertia<T> Ic_[NV];
        // joints on different branches do not couple
        static_for<NV>([&](auto In) {
            static_for<In>([&](auto Jn) {
                if constexpr (!on_path(Jn, In)) M[In][Jn] = T(0);
            });
        });
        static_for<NV>([&](auto Ic) {
            constexpr int i = Ic;
            Ic_[i].m = T(MD::mass[i]);
            Ic_[i].h = mc_<i>();
            static_for<6>([&](auto En) { Ic_[i].I[En] = T(MD::IO[i][En]); });
        });
        static_for<NV>([&](auto Icn) {
            constexpr int i = NV - 1 - Icn;
            constexpr int p = MD::parent[i];
            constexpr int A = MD::axis_id[i], U = (A + 1) % 3, W = (A + 2) % 3;
            // column i of M: force produced by unit acceleration of joint i on composite body i
            {
                Vec3<T> ea = {{T(0), T(0), T(0)}};
                const T sgn = T(MD::axis_sign[i]);
                ea.v[A] = sgn;
                Vec3<T> n, f;
                if constexpr (MD::jtype[i] == 1) {
                    n = cross(Ic_[i].h, ea);
                    f = scale(Ic_[i].m, ea);
                    M[i][i] = Ic_[i].m + T(MD::armature[i]);
                } else {
                    n = {{sgn * Ic_[i].I[sym_idx(0, A)], sgn * Ic_[i].I[sym_idx(1, A)], sgn * Ic_[i].I[sym_idx(2, A)]}};
                    f = cross(ea, Ic_[i].h);
                    M[i][i] = Ic_[i].I[sym_idx(A, A)] + T(MD::armature[i]);
                }
                if constexpr (p >= 0) {
                    Vec3<Pair<T>> fm;
                    static_for<3>([&](auto Cn) { fm.v[Cn] = make_pair_of(f.v[Cn], n.v[Cn]); });
                    walk_up<i, i>(q, fr, fm, M);
                }
            }
            if constexpr (p >= 0) {
                // express composite i in the parent frame and add it to the parent's composite
                const T* I = Ic_[i].I;
                const Vec3<T> h = up<i>(fr, Ic_[i].h);
                T J[6];
                if constexpr (MD::jtype[i] == 2) {
                    // J = R I R^T with a full rotation matrix
                    const T (&R)[9] = rb_of<i>(fr);       // the matrix of link i (not of the loop counter NV - 1 - i)
                    T RI[3][3];
                    static_for<3>([&](auto Rn) {
                        static_for<3>([&](auto Cn) {
                            RI[Rn][Cn] = R[3 * Rn] * I[sym_idx(0, Cn)] + R[3 * Rn + 1] * I[sym_idx(1, Cn)] + R[3 * Rn + 2] * I[sym_idx(2, Cn)];
                        });
                    });
                    static_for<3>([&](auto Rn) {
                        static_for<3 - Rn>([&](auto Dn) {
                            constexpr int c2 = Rn + Dn;
                            J[sym_idx(Rn, c2)] = RI[Rn][0] * R[3 * c2] + RI[Rn][1] * R[3 * c2 + 1] + RI[Rn][2] * R[3 * c2 + 2];
                        });
                    });
                } else {
                    const T c = fr.cs(i), s = fr.sn(i);
                    const T cc = c * c, ss = s * s, sc = s * c;
                    const T Iuu = I[sym_idx(U, U)], Iww = I[sym_idx(W, W)], Iuw = I[sym_idx(U, W)];
                    const T Iau = I[sym_idx(A, U)], Iaw = I[sym_idx(A, W)];
                    J[sym_idx(A, A)] = I[sym_idx(A, A)];
                    J[sym_idx(A, U)] = c * Iau - s * Iaw;
                    J[sym_idx(A, W)] = s * Iau + c * Iaw;
                    J[sym_idx(U, U)] = cc * Iuu - T(2) * sc * Iuw + ss * Iww;
                    J[sym_idx(W, W)] = ss * Iuu + T(2) * sc * Iuw + cc * Iww;
                    J[sym_idx(U, W)] = sc * (Iuu - Iww) + (cc - ss) * Iuw;
                }
                const Vec3<T> r = off_<i>(q);
                const T m = Ic_[i].m;
                const T rr = r.v[0] * r.v[0] + r.v[1] * r.v[1] + r.v[2] * r.v[2];
                const T rh = r.v[0] * h.v[0] + r.v[1] * h.v[1] + r.v[2] * h.v[2];
                const T dg = m * rr + T(2) * rh;
                Ic_[p].m += m;
                Ic_[p].h = add(Ic_[p].h, add(h, scale(m, r)));
                static_for<3>([&](auto ax) {
                    constexpr int a = ax;
                    Ic_[p].I[sym_idx(a, a)] += J[sym_idx(a, a)] + dg - m * r.v[a] * r.v[a] - T(2) * r.v[a] * h.v[a];
                });
                Ic_[p].I[3] += J[3] - m * r.v[0] * r.v[1] - r.v[0] * h.v[1] - h.v[0] * r.v[1];
                Ic_[p].I[4] += J[4] - m * r.v[0] * r.v[2] - r.v[0] * h.v[2] - h.v[0] * r.v[2];
                Ic_[p].I[5] += J[5] - m * r.v[1] * r.v[2] - r.v[1] * h.v[2] - h.v[1] * r.v[2];
            }
        });

    }

    // carry the pair {force, moment about the origin of link I} of column COL up the tree, filling M[COL][ancestor]
    template <int COL, int I>
    static MPPI_HD void walk_up(const T (&q)[NQ], const Frames& fr, const Vec3<Pair<T>>& fm, T (&M)[NV][NV]) {
        constexpr int p = MD::parent[I];
        if constexpr (p >= 0) {
            constexpr int A = MD::axis_id[I];
            const Vec3<T> r = off_<I>(q);
            Vec3<Pair<T>> par = up<I>(fr, fm);
            add_cross_to_second(par, r, par);
            M[COL][p] = T(MD::axis_sign[p]) * par.v[MD::axis_id[p]][MD::jtype[p] == 1 ? 0 : 1];
            walk_up<COL, p>(q, fr, par, M);
        }
    }

    // Impedance, regulariser and reference acceleration of one constraint row.
    //   xs  : signed violation (a limit row: q - clamp(q, lo, hi); the contact row: distance - margin, negative)
    //   bvel: b times the velocity of the joint (limit row) or of J qvel (contact row)
    //   D   : 1 / R, R = (1 - d) / d * invweight
    //   sa  : reference acceleration -b vel - k d xs, in the coordinates bvel and xs are given in
    // The impedance d rises from d0 to dmax over the ramp u = |xs| / width in [0, 1]: y(u) = u^2 / mid below mid,
    // 1 - (1 - u)^2 / (1 - mid) above.  For mid = 1/2 the two branches are one expression,
    // y = 1/2 + 2 p with p = z (1 - |z|), z = u - 1/2, which needs no comparison (ramp_product: absolute values are
    // free operand modifiers of single instructions, so this part is not done in pairs); d and 1 - d are both
    // formed from p (not 1 - d by subtracting d from 1: d is within 5 % of 1 and the difference would lose four bits).
    static constexpr bool RAMP_ONE_EXPRESSION = MD::solimp[3] == 0.5 && MD::solimp[4] == 2.0;
    static MPPI_HD T ramp_product(T xs) {
        const T z = min_t(abs_t(xs) * T(1.0 / MD::solimp[2]) - T(0.5), T(0.5));
        return z - abs_t(z) * z;
    }
    // (E: T, or a pair of rows)
    template <class E> static MPPI_HD void row_from_ramp(E p, E xs, E bvel, E invweight, E& D, E& sa) {
        constexpr double half_rise = 0.5 * (MD::solimp[1] - MD::solimp[0]);
        const E imp = T(MD::solimp[0] + half_rise) + T(4.0 * half_rise) * p;
        const E omi = T(1.0 - MD::solimp[0] - half_rise) - T(4.0 * half_rise) * p;
        sa = -bvel - T(MD::solref_k) * (imp * xs);
        D = imp * rcp_t(omi * invweight);
    }
    static MPPI_HD void row_constants(T xs, T bvel, T invweight, T& D, T& sa) {
        static_assert(MD::solimp[4] == 2.0, "impedance power other than 2 is not compiled in");
        if constexpr (RAMP_ONE_EXPRESSION) {
            row_from_ramp(ramp_product(xs), xs, bvel, invweight, D, sa);
        } else {
            const T d0 = T(MD::solimp[0]), dmax = T(MD::solimp[1]), mid = T(MD::solimp[3]);
            const T u = min_t(abs_t(xs) * T(1.0 / MD::solimp[2]), T(1));
            const T w = T(1) - u;
            const T ylo = u * u * T(1.0 / MD::solimp[3]);
            const T yhi = T(1) - w * w * T(1.0 / (1.0 - MD::solimp[3]));
            const T y = sel(u <= mid, ylo, yhi);
            const T imp = d0 + y * (dmax - d0);
            sa = -bvel - T(MD::solref_k) * (imp * xs);
            D = imp * rcp_t((T(1) - imp) * invweight);
        }
    }

    // ---- dense solver -----------------------------------------------------------------------------
    // A = L D L^T with unit lower-triangular L, written right-looking: a finished column is applied to the rest
    // of the matrix at once, and the forward substitution rides along as one more row.  Every sum runs in the
    // order of the textbook left-looking form.  inv[j] = 1 / D_jj.  (Against Cholesky: a reciprocal instead of a
    // reciprocal square root per pivot, and one scaling by 1 / D per solve instead of a scaling by 1 / L_jj in
    // each of the two substitutions.)
    // Every column is held as pairs of rows {2p, 2p+1}: scaling a column, applying it to a later column and
    // the forward substitution are one packed instruction per pair with the multiplier broadcast (fp64: two
    // instructions, as before).  Halves above the diagonal (and row NV of an odd NV) carry unused values:
    // packed arithmetic never mixes the halves, so they stay where they are.

    // A = M + diag(dg) + dc * Jc Jc^T, lower triangle
    template <bool CONTACT>
    static MPPI_HD void build_system(const T (&M)[NV][NV], const T (&dg)[NV], T dc, const T (&Jc)[NV], PackedSystem& A) {
        static_for<NV>([&](auto Jn) {
            constexpr int j = Jn;
            static_for<NP - j / 2>([&](auto Pn) {
                constexpr int p = j / 2 + Pn;
                auto entry = [&](auto In) -> T {
                    constexpr int i = In;
                    if constexpr (i >= NV || i < j) return T(0);
                    else {
                        T a = M[i][j];
                        if constexpr (i == j) a += dg[i];
                        if constexpr (CONTACT) a += dc * Jc[i] * Jc[j];
                        return a;
                    }
                };
                if constexpr (2 * p < j) A.col[j][p] = half_pair<1>(entry(std::integral_constant<int, 2 * p + 1>{}));
                else if constexpr (2 * p + 1 >= NV) A.col[j][p] = half_pair<0>(entry(std::integral_constant<int, 2 * p>{}));
                else A.col[j][p] = make_pair_of(entry(std::integral_constant<int, 2 * p>{}), entry(std::integral_constant<int, 2 * p + 1>{}));
            });
        });
    }
    // column j of the factor (in place), applied to the columns right of it: A_ik -= L_ij A_kj
    template <int j>
    static MPPI_HD void eliminate_column(PackedSystem& A, T (&inv)[NV]) {
        inv[j] = rcp_t(A.col[j][j / 2][j % 2]);
        Pair<T> raw[NP];                    // the column before scaling: A_kj = L_kj D_jj
        static_for<NP - (j + 1) / 2>([&](auto Pn) {
            constexpr int p = (j + 1) / 2 + Pn;
            raw[p] = A.col[j][p];
            A.col[j][p] *= inv[j];
        });
        static_for<NV - 1 - j>([&](auto Kn) {
            constexpr int k = j + 1 + Kn;
            const Pair<T> m = splat_half<k % 2>(raw[k / 2]);
            static_for<NP - k / 2>([&](auto Pn) { constexpr int p = k / 2 + Pn; A.col[k][p] -= A.col[j][p] * m; });
        });
    }
    // the right-hand side as row pairs
    static MPPI_HD void pack_rhs(const T (&b)[NV], Pair<T> (&bp)[NP]) {
        static_for<NP>([&](auto Pn) {
            constexpr int p = Pn;
            if constexpr (2 * p + 1 < NV) bp[p] = make_pair_of(b[2 * p], b[2 * p + 1]);
            else bp[p] = half_pair<0>(b[2 * p]);
        });
    }
    // forward substitution step j (L y = b): y_j is final, later rows accumulate
    // (the pair that holds row j itself is updated along with the rows below it and is not read again)
    template <int j>
    static MPPI_HD void forward_column(const PackedSystem& L, Pair<T> (&bp)[NP], T (&y)[NV]) {
        y[j] = bp[j / 2][j % 2];
        static_for<NP - (j + 1) / 2>([&](auto Pn) { constexpr int p = (j + 1) / 2 + Pn; bp[p] -= L.col[j][p] * y[j]; });
    }
    // z = y / D, then x_i = z_i - sum_{k > i} L_ki x_k, last row first; y is overwritten with x
    static MPPI_HD void back_substitute(const PackedSystem& L, const T (&inv)[NV], T (&y)[NV]) {
        static_for<NV>([&](auto Kn) {
            constexpr int i = NV - 1 - Kn;
            T acc = y[i] * inv[i];
            static_for<NV - 1 - i>([&](auto Rn) { constexpr int k = NV - 1 - Rn; acc -= L.col[i][k / 2][k % 2] * y[k]; });
            y[i] = acc;
        });
    }

    // L D L^T factor of M + diag(dg) + dc * Jc Jc^T; M (lower triangle) is read, not modified.
    template <bool CONTACT>
    static MPPI_HD void chol_factor(const T (&M)[NV][NV], const T (&dg)[NV], T dc, const T (&Jc)[NV], PackedSystem& L,
                                    T (&inv)[NV]) {
        build_system<CONTACT>(M, dg, dc, Jc, L);
        static_for<NV>([&](auto Jn) { eliminate_column<Jn>(L, inv); });
    }
    // b <- (L D L^T)^-1 b
    static MPPI_HD void chol_apply(const PackedSystem& L, const T (&inv)[NV], T (&b)[NV]) {
        Pair<T> bp[NP];
        pack_rhs(b, bp);
        static_for<NV>([&](auto Jn) { forward_column<Jn>(L, bp, b); });
        back_substitute(L, inv, b);
    }
    // solve (M + diag(dg) + dc * Jc Jc^T) x = b; b is overwritten with x
    template <bool CONTACT>
    static MPPI_HD void chol_solve(const T (&M)[NV][NV], const T (&dg)[NV], T dc, const T (&Jc)[NV], T (&b)[NV]) {
        PackedSystem L;
        T inv[NV];
        Pair<T> bp[NP];
        build_system<CONTACT>(M, dg, dc, Jc, L);
        pack_rhs(b, bp);
        static_for<NV>([&](auto Jn) {
            eliminate_column<Jn>(L, inv);
            forward_column<Jn>(L, bp, b);
        });
        back_substitute(L, inv, b);
    }
    // ---- two systems at once ---------------------------------------------------------------------------
    // System 1 = M + diag(dg) + dc * Jc Jc^T with right-hand side b (SOLVE: b becomes the solution) and
    // system 2 = M + diag(d2), factored in the same right-looking order with every entry of the two held as one
    // pair: scaling and updates are one packed instruction for both systems, with no idle halves.  Half [1]
    // of A and inv2 are the factor of system 2 afterwards.
    template <bool CONTACT, bool SOLVE>
    static MPPI_HD void dual_factor(const T (&M)[NV][NV], const T (&dg)[NV], T dc, const T (&Jc)[NV], T (&b)[NV],
                                    const T (&d2)[NV], DualSystem& A, T (&inv2)[NV]) {
        static_for<NV>([&](auto In) {
            constexpr int i = In;
            static_for<i + 1>([&](auto Jn) {
                constexpr int j = Jn;
                T a1 = M[i][j], a2 = M[i][j];
                if constexpr (i == j) { a1 += dg[i]; a2 += d2[i]; }
                if constexpr (CONTACT) a1 += dc * Jc[i] * Jc[j];
                A.a[i][j] = make_pair_of(a1, a2);
            });
        });
        T inv1[NV];
        static_for<NV>([&](auto Jn) {
            constexpr int j = Jn;
            inv1[j] = rcp_t(A.a[j][j][0]);
            inv2[j] = rcp_t(A.a[j][j][1]);
            const Pair<T> ip = make_pair_of(inv1[j], inv2[j]);
            Pair<T> raw[NV];                // the column before scaling: A_kj = L_kj D_jj
            static_for<NV - 1 - j>([&](auto In) {
                constexpr int i = j + 1 + In;
                raw[i] = A.a[i][j];
                A.a[i][j] *= ip;
            });
            static_for<NV - 1 - j>([&](auto Kn) {
                constexpr int k = j + 1 + Kn;
                static_for<NV - k>([&](auto In) { constexpr int i = k + In; A.a[i][k] -= A.a[i][j] * raw[k]; });
            });
            if constexpr (SOLVE) {
                static_for<NV - 1 - j>([&](auto In) { constexpr int i = j + 1 + In; b[i] -= A.a[i][j][0] * b[j]; });
            }
        });
        if constexpr (SOLVE) {
            static_for<NV>([&](auto Kn) { b[Kn] *= inv1[Kn]; });
            static_for<NV>([&](auto Kn) {
                constexpr int k = NV - 1 - Kn;
                static_for<k>([&](auto In) { constexpr int i = k - 1 - In; b[i] -= A.a[k][i][0] * b[k]; });
            });
        }
    }
    // b <- (L D L^T)^-1 b with the factor of system 2
    static MPPI_HD void dual_apply(const DualSystem& A, const T (&inv)[NV], T (&b)[NV]) {
        static_for<NV>([&](auto Jn) {
            constexpr int j = Jn;
            static_for<NV - 1 - j>([&](auto In) { constexpr int i = j + 1 + In; b[i] -= A.a[i][j][1] * b[j]; });
        });
        static_for<NV>([&](auto Kn) { b[Kn] *= inv[Kn]; });
        static_for<NV>([&](auto Kn) {
            constexpr int k = NV - 1 - Kn;
            static_for<k>([&](auto In) { constexpr int i = k - 1 - In; b[i] -= A.a[k][i][1] * b[k]; });
        });
    }

    // ---- branch-aware solver ---------------------------------------------------------------------------
    // A = M + diag(dg) = L^T L with L lower triangular, eliminated from the last link to the first.  A link
    // couples only to its ancestors, and two ancestors of one link are ancestors of each other, so eliminating
    // leaves first creates no fill-in: L has the zero pattern of M, and limbs of a tree never meet in the
    // factor (root-first Cholesky of the same matrix fills every limb-limb block).  Row k of L is stored in
    // L[k][i], i < k; inv[k] = 1 / L_kk.
    static MPPI_HD void leaf_first_factor(const T (&M)[NV][NV], const T (&dg)[NV], T (&L)[NV][NV], T (&inv)[NV]) {
        static_for<NV>([&](auto In) {
            constexpr int i = In;
            static_for<i + 1>([&](auto Jn) {
                constexpr int j = Jn;
                if constexpr (i == j) L[i][j] = M[i][j] + dg[i];
                else if constexpr (coupled(i, j)) L[i][j] = M[i][j];
            });
        });
        static_for<NV>([&](auto Kn) {
            constexpr int k = NV - 1 - Kn;
            inv[k] = rsqrt_t(L[k][k]);
            static_for<k>([&](auto In) {
                constexpr int i = In;
                if constexpr (coupled(k, i)) L[k][i] *= inv[k];
            });
            static_for<k>([&](auto In) {
                constexpr int i = In;
                if constexpr (coupled(k, i)) {
                    static_for<i + 1>([&](auto Jn) {
                        constexpr int j = Jn;
                        if constexpr (coupled(k, j)) L[i][j] -= L[k][i] * L[k][j];
                    });
                }
            });
        });
    }
    // b <- (L^T L)^-1 b
    static MPPI_HD void leaf_first_apply(const T (&L)[NV][NV], const T (&inv)[NV], T (&b)[NV]) {
        // L^T y = b, from the last link down
        static_for<NV>([&](auto Kn) {
            constexpr int k = NV - 1 - Kn;
            b[k] *= inv[k];
            static_for<k>([&](auto In) {
                constexpr int i = In;
                if constexpr (coupled(k, i)) b[i] -= L[k][i] * b[k];
            });
        });
        // L x = y, from the root up
        static_for<NV>([&](auto Kn) {
            constexpr int k = Kn;
            T acc = b[k];
            static_for<k>([&](auto In) {
                constexpr int i = In;
                if constexpr (coupled(k, i)) acc -= L[k][i] * b[i];
            });
            b[k] = acc * inv[k];
        });
    }
};

}  // namespace mppi
