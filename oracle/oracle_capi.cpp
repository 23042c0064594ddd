// CPU oracle C entry points -- TEST INFRASTRUCTURE (checker only, never shipped).
// Restates the rollout loop of the reference: `do_env_rollout`
// (trajopt/utils.py:59-102) around `Reacher7DOFEnv._step`
// (trajopt/envs/reacher_env.py:43-85) with the simulator step of mjstep.hpp,
// and the scoring of `MPPI.score_trajectory` (trajopt/algos/mppi.py:91-108).
#include "mjstep.hpp"

#include <atomic>
#include <thread>
#include <vector>

using namespace oracle;

// ---- operation-counting scalar ------------------------------------------------
struct Counted {
    double v;
    Counted() : v(0) {}
    explicit Counted(double x) : v(x) {}
};
static thread_local long g_add = 0, g_mul = 0, g_div = 0, g_sqrt = 0, g_trig = 0, g_cmp = 0;
inline Counted operator+(Counted a, Counted b) { g_add++; return Counted(a.v + b.v); }
inline Counted operator-(Counted a, Counted b) { g_add++; return Counted(a.v - b.v); }
inline Counted operator*(Counted a, Counted b) { g_mul++; return Counted(a.v * b.v); }
inline Counted operator/(Counted a, Counted b) { g_div++; return Counted(a.v / b.v); }
inline bool operator<(Counted a, Counted b) { g_cmp++; return a.v < b.v; }
inline bool operator>(Counted a, Counted b) { g_cmp++; return a.v > b.v; }
inline bool operator<=(Counted a, Counted b) { g_cmp++; return a.v <= b.v; }
inline bool operator==(Counted a, Counted b) { return a.v == b.v; }
inline Counted sin(Counted a) { g_trig++; return Counted(std::sin(a.v)); }
inline Counted cos(Counted a) { g_trig++; return Counted(std::cos(a.v)); }
inline Counted sqrt(Counted a) { g_sqrt++; return Counted(std::sqrt(a.v)); }
inline Counted fabs(Counted a) { return Counted(std::fabs(a.v)); }
inline Counted pow(Counted a, Counted b) { g_mul++; return Counted(std::pow(a.v, b.v)); }

extern "C" {

int oracle_model_size(void) { return (int)sizeof(Model); }

void oracle_substep(const Model* md, const double* q, const double* v, const double* ctrl, double* q_out,
                    double* v_out, double* hand3, int* info3) {
    V3<double> hand = {0, 0, 0};
    StepInfo inf;
    substep<double>(*md, q, v, ctrl, q_out, v_out, &hand, &inf);
    if (hand3) { hand3[0] = hand.x; hand3[1] = hand.y; hand3[2] = hand.z; }
    if (info3) { info3[0] = inf.nrows; info3[1] = inf.nactive; info3[2] = inf.sweeps; }
}

void oracle_fk_hand(const Model* md, const double* q, double* hand3) {
    Kin<double>* k = new Kin<double>;
    forward_kinematics<double>(*md, q, *k);
    V3<double> hnd = site_world(*md, *k, md->hand_link, md->hand_pos);
    hand3[0] = hnd.x; hand3[1] = hnd.y; hand3[2] = hnd.z;
    delete k;
}

void oracle_mass_matrix(const Model* md, const double* q, const double* qd, double* M, double* bias) {
    Kin<double>* k = new Kin<double>;
    Dyn<double>* d = new Dyn<double>;
    forward_kinematics<double>(*md, q, *k);
    inertia_and_bias<double>(*md, *k, qd, *d);
    for (int i = 0; i < md->nv; i++) {
        for (int j = 0; j < md->nv; j++) M[i * md->nv + j] = d->M[i][j];
        bias[i] = d->bias[i];
    }
    delete k;
    delete d;
}

// one env.step: frame_skip substeps, then reward/obs from the stale site position.
// `env_timestep` is the value before the step; the reference row used is
// reference[env_timestep + 1] (reacher_env.py:46,61).
void oracle_env_step(const Model* md, const double* q, const double* v, const double* target, const double* act,
                     int env_timestep, int reward_type, const double* reference, int ref_len, double alpha,
                     double* q_out, double* v_out, double* obs, double* reward_out, int* info3) {
    const int nv = md->nv, nq = md->nq;
    double qa[MAXV], va[MAXV], qb[MAXV], vb[MAXV];
    std::memcpy(qa, q, nq * sizeof(double));
    std::memcpy(va, v, nv * sizeof(double));
    V3<double> hand = {0, 0, 0};
    StepInfo inf, acc = {0, 0, 0};
    for (int s = 0; s < md->frame_skip; s++) {
        substep<double>(*md, qa, va, act, qb, vb, &hand, &inf);
        std::memcpy(qa, qb, nq * sizeof(double));
        std::memcpy(va, vb, nv * sizeof(double));
        acc.nrows += inf.nrows; acc.nactive += inf.nactive; if (inf.sweeps > acc.sweeps) acc.sweeps = inf.sweeps;
    }
    double h3[3] = {hand.x, hand.y, hand.z};
    const double* ref_row = nullptr;
    if (reference && ref_len > 0) {
        int idx = env_timestep + 1;
        if (idx >= ref_len) idx = ref_len - 1;
        ref_row = reference + 3 * idx;
    }
    if (reward_out) *reward_out = reward(reward_type, h3, target, va, nv, ref_row, alpha);
    std::memcpy(q_out, qa, nq * sizeof(double));
    std::memcpy(v_out, va, nv * sizeof(double));
    if (obs) {
        for (int j = 0; j < nq; j++) obs[j] = qa[j];
        for (int j = 0; j < nv; j++) obs[nq + j] = va[j];
        for (int c = 0; c < 3; c++) { obs[nq + nv + c] = h3[c]; obs[nq + nv + 3 + c] = target[c]; }
    }
    if (info3) { info3[0] = acc.nrows; info3[1] = acc.nactive; info3[2] = acc.sweeps; }
}

// K rollouts of H env steps from one start state.  act is [K][H][nv].
// returns[k] = sum_t gamma^t r_t accumulated left to right with gamma**t by pow.
// Optional dumps: next_obs [K][H][2nv+6], rewards [K][H].
void oracle_rollout(const Model* md, const double* q0, const double* v0, const double* target, int timestep0,
                    const double* act, int K, int H, int reward_type, const double* reference, int ref_len,
                    double alpha, double gamma, double* returns, double* next_obs, double* rewards,
                    int nthreads) {
    const int nv = md->nv, nq = md->nq;
    const int od = nq + nv + 6;
    if (nthreads < 1) nthreads = 1;
    std::atomic<int> next(0);
    auto work = [&]() {
        std::vector<double> obs(od);
        for (;;) {
            int k = next.fetch_add(1);
            if (k >= K) break;
            double q[MAXV], v[MAXV], q2[MAXV], v2[MAXV];
            std::memcpy(q, q0, nq * sizeof(double));
            std::memcpy(v, v0, nv * sizeof(double));
            double R = 0.0;
            for (int t = 0; t < H; t++) {
                double r;
                oracle_env_step(md, q, v, target, act + ((size_t)k * H + t) * nv, timestep0 + t, reward_type,
                                reference, ref_len, alpha, q2, v2, obs.data(), &r, nullptr);
                std::memcpy(q, q2, nq * sizeof(double));
                std::memcpy(v, v2, nv * sizeof(double));
                R += std::pow(gamma, (double)t) * r;
                if (rewards) rewards[(size_t)k * H + t] = r;
                if (next_obs) std::memcpy(next_obs + ((size_t)k * H + t) * od, obs.data(), od * sizeof(double));
            }
            returns[k] = R;
        }
    };
    if (nthreads == 1) {
        work();
    } else {
        std::vector<std::thread> th;
        for (int i = 0; i < nthreads; i++) th.emplace_back(work);
        for (auto& t : th) t.join();
    }
}

// algorithmic operation count of one substep at the given state
void oracle_count_ops(const Model* md, const double* q, const double* v, const double* ctrl, long* counts6) {
    Counted cq[MAXV], cv[MAXV], cc[MAXV], qo[MAXV], vo[MAXV];
    for (int j = 0; j < md->nq; j++) cq[j] = Counted(q[j]);
    for (int j = 0; j < md->nv; j++) { cv[j] = Counted(v[j]); cc[j] = Counted(ctrl[j]); }
    g_add = g_mul = g_div = g_sqrt = g_trig = g_cmp = 0;
    V3<Counted> hand;
    StepInfo inf;
    substep<Counted>(*md, cq, cv, cc, qo, vo, &hand, &inf);
    counts6[0] = g_add; counts6[1] = g_mul; counts6[2] = g_div; counts6[3] = g_sqrt; counts6[4] = g_trig; counts6[5] = g_cmp;
}

}  // extern "C"
