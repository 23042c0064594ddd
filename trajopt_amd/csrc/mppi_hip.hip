// Device-resident MPPI planning step for MI355X (gfx950): the C ABI and the model-independent kernels.
// The per-model kernels live in kernels.hpp and are compiled once per model (model_ops.hip).
//
// Kernels (one lane = one sampled action sequence; wave = 64 samples):
//   noise_upload_kernel / noise_philox_kernel   sigma scaling + recursive 3-tap filter, writes eps[t][j][k]
//   rollout_kernel<T, MD, LEAN>                 H env steps x FRAME_SKIP substeps, fused reward + discounted return
//   update_rows_kernel<T, NV, ROWS>             max / arg-max over K, S_k = exp(kappa (R_k - Rmax)), sum S, sum S eps:
//                                               the whole update of a single rank, or a shard's statistics
//   update_weights_kernel + update_steps_kernel the same for many planners in one handle (weights once, a sub-block per step)
//   update_chunk_{max,weights,sums,finish}      the same for one planner of more than 4096 samples (rows x chunks of samples)
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
#include <dlfcn.h>

#include <cfloat>
#include <cmath>
#include <climits>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <functional>
#include <mutex>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/mppi_hip.h"
#include "numpy_stream.h"
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

#include "kernels.hpp"

extern "C" void mppi_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
    philox4x32_10(c, key[0], key[1]);
    for (int i = 0; i < 4; i++) out[i] = c[i];
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

// alpha and the reference length of every planner of a handle (the rest of the device state is left alone: it
// may be ahead of the host's copy after device-resident steps)
__global__ void set_reference_fields_kernel(DevState* __restrict__ st, int P, double alpha, int ref_len) {
    for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < P; b += gridDim.x * blockDim.x) {
        st[b].alpha = alpha;
        st[b].ref_len = ref_len;
    }
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
// `MPPI.update` without any allocation: the weights S_k and their sum live in a fixed per-GPU buffer
// ------------------------------------------------------------------------------------------
// [0] Rmax, [1] sum S, [NOALLOC_HEAD + k] S_k.  Every GPU the library runs on holds its own copy of this array
// until the process ends; host code reaches it only through the address hipGetSymbolAddress returns for the
// current device (noalloc_scratch below), never through the array's name.
constexpr int NOALLOC_HEAD = 32;
static __device__ __align__(256) double scratch[MPPI_NOALLOC_MAX_K + NOALLOC_HEAD];

// S_k = exp(kappa (R_k - Rmax)) and sum S, in the summation order of update_direct_kernel (one block of 256)
__global__ void __launch_bounds__(256) noalloc_weights_kernel(const double* __restrict__ R, int K, double kappa,
                                                              double* __restrict__ ws) {
    __shared__ double s_v[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double bv = -DBL_MAX;
    for (int k = tid; k < K; k += 256) bv = R[k] > bv ? R[k] : bv;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { const double o = shfl_xor_d(bv, m); bv = o > bv ? o : bv; }
    if (lane == 0) s_v[wave] = bv;
    __syncthreads();
    bv = s_v[0];
    for (int w = 1; w < 4; w++) bv = s_v[w] > bv ? s_v[w] : bv;
    __syncthreads();
    double den = 0.0;
    for (int k = tid; k < K; k += 256) {
        const double s = exp(kappa * (R[k] - bv));
        ws[NOALLOC_HEAD + k] = s;
        den += s;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) den += shfl_xor_d(den, m);
    if (lane == 0) s_v[wave] = den;
    __syncthreads();
    if (tid == 0) { ws[0] = bv; ws[1] = ((s_v[0] + s_v[1]) + s_v[2]) + s_v[3]; }
}
// out[hm] = sum_k S_k act[k][hm] / (sum S + 1e-6): block hm, same order as update_direct_kernel
__global__ void __launch_bounds__(256) noalloc_sum_kernel(const double* __restrict__ act, const double* __restrict__ ws,
                                                          int K, int HM, double* __restrict__ out) {
    __shared__ double s_v[4];
    const int hm = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double num = 0.0;
    for (int k = tid; k < K; k += 256) {
        const double s = ws[NOALLOC_HEAD + k];
        num += s * act[(size_t)k * HM + hm];
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) num += shfl_xor_d(num, m);
    if (lane == 0) s_v[wave] = num;
    __syncthreads();
    if (tid == 0) out[hm] = (((s_v[0] + s_v[1]) + s_v[2]) + s_v[3]) / (ws[1] + 1e-6);
}

// Device address of `scratch` on the given device (which the caller has made current), resolved once per device.
static double* g_scratch_ptr[16];
static std::mutex g_scratch_mu;
static int noalloc_scratch(int device, double** out) {
    if (device < 0 || device >= 16) return fail(-1, "device ordinal out of range");
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[device]) {
        void* ptr = nullptr;
        HIP_OK(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
        if (!ptr) return fail(-2, "hipGetSymbolAddress returned a null address for the update workspace");
        g_scratch_ptr[device] = (double*)ptr;
    }
    *out = g_scratch_ptr[device];
    return 0;
}

// one table per compiled model (model_ops.hip)
#if defined(MPPI_ONLY_MODEL)          // one-model builds for kernel experiments (never the shipped library)
#define MPPI_OPS_CAT2(a, b) a##b
#define MPPI_OPS_CAT(a, b) MPPI_OPS_CAT2(a, b)
extern const ModelOps MPPI_OPS_CAT(mppi_ops_, MPPI_ONLY_MODEL);
static const ModelOps* const g_ops[] = {&MPPI_OPS_CAT(mppi_ops_, MPPI_ONLY_MODEL)};
#elif defined(MPPI_ONLY_REACHER)    // quick builds for kernel experiments (make quick)
extern const ModelOps mppi_ops_Reacher7Dof;
static const ModelOps* const g_ops[] = {&mppi_ops_Reacher7Dof};
#else
extern const ModelOps mppi_ops_Reacher7Dof;
extern const ModelOps mppi_ops_Reacher7DofFree, mppi_ops_Tree18Dof, mppi_ops_Tree18DofQuat, mppi_ops_Tree18DofRk4;
static const ModelOps* const g_ops[] = {&mppi_ops_Reacher7Dof, &mppi_ops_Reacher7DofFree, &mppi_ops_Tree18Dof,
                                        &mppi_ops_Tree18DofQuat, &mppi_ops_Tree18DofRk4};
#endif

// ------------------------------------------------------------------------------------------
// handles
// ------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------
// RCCL, bound at run time: the process may already hold the copy PyTorch ships, and planners that are not
// sharded never need it
// ------------------------------------------------------------------------------------------
struct RcclUniqueId { char internal[128]; };
struct RcclApi {
    int (*GetUniqueId)(RcclUniqueId*);
    int (*CommInitRank)(void**, int, RcclUniqueId, int);
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t);
    int (*CommDestroy)(void*);
    const char* (*GetErrorString)(int);
};
constexpr int RCCL_FLOAT64 = 8;
static const RcclApi* rccl_api(std::string& why) {
    static std::mutex mu;
    static RcclApi api;
    static bool loaded = false;
    std::lock_guard<std::mutex> lock(mu);
    if (loaded) return &api;
    void* h = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
        h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) { why = "RCCL library not found (librccl.so)"; return nullptr; }
    api.GetUniqueId = (int (*)(RcclUniqueId*))dlsym(h, "ncclGetUniqueId");
    api.CommInitRank = (int (*)(void**, int, RcclUniqueId, int))dlsym(h, "ncclCommInitRank");
    api.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(h, "ncclAllGather");
    api.CommDestroy = (int (*)(void*))dlsym(h, "ncclCommDestroy");
    api.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
    if (!api.GetUniqueId || !api.CommInitRank || !api.AllGather || !api.CommDestroy || !api.GetErrorString) {
        why = "RCCL library lacks an expected entry point";
        return nullptr;
    }
    loaded = true;
    return &api;
}
#define RCCL_OK(api, expr)                                                                             \
    do {                                                                                               \
        const int r_ = (expr);                                                                         \
        if (r_ != 0) return fail(-7, std::string(#expr) + ": " + (api)->GetErrorString(r_));           \
    } while (0)

struct mppi_model {
    const ModelOps* ops;
    mppi_model_desc desc;
    // Per-device scratch and stream of the single-step entry points (made on first use).  Calls on one device take
    // turns (mu): environments of a process share the model handle and may be stepped from several host threads.
    double* d_io[16];
    hipStream_t io_stream[16];
    std::mutex mu[16];
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
    double* h_stage;       // page-locked host buffer of the same size, made on request
    hipStream_t copy_stream;       // uploads that run beside the planner's own work (mppi_draw_noise)
    hipEvent_t ev_raw_free;        // recorded on the planner's stream behind the last kernel that read d_raw
    hipEvent_t ev_upload_done;     // recorded on copy_stream behind the last piece of a streamed upload
    bool raw_pending;              // d_raw holds (or is receiving) draws that mppi_commit_noise has not filtered yet
    double* d_weights;     // the update's scratch (update_scratch_doubles): weights, Rmax, arg-max, ...: large K only
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
    int env_reward_type;   // reward the real environment's step reports (mppi_set_env_reward); -1: the rollouts' own
    double* d_log;         // [LOG_CAP][3M+7] real-env log ring of mppi_advance
    int* d_log_count;
    double* d_last_row;    // [P][3M+7] the newest log row of every planner (mppi_read_step)
    double* h_step;        // page-locked [P][3M+7 + H*M]: where mppi_read_step lands its two copies
    // Host mirror of a single planner's newest step (advance_block writes it): [3M+7 | H*M | count of steps logged] in
    // page-locked host memory the device writes to directly; d_mirror is the same memory as the device sees it.
    double* h_mirror;
    double* d_mirror;
    uint64_t advances;     // real-env steps enqueued so far (= the count the device will have logged after them) ...
    uint64_t mirror_at;    // ... and the value of that count after the last step that was asked to write the mirror
    bool mirror_wanted;    // somebody reads steps back (mppi_read_step has been called): single steps write the mirror
    hipStream_t adv_stream;        // stream the last advance was enqueued on
    int* d_next_step;      // [P] index of the planning step after the one being rolled out
    uint32_t* d_noise_key; // [P][2] stream key the Philox noise in d_eps was drawn with (written by the noise kernels)
    bool eps_philox;       // d_eps holds Philox noise (the update may draw it again instead of reading it)
    bool noise_ahead;      // eps already holds the Philox noise of the coming planning step ...
    uint64_t noise_ahead_seed;   // ... drawn with this seed
    int explicit_act;
    FilterCoefs fc;
    // Captured graphs carry kernel arguments (kappa, gamma, reward type, kernel variant, buffer addresses, critic).
    // Every setter that changes one of them bumps cfg_gen; a graph is replayed only while the generation it was
    // captured at is still the current one, each graph on its own.
    uint64_t cfg_gen;
    hipGraphExec_t graph;           // one iteration: rollouts -> update
    uint64_t graph_gen;             // 0: never captured
    hipGraphExec_t step_graph;      // whole planning step: noise -> niter x (rollout, update) -> advance
    hipGraphExec_t step_graph_run;  // PLAN_RUN consecutive planning steps as one graph (made when a call asks for that many)
    uint64_t step_graph_gen;
    int step_graph_niter;
    uint64_t step_graph_seed;
    // exchange of the shard statistics between the ranks of a sharded planner (mppi_comm_init)
    void* comm;                     // ncclComm_t, or null: the planner is not sharded
    int comm_rank, comm_nranks;
    double* d_gathered;             // [nranks][P][3 + HM]; d_part then points at this rank's row of it (in-place gather)
    double* d_part_alloc;           // the allocation d_part pointed at before
    bool profiling;
    std::vector<hipEvent_t>* prof_events;   // start/stop pairs around every rollout launch
    std::vector<DevState>* h_state;         // host mirror of the fields the host sets (alpha, ref_len, start state)
};

static hipStream_t pick(mppi_planner* p, void* s) { return s ? (hipStream_t)s : p->stream; }

// where the noise in d_eps comes from decides which update kernel a captured iteration holds
static void set_eps_source(mppi_planner* p, bool philox) {
    if (p->eps_philox != philox) p->cfg_gen++;
    p->eps_philox = philox;
}

// Synchronous copies and fills, made on a stream of the caller's and waited for there.  Never on the legacy (null)
// stream: while any thread of the process captures a graph, the runtime refuses every operation on the legacy stream
// ("would make the legacy stream depend on a capturing ... stream") and spoils that capture too, so planners driven
// from different host threads would break each other at allocation and read-back time.
static hipError_t copy_now(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s) {
    const hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, s);
    return e != hipSuccess ? e : hipStreamSynchronize(s);
}
static hipError_t zero_now(void* dst, size_t bytes, hipStream_t s) {
    const hipError_t e = hipMemsetAsync(dst, 0, bytes, s);
    return e != hipSuccess ? e : hipStreamSynchronize(s);
}

// Input arrays may be host memory or device memory (a torch tensor's data_ptr()): bulk copies use hipMemcpyDefault,
// which takes the direction from the addresses; the few small arrays the host itself has to look at come through here.
static bool on_device(const void* ptr) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, ptr) != hipSuccess) {
        (void)hipGetLastError();            // an ordinary host pointer is unknown to the runtime: not an error
        return false;
    }
    return a.type == hipMemoryTypeDevice;
}
static int host_view(const double* src, size_t n, std::vector<double>& tmp, const double** out, hipStream_t s) {
    if (!on_device(src)) { *out = src; return 0; }
    tmp.resize(n);
    HIP_OK(copy_now(tmp.data(), src, n * 8, hipMemcpyDeviceToHost, s));
    *out = tmp.data();
    return 0;
}

// reward type of the real environment's step inside mppi_advance / mppi_plan: the rollouts' own unless
// mppi_set_env_reward chose another
static int env_reward(const mppi_planner* p) { return p->env_reward_type >= 0 ? p->env_reward_type : p->cfg.reward_type; }

// tracking / cooling rewards read reference[timestep + 1 .. timestep + steps]: the reference raises IndexError past
// the end (trajopt/envs/reacher_env.py:61,75); -5 here, from the host's copy of the time step (kept in step with
// device-resident advances)
static int reference_covers(const mppi_planner* p, int steps, int reward_type) {
    if (reward_type != REWARD_TRACKING && reward_type != REWARD_COOLING) return 0;
    // (test hook, looked up at every call: with MPPI_NO_HOST_REF_CHECK=1 the kernels' own report,
    // MPPI_STATUS_REF_END, can be observed)
    const char* e = getenv("MPPI_NO_HOST_REF_CHECK");
    if (e && e[0] == '1') return 0;
    for (int b = 0; b < p->P; b++) {
        const DevState& st = (*p->h_state)[b];
        if (st.ref_len <= 0) return fail(-1, "tracking / cooling reward needs a reference (mppi_set_reference)");
        if (st.timestep + steps >= st.ref_len)
            return fail(-5, "reference index out of range: time step " + std::to_string(st.timestep) + " + " +
                                std::to_string(steps) + " steps, reference of " + std::to_string(st.ref_len) + " rows");
    }
    return 0;
}

extern "C" void mppi_planner_destroy(mppi_planner* p);

static int planner_allocate(mppi_planner* p) {
    const mppi_planner_cfg* cfg = &p->cfg;
    const size_t P = p->P;
    const size_t K = cfg->K * P, HM = p->HM * P, OD = p->nq + p->nv + 6;   // sizes over all planners
    HIP_OK(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
    HIP_OK(hipMalloc(&p->d_state, sizeof(DevState) * P));
    HIP_OK(zero_now(p->d_state, sizeof(DevState) * P, p->stream));
    HIP_OK(hipMalloc(&p->d_base, HM * 8));
    HIP_OK(hipMalloc(&p->d_base_tmp, HM * 8));
    HIP_OK(hipMalloc(&p->d_mean, MPPI_MAXV * 8));
    HIP_OK(copy_now(p->d_mean, cfg->mean, MPPI_MAXV * 8, hipMemcpyHostToDevice, p->stream));
    HIP_OK(hipMalloc(&p->d_eps, (size_t)p->HM * K * p->esz));
    HIP_OK(zero_now(p->d_eps, (size_t)p->HM * K * p->esz, p->stream));
    HIP_OK(hipMalloc(&p->d_R, K * p->esz));
    HIP_OK(zero_now(p->d_R, K * p->esz, p->stream));
    HIP_OK(hipMalloc(&p->d_part, (3 * P + HM) * 8));
    HIP_OK(zero_now(p->d_part, (3 * P + HM) * 8, p->stream));
    HIP_OK(hipMalloc(&p->d_result, 2 * 8 * P));
    HIP_OK(zero_now(p->d_result, 2 * 8 * P, p->stream));
    if (update_wants_weights(cfg->K, cfg->H, (int)P))
        HIP_OK(hipMalloc(&p->d_weights, update_scratch_doubles(cfg->K, cfg->H, (int)P, p->nv) * 8));
    HIP_OK(hipMalloc(&p->d_status, 16));
    HIP_OK(zero_now(p->d_status, 16, p->stream));
    HIP_OK(hipMalloc(&p->d_log, P * MPPI_LOG_CAP * (p->nq + 2 * p->nv + 7) * 8));
    HIP_OK(hipMalloc(&p->d_log_count, 4 * P));
    HIP_OK(zero_now(p->d_log_count, 4 * P, p->stream));
    HIP_OK(hipMalloc(&p->d_next_step, 4 * P));
    HIP_OK(zero_now(p->d_next_step, 4 * P, p->stream));
    HIP_OK(hipMalloc(&p->d_noise_key, 8 * P));
    HIP_OK(zero_now(p->d_noise_key, 8 * P, p->stream));
    HIP_OK(hipMalloc(&p->d_last_row, P * (p->nq + 2 * p->nv + 7) * 8));
    HIP_OK(zero_now(p->d_last_row, P * (p->nq + 2 * p->nv + 7) * 8, p->stream));
    HIP_OK(hipHostMalloc(&p->h_step, (P * (p->nq + 2 * p->nv + 7) + HM) * 8, hipHostMallocDefault));
    if (P == 1) {
        const size_t n = (size_t)(p->nq + 2 * p->nv + 7) + p->HM + 1;
        HIP_OK(hipHostMalloc(&p->h_mirror, n * 8, hipHostMallocMapped | hipHostMallocCoherent));
        std::memset(p->h_mirror, 0, n * 8);
        HIP_OK(hipHostGetDevicePointer((void**)&p->d_mirror, p->h_mirror, 0));
    }
    if (cfg->dump) {
        HIP_OK(hipMalloc(&p->d_obs, K * cfg->H * OD * p->esz));
        HIP_OK(hipMalloc(&p->d_rew, K * cfg->H * p->esz));
    }
    // act_sequence = ones((H, m)) * mean
    std::vector<double> base(HM);
    for (size_t t = 0; t < (size_t)cfg->H * P; t++)
        for (int j = 0; j < p->nv; j++) base[t * p->nv + j] = cfg->mean[j];
    HIP_OK(copy_now(p->d_base, base.data(), HM * 8, hipMemcpyHostToDevice, p->stream));
    for (int j = 0; j < MPPI_MAXV; j++) p->fc.sigma[j] = cfg->sigma[j];
    p->fc.b0 = cfg->beta[0]; p->fc.b1 = cfg->beta[1]; p->fc.b2 = cfg->beta[2];
    return 0;
}

extern "C" {

// Bookkeeping of the host mirror (see mppi_planner::h_mirror): an enqueued real-env step that was asked to write it
// makes it current; anything else that changes act_sequence on the device makes it stale.
static double* mirror_for(mppi_planner* p, bool wanted) { return (wanted && p->mirror_wanted && p->P == 1) ? p->d_mirror : nullptr; }
static void advanced(mppi_planner* p, int steps, bool last_wrote_mirror, hipStream_t s) {
    p->advances += (uint64_t)steps;
    p->mirror_at = (last_wrote_mirror && p->mirror_wanted && p->P == 1 && p->d_mirror) ? p->advances : 0;
    p->adv_stream = s;
    for (DevState& st : *p->h_state) st.timestep += steps;
}
static void act_sequence_changed(mppi_planner* p) { p->mirror_at = 0; }

int mppi_version(void) { return MPPI_ABI_VERSION; }
const char* mppi_last_error(void) { return g_err.c_str(); }

int mppi_model_create(const mppi_model_desc* desc, mppi_model** out) try {
    if (!desc || !out) return fail(-1, "null argument");
    std::string why = "unknown model name";
    for (const ModelOps* op : g_ops) {
        const ModelOps& o = *op;
        if (std::strncmp(desc->name, o.name, sizeof(desc->name)) != 0) continue;
        if (!o.matches(desc, why)) return fail(-3, std::string("model '") + o.name + "': " + why);
        mppi_model* m = new mppi_model();
        m->ops = &o;
        m->desc = *desc;
        for (auto& x : m->d_io) x = nullptr;
        for (auto& x : m->io_stream) x = nullptr;
        *out = m;
        return 0;
    }
    return fail(-3, std::string("no kernel compiled for model '") + std::string(desc->name, strnlen(desc->name, 64)) + "'");
} catch (const std::exception& e) {
    return fail(-8, std::string("host exception: ") + e.what());
}

void mppi_model_destroy(mppi_model* m) {
    if (!m) return;
    for (int d = 0; d < 16; d++)
        if (m->d_io[d]) {
            (void)hipSetDevice(d);
            (void)hipFree(m->d_io[d]);
            if (m->io_stream[d]) (void)hipStreamDestroy(m->io_stream[d]);
        }
    delete m;
}

int mppi_planner_create(const mppi_model* m, const mppi_planner_cfg* cfg, mppi_planner** out) try {
    if (!m || !cfg || !out) return fail(-1, "null argument");
    if (cfg->K < 1 || cfg->H < 1) return fail(-1, "K and H must be positive");
    if (cfg->dtype != MPPI_F32 && cfg->dtype != MPPI_F64) return fail(-1, "dtype must be MPPI_F32 or MPPI_F64");
    if (cfg->sweep_cap < 0) return fail(-1, "sweep_cap must not be negative");
    HIP_OK(hipSetDevice(cfg->device));
    {   // set-up time is when the no-allocation update's workspace address is looked up for this device
        double* ws;
        const int rc = noalloc_scratch(cfg->device, &ws);
        if (rc) return rc;
    }
    mppi_planner* p = new mppi_planner();
    std::memset(p, 0, sizeof(*p));
    p->model = m;
    p->cfg = *cfg;
    p->nv = m->ops->nv;
    p->nq = m->ops->nq;
    p->HM = cfg->H * p->nv;
    p->P = cfg->P > 1 ? cfg->P : 1;
    p->cfg.P = p->P;
    if (p->cfg.sweep_cap == 0) p->cfg.sweep_cap = 16;
    p->esz = cfg->dtype == MPPI_F32 ? 4 : 8;
    p->cfg_gen = 1;
    p->env_reward_type = -1;
    p->prof_events = new std::vector<hipEvent_t>();
    p->h_state = new std::vector<DevState>(p->P);
    std::memset(p->h_state->data(), 0, sizeof(DevState) * p->P);
    // any failure below releases what was allocated before it
    const int rc = planner_allocate(p);
    if (rc) {
        const std::string msg = g_err;
        mppi_planner_destroy(p);
        return fail(rc, msg);
    }
    *out = p;
    return 0;
} catch (const std::exception& e) {
    return fail(-8, std::string("host exception: ") + e.what());
}

void mppi_planner_destroy(mppi_planner* p) {
    if (!p) return;
    (void)hipSetDevice(p->cfg.device);
    if (p->stream) (void)hipStreamSynchronize(p->stream);
    if (p->prof_events) for (hipEvent_t e : *p->prof_events) (void)hipEventDestroy(e);
    delete p->prof_events;
    delete p->h_state;
    if (p->graph) (void)hipGraphExecDestroy(p->graph);
    if (p->step_graph) (void)hipGraphExecDestroy(p->step_graph);
    if (p->step_graph_run) (void)hipGraphExecDestroy(p->step_graph_run);
    if (p->comm) {
        std::string why;
        if (const RcclApi* api = rccl_api(why)) (void)api->CommDestroy(p->comm);
    }
    if (p->d_gathered) (void)hipFree(p->d_gathered);
    if (p->h_stage) (void)hipHostFree(p->h_stage);
    if (p->h_step) (void)hipHostFree(p->h_step);
    if (p->h_mirror) (void)hipHostFree(p->h_mirror);
    if (p->copy_stream) { (void)hipStreamSynchronize(p->copy_stream); (void)hipStreamDestroy(p->copy_stream); }
    if (p->ev_raw_free) (void)hipEventDestroy(p->ev_raw_free);
    if (p->ev_upload_done) (void)hipEventDestroy(p->ev_upload_done);
    if (p->d_weights) (void)hipFree(p->d_weights);
    void* bufs[] = {p->d_state, p->d_base, p->d_base_tmp, p->d_mean, p->d_eps, p->d_raw, p->d_ref, p->d_R,
                    p->d_part_alloc ? p->d_part_alloc : p->d_part, p->d_result, p->d_obs, p->d_rew, p->d_status, p->d_log, p->d_log_count, p->d_fobs, p->d_critic, p->d_next_step, p->d_last_row, p->d_noise_key};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    if (p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
}

int mppi_set_start_state(mppi_planner* p, const double* qpos, const double* qvel, const double* target3,
                         const int32_t* timestep) try {
    if (!p || !qpos || !qvel || !target3 || !timestep) return fail(-1, "null argument");
    HIP_OK(hipSetDevice(p->cfg.device));
    // alpha / ref_len come from the host mirror; q, v, target, timestep are replaced
    std::vector<double> tq, tv, tt;
    int rc = host_view(qpos, (size_t)p->P * p->nq, tq, &qpos, p->stream);
    if (!rc) rc = host_view(qvel, (size_t)p->P * p->nv, tv, &qvel, p->stream);
    if (!rc) rc = host_view(target3, (size_t)p->P * 3, tt, &target3, p->stream);
    if (rc) return rc;
    std::vector<int32_t> ts(p->P);
    if (on_device(timestep)) HIP_OK(copy_now(ts.data(), timestep, (size_t)p->P * 4, hipMemcpyDeviceToHost, p->stream));
    else std::memcpy(ts.data(), timestep, (size_t)p->P * 4);
    timestep = ts.data();
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
} catch (const std::exception& e) {
    return fail(-8, std::string("host exception: ") + e.what());
}

int mppi_set_base_actions(mppi_planner* p, const double* act) try {
    if (!p || !act) return fail(-1, "null argument");
    HIP_OK(hipSetDevice(p->cfg.device));
    HIP_OK(hipMemcpyAsync(p->d_base, act, (size_t)p->HM * p->P * 8, hipMemcpyDefault, p->stream));
    HIP_OK(hipStreamSynchronize(p->stream));
    act_sequence_changed(p);
    return 0;
} catch (const std::exception& e) {
    return fail(-8, std::string("host exception: ") + e.what());
}

int mppi_set_reference(mppi_planner* p, const double* ref, int32_t T, double alpha) try {
    if (!p) return fail(-1, "null argument");
    if (T < 0 || (T > 0 && !ref)) return fail(-1, "bad reference");
    HIP_OK(hipSetDevice(p->cfg.device));
    HIP_OK(hipStreamSynchronize(p->stream));      // rollouts in flight may still read the old rows
    const bool had_ref = (*p->h_state)[0].ref_len > 0;
    if (T > p->ref_cap) {
        // the new buffer first: if it cannot be had, the planner keeps the reference it has
        double* fresh = nullptr;
        HIP_OK(hipMalloc(&fresh, (size_t)T * 3 * 8));
        if (p->d_ref) (void)hipFree(p->d_ref);
        p->d_ref = fresh;
        p->ref_cap = T;
        p->cfg_gen++;                             // captured launches hold the old address
    }
    if ((T > 0) != had_ref) p->cfg_gen++;         // with and without a reference are different kernel variants
    if (T > 0) HIP_OK(hipMemcpyAsync(p->d_ref, ref, (size_t)T * 3 * 8, hipMemcpyDefault, p->stream));
    // alpha and the length are fields of the device-resident state, which captured launches read at run time
    for (int b = 0; b < p->P; b++) { (*p->h_state)[b].alpha = alpha; (*p->h_state)[b].ref_len = T; }
    hipLaunchKernelGGL(set_reference_fields_kernel, dim3((p->P + 255) / 256), dim3(256), 0, p->stream, p->d_state, p->P, alpha, (int)T);
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(p->stream));      // the caller may reuse its buffer
    return 0;
} catch (const std::exception& e) {
    return fail(-8, std::string("host exception: ") + e.what());
}

int mppi_set_critic(mppi_planner* p, int32_t dim, int32_t inner, const float* W1, const float* b1, const float* W2,
                    const float* b2, const float* W3, float b3) try {
    if (!p) return fail(-1, "null argument");
    HIP_OK(hipSetDevice(p->cfg.device));
    HIP_OK(hipStreamSynchronize(p->stream));
    if (dim == 0) {
        if (p->critic_dim != 0) p->cfg_gen++;
        p->critic_dim = 0;
        return 0;
    }
    if (dim != 14 && dim != 15 && dim != 17 && dim != 3 && dim != 6) return fail(-1, "critic input dim must be 14, 15, 17, 3 or 6");
    if (p->nv != 7) return fail(-1, "critic input slices are defined for the 20-dim reacher observation");
    if (inner != CRITIC_INNER) return fail(-1, "only the 128-wide critic has a compiled kernel");
    if (!W1 || !b1 || !W2 || !b2 || !W3) return fail(-1, "null critic weights");
    std::vector<float> w(CRITIC_W_FLOATS, 0.f);
    pack_critic_weights(dim, W1, b1, W2, b2, W3, b3, w.data());
    if (!p->d_critic) HIP_OK(hipMalloc(&p->d_critic, CRITIC_W_FLOATS * 4));
    if (!p->d_fobs) HIP_OK(hipMalloc(&p->d_fobs, (size_t)p->cfg.K * p->P * (p->nq + p->nv + 6) * p->esz));
    HIP_OK(copy_now(p->d_critic, w.data(), CRITIC_W_FLOATS * 4, hipMemcpyHostToDevice, p->stream));
    if (p->critic_dim != dim) p->cfg_gen++;       // the weights are read at run time; the input slice is an argument
    p->critic_dim = dim;
    return 0;
} catch (const std::exception& e) {
    return fail(-8, std::string("host exception: ") + e.what());
}

int mppi_set_params(mppi_planner* p, double kappa, double gamma, int32_t reward_type) try {
    if (!p) return fail(-1, "null argument");
    if (reward_type < 0 || reward_type > 3) return fail(-1, "bad reward_type");
    if (p->cfg.kappa != kappa || p->cfg.gamma != gamma || p->cfg.reward_type != reward_type) p->cfg_gen++;
    p->cfg.kappa = kappa; p->cfg.gamma = gamma; p->cfg.reward_type = reward_type;
    return 0;
} catch (const std::exception& e) {
    return fail(-8, std::string("host exception: ") + e.what());
}

int mppi_set_env_reward(mppi_planner* p, int32_t reward_type) try {
    if (!p) return fail(-1, "null argument");
    if (reward_type < -1 || reward_type > 3) return fail(-1, "bad reward_type");
    if (p->env_reward_type != reward_type) p->cfg_gen++;      // the advance launch of a captured step carries it
    p->env_reward_type = reward_type;
    return 0;
} catch (const std::exception& e) {
    return fail(-8, std::string("host exception: ") + e.what());
}

int mppi_host_staging(mppi_planner* p, double** out) try {
    if (!p || !out) return fail(-1, "null argument");
    HIP_OK(hipSetDevice(p->cfg.device));
    if (!p->h_stage) HIP_OK(hipHostMalloc(&p->h_stage, (size_t)p->HM * p->cfg.K * p->P * 8, hipHostMallocDefault));
    *out = p->h_stage;
    return 0;
} catch (const std::exception& e) {
    return fail(-8, std::string("host exception: ") + e.what());
}

int mppi_upload_noise(mppi_planner* p, const double* raw, void* stream) try {
    if (!p || !raw) return fail(-1, "null argument");
    HIP_OK(hipSetDevice(p->cfg.device));
    hipStream_t s = pick(p, stream);
    if (!p->d_raw) HIP_OK(hipMalloc(&p->d_raw, (size_t)p->HM * p->cfg.K * p->P * 8));      // host-upload staging, on first use
    if (p->copy_stream) HIP_OK(hipStreamSynchronize(p->copy_stream));      // a streamed upload may still be landing in d_raw
    p->raw_pending = false;
    HIP_OK(hipMemcpyAsync(p->d_raw, raw, (size_t)p->HM * p->cfg.K * p->P * 8, hipMemcpyDefault, s));
    HIP_OK(p->model->ops->noise_upload(p->cfg.dtype, p->d_raw, p->cfg.K, p->cfg.H, p->P, p->fc, 1, p->d_eps, s));
    HIP_OK(hipStreamSynchronize(s));     // the host buffer may be reused by the caller
    p->explicit_act = 0;
    p->noise_ahead = false;
    set_eps_source(p, false);
    return 0;
} catch (const std::exception& e) {
    return fail(-8, std::string("host exception: ") + e.what());
}

// The draws of `generate_paths` for all workers (trajopt/utils.py:135-139, 151-152), made on the host and sent to the
// device while they are being made: worker i's stream (np.random.seed(seeds[i])) fills samples [i*K/nstreams,
// (i+1)*K/nstreams) of the planner's page-locked staging buffer, and every finished stretch of a stream is copied to
// the device on a side stream at once, so that the copy ends with the generator.  The planner's own stream is not
// touched: it may be rolling out the previous planning step meanwhile.
int mppi_draw_noise(mppi_planner* p, const uint32_t* seeds, int32_t nstreams, int32_t nthreads, uint32_t* state_out) try {
    if (!p || !seeds || nstreams < 1) return fail(-1, "bad argument");
    if (p->P != 1) return fail(-4, "mppi_draw_noise serves single planners");
    const size_t total = (size_t)p->HM * p->cfg.K;
    if (p->cfg.K % nstreams != 0) return fail(-1, "the streams must share the samples evenly");
    if (state_out && nstreams != 1) return fail(-1, "the generator state is reported for a single stream only");
    const size_t n_each = total / nstreams;
    HIP_OK(hipSetDevice(p->cfg.device));
    if (!p->h_stage) HIP_OK(hipHostMalloc(&p->h_stage, total * 8, hipHostMallocDefault));
    if (!p->d_raw) HIP_OK(hipMalloc(&p->d_raw, total * 8));
    if (!p->copy_stream) {
        HIP_OK(hipStreamCreateWithFlags(&p->copy_stream, hipStreamNonBlocking));
        HIP_OK(hipEventCreateWithFlags(&p->ev_raw_free, hipEventDisableTiming));
        HIP_OK(hipEventCreateWithFlags(&p->ev_upload_done, hipEventDisableTiming));
        HIP_OK(hipEventRecord(p->ev_raw_free, p->stream));
    }
    // d_raw is free once the filter kernel of the previous commit has read it
    HIP_OK(hipStreamWaitEvent(p->copy_stream, p->ev_raw_free, 0));
    p->raw_pending = true;
    // every stretch of at least PIECE finished values of a stream (and whatever is left at its end) is sent by the
    // generator thread that finished it
    constexpr size_t PIECE = 64 * 1024;                // values: 512 KiB per copy
    std::vector<size_t> sent(nstreams, 0);
    std::atomic<int> copy_err{(int)hipSuccess};
    const std::function<void(int, size_t)> on_ready = [&](int i, size_t have) {
        if (have - sent[i] < PIECE && have < n_each) return;
        const size_t at = (size_t)i * n_each + sent[i];
        const hipError_t e = hipMemcpyAsync(p->d_raw + at, p->h_stage + at, (have - sent[i]) * 8, hipMemcpyHostToDevice, p->copy_stream);
        if (e != hipSuccess) copy_err.store((int)e);
        sent[i] = have;
    };
    const int rc = mppi_numpy_normals_streams(seeds, nstreams, n_each, p->h_stage, nthreads, state_out, &on_ready);
    if (copy_err.load() != (int)hipSuccess)
        return fail(-2, std::string("streamed upload: ") + hipGetErrorString((hipError_t)copy_err.load()));
    if (rc) return fail(rc, "the host generator failed");
    HIP_OK(hipEventRecord(p->ev_upload_done, p->copy_stream));
    return 0;
} catch (const std::exception& e) {
    return fail(-8, std::string("host exception: ") + e.what());
}

// sigma scaling and recursive filter (trajopt/utils.py:121-125) of the draws mppi_draw_noise sent, on the planner's
// stream behind the last piece of the upload; the next rollouts perturb the base actions with them
int mppi_commit_noise(mppi_planner* p, void* stream) try {
    if (!p) return fail(-1, "null argument");
    if (!p->raw_pending) return fail(-4, "no draws waiting on the device: call mppi_draw_noise first");
    HIP_OK(hipSetDevice(p->cfg.device));
    hipStream_t s = pick(p, stream);
    HIP_OK(hipStreamWaitEvent(s, p->ev_upload_done, 0));
    HIP_OK(p->model->ops->noise_upload(p->cfg.dtype, p->d_raw, p->cfg.K, p->cfg.H, p->P, p->fc, 1, p->d_eps, s));
    HIP_OK(hipEventRecord(p->ev_raw_free, s));
    p->raw_pending = false;
    p->explicit_act = 0;
    p->noise_ahead = false;
    set_eps_source(p, false);
    return 0;
} catch (const std::exception& e) {
    return fail(-8, std::string("host exception: ") + e.what());
}

static int generate_noise(mppi_planner* p, uint64_t seed, int seed_from_state, void* stream) {
    if (!p) return fail(-1, "null argument");
    HIP_OK(hipSetDevice(p->cfg.device));
    HIP_OK(p->model->ops->noise_philox(p->cfg.dtype, p->cfg.K, p->cfg.H, p->P, p->cfg.k_offset, seed, p->d_state, seed_from_state,
                                       p->fc, p->d_eps, p->d_noise_key, pick(p, stream)));
    p->explicit_act = 0;
    p->noise_ahead = false;
    set_eps_source(p, true);
    return 0;
}
int mppi_generate_noise(mppi_planner* p, uint64_t seed, void* stream) { return generate_noise(p, seed, 0, stream); }
int mppi_generate_noise_step(mppi_planner* p, uint64_t seed, void* stream) { return generate_noise(p, seed, 1, stream); }

int mppi_advance(mppi_planner* p, void* stream) try {
    if (!p) return fail(-1, "null argument");
    if (p->HM > 256 * 16) return fail(-1, "H*M too large for mppi_advance");
    HIP_OK(hipSetDevice(p->cfg.device));
    const int rc = reference_covers(p, 1, env_reward(p));
    if (rc) return rc;
    hipStream_t s = pick(p, stream);
    HIP_OK(p->model->ops->advance(p->d_state, p->d_base, p->d_ref, env_reward(p), p->cfg.H, p->P, p->cfg.default_act, p->d_mean,
                                  p->d_log, MPPI_LOG_CAP, p->d_log_count, p->d_last_row, p->d_status, mirror_for(p, true), s));
    advanced(p, 1, true, s);
    p->noise_ahead = false;          // whatever noise the device holds was drawn for the step that has just ended
    return 0;
} catch (const std::exception& e) {
    return fail(-8, std::string("host exception: ") + e.what());
}

int mppi_advance_noise(mppi_planner* p, uint64_t seed, void* stream) try {
    if (!p) return fail(-1, "null argument");
    if (p->HM > 256 * 16) return fail(-1, "H*M too large for mppi_advance_noise");
    HIP_OK(hipSetDevice(p->cfg.device));
    const int rc = reference_covers(p, 1, env_reward(p));
    if (rc) return rc;
    hipStream_t s = pick(p, stream);
    HIP_OK(p->model->ops->advance_noise(p->cfg.dtype, p->d_state, p->d_base, p->d_ref, env_reward(p), p->cfg.H, p->P,
                                        p->cfg.default_act, p->d_mean, p->d_log, MPPI_LOG_CAP, p->d_log_count, p->d_last_row,
                                        p->d_status, p->cfg.K, p->cfg.k_offset, seed, p->d_next_step, p->fc, p->d_eps,
                                        p->d_noise_key, mirror_for(p, true), s));
    advanced(p, 1, true, s);
    set_eps_source(p, true);
    p->explicit_act = 0;
    p->noise_ahead = true;
    p->noise_ahead_seed = seed;
    return 0;
} catch (const std::exception& e) {
    return fail(-8, std::string("host exception: ") + e.what());
}

int mppi_read_log(mppi_planner* p, int32_t planner, double* rows, int32_t max_rows, int32_t* n_rows, int32_t* total) try {
    if (!p || !rows || !n_rows || max_rows < 0 || planner < 0 || planner >= p->P) return fail(-1, "bad argument");
    HIP_OK(hipSetDevice(p->cfg.device));
    HIP_OK(hipStreamSynchronize(p->stream));      // (the planner's own stream: other planners are left alone)
    int count = 0;
    HIP_OK(copy_now(&count, p->d_log_count + planner, 4, hipMemcpyDeviceToHost, p->stream));
    const int row = p->nq + 2 * p->nv + 7;
    const double* ring = p->d_log + (size_t)planner * MPPI_LOG_CAP * row;
    int n = count < MPPI_LOG_CAP ? count : MPPI_LOG_CAP;
    if (n > max_rows) n = max_rows;
    // the n most recent rows, oldest first: one copy, or two when the run wraps around the end of the ring
    if (n > 0) {
        const int first = (count - n) % MPPI_LOG_CAP;
        const int head = n < MPPI_LOG_CAP - first ? n : MPPI_LOG_CAP - first;
        HIP_OK(copy_now(rows, ring + (size_t)first * row, (size_t)head * row * 8, hipMemcpyDeviceToHost, p->stream));
        if (head < n)
            HIP_OK(copy_now(rows + (size_t)head * row, ring, (size_t)(n - head) * row * 8, hipMemcpyDeviceToHost, p->stream));
    }
    *n_rows = n;
    if (total) *total = count;
    return 0;
} catch (const std::exception& e) {
    return fail(-8, std::string("host exception: ") + e.what());
}

int mppi_upload_actions(mppi_planner* p, const double* act, void* stream) try {
    if (!p || !act) return fail(-1, "null argument");
    HIP_OK(hipSetDevice(p->cfg.device));
    hipStream_t s = pick(p, stream);
    if (!p->d_raw) HIP_OK(hipMalloc(&p->d_raw, (size_t)p->HM * p->cfg.K * p->P * 8));
    if (p->copy_stream) HIP_OK(hipStreamSynchronize(p->copy_stream));
    p->raw_pending = false;
    HIP_OK(hipMemcpyAsync(p->d_raw, act, (size_t)p->HM * p->cfg.K * p->P * 8, hipMemcpyDefault, s));
    HIP_OK(p->model->ops->noise_upload(p->cfg.dtype, p->d_raw, p->cfg.K, p->cfg.H, p->P, p->fc, 0, p->d_eps, s));
    HIP_OK(hipStreamSynchronize(s));
    p->explicit_act = 1;
    p->noise_ahead = false;
    set_eps_source(p, false);
    return 0;
} catch (const std::exception& e) {
    return fail(-8, std::string("host exception: ") + e.what());
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
    a.sweep_cap = p->cfg.sweep_cap;
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

// true if the update draws the noise again instead of reading it (batched planners on device noise).  Both routes give the
// same bits; drawing a normal again costs a CU about as long as HBM takes to deliver a stored one (283 against 276 us at
// 1024 planners x 512 samples), so the stored-noise kernel stays the default and MPPI_UPDATE_REGEN=1 selects this one
// (a planner whose noise buffer could be dropped would need it).
static bool update_regenerates(const mppi_planner* p) {
    const char* e = getenv("MPPI_UPDATE_REGEN");        // (looked up at every launch or capture: tests flip it)
    const bool allowed = e && e[0] == '1';
    return allowed && p->eps_philox && !p->explicit_act && p->model->ops->update_regen_fits(p->cfg.K, p->cfg.H, p->P);
}

static int launch_partials(mppi_planner* p, hipStream_t s) {
    if (update_regenerates(p)) {
        HIP_OK(p->model->ops->update_regen(p->cfg.dtype, p->d_R, p->cfg.K, p->cfg.H, p->P, p->cfg.kappa, p->cfg.k_offset, 0, p->d_base,
                                           p->d_part, p->d_result, p->d_noise_key, p->fc, s));
        return 0;
    }
    HIP_OK(p->model->ops->update_rows(p->cfg.dtype, p->d_R, p->d_eps, p->cfg.K, p->cfg.H, p->P, p->cfg.kappa, p->cfg.k_offset, 0,
                                      p->d_weights, p->d_base, p->d_part, p->d_result, s));
    return 0;
}

static int launch_update_fused(mppi_planner* p, hipStream_t s) {
    if (update_regenerates(p)) {
        HIP_OK(p->model->ops->update_regen(p->cfg.dtype, p->d_R, p->cfg.K, p->cfg.H, p->P, p->cfg.kappa, p->cfg.k_offset, 1, p->d_base,
                                           p->d_part, p->d_result, p->d_noise_key, p->fc, s));
        return 0;
    }
    HIP_OK(p->model->ops->update_rows(p->cfg.dtype, p->d_R, p->d_eps, p->cfg.K, p->cfg.H, p->P, p->cfg.kappa, p->cfg.k_offset, 1,
                                      p->d_weights, p->d_base, p->d_part, p->d_result, s));
    return 0;
}

static int launch_merge(mppi_planner* p, const double* gathered, int nranks, hipStream_t s);

// `MPPI.update` of one iteration: a single launch on a planner that owns all samples; on a sharded planner
// this shard's statistics, one all-gather of 3 + H*M doubles per planner and rank, and the rank-ordered merge
// that every rank applies identically.  All of it is stream work and can be captured.
static int launch_update(mppi_planner* p, hipStream_t s) {
    if (!p->comm) return launch_update_fused(p, s);
    int rc = launch_partials(p, s);
    if (rc) return rc;
    std::string why;
    const RcclApi* api = rccl_api(why);
    if (!api) return fail(-7, why);
    const size_t row = (size_t)(3 + p->HM) * p->P;
    RCCL_OK(api, api->AllGather(p->d_part, p->d_gathered, row, RCCL_FLOAT64, p->comm, s));      // in place
    return launch_merge(p, p->d_gathered, p->comm_nranks, s);
}

static int launch_merge(mppi_planner* p, const double* gathered, int nranks, hipStream_t s) {
    hipLaunchKernelGGL(merge_kernel, dim3((p->HM + 255) / 256, p->P), dim3(256), 0, s, gathered, nranks, p->HM, p->cfg.kappa,
                       p->d_base, p->d_result);
    HIP_OK(hipGetLastError());
    return 0;
}

int mppi_rollout(mppi_planner* p, void* stream) try {
    if (!p) return fail(-1, "null argument");
    {
        const int rc = reference_covers(p, p->cfg.H, p->cfg.reward_type);
        if (rc) return rc;
    }
    HIP_OK(hipSetDevice(p->cfg.device));
    return launch_rollout_only(p, pick(p, stream));
} catch (const std::exception& e) {
    return fail(-8, std::string("host exception: ") + e.what());
}

int mppi_local_partials(mppi_planner* p, void* stream) try {
    if (!p) return fail(-1, "null argument");
    if (p->explicit_act) return fail(-4, "update needs noise-relative samples; call mppi_upload_noise or mppi_generate_noise");
    {
        const int rc = reference_covers(p, p->cfg.H, p->cfg.reward_type);
        if (rc) return rc;
    }
    HIP_OK(hipSetDevice(p->cfg.device));
    hipStream_t s = pick(p, stream);
    int rc = launch_rollout_only(p, s);
    if (rc) return rc;
    return launch_partials(p, s);
} catch (const std::exception& e) {
    return fail(-8, std::string("host exception: ") + e.what());
}

int mppi_partials_ptr(mppi_planner* p, void** ptr, size_t* bytes) try {
    if (!p || !ptr || !bytes) return fail(-1, "null argument");
    *ptr = p->d_part;
    *bytes = (size_t)(3 + p->HM) * p->P * 8;
    return 0;
} catch (const std::exception& e) {
    return fail(-8, std::string("host exception: ") + e.what());
}

int mppi_merge(mppi_planner* p, const void* gathered, int32_t nranks, void* stream) try {
    if (!p || !gathered || nranks < 1) return fail(-1, "bad argument");
    HIP_OK(hipSetDevice(p->cfg.device));
    act_sequence_changed(p);
    return launch_merge(p, (const double*)gathered, nranks, pick(p, stream));
} catch (const std::exception& e) {
    return fail(-8, std::string("host exception: ") + e.what());
}

static int build_graph(mppi_planner* p) {
    if (p->graph) { HIP_OK(hipGraphExecDestroy(p->graph)); p->graph = nullptr; }
    p->graph_gen = 0;
    HIP_OK(hipStreamBeginCapture(p->stream, hipStreamCaptureModeThreadLocal));
    int rc = launch_rollout_only(p, p->stream);
    if (!rc) rc = launch_update(p, p->stream);
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(p->stream, &g);
    if (rc) {
        if (e == hipSuccess && g) (void)hipGraphDestroy(g);
        return rc;
    }
    HIP_OK(e);
    e = hipGraphInstantiate(&p->graph, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    HIP_OK(e);
    p->graph_gen = p->cfg_gen;
    return 0;
}

int mppi_iterate(mppi_planner* p, int32_t niter, void* stream) try {
    if (!p || niter < 0) return fail(-1, "bad argument");
    if (p->explicit_act) return fail(-4, "update needs noise-relative samples; call mppi_upload_noise or mppi_generate_noise");
    if (niter > 0) {
        const int rc = reference_covers(p, p->cfg.H, p->cfg.reward_type);
        if (rc) return rc;
        act_sequence_changed(p);
    }
    HIP_OK(hipSetDevice(p->cfg.device));
    hipStream_t s = pick(p, stream);
    static const bool use_graph = [] { const char* e = getenv("MPPI_GRAPH"); return !(e && e[0] == '0'); }();
    if (use_graph && !p->cfg.dump && !p->profiling) {
        if (p->graph_gen != p->cfg_gen) { int rc = build_graph(p); if (rc) return rc; }
        for (int i = 0; i < niter; i++) HIP_OK(hipGraphLaunch(p->graph, s));
        return 0;
    }
    for (int i = 0; i < niter; i++) {
        int rc = launch_rollout_only(p, s);
        if (!rc) rc = launch_update(p, s);
        if (rc) return rc;
    }
    return 0;
} catch (const std::exception& e) {
    return fail(-8, std::string("host exception: ") + e.what());
}

// one whole planning step enqueued on s (no host synchronisation)
// The noise of the coming step is drawn in the advance launch of the previous one when the handle is small enough
// for that launch to have idle CUs (the fp64 step next to it sets the kernel's register budget, which would slow a
// GPU-filling noise pass down).
static bool plan_draws_ahead(const mppi_planner* p) { return (long)p->cfg.K * p->P <= 32768; }

// niter x (rollouts, update), then the real-env step; the noise of this step must be in place
static int enqueue_plan_body(mppi_planner* p, uint64_t seed, int niter, bool mirror, hipStream_t s) {
    for (int i = 0; i < niter; i++) {
        int rc = launch_rollout_only(p, s);
        if (!rc) rc = launch_update(p, s);
        if (rc) return rc;
    }
    if (plan_draws_ahead(p))
        HIP_OK(p->model->ops->advance_noise(p->cfg.dtype, p->d_state, p->d_base, p->d_ref, env_reward(p), p->cfg.H, p->P,
                                            p->cfg.default_act, p->d_mean, p->d_log, MPPI_LOG_CAP, p->d_log_count, p->d_last_row,
                                            p->d_status, p->cfg.K, p->cfg.k_offset, seed, p->d_next_step, p->fc, p->d_eps,
                                            p->d_noise_key, mirror_for(p, mirror), s));
    else
        HIP_OK(p->model->ops->advance(p->d_state, p->d_base, p->d_ref, env_reward(p), p->cfg.H, p->P, p->cfg.default_act,
                                      p->d_mean, p->d_log, MPPI_LOG_CAP, p->d_log_count, p->d_last_row, p->d_status,
                                      mirror_for(p, mirror), s));
    return 0;
}
static int enqueue_plan_noise(mppi_planner* p, uint64_t seed, hipStream_t s) {
    HIP_OK(p->model->ops->noise_philox(p->cfg.dtype, p->cfg.K, p->cfg.H, p->P, p->cfg.k_offset, seed, p->d_state, 1, p->fc,
                                       p->d_eps, p->d_noise_key, s));
    return 0;
}

// `steps` consecutive planning steps captured on the planner's stream and instantiated
constexpr int PLAN_RUN = 8;
static int capture_steps(mppi_planner* p, uint64_t seed, int niter, int steps, bool ahead, hipGraphExec_t* out) {
    hipGraph_t g = nullptr;
    HIP_OK(hipStreamBeginCapture(p->stream, hipStreamCaptureModeThreadLocal));
    int rc = 0;
    for (int i = 0; i < steps && !rc; i++) {
        rc = ahead ? 0 : enqueue_plan_noise(p, seed, p->stream);
        // a captured single step writes the host mirror (its caller reads every step back); the steps of a captured
        // run do not: nobody reads between them, and a write to host memory is not free
        if (!rc) rc = enqueue_plan_body(p, seed, niter, steps == 1, p->stream);
    }
    hipError_t e = hipStreamEndCapture(p->stream, &g);
    if (rc) {
        if (e == hipSuccess && g) (void)hipGraphDestroy(g);
        return rc;
    }
    HIP_OK(e);
    e = hipGraphInstantiate(out, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    HIP_OK(e);
    return 0;
}

int mppi_plan(mppi_planner* p, uint64_t seed, int32_t niter, int32_t nsteps, void* stream) try {
    if (!p || niter < 1 || nsteps < 0) return fail(-1, "bad argument");
    if (p->HM > 256 * 16) return fail(-1, "H*M too large for mppi_plan");
    if (nsteps == 0) return 0;
    {   // the last of the steps rolls out from time step t + nsteps - 1
        int rc = reference_covers(p, nsteps - 1 + p->cfg.H, p->cfg.reward_type);
        if (!rc) rc = reference_covers(p, nsteps, env_reward(p));
        if (rc) return rc;
    }
    HIP_OK(hipSetDevice(p->cfg.device));
    hipStream_t s = pick(p, stream);
    p->explicit_act = 0;
    set_eps_source(p, true);
    const bool ahead = plan_draws_ahead(p);
    // with noise drawn ahead only the first step of a run needs a launch of its own
    const bool have_noise = ahead && p->noise_ahead && p->noise_ahead_seed == seed;
    bool last_wrote_mirror = false;
    static const bool use_graph = [] { const char* e = getenv("MPPI_GRAPH"); return !(e && e[0] == '0'); }();
    if (!use_graph || p->profiling) {
        for (int i = 0; i < nsteps; i++) {
            int rc = (ahead && (i > 0 || have_noise)) ? 0 : enqueue_plan_noise(p, seed, s);
            if (!rc) rc = enqueue_plan_body(p, seed, niter, i == nsteps - 1, s);
            if (rc) return rc;
        }
        last_wrote_mirror = true;
    } else {
        if (!p->step_graph || p->step_graph_gen != p->cfg_gen || p->step_graph_niter != niter || p->step_graph_seed != seed) {
            if (p->step_graph) { HIP_OK(hipGraphExecDestroy(p->step_graph)); p->step_graph = nullptr; }
            if (p->step_graph_run) { HIP_OK(hipGraphExecDestroy(p->step_graph_run)); p->step_graph_run = nullptr; }
            p->step_graph_gen = 0;
            const int rc = capture_steps(p, seed, niter, 1, ahead, &p->step_graph);
            if (rc) return rc;
            p->step_graph_gen = p->cfg_gen;
            p->step_graph_niter = niter;
            p->step_graph_seed = seed;
        }
        // A run of steps is replayed PLAN_RUN at a time: the kernels take the time step from the device-resident state,
        // so the same launches serve every step, and the boundary between two replays costs the stream more than the
        // boundary between two kernels of one replay.
        if (nsteps >= PLAN_RUN && !p->step_graph_run) {
            const int rc = capture_steps(p, seed, niter, PLAN_RUN, ahead, &p->step_graph_run);
            if (rc) return rc;
        }
        if (ahead && !have_noise) { const int rc = enqueue_plan_noise(p, seed, s); if (rc) return rc; }
        int left = nsteps;
        for (; left >= PLAN_RUN; left -= PLAN_RUN) HIP_OK(hipGraphLaunch(p->step_graph_run, s));
        last_wrote_mirror = left > 0;
        for (; left > 0; left--) HIP_OK(hipGraphLaunch(p->step_graph, s));
    }
    p->noise_ahead = ahead;
    p->noise_ahead_seed = seed;
    advanced(p, nsteps, last_wrote_mirror, s);
    return 0;
} catch (const std::exception& e) {
    return fail(-8, std::string("host exception: ") + e.what());
}

// Waits until the device has written the outcome of real-env step number `expect` to the host mirror.  false: the
// stream drained (or failed) without that happening -- the caller then copies the slow way, which also surfaces errors.
static bool mirror_arrived(mppi_planner* p, uint64_t expect) {
    const size_t at = (size_t)(p->nq + 2 * p->nv + 7) + p->HM;
    const uint64_t* count = reinterpret_cast<const uint64_t*>(p->h_mirror + at);
    for (uint64_t spins = 1;; spins++) {
        if (__atomic_load_n(count, __ATOMIC_ACQUIRE) == expect) return true;
        if ((spins & 1023) == 0) {
            // a kernel that never runs must not hang the host: look at the stream now and then
            const hipError_t q = hipStreamQuery(p->adv_stream ? p->adv_stream : p->stream);
            if (q != hipErrorNotReady) return __atomic_load_n(count, __ATOMIC_ACQUIRE) == expect;
            (void)hipGetLastError();              // "not ready" is an answer, not an error to be found by a later check
        }
        if (spins < 4096) __builtin_ia32_pause();
        else std::this_thread::yield();           // a long step (many samples): leave the processor to others
    }
}

int mppi_read_step(mppi_planner* p, double* rows, double* act_seq) try {
    if (!p || (!rows && !act_seq)) return fail(-1, "null argument");
    HIP_OK(hipSetDevice(p->cfg.device));
    const size_t nrow = (size_t)p->P * (p->nq + 2 * p->nv + 7), nseq = (size_t)p->HM * p->P;
    // A single planner whose newest step wrote the host mirror and whose act_sequence nothing has touched since: the
    // data arrive in host memory by themselves, the count last (no copy, no wait on the stream).
    if (p->h_mirror && p->mirror_at != 0 && p->mirror_at == p->advances && mirror_arrived(p, p->mirror_at)) {
        if (rows) std::memcpy(rows, p->h_mirror, nrow * 8);
        if (act_seq) std::memcpy(act_seq, p->h_mirror + nrow, nseq * 8);
        return 0;
    }
    // The first read-back of a planner says that its caller reads steps back: from now on single steps write the
    // mirror (a captured step is captured again for that; callers that never read pay nothing for it).
    if (p->h_mirror && !p->mirror_wanted) { p->mirror_wanted = true; p->cfg_gen++; }
    // two small copies into page-locked memory behind whatever the stream of the last step still holds, one wait
    hipStream_t s = p->adv_stream ? p->adv_stream : p->stream;
    if (rows) HIP_OK(hipMemcpyAsync(p->h_step, p->d_last_row, nrow * 8, hipMemcpyDeviceToHost, s));
    if (act_seq) HIP_OK(hipMemcpyAsync(p->h_step + nrow, p->d_base, nseq * 8, hipMemcpyDeviceToHost, s));
    HIP_OK(hipStreamSynchronize(s));
    if (rows) std::memcpy(rows, p->h_step, nrow * 8);
    if (act_seq) std::memcpy(act_seq, p->h_step + nrow, nseq * 8);
    return 0;
} catch (const std::exception& e) {
    return fail(-8, std::string("host exception: ") + e.what());
}

int mppi_shift(mppi_planner* p, const double* override_seq, void* stream) try {
    if (!p) return fail(-1, "null argument");
    HIP_OK(hipSetDevice(p->cfg.device));
    hipStream_t s = pick(p, stream);
    act_sequence_changed(p);
    if (override_seq) {
        HIP_OK(hipMemcpyAsync(p->d_base, override_seq, (size_t)p->HM * p->P * 8, hipMemcpyDefault, s));
        HIP_OK(hipStreamSynchronize(s));
    }
    hipLaunchKernelGGL(shift_kernel, dim3((p->HM + 255) / 256, p->P), dim3(256), 0, s, p->d_base, p->cfg.H, p->nv, p->cfg.default_act,
                       p->d_mean, p->d_base_tmp);
    HIP_OK(hipGetLastError());
    HIP_OK(hipMemcpyAsync(p->d_base, p->d_base_tmp, (size_t)p->HM * p->P * 8, hipMemcpyDeviceToDevice, s));
    return 0;
} catch (const std::exception& e) {
    return fail(-8, std::string("host exception: ") + e.what());
}

int mppi_profile(mppi_planner* p, int32_t enable) try {
    if (!p) return fail(-1, "null argument");
    p->profiling = enable != 0;
    return 0;
} catch (const std::exception& e) {
    return fail(-8, std::string("host exception: ") + e.what());
}

int mppi_profile_read(mppi_planner* p, double* rollout_ms_total, int32_t* launches) try {
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
} catch (const std::exception& e) {
    return fail(-8, std::string("host exception: ") + e.what());
}

// The status word is sticky: bits stay set until this call (enqueued on the planner's stream, behind whatever may
// still set them).
int mppi_clear_status(mppi_planner* p) try {
    if (!p) return fail(-1, "null argument");
    HIP_OK(hipSetDevice(p->cfg.device));
    HIP_OK(hipMemsetAsync(p->d_status, 0, 16, p->stream));
    return 0;
} catch (const std::exception& e) {
    return fail(-8, std::string("host exception: ") + e.what());
}

int mppi_synchronize(mppi_planner* p) try {
    if (!p) return fail(-1, "null argument");
    HIP_OK(hipSetDevice(p->cfg.device));
    HIP_OK(hipStreamSynchronize(p->stream));
    return 0;
} catch (const std::exception& e) {
    return fail(-8, std::string("host exception: ") + e.what());
}

static int read_as_double(mppi_planner* p, const void* dsrc, size_t n, double* dst) {
    if (p->cfg.dtype == MPPI_F64) {
        HIP_OK(copy_now(dst, dsrc, n * 8, hipMemcpyDeviceToHost, p->stream));
    } else {
        std::vector<float> tmp(n);
        HIP_OK(copy_now(tmp.data(), dsrc, n * 4, hipMemcpyDeviceToHost, p->stream));
        for (size_t i = 0; i < n; i++) dst[i] = (double)tmp[i];
    }
    return 0;
}

int mppi_read(mppi_planner* p, int32_t what, void* dst, size_t bytes) try {
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
        HIP_OK(copy_now(dst, p->d_base, P * HM * 8, hipMemcpyDeviceToHost, p->stream));
        return 0;
    case MPPI_READ_ARGMAX:
    case MPPI_READ_RMAX: {
        if (!need(P * 8)) return fail(-1, "ARGMAX / RMAX need P values");
        std::vector<double> r(2 * P);
        HIP_OK(copy_now(r.data(), p->d_result, 16 * P, hipMemcpyDeviceToHost, p->stream));
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
        HIP_OK(copy_now(base.data(), p->d_base, P * HM * 8, hipMemcpyDeviceToHost, p->stream));
        double* o = (double*)dst;
        const bool explicit_act = p->explicit_act != 0, f32 = p->cfg.dtype == MPPI_F32;
        // device layout [step, dimension][sample] -> (sample, step, dimension), in tiles of 64 samples so that reads run
        // along the samples and the 64 output rows being filled stay in cache; tiles are spread over a few threads
        constexpr size_t TILE = 64;
        const size_t tiles = (K + TILE - 1) / TILE, jobs = P * tiles;
        auto run = [&](size_t lo, size_t hi) {
            for (size_t job = lo; job < hi; job++) {
                const size_t b = job / tiles, k0 = (job % tiles) * TILE, k1 = k0 + TILE < K ? k0 + TILE : K;
                for (size_t hm = 0; hm < HM; hm++) {
                    const double bb = base[b * HM + hm];
                    const double* x = &e[(b * HM + hm) * K];
                    for (size_t k = k0; k < k1; k++) {
                        // same arithmetic as the kernel: the sum is formed in the rollout dtype
                        double a;
                        if (explicit_act) a = x[k];
                        else if (f32) a = (double)((float)bb + (float)x[k]);
                        else a = bb + x[k];
                        o[(b * K + k) * HM + hm] = a;
                    }
                }
            }
        };
        const size_t nthreads = jobs >= 32 ? 8 : 1, per = (jobs + nthreads - 1) / nthreads;
        {
            // helpers are joined on every way out; a helper that cannot be started leaves its share to this thread
            struct Pool {
                std::vector<std::thread> threads;
                ~Pool() { for (auto& t : threads) if (t.joinable()) t.join(); }
            } pool;
            size_t started = 1;
            try {
                for (; started < nthreads && started * per < jobs; started++)
                    pool.threads.emplace_back(run, started * per, (started + 1) * per < jobs ? (started + 1) * per : jobs);
            } catch (const std::system_error&) {
            }
            run(0, per < jobs ? per : jobs);
            if (started < nthreads && started * per < jobs) run(started * per, jobs);
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
        HIP_OK(copy_now(dst, p->d_status, 4, hipMemcpyDeviceToHost, p->stream));
        return 0;
    case MPPI_READ_STATUS_DETAIL:
        if (!need(16)) return fail(-1, "STATUS_DETAIL needs four int32");
        HIP_OK(copy_now(dst, p->d_status, 16, hipMemcpyDeviceToHost, p->stream));
        return 0;
    case MPPI_READ_PARTIALS:
        if (!need(P * (3 + HM) * 8)) return fail(-1, "PARTIALS needs P*(3+H*M) doubles");
        HIP_OK(copy_now(dst, p->d_part, P * (3 + HM) * 8, hipMemcpyDeviceToHost, p->stream));
        return 0;
    }
    return fail(-1, "unknown read selector");
} catch (const std::exception& e) {
    return fail(-8, std::string("host exception: ") + e.what());
}

// scratch and stream of the single-step entry points on one device; the caller holds m->mu[device]
static int env_io(const mppi_model* m, int device, double** io, hipStream_t* stream) {
    HIP_OK(hipSetDevice(device));
    mppi_model* mm = const_cast<mppi_model*>(m);
    if (!mm->io_stream[device]) HIP_OK(hipStreamCreateWithFlags(&mm->io_stream[device], hipStreamNonBlocking));
    if (!mm->d_io[device]) HIP_OK(hipMalloc(&mm->d_io[device], 512 * 8));
    *io = mm->d_io[device];
    *stream = mm->io_stream[device];
    return 0;
}

int mppi_env_step(const mppi_model* m, int32_t device, const double* qpos, const double* qvel, const double* target3,
                  const double* act, int32_t timestep, int32_t reward_type, const double* ref, int32_t T, double alpha,
                  double* qpos_out, double* qvel_out, double* obs_out, double* reward_out) try {
    if (!m || !qpos || !qvel || !target3 || !act) return fail(-1, "null argument");
    if ((reward_type == MPPI_REWARD_TRACKING || reward_type == MPPI_REWARD_COOLING) && (!ref || T < 1))
        return fail(-1, "tracking / cooling reward needs a reference");
    const int nv = m->ops->nv, nq = m->ops->nq, ns = nq + nv;
    if (device < 0 || device >= 16) return fail(-1, "device ordinal out of range");
    std::lock_guard<std::mutex> turn(const_cast<mppi_model*>(m)->mu[device]);
    double* io;
    hipStream_t st;
    int rc = env_io(m, device, &io, &st);
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
    HIP_OK(hipMemcpyAsync(io, in, n_in * 8, hipMemcpyHostToDevice, st));
    HIP_OK(m->ops->env_step(io, reward_type, has_ref, 0, io + 256, st));
    double out[4 * MPPI_MAXV + 16];
    HIP_OK(hipMemcpyAsync(out, io + 256, n_out * 8, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    if (qpos_out) for (int i = 0; i < nq; i++) qpos_out[i] = out[i];
    if (qvel_out) for (int i = 0; i < nv; i++) qvel_out[i] = out[nq + i];
    if (obs_out) for (int i = 0; i < ns + 6; i++) obs_out[i] = out[ns + i];
    if (reward_out) *reward_out = out[2 * ns + 6];
    return 0;
} catch (const std::exception& e) {
    return fail(-8, std::string("host exception: ") + e.what());
}

int mppi_env_forward(const mppi_model* m, int32_t device, const double* qpos, double* hand3_out) try {
    if (!m || !qpos || !hand3_out) return fail(-1, "null argument");
    const int nv = m->ops->nv, nq = m->ops->nq, ns = nq + nv;
    if (device < 0 || device >= 16) return fail(-1, "device ordinal out of range");
    std::lock_guard<std::mutex> turn(const_cast<mppi_model*>(m)->mu[device]);
    double* io;
    hipStream_t st;
    int rc = env_io(m, device, &io, &st);
    if (rc) return rc;
    double in[4 * MPPI_MAXV + 16] = {0};
    for (int i = 0; i < nq; i++) in[i] = qpos[i];
    HIP_OK(hipMemcpyAsync(io, in, (ns + nv + 7) * 8, hipMemcpyHostToDevice, st));
    HIP_OK(m->ops->env_step(io, 0, 0, 1, io + 256, st));
    double out[4 * MPPI_MAXV + 16];
    HIP_OK(hipMemcpyAsync(out, io + 256, (2 * ns + 7) * 8, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    for (int c = 0; c < 3; c++) hand3_out[c] = out[2 * ns + c];
    return 0;
} catch (const std::exception& e) {
    return fail(-8, std::string("host exception: ") + e.what());
}

int mppi_comm_unique_id(void* id_out, size_t bytes) try {
    if (!id_out || bytes != sizeof(RcclUniqueId)) return fail(-1, "the exchange id is 128 bytes");
    std::string why;
    const RcclApi* api = rccl_api(why);
    if (!api) return fail(-7, why);
    RCCL_OK(api, api->GetUniqueId((RcclUniqueId*)id_out));
    return 0;
} catch (const std::exception& e) {
    return fail(-8, std::string("host exception: ") + e.what());
}

int mppi_comm_init(mppi_planner* p, const void* id, size_t bytes, int32_t rank, int32_t nranks) try {
    if (!p || !id || bytes != sizeof(RcclUniqueId)) return fail(-1, "bad argument (the exchange id is 128 bytes)");
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(-1, "rank must lie in [0, nranks)");
    if (p->comm) return fail(-4, "the planner already has an exchange");
    std::string why;
    const RcclApi* api = rccl_api(why);
    if (!api) return fail(-7, why);
    HIP_OK(hipSetDevice(p->cfg.device));
    HIP_OK(hipStreamSynchronize(p->stream));
    RcclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    void* comm = nullptr;
    RCCL_OK(api, api->CommInitRank(&comm, nranks, uid, rank));
    const size_t row = (size_t)(3 + p->HM) * p->P;
    const hipError_t e = hipMalloc(&p->d_gathered, row * nranks * 8);
    if (e != hipSuccess) {
        (void)api->CommDestroy(comm);
        return fail(-2, std::string("hipMalloc of the gathered rows: ") + hipGetErrorString(e));
    }
    // from here on a failure leaves the planner as it was: single-rank, its own statistics buffer
    auto undo = [&]() {
        (void)api->CommDestroy(comm);
        (void)hipFree(p->d_gathered);
        p->d_gathered = nullptr;
        if (p->d_part_alloc) { p->d_part = p->d_part_alloc; p->d_part_alloc = nullptr; }
        p->comm = nullptr; p->comm_rank = 0; p->comm_nranks = 0;
        p->cfg_gen++;
    };
    hipError_t he = zero_now(p->d_gathered, row * nranks * 8, p->stream);
    if (he != hipSuccess) { undo(); return fail(-2, std::string("hipMemset of the gathered rows: ") + hipGetErrorString(he)); }
    p->d_part_alloc = p->d_part;
    p->d_part = p->d_gathered + row * rank;       // the statistics kernel writes where the gather expects them
    p->comm = comm; p->comm_rank = rank; p->comm_nranks = nranks;
    // One gather outside any capture: RCCL sets up its connections and buffers at the first collective of a
    // communicator, which must not happen while a stream is being captured.
    const int ar = api->AllGather(p->d_part, p->d_gathered, row, RCCL_FLOAT64, comm, p->stream);
    if (ar != 0) { undo(); return fail(-7, std::string("ncclAllGather (first exchange): ") + api->GetErrorString(ar)); }
    he = hipStreamSynchronize(p->stream);
    if (he != hipSuccess) { undo(); return fail(-2, std::string("first exchange: ") + hipGetErrorString(he)); }
    p->cfg_gen++;                   // captured updates were the single-rank kind
    return 0;
} catch (const std::exception& e) {
    return fail(-8, std::string("host exception: ") + e.what());
}

int mppi_update_from_returns(int32_t device, const double* act, const double* R, int32_t K, int32_t HM, double kappa,
                             double* act_seq_out) try {
    if (!act || !R || !act_seq_out || K < 1 || HM < 1) return fail(-1, "bad argument");
    HIP_OK(hipSetDevice(device));
    struct Owned {            // released on every way out
        double* p = nullptr;
        ~Owned() { if (p) (void)hipFree(p); }
    } d_act, d_R, d_out;
    struct OwnStream {        // a stream of this call's own (see copy_now)
        hipStream_t s = nullptr;
        ~OwnStream() { if (s) (void)hipStreamDestroy(s); }
    } us;
    HIP_OK(hipStreamCreateWithFlags(&us.s, hipStreamNonBlocking));
    HIP_OK(hipMalloc(&d_act.p, (size_t)K * HM * 8));
    HIP_OK(hipMalloc(&d_R.p, (size_t)K * 8));
    HIP_OK(hipMalloc(&d_out.p, (size_t)HM * 8));
    HIP_OK(copy_now(d_act.p, act, (size_t)K * HM * 8, hipMemcpyHostToDevice, us.s));
    HIP_OK(copy_now(d_R.p, R, (size_t)K * 8, hipMemcpyHostToDevice, us.s));
    hipLaunchKernelGGL(update_direct_kernel, dim3(HM), dim3(256), 0, us.s, d_act.p, d_R.p, K, (int)HM, kappa, d_out.p);
    HIP_OK(hipGetLastError());
    HIP_OK(copy_now(act_seq_out, d_out.p, (size_t)HM * 8, hipMemcpyDeviceToHost, us.s));
    return 0;
} catch (const std::exception& e) {
    return fail(-8, std::string("host exception: ") + e.what());
}

int mppi_update_from_returns_noalloc(int32_t device, const double* d_act, const double* d_R, int32_t K, int32_t HM,
                                     double kappa, double* d_act_seq_out, void* stream) try {
    if (!d_act || !d_R || !d_act_seq_out || K < 1 || HM < 1) return fail(-1, "bad argument");
    if (K > MPPI_NOALLOC_MAX_K)
        return fail(-6, "mppi_update_from_returns_noalloc: K = " + std::to_string(K) + " exceeds the fixed workspace of " +
                            std::to_string(MPPI_NOALLOC_MAX_K) + " samples; use mppi_update_from_returns");
    HIP_OK(hipSetDevice(device));
    double* ws;
    const int rc = noalloc_scratch(device, &ws);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    // both kernels overwrite every workspace entry they later read (Rmax, sum S, S_0 .. S_{K-1}): nothing has to be cleared
    hipLaunchKernelGGL(noalloc_weights_kernel, dim3(1), dim3(256), 0, s, d_R, (int)K, kappa, ws);
    HIP_OK(hipGetLastError());
    hipLaunchKernelGGL(noalloc_sum_kernel, dim3(HM), dim3(256), 0, s, d_act, (const double*)ws, (int)K, (int)HM, d_act_seq_out);
    HIP_OK(hipGetLastError());
    return 0;
} catch (const std::exception& e) {
    return fail(-8, std::string("host exception: ") + e.what());
}

}  // extern "C"
