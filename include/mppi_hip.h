/* C ABI of the device-resident MPPI planning step for AMD MI355X (gfx950).
 *
 * The reference is pure Python and has no FFI of its own; each entry point
 * below names the reference interface it stands in for (paths relative to the
 * reference tree).  A maintainer binds these with ctypes; trajopt_amd/_lib.py
 * holds the prototypes and structure layouts of such a binding.
 *
 * Conventions: every function returns 0 on success and a negative code on
 * failure (-1 bad argument, -2 HIP error, -3 no kernel for the model, -4 wrong
 * state for the call, -5 reference index out of range, -6 size limit, -7 RCCL
 * error, -8 host exception such as out of memory; nothing is thrown across
 * this boundary); the message is available from mppi_last_error() (thread local,
 * valid until the next call on that thread).  Host pointers are copied before
 * the call returns.  The array arguments of mppi_set_start_state,
 * mppi_set_base_actions, mppi_set_reference, mppi_upload_noise,
 * mppi_upload_actions and mppi_shift may also be device memory (for instance a
 * torch tensor's data_ptr()): the library looks the address up and copies from
 * where it lies.  `stream` arguments take a hipStream_t (or NULL for the
 * planner's own stream).  A planner is not re-entrant; distinct planners may
 * be driven from distinct host threads (the library keeps off the legacy
 * stream, which the runtime blocks for the whole process while any thread
 * captures a graph).  A model handle may be shared between
 * threads: mppi_env_step / mppi_env_forward calls on one model and device take
 * turns inside the library (their own stream, no device-wide synchronisation).
 */
#ifndef MPPI_HIP_H
#define MPPI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPPI_MAXV 32
#define MPPI_ABI_VERSION 8
#define MPPI_LOG_CAP 4096
#define MPPI_NOALLOC_MAX_K 262144   /* most samples mppi_update_from_returns_noalloc accepts */

enum { MPPI_F32 = 0, MPPI_F64 = 1 };
enum { MPPI_REWARD_DENSE = 0, MPPI_REWARD_SPARSE = 1, MPPI_REWARD_TRACKING = 2, MPPI_REWARD_COOLING = 3 };
enum { MPPI_TAIL_REPEAT = 0, MPPI_TAIL_MEAN = 1 };
enum {
    /* sizes are per planner; a handle with P batched planners returns P of each, back to back */
    MPPI_READ_RETURNS = 0,      /* double[K]            discounted return of every local sample        */
    MPPI_READ_ACT_SEQ = 1,      /* double[H*M]          current act_sequence                            */
    MPPI_READ_ARGMAX = 2,       /* int64[1]             global index of the best sample (first on ties) */
    MPPI_READ_ACTIONS = 3,      /* double[K*H*M]        perturbed actions, sample-major (K,H,M)         */
    MPPI_READ_NEXT_OBS = 4,     /* double[K*H*(nq+M+6)] next_observations (needs cfg.dump); nq = M without balls */
    MPPI_READ_REWARDS = 5,      /* double[K*H]          per-step rewards (needs cfg.dump)               */
    MPPI_READ_STATUS = 6,       /* int32[1]             MPPI_STATUS_* bits, sticky since the planner was created */
    MPPI_READ_RMAX = 7,         /* double[1]            best return                                      */
    MPPI_READ_PARTIALS = 8,     /* double[3+H*M]        this shard's (Rmax, argmax, den, num[H*M])      */
    MPPI_READ_STATUS_DETAIL = 9 /* int32[4]             status bits; for the first capped solve reported by a     */
                                /*                      rollout with dumps, a reference or a sparse reward: 1 +    */
                                /*                      first sample of its wave, env step, planner (else zeros)   */
};

enum {
    MPPI_STATUS_NONFINITE = 1,  /* a rollout returned NaN or infinity (the simulation diverged); such a sample    */
                                /* has weight zero in the planner's updates                                      */
    MPPI_STATUS_SWEEP_CAP = 2,  /* an active-set solve of the constraint rows stopped at cfg.sweep_cap sweeps     */
    MPPI_STATUS_REF_END = 4     /* a tracking / cooling reward asked for a row past the end of the reference      */
                                /* trajectory (IndexError in trajopt/envs/reacher_env.py:61,75); the last row was */
                                /* used.  Calls return -5 instead whenever the host already knows the time step;   */
                                /* the bit covers steps whose time step advanced on the device (mppi_plan).        */
};

/* Flat model: one link per hinge or slide joint (welded bodies merged), the data the reference hands
 * to its simulator through `load_model_from_path` (trajopt/envs/mujoco_env.py:39-40). */
typedef struct mppi_model_desc {
    char name[64];
    int32_t nv;
    int32_t frame_skip;
    int32_t parent[MPPI_MAXV];
    int32_t limited[MPPI_MAXV];
    int32_t contact_link;
    int32_t hand_link;
    double h;
    double pos[MPPI_MAXV][3];
    double axis[MPPI_MAXV][3];
    double mass[MPPI_MAXV];
    double com[MPPI_MAXV][3];
    double inertia[MPPI_MAXV][6];
    double armature[MPPI_MAXV];
    double damping[MPPI_MAXV];
    double range_lo[MPPI_MAXV];
    double range_hi[MPPI_MAXV];
    double gear[MPPI_MAXV];
    double ctrl_lo[MPPI_MAXV];
    double ctrl_hi[MPPI_MAXV];
    double dof_invweight[MPPI_MAXV];
    double contact_pos[3];
    double contact_radius;
    double contact_plane_z;
    double contact_margin;
    double contact_invweight;
    double hand_pos[3];
    double solref_k;
    double solref_b;
    double solimp[5];
    double gravity[3];
    int32_t jtype[MPPI_MAXV];   /* 0 hinge, 1 slide along `axis`, 2/3/4 x/y/z component of a ball joint */
    int32_t nq;                 /* length of qpos: nv + one per ball joint (unit quaternion w, x, y, z)  */
    int32_t qadr[MPPI_MAXV];    /* qpos index of the joint coordinate; a ball's three links: its quaternion */
    int32_t integrator;         /* 0 semi-implicit Euler with implicit joint damping, 1 fourth-order Runge-Kutta */
} mppi_model_desc;

/* Planner configuration: the constructor arguments of `MPPI.__init__`
 * (trajopt/algos/mppi.py:16-54) that the device needs. */
typedef struct mppi_planner_cfg {
    int32_t K;             /* samples rolled out by this instance (this rank's shard)      */
    int32_t H;             /* horizon                                                      */
    int32_t dtype;         /* MPPI_F32 | MPPI_F64: arithmetic of the rollouts              */
    int32_t device;        /* HIP device ordinal                                           */
    int32_t reward_type;   /* MPPI_REWARD_*                                                */
    int32_t default_act;   /* MPPI_TAIL_*                                                  */
    int32_t k_offset;      /* global index of local sample 0 (sharded planners)            */
    int32_t dump;          /* keep next_observations / rewards of every rollout            */
    int32_t P;             /* independent planners batched in this handle (0 or 1: one).   */
                           /*   Every per-planner array argument / result then holds the P */
                           /*   planners back to back; noise streams differ per planner.   */
    int32_t sweep_cap;     /* most sweeps of the constraint rows' active-set iteration per  */
                           /*   physics substep; 0: the default, 16                        */
    double kappa;
    double gamma;
    double sigma[MPPI_MAXV];   /* filter_coefs[0]                                          */
    double beta[3];            /* filter_coefs[1:4]                                        */
    double mean[MPPI_MAXV];
} mppi_planner_cfg;

typedef struct mppi_model mppi_model;
typedef struct mppi_planner mppi_planner;

int mppi_version(void);
const char* mppi_last_error(void);

/* Binds a model description to the kernels compiled for it; fails if no kernel was built
 * for this model.  Stands in for `get_environment` + `MjSim` construction inside every
 * rollout worker (trajopt/utils.py:66, trajopt/envs/mujoco_env.py:39-40). */
int mppi_model_create(const mppi_model_desc* desc, mppi_model** out);
void mppi_model_destroy(mppi_model* m);

/* `MPPI.__init__` (trajopt/algos/mppi.py:16-54): allocates the device-resident state. */
int mppi_planner_create(const mppi_model* m, const mppi_planner_cfg* cfg, mppi_planner** out);
void mppi_planner_destroy(mppi_planner* p);

/* `start_state` of `gather_paths_parallel` / `set_env_state` at the top of every rollout
 * (trajopt/utils.py:76, trajopt/envs/reacher_env.py:151-160). */
int mppi_set_start_state(mppi_planner* p, const double* qpos, const double* qvel, const double* target3,
                         const int32_t* timestep);
/* `base_act` of `gather_paths_parallel` = `MPPI.act_sequence` (trajopt/algos/mppi.py:47,113). */
int mppi_set_base_actions(mppi_planner* p, const double* act_HxM);
/* `reference` / `alpha` of the tracking and cooling rewards (trajopt/envs/reacher_env.py:60-79);
 * ref may be NULL with T = 0. */
int mppi_set_reference(mppi_planner* p, const double* ref_Tx3, int32_t T, double alpha);
int mppi_set_params(mppi_planner* p, double kappa, double gamma, int32_t reward_type);
/* Reward the planner's own step of the real environment reports (mppi_advance, mppi_plan: `self.env.step` of
 * `MPPI.advance_time`, trajopt/algos/mppi.py:79) when that environment was built with another `reward_type` than the
 * rollouts are scored with (trajopt/envs/utils.py:8, trajopt/algos/mppi.py:117: the actor-critic scripts plan on
 * 'cooling' and log the 'sparse' reward).  MPPI_REWARD_* or -1 (the default): the rollouts' own.  A tracking or cooling
 * reward reads the reference of mppi_set_reference at time step + 1; -5 from the advancing calls past its end. */
int mppi_set_env_reward(mppi_planner* p, int32_t reward_type);
/* Terminal-value critic of `MPPI.train_step(critic=...)` (trajopt/algos/mppi.py:134-159, 95-101;
 * network trajopt/models/critic_nets.py:47-60): tanh MLP dim -> inner -> inner -> 1 in fp32, row-major
 * weights as torch stores them (W1 [inner][dim], W2 [inner][inner], W3 [inner]).  While set, every
 * rollout scores its last step with the critic's value of the final observation instead of the last
 * reward.  dim selects the input slice (14, 15, 17, 3, 6); dim = 0 removes the critic.  inner must be 128. */
int mppi_set_critic(mppi_planner* p, int32_t dim, int32_t inner, const float* W1, const float* b1, const float* W2,
                    const float* b2, const float* W3, float b3);

/* Noise of `generate_perturbed_actions` (trajopt/utils.py:118-126).
 * upload: raw N(0,1) draws made on the host, sample-major (K,H,M) -- the exact-comparison
 * mode; the sigma scaling and the recursive filter run on the device.
 * generate: counter-based Philox-4x32-10 stream keyed by seed, counter = global sample index. */
int mppi_upload_noise(mppi_planner* p, const double* raw_KxHxM, void* stream);
/* A page-locked host buffer of P*K*H*M doubles that belongs to the planner (made on the first call, freed with the
 * planner): the place to write the draws of `np.random.normal(size=(H, m))` (trajopt/utils.py:120) for
 * mppi_upload_noise / mppi_upload_actions, which copy from it at the bus rate instead of the pageable-memory rate.
 * The uploads return after the copy, so the buffer may be rewritten as soon as they have returned. */
int mppi_host_staging(mppi_planner* p, double** buffer_out);
/* The host draws of all workers of `gather_paths_parallel` made and uploaded in one call, the upload running while
 * the generator works: worker i (trajopt/utils.py:151-152: seed base_seed + i * paths_per_cpu -> seeds[i]) fills
 * samples [i*K/nstreams, (i+1)*K/nstreams) of the staging buffer (mppi_host_staging) with numpy's legacy stream, and
 * every finished stretch goes to the device on a side stream.  The planner's own stream is left alone, so the call may
 * overlap the rollouts of the previous planning step; mppi_commit_noise then applies sigma and the filter on the
 * planner's stream (behind the upload) and makes the draws the noise of the next rollouts.  state_out as for
 * mppi_numpy_normals (one stream only), or NULL.  Single planners (P = 1). */
int mppi_draw_noise(mppi_planner* p, const uint32_t* seeds, int32_t nstreams, int32_t nthreads, uint32_t* state_out);
int mppi_commit_noise(mppi_planner* p, void* stream);
int mppi_generate_noise(mppi_planner* p, uint64_t seed, void* stream);
/* Same, with the stream key = seed + the planner's current timestep read on the device: the
 * reference re-seeds every planning step with `self.seed + t` (trajopt/algos/mppi.py:130-132). */
int mppi_generate_noise_step(mppi_planner* p, uint64_t seed, void* stream);
/* `act_list` of `do_env_rollout` (trajopt/utils.py:59): explicit action sequences (K,H,M);
 * the next mppi_rollout uses them verbatim. */
int mppi_upload_actions(mppi_planner* p, const double* act_KxHxM, void* stream);

/* `do_env_rollout` + `score_trajectory` for all K samples (trajopt/utils.py:59-102,
 * trajopt/algos/mppi.py:91-108). */
int mppi_rollout(mppi_planner* p, void* stream);
/* niter x [rollouts, `MPPI.update`] with the same noise, as the loop of `MPPI.train_step`
 * (trajopt/algos/mppi.py:131-193, 56-65); single-rank form. */
int mppi_iterate(mppi_planner* p, int32_t niter, void* stream);
/* Sharded form of one iteration: rollouts + this shard's sufficient statistics
 * (Rmax, argmax, sum S, sum S*eps) into the buffer at mppi_partials_ptr; after the caller
 * has all-gathered the nranks buffers (rank order), mppi_merge applies `MPPI.update`.
 * Replaces the result concatenation of `gather_paths_parallel` (trajopt/utils.py:158-161). */
int mppi_local_partials(mppi_planner* p, void* stream);
int mppi_partials_ptr(mppi_planner* p, void** device_ptr, size_t* bytes);
int mppi_merge(mppi_planner* p, const void* gathered_device_ptr, int32_t nranks, void* stream);

/* The same exchange inside the library: after mppi_comm_init, mppi_iterate and mppi_plan run the sharded form by
 * themselves -- statistics, one RCCL all-gather of 3 + H*M doubles per planner and rank on the planner's stream,
 * merge -- so that a whole planning step, exchange included, is one captured hipGraph and every rank ends an
 * iteration with the identical act_sequence.  Rank r must own the samples [r*K, (r+1)*K) (cfg.k_offset = r*K),
 * and all ranks the same start state, base actions and parameters.  mppi_comm_unique_id makes the 128-byte id on
 * one rank; the caller hands it to the others (any channel: MPI, torch.distributed, a file) and every rank calls
 * mppi_comm_init, which is collective.  RCCL is looked up at run time (librccl.so); -7 reports its errors.
 * Replaces the worker pool's result collection (trajopt/utils.py:157-161). */
int mppi_comm_unique_id(void* id_out_128, size_t bytes);
int mppi_comm_init(mppi_planner* p, const void* id_128, size_t bytes, int32_t rank, int32_t nranks);

/* Horizon shift and tail fill of `MPPI.advance_time` (trajopt/algos/mppi.py:85-89);
 * override_seq_HxM may be NULL (use the planner's own sequence). */
int mppi_shift(mppi_planner* p, const double* override_seq_HxM, void* stream);

/* Device-resident `MPPI.advance_time` (trajopt/algos/mppi.py:67-89): steps the real environment
 * in fp64 from the planner's own start state with act_sequence[0], stores the new state as the
 * next start state (timestep + 1), appends a row [qpos', qvel', hand, target, reward, action]
 * (3M+7 doubles) to a ring of MPPI_LOG_CAP rows, and shifts the horizon.  No host round trip.
 * Horizons up to H * M = 4096 (the shifted sequence travels through registers of one block); longer ones return -1 from
 * mppi_advance, mppi_advance_noise and mppi_plan, and their callers step with mppi_env_step and mppi_shift instead.
 * mppi_read_log copies the most recent rows (oldest first); total = rows ever written. */
int mppi_advance(mppi_planner* p, void* stream);
/* nsteps whole planning steps -- the loop `for t: agent.train_step(niter)` of the example scripts
 * (sandbox/examples/reacher_7dof_mppi.py:49-50) -- enqueued as replays of one captured hipGraph:
 * Philox noise keyed by seed + timestep, niter x (rollouts, update), device-resident advance.
 * A call for eight steps or more replays a graph of eight consecutive steps (the kernels read the
 * time step from the device-resident state, so one capture serves every step); same results.
 * Nothing returns to the host until mppi_read_log / mppi_read.  On handles of up to 32768 samples the
 * noise of step t+1 is drawn in the advance launch of step t (see mppi_advance_noise); afterwards the
 * noise buffer (MPPI_READ_ACTIONS) therefore belongs to the step that has not been planned yet. */
int mppi_plan(mppi_planner* p, uint64_t seed, int32_t niter, int32_t nsteps, void* stream);
/* mppi_advance and, in the same launch on otherwise idle CUs, mppi_generate_noise_step for the NEXT
 * planning step (stream key seed + new timestep); must follow the rollouts of the current step.
 * A loop  generate_noise_step; { iterate; advance_noise; }*  produces the same trajectory as
 * { generate_noise_step; iterate; advance; }*.  No counterpart in the reference (its noise is drawn
 * inside every worker, trajopt/utils.py:135-139). */
int mppi_advance_noise(mppi_planner* p, uint64_t seed, void* stream);
/* The status word (MPPI_READ_STATUS) is sticky: a bit stays set until it is cleared here.  No counterpart in the
 * reference (its simulator prints a warning and carries on). */
int mppi_clear_status(mppi_planner* p);
int mppi_read_log(mppi_planner* p, int32_t planner, double* rows, int32_t max_rows, int32_t* n_rows, int32_t* total);
/* What one `MPPI.train_step` hands back to its caller (trajopt/algos/mppi.py:74-89), after mppi_advance,
 * mppi_advance_noise or mppi_plan: the newest log row of every planner (P x (3M+7) doubles, layout as in
 * mppi_advance) and the shifted act_sequence (P x H*M doubles).  Either pointer may be NULL.
 * A single planner (P = 1) whose newest step was a lone advance or a single step of mppi_plan gets them without a copy
 * from its second read-back on (the first one tells the library that steps are read back):
 * that step's kernel also writes both, and then the count of steps logged, into page-locked host memory of the planner,
 * and this call only waits for the count (it looks at the stream now and then, so a launch that failed cannot hang
 * it).  Every other case (batched planners, the last step inside a multi-step replay, act_sequence changed since by
 * mppi_set_base_actions / mppi_iterate / mppi_merge / mppi_shift) is two copies behind the stream of the last step and
 * one wait. */
int mppi_read_step(mppi_planner* p, double* rows_out, double* act_seq_out);

int mppi_read(mppi_planner* p, int32_t what, void* dst, size_t bytes);

/* Timing hook (the reference only prints wall-clock around the whole experiment,
 * sandbox/examples/reacher_7dof_mppi.py:47,61): while enabled, every rollout launch is
 * bracketed by HIP events on its stream; read returns the summed kernel time and the
 * number of launches since the last read. */
int mppi_profile(mppi_planner* p, int32_t enable);
int mppi_profile_read(mppi_planner* p, double* rollout_ms_total, int32_t* launches);
int mppi_synchronize(mppi_planner* p);

/* One `env.step(a)` / `sim.forward()` of the real environment in fp64 on the device
 * (trajopt/envs/reacher_env.py:43-85, 151-160).  obs_out holds 2*nv+6 values
 * [qpos, qvel, hand(stale), target]; ref may be NULL. */
int mppi_env_step(const mppi_model* m, int32_t device, const double* qpos, const double* qvel, const double* target3,
                  const double* act, int32_t timestep, int32_t reward_type, const double* ref_Tx3, int32_t T,
                  double alpha, double* qpos_out, double* qvel_out, double* obs_out, double* reward_out);
int mppi_env_forward(const mppi_model* m, int32_t device, const double* qpos, double* hand3_out);

/* `MPPI.update(paths)` for caller-supplied trajectories (trajopt/algos/mppi.py:56-65):
 * actions (K, H*M), returns R (K) -> act_sequence (H*M), computed on the device. */
int mppi_update_from_returns(int32_t device, const double* act_KxHM, const double* R, int32_t K, int32_t HM,
                             double kappa, double* act_seq_out);

/* The arithmetic of mppi_update_from_returns for callers that cannot allocate device memory: code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc.  No counterpart in the reference.
 * Inputs and the output are DEVICE pointers (actions (K, H*M), returns (K), act_sequence (H*M)); the two kernels
 * are enqueued on `stream` (NULL: the default stream) and nothing is synchronised, allocated or freed.  The
 * weights S_k live in a fixed array in the library's device image, (MPPI_NOALLOC_MAX_K + 32) doubles = 2 MiB on
 * every GPU the library runs on, held until the process ends; K above MPPI_NOALLOC_MAX_K is rejected (-6).
 * All calls on one GPU share that array: calls whose kernels may overlap on one GPU (different streams or
 * host threads) race and must be ordered by the caller.  The array's address is looked up once per device, at
 * mppi_planner_create or at the first call on that device (which should therefore be made before a capture). */
int mppi_update_from_returns_noalloc(int32_t device, const double* d_act_KxHM, const double* d_R, int32_t K, int32_t HM,
                                     double kappa, double* d_act_seq_out, void* stream);

/* Host utility: the first n values of numpy's legacy normal stream after `np.random.seed(seed)` -- what the
 * reference's rollout workers draw (trajopt/utils.py:118-139) -- bit for bit, generated in bulk on nthreads host
 * threads (several times faster than numpy's one-value-at-a-time generator).  state_out, if not NULL, receives 628
 * words: what np.random.get_state() would hold afterwards (key[624], pos, has_gauss, cached variate as two words). */
int mppi_numpy_normals(uint32_t seed, size_t n, double* out, int32_t nthreads, uint32_t* state_out);
/* The streams of several pool workers side by side (trajopt/utils.py:151-163: worker i re-seeds its own generator with
 * seeds[i]): n_each values per stream into out + i * n_each, the threads shared between the streams. */
int mppi_numpy_normals_many(const uint32_t* seeds, int32_t nstreams, size_t n_each, double* out, int32_t nthreads);

/* Test hook of the host generator: scales the surplus of rejection-sampling attempts its first round makes (1: normal;
 * negative: the first round falls short and the stream is made again with a larger surplus). */
void mppi_numpy_normals_test_margin(double scale);

/* Philox-4x32-10 block function (host), for known-answer tests of the device generator. */
void mppi_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

#ifdef __cplusplus
}
#endif
#endif
