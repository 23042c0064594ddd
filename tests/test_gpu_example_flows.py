"""The flows of the reference's example scripts on the GPU planner, step by step against the CPU oracle.

The main example (sandbox/examples/reacher_7dof_mppi.py): tracking the stored dense agent's hand trajectory -- see
test_main_example_script_tracks_the_stored_agent.

The actor-critic use of the planner as the reference's script wires it (sandbox/examples/ac_reacher_7dof_mppi.py:
146-229): the rollouts are scored with the 'cooling' reward against the hand positions of a stored dense solution
(alpha cooled from epoch to epoch) and end in the critic's value (trajopt/algos/mppi.py:95-101, 134-159), the real
environment reports the sparse reward, every epoch builds a new `MPPI`, the solution goes through
`Critic.compress_agent` into a `ReplayBuffer`, and the critic is trained between the epochs (so the device critic has to
follow the new weights).  Every planning step is checked against the same step on the CPU oracle."""
import os

import numpy as np
import pytest

from oracle import ref_mppi

pytestmark = pytest.mark.gpu

FILTER = [np.ones(7), 0.25, 0.8, 0.0]
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
H, K, NITER, SEED = 16, 40, 5, 12345


def _trained_critic(dim=14):
    """The critic the reference trained and stored for this input (tests/golden/critic_vectors.npz)."""
    import torch
    from trajopt.models.critic_nets import Critic
    g = np.load(os.path.join(GOLDEN, "critic_vectors.npz"))
    critic = Critic(num_iters=30, input_dim=dim, inner_layer=128, batch_size=128, gamma=0.9)
    critic.load_state_dict({"linear1.weight": torch.tensor(g["w%d_W1" % dim]), "linear1.bias": torch.tensor(g["w%d_b1" % dim]),
                            "linear2.weight": torch.tensor(g["w%d_W2" % dim]), "linear2.bias": torch.tensor(g["w%d_b2" % dim]),
                            "linear3.weight": torch.tensor(g["w%d_W3" % dim]), "linear3.bias": torch.tensor(g["w%d_b3" % dim])})
    critic.eval()
    critic.float()
    return critic


def _stored_dense_solution():
    """Joint positions of a dense-reward solution (the oracle's closed loop of the example script, goal (0, 0, 0))."""
    g = np.load(os.path.join(GOLDEN, "closed_loop_oracle.npz"))
    return np.concatenate((np.zeros((1, 7)), g["qpos"][0])), np.concatenate((np.zeros((1, 7)), g["qvel"][0]))


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_actor_critic_epochs_against_the_oracle(oracle_sim, dtype):
    import torch
    from trajopt.algos.mppi import MPPI
    from trajopt.envs.utils import get_environment
    from trajopt.utils import ReplayBuffer, Tuple
    goal = np.zeros(3)
    # the tracked reference: hand positions of the stored solution, read through the environment like the script does
    qp, qv = _stored_dense_solution()
    probe = get_environment('reacher_7dof')
    reference = []
    for t in range(len(qp)):
        probe.set_env_state(dict(qp=qp[t], qv=qv[t], target_pos=goal, timestep=t))
        reference.append(np.array(probe.data.site_xpos[probe.hand_sid]))
    reference = np.array(reference)
    assert np.abs(reference[7] - oracle_sim.fk_hand(qp[7])).max() < 1e-12

    e = get_environment('reacher_7dof', reward_type='sparse')
    critic = _trained_critic(14)
    optimizer = torch.optim.Adam(critic.parameters(), lr=1e-3)
    replay = ReplayBuffer(max_size=10000)
    steps = 24 if dtype == "f64" else 12
    alpha = 1.0
    # The critic is an fp32 network on both sides, evaluated by different code (matrix-core kernel with hardware
    # exponentials against torch): values of a trained critic differ by a few 1e-5 of their range, and exp(25 dV) turns
    # that into 1e-3 on the action sequence after five updates.  So the first epoch runs one update per step and is
    # taken apart: the device's returns against the oracle's (critic rounding allowed for), the new sequence against
    # the reference update OF THE DEVICE'S RETURNS (1e-9), the real environment's step against the oracle's from the
    # action the device applied (1e-8).  The second epoch -- a new planner object, a critic that has been trained in
    # between -- starts with one such step (the new weights are the ones on the device) and goes on with the script's
    # five updates per step against the free oracle step, inside the band that the critic's rounding needs.
    f64 = dtype == "f64"
    for epoch in range(2):
        e.reset_model(seed=None, goal=(0.0, 0.0, 0.0), alpha=alpha)
        ag = MPPI(e, H=H, paths_per_cpu=K, num_cpu=1, kappa=25.0, gamma=1.0, mean=np.zeros(7), filter_coefs=FILTER,
                  default_act='mean', seed=SEED, reward_type='cooling', reference=reference, dtype=dtype)
        worst_state = worst_seq = worst_seq5 = worst_R = 0.0
        for t in range(steps):
            niter = 1 if (epoch == 0 or t == 0) else NITER
            s0, seq = ag.sol_state[-1], ag.act_sequence.copy()
            assert ag.train_step(critic=critic, niter=niter, goal=goal, dim=14) == []
            if not f64 and t > 0:
                continue                 # the closed loop amplifies fp32 rounding: only the first step is compared
            eps = ref_mppi.filter_noise(ref_mppi.all_raw_normals(SEED + t, K, 1, H, 7), FILTER)
            for _ in range(niter):
                _, nobs, rew = oracle_sim.rollout(s0["qp"], s0["qv"], goal, t, seq + eps, reward_type="cooling",
                                                  reference=reference, alpha=alpha, dump=True)
                with torch.no_grad():
                    V = critic(torch.tensor(nobs[:, -1, :14], dtype=torch.float32)).numpy().reshape(-1).astype(np.float64)
                R = ref_mppi.score(rew, 1.0, critic_rewards=np.broadcast_to(V[:, None], rew.shape))
                if niter == 1:
                    R_dev = ag._planner.returns()
                    worst_R = max(worst_R, np.abs(R_dev - R).max() / max(np.abs(V).max(), 1.0))
                    assert int(np.argmax(R_dev)) == int(np.argmax(R)) or np.sort(R)[-1] - np.sort(R)[-2] < 1e-3
                    if f64:
                        R = R_dev        # the update is checked on the returns the device used
                seq = ref_mppi.update(seq + eps, R, 25.0)
            applied = e.data.ctrl.copy()             # the first action of the device's new sequence
            q, v, _, r, _ = oracle_sim.env_step(s0["qp"], s0["qv"], goal, applied, t, reward_type="sparse")
            worst_state = max(worst_state, np.abs(ag.sol_state[-1]["qp"] - q).max(), np.abs(ag.sol_state[-1]["qv"] - v).max())
            dseq = max(np.abs(applied - seq[0]).max(), np.abs(ag.act_sequence - ref_mppi.shift(seq, "mean", np.zeros(7))).max())
            if niter == 1:
                worst_seq = max(worst_seq, dseq)
            else:
                worst_seq5 = max(worst_seq5, dseq)
            assert ag.sol_reward[-1] == r            # sparse: 100 inside the radius, -100 far away, -10 otherwise
        assert worst_state < 1e-8, (epoch, worst_state)
        assert worst_R < (2e-5 if f64 else 5e-4), (epoch, worst_R)      # of the critic's range
        assert worst_seq < (1e-9 if f64 else 1.0) and worst_seq5 < 5e-2, (epoch, worst_seq, worst_seq5)
        assert len(ag.sol_reward) == steps and set(np.unique(ag.sol_reward)) <= {-10.0, 100.0, -100.0}
        # solution -> replay buffer (twice, as the script does) -> temporal-difference steps of the critic
        samples = critic.compress_agent(ag, dim=14)
        assert len(samples) == steps and type(samples[0]) is Tuple and samples[0].state.shape == (14,)
        assert np.array_equal(samples[3].next_state, ag.sol_obs[4][:14]) and samples[3].reward == ag.sol_reward[3]
        replay.concatenate(samples + critic.compress_agent(ag, dim=14))
        assert len(replay.buffer) == 2 * steps * (epoch + 1)
        before = critic.linear3.weight.detach().clone()
        for _ in range(critic.num_iters):
            batch = replay.get_minibatch(size=min(critic.batch_size, len(replay.buffer)))
            state = torch.tensor(np.array([d.state for d in batch]), dtype=torch.float32)
            nxt = torch.tensor(np.array([d.next_state for d in batch]), dtype=torch.float32)
            reward = torch.tensor([float(d.reward) for d in batch], dtype=torch.float32).unsqueeze(1)
            with torch.no_grad():
                y = reward + critic.gamma * critic(nxt)
            loss = torch.nn.functional.mse_loss(critic(state), y)
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
        assert not torch.equal(before, critic.linear3.weight.detach())      # the next epoch plans with new weights
        alpha *= 0.9


def test_example_actor_critic_script():
    """examples/actor_critic_reacher.py end to end (short): dense solution, two cooled epochs, critic steps."""
    import importlib.util
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "actor_critic_reacher.py")
    spec = importlib.util.spec_from_file_location("actor_critic_reacher", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    results = mod.main(["--epochs", "2", "--steps", "40", "--iters", "20"])
    assert [round(r[0], 6) for r in results] == [1.0, 0.9]
    for alpha, ret, dist, loss in results:
        assert -40.0 * 100.0 <= ret <= 40.0 * 100.0 and np.isfinite(dist) and dist < 0.6 and np.isfinite(loss)


def _stored_dense_agent_states(transitions):
    """qpos / qvel of the 117 states of the reference's stored `dense_agent.pickle` (its 116 transitions are part of
    tests/golden/reference_transitions.npz)."""
    d = transitions
    m = d["file_id"] == [str(n) for n in d["file_names"]].index("dense_agent.pickle")
    order = np.argsort(d["step"][m])
    qp = np.concatenate((d["qp"][m][order], d["qp_next"][m][order][-1:]))
    qv = np.concatenate((d["qv"][m][order], d["qv_next"][m][order][-1:]))
    assert len(qp) == 117 and np.array_equal(qp[1:-1], d["qp_next"][m][order][:-1])
    return qp, qv


def test_main_example_script_tracks_the_stored_agent(oracle_sim, transitions):
    """The reference's main example (sandbox/examples/reacher_7dof_mppi.py:19-50): the hand trajectory of the stored
    dense agent, read through `env.data.site_xpos`, is the reference of a 'tracking' environment and planner (H = 16,
    K = 40, five updates per step, seed 3, 100 steps).  fp64, every step against the oracle restarted from the agent's
    own state; the step after the reference runs out raises IndexError like `self.reference[self.env_timestep]`."""
    from trajopt.algos.mppi import MPPI
    from trajopt.envs.utils import get_environment
    qp, qv = _stored_dense_agent_states(transitions)
    goal = np.zeros(3)
    probe = get_environment('reacher_7dof')
    reference = []
    for t in range(len(qp)):
        probe.set_env_state(dict(qp=qp[t], qv=qv[t], target_pos=goal, timestep=t))
        reference.append(np.array(probe.data.site_xpos[probe.hand_sid]))
    reference = np.array(reference)
    assert reference.shape == (117, 3) and np.linalg.norm(reference[-1]) < 0.05        # the stored agent reached the goal

    e = get_environment('reacher_7dof', reward_type='tracking', reference=reference)
    e.reset_model(seed=3, goal=(0, 0, 0), alpha=1.0)
    ag = MPPI(e, H=H, paths_per_cpu=K, num_cpu=1, kappa=25.0, gamma=1.0, mean=np.zeros(7), filter_coefs=FILTER,
              default_act='mean', seed=3, reward_type='tracking', reference=reference, dtype="f64")
    assert ag._real_env_steps_on_the_device()
    worst = worst_seq = 0.0
    for t in range(100):
        s0, seq = ag.sol_state[-1], ag.act_sequence.copy()
        ag.train_step(niter=NITER, goal=(0, 0, 0))
        eps = ref_mppi.filter_noise(ref_mppi.all_raw_normals(3 + t, K, 1, H, 7), FILTER)
        for _ in range(NITER):
            R = oracle_sim.rollout(s0["qp"], s0["qv"], goal, t, seq + eps, reward_type="tracking", reference=reference)
            seq = ref_mppi.update(seq + eps, R, 25.0)
        q, v, _, r, _ = oracle_sim.env_step(s0["qp"], s0["qv"], goal, seq[0], t, reward_type="tracking", reference=reference)
        worst = max(worst, np.abs(ag.sol_state[-1]["qp"] - q).max(), np.abs(ag.sol_state[-1]["qv"] - v).max(),
                    abs(ag.sol_reward[-1] - r))
        worst_seq = max(worst_seq, np.abs(ag.act_sequence - ref_mppi.shift(seq, "mean", np.zeros(7))).max())
    assert worst < 1e-8 and worst_seq < 1e-6, (worst, worst_seq)
    # the planner follows the stored hand trajectory: mean distance to the tracked point of a few centimetres
    hand = np.array([o[14:17] for o in ag.sol_obs[1:]])
    assert np.linalg.norm(hand - reference[1:101], axis=1).mean() < 0.1
    assert len(ag.sol_state) == 101 and ag.sol_state[-1]["timestep"] == 100
    ag.train_step(niter=1, goal=(0, 0, 0))               # time step 100: rows up to 116 are read
    with pytest.raises(IndexError):
        ag.train_step(niter=1, goal=(0, 0, 0))           # time step 101 would read row 117


def test_example_tracking_script():
    """examples/reacher_tracking.py end to end (short): a dense solution planned, then tracked."""
    import importlib.util
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "reacher_tracking.py")
    spec = importlib.util.spec_from_file_location("reacher_tracking", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    total, mean_gap, final_gap = mod.main(["--steps", "50"])
    assert np.isfinite(total) and mean_gap < 0.15 and final_gap < 0.15
