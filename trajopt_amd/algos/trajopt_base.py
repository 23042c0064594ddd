"""`Trajectory`: the planner base class callers may subclass
(reference: trajopt/algos/trajopt_base.py:7-53).

State kept by every planner: the env handle, the solution logs (`sol_state`, `sol_act`,
`sol_reward`, `sol_obs`) and the open-loop `act_sequence` of shape (H, m).  The two
`animate_*` methods replay logged states through the env; with the headless env of this
package they only step it (its render hooks do nothing).
"""
import numpy as np


class Trajectory:
    def __init__(self, env, H=32, seed=123):
        self.env = env
        self.seed = seed
        self.H = H
        self.n = env.observation_dim
        self.m = env.action_dim
        env.reset_model(seed=seed)
        self.sol_state = [env.get_env_state()]
        self.sol_obs = [env._get_obs()]
        self.sol_act = []
        self.sol_reward = []
        self.act_sequence = np.zeros((H, self.m))

    def update(self, paths):
        """Fold a set of rolled-out trajectories into `act_sequence`; planners implement it."""
        raise NotImplementedError

    def _replay(self, first, actions, verbose):
        # every step restarts from the logged state, exactly one env step per logged action
        env = self.env
        env.mujoco_render_frames = True
        try:
            for offset, action in enumerate(actions):
                env.set_env_state(self.sol_state[first + offset])
                env.step(action)
                if verbose:
                    print(env.env_timestep)
                    print(env.real_step)
        finally:
            env.mujoco_render_frames = False

    def animate_rollout(self, t, act):
        """Replay the action rows of `act` from solution step t."""
        self.env.set_env_state(self.sol_state[t])
        self._replay(t, list(act), verbose=True)

    def animate_result(self):
        """Replay the whole logged solution."""
        self._replay(0, self.sol_act, verbose=False)
