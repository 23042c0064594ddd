"""Known-answer vectors for the terminal-value critic from the reference's own trained weights.

Reads three of the reference's critic `state_dict`s (input width 6, 14 and 17; `torch.load(weights_only=True)`,
nothing in the files is executed), instantiates the reference's `Critic` (trajopt/models/critic_nets.py:12-60,
imported with the simulator modules stubbed as in make_planner_vectors.py), and evaluates it on the input slices
`MPPI.train_step` feeds it (trajopt/algos/mppi.py:138-154) for 96 synthetic 20-dim observations.

Runs only where the reference tree is mounted.   python tests/golden/make_critic_vectors.py [reference_root]
writes tests/golden/critic_vectors.npz:  obs (96, 20);  per width d in (6, 14, 17):  w<d>_W1, _b1, _W2, _b2, _W3,
_b3 (float32, torch layout), v<d> (96,) float32 values of the reference module, src<d> the file it came from.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_planner_vectors import import_reference  # noqa: E402

FILES = {
    6: "trajopt/sandbox/examples/ac_tests/ac_cooling_dim6/ac_test_critic_99_0.pt",
    14: "trajopt/sandbox/examples/ac_tests/ac_sparse_005_cooling_05/ac_test_critic_99.pt",
    17: "trajopt/sandbox/examples/ac_tests/ac_cooling_goals_3_critic_None_10e-3/ac_test_critic_0_0.pt",
}


def slice_obs(obs, dim):
    """Input of the critic for every row of obs, as train_step builds it (mppi.py:138-154)."""
    o = torch.tensor(obs, dtype=torch.float32)
    if dim == 14:
        return o[:, :14]
    if dim == 17:
        return torch.cat((o[:, :14], o[:, -3:]), dim=1)
    if dim == 6:
        return o[:, -6:]
    raise ValueError(dim)


def main():
    root = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    import_reference(root)
    from trajopt.models.critic_nets import Critic
    rs = np.random.RandomState(7)
    # observations of the reacher's range: joint angles, velocities, hand and target positions
    obs = np.concatenate([rs.uniform(-1.5, 1.5, (96, 7)), rs.randn(96, 7) * 2.0, rs.uniform(-0.8, 0.8, (96, 3)),
                          rs.uniform(-0.1, 0.1, (96, 3))], axis=1)
    out = {"obs": obs}
    for dim, rel in FILES.items():
        sd = torch.load(os.path.join(root, rel), weights_only=True, map_location="cpu")
        critic = Critic(input_dim=dim)
        critic.load_state_dict(sd)
        with torch.no_grad():
            v = critic(slice_obs(obs, dim)).numpy().reshape(-1)
        for name, key in (("W1", "linear1.weight"), ("b1", "linear1.bias"), ("W2", "linear2.weight"),
                          ("b2", "linear2.bias"), ("W3", "linear3.weight"), ("b3", "linear3.bias")):
            out["w%d_%s" % (dim, name)] = sd[key].numpy().astype(np.float32)
        out["v%d" % dim] = v.astype(np.float32)
        out["src%d" % dim] = np.array(rel)
        print(dim, rel, "values", v[:3], "range", v.min(), v.max())
    np.savez_compressed(os.path.join(HERE, "critic_vectors.npz"), **out)


if __name__ == "__main__":
    main()
