"""Pick the committed subset of the reference's stored MuJoCo transitions.

Run where the reference tree is mounted:
    python tests/golden/make_transition_fixture.py [reference_root]
writes tests/golden/reference_transitions.npz (about 1 MB): every transition of
dense_agent.pickle and 116_agent.pickle, every 8th transition of the ac_tests
agents, plus every transition whose start or end state has the wrist sphere
within the table contact margin (the rare contact row) and the known-answer
transition dense_agent k=11.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from extract_reference_transitions import load_all  # noqa: E402
from trajopt_amd.model import get_model, mass_matrix  # noqa: E402


def contact_dist(md, q):
    _, (R, p) = mass_matrix(md, q)
    c = p[md.contact_link] + R[md.contact_link] @ md.contact_pos
    return c[2] - md.contact_radius - md.contact_plane_z


if __name__ == "__main__":
    root = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    data, configs = load_all(root)
    md = get_model("reacher_7dof")
    n = len(data["reward"])
    keep = np.zeros(n, bool)
    keep[data["file_id"] <= 1] = True
    keep[np.arange(n) % 8 == 0] = True
    near = np.array([min(contact_dist(md, data["qp"][i]), contact_dist(md, data["qp_next"][i])) < 0.004
                     for i in range(n)])
    keep |= near
    out = {k: v[keep] for k, v in data.items() if k != "file_names"}
    out["file_names"] = data["file_names"]
    out["n_total"] = np.int64(n)
    np.savez_compressed(os.path.join(HERE, "reference_transitions.npz"), **out)
    print("kept", int(keep.sum()), "of", n, "; near-contact", int(near.sum()))
