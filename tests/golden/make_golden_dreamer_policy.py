"""Fixture generator (run in the BUILD container, where /root/reference exists; the outputs are committed, this script never runs
on the GPU box): the weights of the reference's own trained Dreamer agents - ros_agent/checkpoints/<name>/{rssm,actor}.pkl, the
checkpoints its ROS node deploys (ros_agent/models/dreamer/racing_dreamer.py:30-36) - as plain float32 arrays.

    python tests/golden/make_golden_dreamer_policy.py     ->  tests/golden/dreamer_policy_{austria,treitlstrasse}.npz

The pickles are read with an allow-list unpickler: nothing but NumPy array reconstruction may be named in them (they are
`pickle.dump(tf.Module.variables as numpy)`, dreamer/tools.py:26-33).  Array order = `tf.Module.variables` order, identified
by shape: GRU cell (kernel, recurrent kernel, bias [2, 600] = reset_after), Dense img1, img2, img3, obs1 (1 280 = deter 200 +
the 1 080-beam scan: the deployed agent has no encoder, encoder.pkl is empty), obs2; actor h0..h3, hout; reward head h0, h1, hout.
"""
import hashlib
import os
import pickle
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.dreamer_policy_port import ACTOR_KEYS, ACTOR_NORM_KEYS, DECODER_KEYS, REWARD_KEYS, RSSM_KEYS   # noqa: E402

CHECKPOINTS = "/root/reference/ros_agent/checkpoints"
RSSM_SHAPES = [(200, 600), (200, 600), (2, 600), (32, 200), (200,), (200, 200), (200,), (200, 60), (60,), (1280, 200), (200,), (200, 60), (60,)]
ACTOR_SHAPES = [(230, 400), (400,), (400, 400), (400,), (400, 400), (400,), (400, 400), (400,), (400, 4), (4,)]
REWARD_SHAPES = [(230, 400), (400,), (400, 400), (400,), (400, 1), (1,)]        # the reward head (DenseDecoder, models.py:301-318)
ACTOR_NORM_SHAPES = ACTOR_SHAPES[:8] + [(4,)] * 4 + ACTOR_SHAPES[8:]             # actor_version "normalized": + batch normalisation
DECODER_SHAPES = [(230, 64), (64,), (5, 5, 32, 64), (32,), (5, 5, 16, 32), (16,), (6, 6, 8, 16), (8,), (6, 6, 1, 8), (1,)]


class ArraysOnly(pickle.Unpickler):
    ALLOWED = {("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"), ("numpy", "ndarray"), ("numpy", "dtype")}

    def find_class(self, module, name):
        if (module, name) in self.ALLOWED:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"refused to load {module}.{name}: the checkpoint may only hold NumPy arrays")


def read(path, shapes):
    with open(path, "rb") as f:
        raw = f.read()
    arrays = ArraysOnly(__import__("io").BytesIO(raw)).load()
    assert [a.shape for a in arrays] == shapes and all(a.dtype == np.float32 for a in arrays), path
    return list(arrays), hashlib.sha256(raw).hexdigest()


def main():
    for name, directory in (("austria", "austria_dreamer"), ("treitlstrasse", "treitlstrasse_dreamer")):
        rssm, h1 = read(os.path.join(CHECKPOINTS, directory, "rssm.pkl"), RSSM_SHAPES)
        actor, h2 = read(os.path.join(CHECKPOINTS, directory, "actor.pkl"), ACTOR_SHAPES)
        reward, h3 = read(os.path.join(CHECKPOINTS, directory, "reward.pkl"), REWARD_SHAPES)
        out = dict(zip(RSSM_KEYS, rssm))
        out.update(zip(ACTOR_KEYS, actor))
        out.update(zip(REWARD_KEYS, reward))
        out["source"] = np.array(f"ros_agent/checkpoints/{directory}/rssm.pkl sha256 {h1}; actor.pkl sha256 {h2}; reward.pkl sha256 {h3}")
        path = os.path.join(ROOT, "tests", "golden", f"dreamer_policy_{name}.npz")
        np.savez_compressed(path, **out)
        print(path, os.path.getsize(path), "bytes")


def occupancy_agent(directory="treitlstrasse_dreamer_20210224", name="treitlstrasse_occupancy"):
    """treitlstrasse_dreamer_20210224 / _20210220: the agents trained with the lidar_occupancy reconstruction (LidarOccupancyDecoder)."""
    rssm, h1 = read(os.path.join(CHECKPOINTS, directory, "rssm.pkl"), RSSM_SHAPES)
    actor, h2 = read(os.path.join(CHECKPOINTS, directory, "actor.pkl"), ACTOR_NORM_SHAPES)
    decoder, h3 = read(os.path.join(CHECKPOINTS, directory, "decoder.pkl"), DECODER_SHAPES)
    out = dict(zip(RSSM_KEYS, rssm))
    out.update(zip(ACTOR_NORM_KEYS, actor))
    out.update(zip(DECODER_KEYS, decoder))
    assert (out["hnorm_var"] > 0).all()
    out["source"] = np.array(f"ros_agent/checkpoints/{directory}/rssm.pkl sha256 {h1}; actor.pkl sha256 {h2}; decoder.pkl sha256 {h3}")
    path = os.path.join(ROOT, "tests", "golden", f"dreamer_policy_{name}.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
    occupancy_agent()
    occupancy_agent("treitlstrasse_dreamer_20210220", "treitlstrasse_20210220")      # (round 5: the fourth shipped checkpoint, for G12)
