"""Drop-in shim: makes this package importable under the names the reference's callers use.

`install()` puts `racing_dreamer_amd/compat` at the front of sys.path, so that
`import racecar_gym` (dreamer/wrappers.py:5, baselines/.../sb_experiment.py:9-10) and
`from agents.gap_follower import GapFollower` (dreamer/dream.py:16) resolve to the MI355X-backed
implementations in this directory.  Nothing else in the reference has to change.
"""
import os
import sys

COMPAT_DIR = os.path.dirname(os.path.abspath(__file__))


def install() -> str:
    if COMPAT_DIR not in sys.path:
        sys.path.insert(0, COMPAT_DIR)
    return COMPAT_DIR
