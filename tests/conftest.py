import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_lib():
    """The built C-ABI library (built on demand; hipcc cross-compiles without a GPU)."""
    from racing_dreamer_amd import build, _lib
    build.build(verbose=False)
    return _lib.load_library()


@pytest.fixture(autouse=True, scope="session")
def _band_knob_from_environment():
    """Validation hook of tools/band_validation.sh: RC_TEST_BAND_LOG2=<l2> narrows the scan's exact-count band on every
    env the GPU tests create (through the explicit debug call - the library itself reads no environment variable), to
    show that the corner-aimed parity tests catch a band below the rounding bound."""
    l2 = os.environ.get("RC_TEST_BAND_LOG2")
    if not l2:
        yield
        return
    from racing_dreamer_amd import batched_env
    orig = batched_env.BatchedRaceEnv.__init__

    def patched(self, *a, **k):
        orig(self, *a, **k)
        self.debug_set("band_log2", int(l2))

    batched_env.BatchedRaceEnv.__init__ = patched
    yield
    batched_env.BatchedRaceEnv.__init__ = orig
