import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "gpu_slow: GPU tests that take tens of seconds each (the NumPy oracle beside large batches, soaks): "
                                        "NOT part of `-m gpu`, which has to fit the driver's step limit; run them with `-m gpu_slow` on the GPU box")


def pytest_collection_modifyitems(config, items):
    """`gpu_slow` tests are GPU tests too (they carry `gpu` as well), but `-m gpu` - the driver's run, which has a step limit -
    SKIPS them: they run only when the marker expression names them (`-m gpu_slow`)."""
    asked = "gpu_slow" in (config.getoption("-m") or "")
    skip = pytest.mark.skip(reason="gpu_slow: run with `pytest -m gpu_slow` on the GPU box")
    for item in items:
        if "gpu_slow" in item.keywords and not asked:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def hip_lib():
    """The built C-ABI library (built on demand; hipcc cross-compiles without a GPU)."""
    from racing_dreamer_amd import build, _lib
    build.build(verbose=False)
    return _lib.load_library()


REF = "/root/reference"


@pytest.fixture()
def ref_wrappers(monkeypatch):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden
    saved = {k: sys.modules.get(k) for k in ("gym", "gym.spaces", "gym.wrappers", "racecar_gym", "racecar_gym.envs",
                                             "racecar_gym.envs.multi_agent_race", "wrappers")}
    make_golden.install_stubs()                      # gym names only
    for k in [m for m in sys.modules if m == "racecar_gym" or m.startswith("racecar_gym.")]:
        del sys.modules[k]                           # the shim provides racecar_gym, not the stub
    from racing_dreamer_amd import compat
    compat.install()
    import racecar_gym.envs.multi_agent_race as mar
    from oracle_backend import OracleBackend
    monkeypatch.setattr(mar, "_BACKEND", OracleBackend)
    monkeypatch.chdir(os.path.join(REF, "dreamer"))  # RaceCarBaseEnv loads "scenarios/{task}/{track}.yml"
    sys.path.insert(0, os.path.join(REF, "dreamer"))
    sys.modules.pop("wrappers", None)
    import wrappers as W
    W.envs.clear()
    yield W
    sys.path.remove(os.path.join(REF, "dreamer"))
    for k, v in saved.items():
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v


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
