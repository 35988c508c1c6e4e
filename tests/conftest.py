import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")
LABELS = ["big_dog", "combination", "empty", "small_dog"]  # src/model/test_model.labels:1
CLIPS = ["test_empty", "test_smalldog", "test_bigdog", "test_combo"]  # src/model/model_test.go:54-71


# the tests compare kernel forms through the library's developer switches (APR_WINOGRAD=0 ...); a release process ignores those
# unless APR_DEV_KNOBS=1 (csrc/engine.hip: dev_getenv).  test_release_library_ignores_developer_switches checks the other side.
os.environ["APR_DEV_KNOBS"] = "1"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu() -> bool:
    try:
        from apr_amd import _capi
        return _capi.lib().apr_device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def clips():
    return {c: np.fromfile(os.path.join(GOLDEN, c + ".dat"), np.uint8) for c in CLIPS}


@pytest.fixture(scope="session")
def he_weights():
    from apr_amd import weights as W
    return W.random_init(4, 0, "he")


@pytest.fixture(scope="session")
def ramp_pcm():
    # createTestAudioBuffer(true), src/model/model_test.go:15-23
    return (np.arange(192000) % 256).astype(np.uint8)
