import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_manifest():
    with open(os.path.join(GOLDEN, "MANIFEST.json")) as f:
        return json.load(f)


def golden_cases():
    """[(case, env, kwargs)] with tuple-valued kwargs restored."""
    out = []
    for case, env, kw in golden_manifest()["cases"]:
        kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in kw.items()}
        out.append((case, env, kw))
    return out


def load_golden(mode, case):
    return dict(np.load(os.path.join(GOLDEN, "mode%s_%s.npz" % (mode, case))))


def saturate_tag_compact(env, state):
    """The packed Tag state keeps num_opp in 7 signed bits, saturating at -64
    (every negative value behaves identically)."""
    if env == "tag":
        state = np.array(state, dtype=np.int64)
        state[..., -1] = np.maximum(state[..., -1], -64)
    return state


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import oracle_lib as ol
    ol.build()
    return ol
