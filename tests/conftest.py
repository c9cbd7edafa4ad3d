import json
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


# MB_ROLLING: the library sends small batches of a rolling Forward through the tile pipeline (faster when there are fewer
# pairs than CUs).  The tests want the rolling kernel itself, so the threshold is switched off here; one test
# (test_rolling_small_batch_uses_pipeline) restores it.
os.environ.setdefault("MB_ROLLING_MIN_PAIRS", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_path(*parts):
    return os.path.join(GOLDEN, *parts)


def load_json(*parts):
    with open(golden_path(*parts)) as f:
        return json.load(f)


def load_matrix_json(*parts):
    """DPMatrix::writeJson output (src/dpmatrix.defs.h:39-53) contains bare -inf, which is not JSON."""
    txt = open(golden_path(*parts)).read()
    cells = {}
    for m in re.finditer(r'"inPos":\s*(\d+),\s*"outPos":\s*(\d+),\s*"state":\s*("[^"]*"|\d+),\s*"logLike":\s*([-+\w.]+)', txt):
        v = m.group(4)
        cells[(int(m.group(1)), int(m.group(2)), m.group(3))] = float("-inf") if v == "-inf" else float(v)
    return cells


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle
    oracle.build()
    return oracle


def _machine(name, params=None, useDefaults=False, preset=False):
    from machineboss_amd.machine import Machine
    from machineboss_amd.evalmachine import EvaluatedMachine
    m = Machine.fromFile(golden_path("preset" if preset else "machine", name + ".json"))
    return m, EvaluatedMachine.fromMachine(m, params, useDefaults=useDefaults)


@pytest.fixture(scope="session")
def machines():
    return _machine
