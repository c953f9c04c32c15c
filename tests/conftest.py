import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
# the tests reach every kernel variant through the library's experiment knobs (EICOS_THREADS, EICOS_TILES, ...): those are
# honoured only under this opt-in (eicos_amd/csrc/envknob.hpp), so that a host application's environment cannot switch paths
os.environ["EICOS_EXPERIMENT"] = "1"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def expected():
    with open(os.path.join(GOLDEN, "expected.json")) as f:
        return json.load(f)


def load_fixture(name):
    from eicos_amd.problem_io import read_epb
    return read_epb(os.path.join(GOLDEN, name + ".epb"))


ALL_FIXTURES = ["MPC02", "update_data", "lp_25fv47", "lp_adlittle", "lp_afiro", "lp_agg", "lp_agg2", "lp_agg3",
                "lp_bandm", "lp_beaconfd", "lp_blend", "lp_bnl1", "unboundedLP1", "unboundedMaxSqrt", "infeasible1",
                "emptyProblem", "feas", "issue98"]
