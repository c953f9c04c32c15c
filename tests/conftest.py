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


def fuzz_case_r3(seed, scale=1):
    """Pattern + the three instances of case `seed` of tools/dev/fuzz_gpu.py as the round-3 campaigns drew it (FUZZ_SCALE = scale)."""
    import numpy as np
    from eicos_amd.generate import feasible_batch, random_socp_pattern
    rng = np.random.default_rng(seed)
    n = int(rng.integers(2, 70 * scale)); p = int(rng.integers(0, max(1, n // 2))); l = int(rng.integers(0, 50 * scale))
    nc = int(rng.integers(0, 5 * scale))
    q = [int(rng.choice([1, 2, 3, 4, 7, 12, 33, 40, 64])) for _ in range(nc)]
    if l + sum(q) == 0:
        l = 3
    dens = float(rng.choice([0.05, 0.15, 0.3, 0.6])) / scale
    pat, base = random_socp_pattern(n, p, l, q, density=dens, seed=seed)
    return pat, feasible_batch(pat, base, 0, 3, seed=seed)
