import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class Params(object):
    """The attributes of the reference's params object that the hot path reads."""

    def __init__(self, nr_cores=1, neighbor_search_depth=2 ** 32):
        self.nr_cores = nr_cores
        self.neighbor_search_depth = neighbor_search_depth
        self.verbose = False
        self.develop_logfile = None


@pytest.fixture
def params_cls():
    return Params


def golden(name):
    import json
    with open(os.path.join(ROOT, "tests", "golden", name)) as f:
        return json.load(f)


def list_to_dd(lst):
    """[[k1, [[k2, v], ...]], ...] -> ordered dict-of-dict (inverse of make_golden.dd_to_list)."""
    return {k1: {k2: (tuple(v) if isinstance(v, list) else v) for k2, v in inner} for k1, inner in lst}


def ordered(dd):
    """dict-of-dict -> nested list, for comparisons that include key order."""
    return [[k1, list(inner.items())] for k1, inner in dd.items()]
