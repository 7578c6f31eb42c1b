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


def g17(which):
    """The whole-graph fixture of a BASELINE configuration (tests/golden/make_golden_g17.py: every row recomputed with the oracle's
    statement of the reference loop NNG:110-198 on the CPU) and the entries it belongs to, in the order every test uses:
    (seqs, best int32[n], row_ptr int64[n + 1], cols uint32[])."""
    import hashlib
    import numpy as np
    from isocon_amd import synth
    args = {"c2": (5000, 1500, 3, 20001), "c3": (50000, 2500, 10, 30001)}[which]
    z = np.load(os.path.join(ROOT, "tests", "golden", "g17_%s_graph.npz" % which))
    accs, seqs, _ = synth.make_reads(*args)
    seqs = sorted(dict.fromkeys(seqs), key=len)
    h = hashlib.sha1()
    for x in seqs:
        h.update(x.encode())
        h.update(b"\n")
    assert h.hexdigest() == str(z["inputs_sha1"]), "the synthetic generator no longer produces the set the fixture was made from"
    return seqs, z["best"], z["row_ptr"], z["cols"]
