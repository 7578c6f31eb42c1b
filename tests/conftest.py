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


def g19(which):
    """The 2-set whole-graph fixture of a BASELINE configuration (tests/golden/make_golden_g19.py: the reads against a seeded candidate set,
    every read's row by the oracle's statement of the reference loop NNG:341-424 on the CPU): (X, C, merged [(seq, acc)], arrays dict)."""
    import importlib.util
    import numpy as np
    spec = importlib.util.spec_from_file_location("make_golden_g19", os.path.join(ROOT, "tests", "golden", "make_golden_g19.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    X, C = mod.candidates(which)
    merged = mod.merged_list(X, C)
    z = np.load(os.path.join(ROOT, "tests", "golden", "g19_%s_graph_2set.npz" % which))
    assert str(z["inputs_sha1"]) == mod.inputs_sha1(merged), "the fixture belongs to another read / candidate set"
    return X, C, merged, {k: z[k] for k in z.files}


def g18(which):
    """The alignment fixture of a BASELINE configuration (tests/golden/make_golden_g18.py: every pair aligned on the CPU by the oracle's
    statement of SWM:64-86, tie policy 0): the npz as a dict of arrays.  Ids index the configuration's entries (g17's order)."""
    import numpy as np
    z = np.load(os.path.join(ROOT, "tests", "golden", "g18_%s_sw.npz" % which))
    return {k: z[k] for k in z.files}


def ops_of_alignment(a1, a2):
    """two gapped strings -> run-length ops (len << 4 | code; 0 '=', 1 'X', 2 'I' = gap in the second, 3 'D' = gap in the first),
    maximal runs, as include/isocon_hip.h defines them"""
    import numpy as np
    x = np.frombuffer(a1.encode("ascii"), dtype=np.uint8)
    y = np.frombuffer(a2.encode("ascii"), dtype=np.uint8)
    code = np.where(y == 45, 2, np.where(x == 45, 3, np.where(x == y, 0, 1))).astype(np.uint32)
    if len(code) == 0:
        return np.zeros(0, np.uint32)
    cut = np.flatnonzero(np.diff(code) != 0) + 1
    starts = np.concatenate([[0], cut])
    lens = np.diff(np.concatenate([starts, [len(code)]])).astype(np.uint32)
    return (lens << 4) | code[starts]
