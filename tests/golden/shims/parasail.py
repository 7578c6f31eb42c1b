"""Stand-in for the absent third-party `parasail` module (see edlib.py in this directory)."""
from oracle import oracle as _O

TIE_POLICY = 0


class _Matrix(object):
    def __init__(self, alphabet, match, mismatch):
        self.alphabet, self.match, self.mismatch = alphabet, match, mismatch


def matrix_create(alphabet, match, mismatch):
    return _Matrix(alphabet, match, mismatch)


class _Cigar(object):
    def __init__(self, text):
        self.decode = text.encode("utf-8")


class _Result(object):
    def __init__(self, r):
        self.saturated = False
        self.score = r["score"]
        self.end_query = r["end_query"]
        self.end_ref = r["end_ref"]
        self.cigar = _Cigar(r["cigar"])


def _sg(s1, s2, open_, ext, matrix):
    return _Result(_O.sg_trace(s1, s2, matrix.match, matrix.mismatch, open_, ext, TIE_POLICY))


sg_trace_scan_16 = _sg
sg_trace_scan_32 = _sg
