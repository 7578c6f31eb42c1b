"""Stand-in for the absent third-party `edlib` module, used ONLY by tests/golden/make_golden.py when it
imports the reference's orchestration in the build container.  Arithmetic is delegated to the CPU oracle."""
from oracle import oracle as _O


def align(query, target, mode="NW", task="distance", k=-1):
    if mode != "NW":
        raise NotImplementedError("shim supports the hot path's NW mode only")
    ed = _O.ed_bounded(query, target, k)
    return {"editDistance": ed, "alphabetLength": len(set(query) | set(target)),
            "locations": [(None, len(target) - 1)] if ed >= 0 else [], "cigar": None}
