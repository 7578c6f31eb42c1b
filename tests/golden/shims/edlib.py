"""Stand-in for the absent third-party `edlib` module, used ONLY by tests/golden/make_golden.py when it
imports the reference's orchestration in the build container.  Arithmetic is delegated to the CPU oracle."""
from oracle import oracle as _O


def nw_path(query, target):
    """Extended CIGAR (=, X, I, D) of one optimal global unit-cost alignment.  Which optimum edlib reports is not pinned
    by anything in the reference ("parity unpinned"); this stand-in backtracks from the end and prefers, in this order,
    a query-only step ('I'), a target-only step ('D'), a diagonal step -- believed to be edlib's order."""
    n, m = len(query), len(target)
    D = [[0] * (m + 1) for _ in range(n + 1)]
    for i in range(1, n + 1):
        D[i][0] = i
    for j in range(1, m + 1):
        D[0][j] = j
    for i in range(1, n + 1):
        qi = query[i - 1]
        for j in range(1, m + 1):
            D[i][j] = min(D[i - 1][j - 1] + (qi != target[j - 1]), D[i - 1][j] + 1, D[i][j - 1] + 1)
    ops = []
    i, j = n, m
    while i > 0 or j > 0:
        if i > 0 and D[i - 1][j] + 1 == D[i][j]:
            ops.append("I"); i -= 1
        elif j > 0 and D[i][j - 1] + 1 == D[i][j]:
            ops.append("D"); j -= 1
        else:
            ops.append("=" if query[i - 1] == target[j - 1] else "X"); i -= 1; j -= 1
    ops.reverse()
    out, k = [], 0
    while k < len(ops):
        e = k
        while e < len(ops) and ops[e] == ops[k]:
            e += 1
        out.append("%d%s" % (e - k, ops[k]))
        k = e
    return D[n][m], "".join(out)


def align(query, target, mode="NW", task="distance", k=-1):
    if mode == "HW" and task == "path":
        # infix mode with location and path: oracle/isocon_oracle.c section 5 (edlib's published semantics; the choice
        # among equally good paths is "parity unpinned")
        r = _O.hw_path(query, target, k)
        r["alphabetLength"] = len(set(query) | set(target))
        return r
    if mode != "NW":
        raise NotImplementedError("shim supports NW, and HW with task='path'")
    if task == "path":
        ed, cigar = nw_path(query, target)
        return {"editDistance": ed, "alphabetLength": len(set(query) | set(target)), "locations": [(0, len(target) - 1)], "cigar": cigar}
    ed = _O.ed_bounded(query, target, k)
    return {"editDistance": ed, "alphabetLength": len(set(query) | set(target)),
            "locations": [(None, len(target) - 1)] if ed >= 0 else [], "cigar": None}
