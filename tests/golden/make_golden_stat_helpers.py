#!/usr/bin/env python3
"""Golden fixture tests/golden/g16_stat_helpers.json: the reference's own modules/functions.py helpers of the statistical
test -- get_variant_coordinates (:89-146), get_support (:149-201), get_read_errors (:204-216),
get_empirical_error_probabilities (:435-466), get_read_ccs_probabilities_c / _t (:240-433) -- and
modules/ccs_info.py::fix_quality_values / CCS.read_aln_to_ccs_coord, called directly on random candidate / reference /
read alignments with base qualities (alignments from the oracle; these helpers are pure functions of their inputs).
Floats are stored by repr: the functions are deterministic (no set iteration inside).  Build container only."""
import contextlib
import io
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def main():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(HERE, "shims"))
    sys.path.insert(0, REF)
    import networkx
    if not hasattr(networkx.Graph, "node"):
        networkx.Graph.node = property(lambda g: g.nodes)
    with contextlib.redirect_stdout(io.StringIO()):
        from modules import ccs_info as RC
        from modules import functions as R
    from oracle import oracle as O
    rng = random.Random(16)

    def mut(b, n):
        v = list(b)
        for _ in range(n):
            p = rng.randrange(len(v))
            r = rng.random()
            if r < 0.4:
                v[p] = rng.choice("ACGT")
            elif r < 0.7:
                del v[p]
            else:
                v.insert(p, v[p] if rng.random() < 0.5 else rng.choice("ACGT"))
        return "".join(v)

    def aln(a, b, **kw):
        return list(O.parasail_alignment(a, b, 0, 0, **kw)[2])

    cases = []
    while len(cases) < 70:
        t = "".join(rng.choice("AACGT") for _ in range(rng.randint(40, 130)))
        c = mut(t, rng.randint(1, 4))
        if rng.random() < 0.25:
            c = c[rng.randint(0, 5):]
        if c == t:
            continue
        aln_t, aln_c, _ = aln(t, c, opening_penalty=3, mismatch_penalty=-3, gap_ext=1)
        start, end = R.get_mask_start_and_end(aln_t, aln_c)
        variants = [(i, a, b) for i, (a, b) in enumerate(zip(aln_t, aln_c)) if a != b and start <= i < end]
        if not variants:
            continue
        vt, vc, ac2t, at2c = R.get_variant_coordinates(t, c, aln_t, aln_c, variants)
        reads, rc, rt = {}, {}, {}
        for k in range(8):
            src = c if k % 2 == 0 else t
            x = src if rng.random() < 0.5 else mut(src, rng.randint(0, 2))
            reads["r%d" % k] = x
            (rc if k < 4 else rt)["r%d" % k] = aln(c if k < 4 else t, x)
        qual = {a: [rng.randint(3, 60) for _ in s] for a, s in reads.items()}
        out = {"t": t, "c": c, "aln_t": aln_t, "aln_c": aln_c, "variants": [list(v) for v in variants], "reads_c": rc, "reads_t": rt, "qual": qual,
               "variant_coords_t": [[k, list(v)] for k, v in vt.items()], "variant_coords_c": [[k, list(v)] for k, v in vc.items()],
               "alignment_c_to_t": [[k, v] for k, v in ac2t.items()], "alignment_t_to_c": [[k, v] for k, v in at2c.items()]}
        rcx = {a: (v[0], v[1], tuple(v[2])) for a, v in rc.items()}
        rtx = {a: (v[0], v[1], tuple(v[2])) for a, v in rt.items()}
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                out["support"] = R.get_support(rcx, vc, rtx, vt, ac2t)
        except IndexError:
            out["support"] = "IndexError"
        errors = R.get_read_errors(rcx, rtx)
        out["errors"] = [[a, list(e)] for a, e in errors.items()]
        out["empirical"] = [[a, repr(p)] for a, p in R.get_empirical_error_probabilities(len(t), errors, vt).items()]
        ccs = {a: RC.CCS(a, reads[a], qual[a], "NA") for a in reads}
        for key, fn, ra, v, sn in (("ccs_c", R.get_read_ccs_probabilities_c, rcx, vc, at2c), ("ccs_t", R.get_read_ccs_probabilities_t, rtx, vt, ac2t)):
            try:
                with contextlib.redirect_stdout(io.StringIO()):
                    pr, non = fn(ra, v, sn, ccs, errors, 43)
                out[key] = {"prob": [[a, repr(p)] for a, p in pr.items()], "non_informative": sorted(non)}
            except (AssertionError, IndexError) as e:
                out[key] = type(e).__name__
        cases.append(out)
    fixq = []
    for _ in range(40):
        s = "".join(rng.choice("AACCGT") for _ in range(rng.randint(1, 40)))
        q = [rng.randint(0, 93) for _ in s]
        fixq.append([s, q, RC.fix_quality_values(s, q)])
    json.dump({"generator": "tests/golden/make_golden_stat_helpers.py", "cases": cases, "fix_quality_values": fixq}, open(os.path.join(HERE, "g16_stat_helpers.json"), "w"), separators=(",", ":"))
    print(len(cases), "cases;", sum(isinstance(c["ccs_c"], dict) for c in cases), "with quality probabilities (c);", sum(c["support"] == "IndexError" for c in cases), "IndexError cases;",
          sum(len(c["ccs_c"]["non_informative"]) + len(c["ccs_t"]["non_informative"]) for c in cases if isinstance(c["ccs_c"], dict) and isinstance(c["ccs_t"], dict)), "non-informative reads")


if __name__ == "__main__":
    main()
