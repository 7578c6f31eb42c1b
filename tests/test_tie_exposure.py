"""The tie-exposure table (scripts/tie_exposure.py -> profiles/r04_tie_exposure.txt): what moves in IsoCon's output when the
trace-back tie rules that cannot be pinned here (parasail, gap-extend 0) are varied.  CPU only, on the oracle's
kernels.  Recomputes the rows of one input for two policies and compares them with the committed table; checks the one thing every
policy must agree on: the optimal score, and alignments that reproduce their inputs (correction_module.py:273-275)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def committed_rows(name):
    rows = {}
    take = False
    for line in open(os.path.join(ROOT, "profiles", "r04_tie_exposure.txt")):
        if line.startswith(name + ":"):
            take = True
            continue
        if take:
            f = line.split()
            if not f or not f[0].isdigit():
                if f and f[0] == "policy":
                    continue
                break
            rows[int(f[0])] = f[1:]
    return rows


def test_rows_of_the_committed_table_are_reproducible(monkeypatch):
    import tie_exposure as T
    from isocon_amd import correction_module as COR
    from isocon_amd import edlib_alignment_module as EAM
    from isocon_amd import graphs
    from isocon_amd import isocon_get_candidates as IGC
    import isocon_amd.SW_alignment_module as SWM
    from oracle import correction as OC
    from oracle import oracle as O
    monkeypatch.setattr(COR, "_correct_on_device", OC.correct_rows)
    monkeypatch.setattr(graphs, "nearest_neighbor_graph", O)
    monkeypatch.setattr(IGC, "edlib_align_sequences", O.edlib_align_sequences)
    monkeypatch.setattr(IGC, "sw_align_sequences", O.sw_align_sequences)
    monkeypatch.setattr(EAM, "edlib_align_sequences_keeping_accession", O.edlib_align_sequences_keeping_accession)
    monkeypatch.setattr(SWM, "sw_align_sequences_keeping_accession", O.sw_align_sequences_keeping_accession)
    monkeypatch.setattr(O, "TIE_POLICY", 0)
    name, S = [x for x in T.inputs(small=True)][0]
    want = committed_rows(name)
    assert sorted(want) == list(range(1, 8))
    base = T.run(S, 0)
    for policy in (1, 4):
        d = T.diff(base, T.run(S, policy))
        assert [str(d[c]) for c in T.COLS] == want[policy], policy
    # this input is the one where the risk shows: the rule for "open or extend" (bit 0) changes the final candidates
    assert int(want[1][T.COLS.index("cands")]) > 0 and int(want[4][T.COLS.index("cands")]) == 0


def test_scores_and_coverage_do_not_depend_on_the_policy():
    import random
    from oracle import oracle as O
    rng = random.Random(8)
    for _ in range(20):
        a = "".join(rng.choice("ACGT") * rng.randrange(1, 4) for _ in range(rng.randrange(20, 60)))
        b = list(a)
        for _ in range(rng.randrange(1, 6)):
            i = rng.randrange(len(b))
            b[i:i + rng.randrange(0, 3)] = rng.choice(["", "A", "TT", "G"])
        b = "".join(b)
        ref = O.sg_trace(a, b, 2, -2, 2, 0, 0)
        for policy in range(8):
            r = O.sg_trace(a, b, 2, -2, 2, 0, policy)
            assert r["score"] == ref["score"]
            x, y = O.cigar_to_seq(r["cigar"], a, b)
            assert x.replace("-", "") == a and y.replace("-", "") == b and len(x) == len(y)
