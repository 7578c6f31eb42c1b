"""GPU parity: edlib_alignment_module mirror against fixtures produced by the reference's EAM (under the shim)."""
import pytest

from conftest import golden, list_to_dd, ordered

pytestmark = pytest.mark.gpu


def test_edlib_align_sequences_dict_input():
    from isocon_amd import edlib_alignment_module as EAM
    g = golden("g3_edlib_align.json")
    matches = {k: {s: 0 for s in v} for k, v in g["dict_input"]}
    for cores in ("1", "2"):
        got = EAM.edlib_align_sequences(matches, nr_cores=int(cores))
        assert ordered(got) == ordered(list_to_dd(g["dict_expected"][cores]))


def test_edlib_align_sequences_set_input():
    from isocon_amd import edlib_alignment_module as EAM
    g = golden("g3_edlib_align.json")
    matches = {k: set(v) for k, v in g["set_input"]}
    got = EAM.edlib_align_sequences(matches)
    assert got == list_to_dd(g["set_expected"])          # inner order = set iteration order: compare as mappings
    assert list(got) == [k for k, _ in g["set_expected"]]


def test_edlib_align_sequences_keeping_accession():
    from isocon_amd import edlib_alignment_module as EAM
    g = golden("g3_edlib_align.json")
    matches = {a1: {a2: tuple(v) for a2, v in inner} for a1, inner in g["acc_input"]}
    for cores in ("1", "2"):
        got = EAM.edlib_align_sequences_keeping_accession(matches, nr_cores=int(cores))
        assert ordered(got) == ordered(list_to_dd(g["acc_expected"][cores]))


def test_single_pair_helpers():
    from isocon_amd import edlib_alignment_module as EAM
    from isocon_amd import nearest_neighbor_graph as NNG
    assert EAM.edlib_alignment("ACGTACGT", "ACGTTCGT", 0, 0) == ("ACGTACGT", "ACGTTCGT", 1)
    assert EAM.edlib_alignment("ACGT", "AGT", 0, 0, x_acc="x", y_acc="y") == ("x", "y", ("ACGT", "AGT", 1))
    assert NNG.edlib_ed("ACGTACGT", "TTTTTTTT", k=2) == -1
    assert NNG.edlib_ed("ACGTACGT", "ACGTACGA", k=2) == 1
    assert EAM.edlib_align_sequences({}) == {}


@pytest.mark.gpu
def test_edlib_traceback_nw_mode():
    """EAM.edlib_traceback (row a11, no live caller): distance from the GPU, path by the oracle's tie rule; above k edlib's
    (-1, [], None)."""
    import random
    from isocon_amd import edlib_alignment_module as EAM
    from oracle import oracle as O
    rng = random.Random(2)
    for _ in range(25):
        x = "".join(rng.choice("ACGT") for _ in range(rng.randint(5, 120)))
        y = list(x)
        for _ in range(rng.randint(0, 6)):
            p = rng.randrange(len(y))
            r = rng.random()
            if r < 0.4:
                y[p] = rng.choice("ACGT")
            elif r < 0.7 and len(y) > 2:
                del y[p]
            else:
                y.insert(p, rng.choice("ACGT"))
        y = "".join(y)
        ed, ops = O.nw_path(x, y)
        cigar = "".join("%d%s" % o for o in ops)
        assert EAM.edlib_traceback(x, y, mode="NW", task="path", k=10) == ((ed, [(0, len(y) - 1)], cigar) if ed <= 10 else (-1, [], None))
    with pytest.raises(NotImplementedError):
        EAM.edlib_traceback("ACGT", "ACGT", mode="HW")
