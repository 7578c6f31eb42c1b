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
