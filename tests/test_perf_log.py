"""ISOCON_PERF_LOG: one JSON line per device-backed call (isocon_amd/perf_log.py)."""
import json

import pytest

from isocon_amd import perf_log


def test_nothing_is_written_without_the_variable(tmp_path, monkeypatch):
    monkeypatch.delenv("ISOCON_PERF_LOG", raising=False)
    with perf_log.call("x", pairs=3) as rec:
        rec.add(kernel_ms=1.0)
    assert list(tmp_path.iterdir()) == []


def test_one_line_per_call_also_when_the_call_raises(tmp_path, monkeypatch):
    path = tmp_path / "perf.jsonl"
    monkeypatch.setenv("ISOCON_PERF_LOG", str(path))
    with perf_log.call("a", pairs=3) as rec:
        rec.add(kernel_ms=1.5)
    with pytest.raises(RuntimeError):
        with perf_log.call("b", sequences=7):
            raise RuntimeError("boom")
    recs = [json.loads(line) for line in path.read_text().splitlines()]
    assert [r["call"] for r in recs] == ["a", "b"]
    assert recs[0]["pairs"] == 3 and recs[0]["kernel_ms"] == 1.5 and recs[0]["ok"] is True and recs[0]["wall_s"] >= 0
    assert recs[1]["sequences"] == 7 and recs[1]["ok"] is False


@pytest.mark.gpu
def test_the_four_wrappers_record_their_calls(tmp_path, monkeypatch):
    import random
    from isocon_amd import SW_alignment_module, edlib_alignment_module, nearest_neighbor_graph
    rng = random.Random(5)
    base = "".join(rng.choice("ACGT") for _ in range(300))
    seqs = {}
    for i in range(40):
        s = list(base)
        for _ in range(rng.randrange(6)):
            s[rng.randrange(len(s))] = rng.choice("ACGT")
        seqs["r%d" % i] = "".join(s) + "A" * i
    path = tmp_path / "perf.jsonl"
    monkeypatch.setenv("ISOCON_PERF_LOG", str(path))
    params = type("P", (), {"nr_cores": 1, "neighbor_search_depth": 2 ** 32})()
    G, _isolated = nearest_neighbor_graph.compute_nearest_neighbor_graph(seqs, set(), params)
    matches = {s: {t: 0 for t in list(seqs.values())[:3] if t != s} for s in list(seqs.values())[3:8]}
    ed = edlib_alignment_module.edlib_align_sequences(matches, nr_cores=1)
    SW_alignment_module.sw_align_sequences(ed, nr_cores=1)
    calls = [json.loads(line)["call"] for line in path.read_text().splitlines()]
    assert any(c.startswith("nearest_neighbor_graph") for c in calls)
    assert "edlib_alignment_module.distances" in calls and "SW_alignment_module.alignments" in calls
    assert all(json.loads(line)["ok"] for line in path.read_text().splitlines())
    assert G
