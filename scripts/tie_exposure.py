"""How much of IsoCon's OUTPUT depends on the trace-back tie rules that cannot be pinned here (parasail's three decisions with
gap-extend 0; `tie_policy` 0..7 of oracle/isocon_oracle.c and of isocon_sg_trace_batch).

CPU only: the callers of this repo (isocon_amd.partitions / isocon_get_candidates / correction_module: the mirrors of the reference's
modules) run on the oracle's kernels, once per tie policy, on the inputs of the fixtures g8 / g11 / g12 and on a tie-heavy set
(homopolymer-biased indels).  Policy 0 is the baseline; for every other policy the table says what moved:

  stage 1 (first iteration: partition_strings -> get_partition_alignments, isocon_get_candidates.py:37-81)
      pairs         centre/member pairs aligned
      aln           pairs whose gapped strings differ
      edit          pairs whose (mismatches + indels) differs             -- what convergence tests and candidates_step files see
      exon          sequences whose exon-filter verdict differs            (functions.py:23-50)
  stage 2 (correct_strings on stage 1's alignments, correction_module.py:12-75)
      corrected     reads whose corrected sequence differs
  whole phase (find_candidate_transcripts, isocon_get_candidates.py:85-312)
      steps         correction steps until convergence
      cands         candidates (set of sequences) only in one of the two runs
      assign        reads assigned to a different candidate sequence
      read_aln      read->candidate alignments whose gapped strings differ
      read_counts   read->candidate alignments whose (matches, mismatches, indels) differ

Usage: python scripts/tie_exposure.py [out.txt]          (writes profiles/r04_tie_exposure.txt by default)"""
import contextlib
import glob
import io
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from isocon_amd import correction_module as COR  # noqa: E402
from isocon_amd import edlib_alignment_module as EAM  # noqa: E402
from isocon_amd import graphs  # noqa: E402
from isocon_amd import isocon_get_candidates as IGC  # noqa: E402
from isocon_amd import partitions, synth  # noqa: E402
import isocon_amd.SW_alignment_module as SWM  # noqa: E402
from oracle import correction as OC  # noqa: E402
from oracle import oracle as O  # noqa: E402


def use_oracle_kernels():
    """the product modules on the CPU checker instead of the GPU (as tests/test_candidates.py does)"""
    COR._correct_on_device = OC.correct_rows
    graphs.nearest_neighbor_graph = O
    IGC.edlib_align_sequences = O.edlib_align_sequences
    IGC.sw_align_sequences = O.sw_align_sequences
    EAM.edlib_align_sequences_keeping_accession = O.edlib_align_sequences_keeping_accession
    SWM.sw_align_sequences_keeping_accession = O.sw_align_sequences_keeping_accession


class Params(object):
    nr_cores = 1
    neighbor_search_depth = 2 ** 32
    verbose = False
    develop_logfile = None
    logfile = None
    min_exon_diff = 20
    ignore_ends_len = 15
    min_candidate_support = 2
    is_fastq = False
    ccs = None
    outfolder = None


def inputs(small=False):
    g12 = json.load(open(os.path.join(ROOT, "tests", "golden", "g12_candidates.json")))
    out = []
    for name, S in g12["inputs"].items():
        out.append((name, dict(S)))
    accs, seqs, _ = synth.make_reads(150, 500, 3, seed=4401, profile=synth.ONT_PROFILE)
    out.append(("tie_heavy_ont_150x500_3iso", dict(zip(accs, seqs))))
    if small:
        out = [x for x in out if x[0].startswith("synth")]
    return out


def run(S, policy):
    O.TIE_POLICY = policy
    P = Params()
    exon_filtered = set()
    with contextlib.redirect_stdout(io.StringIO()):
        G, partition, M, converged = partitions.partition_strings(dict(S), P)
        pa = IGC.get_partition_alignments(partition, M, G, exon_filtered, P)
        seq_to_acc = IGC.get_unique_seq_accessions(S)
        S_prime, _ = COR.correct_strings(pa, seq_to_acc, {}, 1, nr_cores=1, verbose=False)
        with tempfile.TemporaryDirectory() as tmp:
            P.outfolder = tmp
            read_file = os.path.join(tmp, "reads.fa")
            with open(read_file, "w") as fh:
                for acc, seq in S.items():
                    fh.write(">%s\n%s\n" % (acc, seq))
            cand_file, read_partition, to_realign = IGC.find_candidate_transcripts(read_file, P)
            steps = 1 + len(glob.glob(os.path.join(tmp, "candidates_step_*.fa")))
            cands = {}
            acc = None
            for line in open(cand_file):
                if line.startswith(">"):
                    acc = line[1:].strip()
                else:
                    cands[acc] = line.strip()
    return {"pairs": {(m, s): t for m in pa for s, t in pa[m].items() if s != m}, "exon": set(exon_filtered), "corrected": dict(S_prime),
            "steps": steps, "cands": set(cands.values()),
            "assign": {r: cands[c] for c in read_partition for r in read_partition[c]},
            "read_aln": {r: t for c in read_partition for r, t in read_partition[c].items()}}


def diff(a, b):
    keys = set(a["pairs"]) & set(b["pairs"])
    ra = set(a["read_aln"]) & set(b["read_aln"])
    return {"pairs": len(a["pairs"]),
            "aln": sum(a["pairs"][k][1:3] != b["pairs"][k][1:3] for k in keys) + len(set(a["pairs"]) ^ set(b["pairs"])),
            "edit": sum(a["pairs"][k][0] != b["pairs"][k][0] for k in keys),
            "exon": len(a["exon"] ^ b["exon"]),
            "corrected": sum(a["corrected"].get(k) != b["corrected"].get(k) for k in set(a["corrected"]) | set(b["corrected"])),
            "steps": "%d/%d" % (a["steps"], b["steps"]),
            "cands": len(a["cands"] ^ b["cands"]),
            "assign": sum(a["assign"].get(r) != b["assign"].get(r) for r in set(a["assign"]) | set(b["assign"])),
            "read_aln": sum(a["read_aln"][r][:2] != b["read_aln"][r][:2] for r in ra),
            "read_counts": sum(tuple(a["read_aln"][r][2]) != tuple(b["read_aln"][r][2]) for r in ra)}


COLS = ("pairs", "aln", "edit", "exon", "corrected", "steps", "cands", "assign", "read_aln", "read_counts")


def table(small=False):
    use_oracle_kernels()
    O.build()
    lines = []
    moved = {}
    for name, S in inputs(small):
        base = run(S, 0)
        lines.append("%s: %d reads, %d candidates, %d correction steps under policy 0" % (name, len(S), len(base["cands"]), base["steps"]))
        lines.append("  policy " + " ".join("%11s" % c for c in COLS))
        for policy in range(1, 8):
            d = diff(base, run(S, policy))
            moved[(name, policy)] = d
            lines.append("  %6d " % policy + " ".join("%11s" % d[c] for c in COLS))
    O.TIE_POLICY = 0
    return lines, moved


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r04_tie_exposure.txt")
    lines, moved = table()
    head = [ln for ln in __doc__.splitlines()[:26]]
    any_cand = sum(d["cands"] for d in moved.values())
    any_edit = sum(d["edit"] for d in moved.values())
    tail = ["", "summary: over %d (input, policy) runs, candidates differ in %d, the (mismatches + indels) of a first-iteration pair in %d"
            % (len(moved), sum(1 for d in moved.values() if d["cands"]), sum(1 for d in moved.values() if d["edit"])),
            "         (total candidate differences %d, total edit differences %d)" % (any_cand, any_edit)]
    text = "\n".join(head + [""] + lines + tail) + "\n"
    with open(out, "w") as fh:
        fh.write(text)
    print(text)
