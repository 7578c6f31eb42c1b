"""One-shot probe: do the third-party wheels the reference's arithmetic lives in -- edlib, parasail --
import on the GPU box?  If they do, tie-heavy known-answer vectors are generated FROM THE REAL LIBRARIES (third-party wheels,
nothing of /root/reference) into gpurun_out/real_lib_vectors.json, to be committed under tests/golden/ and used to freeze
TIE_POLICY and the HW location / path rules.  If they do not, the ImportErrors are printed and recorded in DESIGN.md."""
import json, os, sys
import numpy as np

out = {"python": sys.version.split()[0]}
mods = {}
for name in ("edlib", "parasail", "pysam"):
    try:
        m = __import__(name)
        mods[name] = m
        out[name] = {"imports": True, "version": getattr(m, "__version__", "?"), "file": getattr(m, "__file__", "?")}
    except Exception as e:
        out[name] = {"imports": False, "error": repr(e)}
print(json.dumps(out, indent=1))


def tie_heavy_pairs(rng, n):
    """homopolymer runs, tandem repeats, exon-length gaps, unequal lengths: alignments with many co-optimal paths"""
    pairs = []
    for i in range(n):
        L = int(rng.integers(30, 400))
        kind = i % 5
        if kind == 0:       # low-complexity
            a = "".join(rng.choice(list("AC"), L))
        elif kind == 1:     # homopolymer blocks
            a = "".join(c * int(rng.integers(1, 9)) for c in rng.choice(list("ACGT"), L // 4))
        elif kind == 2:     # tandem repeat
            u = "".join(rng.choice(list("ACGT"), int(rng.integers(2, 6))))
            a = (u * (L // len(u) + 1))[:L]
        else:
            a = "".join(rng.choice(list("ACGT"), L))
        b = list(a)
        for _ in range(int(rng.integers(0, max(2, L // 12)))):
            p = int(rng.integers(0, max(1, len(b))))
            r = rng.random()
            if r < 0.4 and b:
                del b[p]
            elif r < 0.8:
                b.insert(p, str(rng.choice(list("ACGT"))))
            elif b:
                b[p] = str(rng.choice(list("ACGT")))
        if kind == 4 and len(b) > 80:      # exon-length deletion
            p = int(rng.integers(10, len(b) - 60))
            del b[p:p + int(rng.integers(20, 50))]
        b = "".join(b)
        if i % 7 == 0:
            b = b[int(rng.integers(0, 6)):len(b) - int(rng.integers(0, 6))]
        if a and b:
            pairs.append((a, b))
    return pairs


vec = {"probe": out, "parasail": [], "edlib_nw": [], "edlib_hw": []}
rng = np.random.Generator(np.random.PCG64(4242))
pairs = tie_heavy_pairs(rng, 600)
if "parasail" in mods:
    ps = mods["parasail"]
    for (a, b) in pairs:
        for (mm, op, ex) in ((-1, 2, 0), (-2, 2, 0), (-4, 2, 0), (-3, 3, 0), (-3, 3, 1)):
            mat = ps.matrix_create("ACGT", 2, mm)
            r = ps.sg_trace_scan_16(a, b, op, ex, mat)
            if r.saturated:
                r = ps.sg_trace_scan_32(a, b, op, ex, mat)
            vec["parasail"].append({"s1": a, "s2": b, "match": 2, "mismatch": mm, "open": op, "ext": ex, "score": int(r.score),
                                    "end_query": int(r.end_query), "end_ref": int(r.end_ref), "cigar": r.cigar.decode.decode() if isinstance(r.cigar.decode, bytes) else str(r.cigar.decode)})
if "edlib" in mods:
    ed = mods["edlib"]
    for (a, b) in pairs:
        r = ed.align(a, b, mode="NW", task="path")
        vec["edlib_nw"].append({"q": a, "t": b, "ed": r["editDistance"], "locations": r["locations"], "cigar": r["cigar"]})
        for k in (10, 25, 40):
            r = ed.align(a, b, mode="HW", task="path", k=k)
            vec["edlib_hw"].append({"q": a, "t": b, "k": k, "ed": r["editDistance"], "locations": r["locations"], "cigar": r["cigar"]})
os.makedirs("gpurun_out", exist_ok=True)
if vec["parasail"] or vec["edlib_nw"]:
    json.dump(vec, open("gpurun_out/real_lib_vectors.json", "w"))
    print("wrote gpurun_out/real_lib_vectors.json: %d parasail, %d edlib NW, %d edlib HW vectors" % (len(vec["parasail"]), len(vec["edlib_nw"]), len(vec["edlib_hw"])))
else:
    json.dump(out, open("gpurun_out/real_lib_probe.json", "w"), indent=1)
    print("neither library imports: parity of path ties stays unpinned")
