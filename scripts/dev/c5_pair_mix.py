"""Who are the pairs of the wide-band phase?  C5-shaped reads (ONT profile, 1-5 kb, 50 isoforms in 5 families): for every read the partners inside
its length window |len difference| <= threshold (its final NN distance, capped at 511: what the phase can at best know), by relation -- same isoform,
same family but another isoform, another family -- and what a positional q-gram bound could remove."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from isocon_amd import synth
from isocon_amd.store import SeqStore
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
accs, seqs_all, iso = synth.make_reads(n, 0, 50, 50001, profile=synth.ONT_PROFILE, families=5, length_range=(1000, 5000))
first = {}
for a, s in zip(accs, seqs_all):
    first.setdefault(s, int(a.rsplit("_", 1)[1]))
seqs = sorted(first, key=len)
isoform = np.array([first[s] for s in seqs])
family = isoform // 10
lens = np.array([len(s) for s in seqs], dtype=np.int64)
print("isoform lengths by family:", [[len(iso[f * 10 + i]) for i in range(10)] for f in range(5)])
st = SeqStore(seqs)
t0 = time.time(); best, rp, cols, stats = st.nn_graph(); print("graph %.2f s, kernels %.1f ms, pairs evaluated %.4g, columns %.4g" % (time.time() - t0, stats["kernel_ms"], stats["pairs_evaluated"], stats["cells_columns"]))
k = np.minimum(np.where(best >= 0, best, lens), 511)
k = np.minimum(k, lens)
print("thresholds: median %d, p10 %d, p90 %d; k / len median %.3f" % (np.median(k), np.percentile(k, 10), np.percentile(k, 90), np.median(k / lens)))
tot = {"same isoform": 0, "same family": 0, "other family": 0}
def count_in(win_lo, win_hi, sel_lens_sorted):
    return np.searchsorted(sel_lens_sorted, win_hi, "right") - np.searchsorted(sel_lens_sorted, win_lo, "left")
all_sorted = lens
for i_iso in range(50):
    q = np.flatnonzero(isoform == i_iso)
    if not len(q):
        continue
    lo, hi = lens[q] - k[q], lens[q] + k[q]
    same_iso = count_in(lo, hi, np.sort(lens[isoform == i_iso])) - 1
    same_fam = count_in(lo, hi, np.sort(lens[family == i_iso // 10])) - 1 - same_iso
    everybody = count_in(lo, hi, all_sorted) - 1
    tot["same isoform"] += int(same_iso.sum()); tot["same family"] += int(same_fam.sum()); tot["other family"] += int((everybody - same_iso - same_fam).sum())
s = sum(tot.values())
print("window pairs (both directions):", {a: "%.3g (%.1f %%)" % (b, 100.0 * b / s) for a, b in tot.items()}, "total %.4g" % s)
# what the alignments cost by relation: a sample of pairs per class through the unbounded distance -> the column at which a band of threshold k is abandoned ~ k / (d / len)
rng = np.random.default_rng(1)
for name, pick in (("same isoform", lambda i: np.flatnonzero(isoform == isoform[i])), ("same family", lambda i: np.flatnonzero((family == family[i]) & (isoform != isoform[i]))),
                   ("other family", lambda i: np.flatnonzero(family != family[i]))):
    a, b = [], []
    for i in rng.choice(len(seqs), 400, replace=False).tolist():
        c = pick(i)
        c = c[np.abs(lens[c] - lens[i]) <= k[i]]
        if len(c):
            a.append(i); b.append(int(rng.choice(c)))
    if not a:
        print(name, "no pairs in the window"); continue
    a, b = np.array(a), np.array(b)
    d = st.ed_pairs(a, b, None)
    rate = d / np.minimum(lens[a], lens[b])
    est_cols = np.minimum(np.minimum(lens[a], lens[b]), k[a] / np.maximum(rate, 1e-9))
    print("%-13s %4d sampled pairs: distance / length median %.3f; within threshold %.1f %%; estimated columns before the band is abandoned: median %.0f of %.0f" % (
        name, len(a), np.median(rate), 100.0 * (d <= k[a]).mean(), np.median(est_cols), np.median(np.minimum(lens[a], lens[b]))))
st.close()
