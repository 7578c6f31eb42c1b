import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from isocon_amd import synth
from isocon_amd import SW_alignment_module as SWM
from isocon_amd.edlib_alignment_module import _intern
from isocon_amd.store import SeqStore
accs, seqs, iso = synth.make_reads(50000, 2500, 10, 30001)
pairs = [(iso[int(a.split("_")[-1])], s) for a, s in zip(accs, seqs)]
for rep in range(2):
    t0 = time.time(); sq, a, b = _intern(pairs); t1 = time.time()
    st = SeqStore(sq); t2 = time.time()
    mm = np.full(len(pairs), -2, np.int8)
    aln_a, aln_b, ptr, res = st.sg_strings(a, b, mm); t3 = time.time()
    A = aln_a.decode("ascii"); B = aln_b.decode("ascii"); p = ptr.tolist(); c = res[:, 3:6].tolist()
    out = [(A[p[i]:p[i + 1]], B[p[i]:p[i + 1]], tuple(c[i])) for i in range(len(pairs))]; t4 = time.time()
    st.close(); t5 = time.time()
    print("intern %.2f store %.2f sg_strings %.2f python-build %.2f close %.2f" % (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4))
