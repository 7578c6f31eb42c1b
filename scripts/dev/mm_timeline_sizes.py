"""k_qgram_mm's tile timeline (a -DISOCON_QM_TIMELINE build through ISOCON_LIB) on sets of growing size: is the K loop of a tile slower
when the operand panels no longer fit the caches (410 MB of profiles at 50 000 reads against 4 MB of L2 per XCD and 256 MB of MALL)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from isocon_amd import synth
from isocon_amd.store import SeqStore
for n in (2000, 6000, 12000, 25000, 50000):
    accs, seqs, _ = synth.make_reads(n, 2500, 10, 30001)
    seqs = sorted(dict.fromkeys(seqs), key=len)
    st = SeqStore(seqs)
    st.nn_graph()
    sys.stderr.write("== n = %d\n" % n); sys.stderr.flush()
    st.nn_graph()
    st.close()
