"""one C3 bound matrix under a -DISOCON_QM_TIMELINE build (ISOCON_LIB): prints k_qgram_mm's tile timeline and the kernel's HIP-event time"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from isocon_amd import synth
from isocon_amd.store import SeqStore
accs, seqs, _ = synth.make_reads(50000, 2500, 10, 30001)
seqs = sorted(dict.fromkeys(seqs), key=len)
st = SeqStore(seqs)
import numpy as np
a = np.arange(0, 1000, dtype=np.uint32); b = a + 1
for rep in range(3):
    rp, vals = st.qgram_bound_matrix()
print("bytes", len(vals))
