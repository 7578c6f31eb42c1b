"""SW batch timing at C3 scale: 8192 (centre, read) pairs of ~2.5 kb, with and without the edit-distance band hint."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from isocon_amd import synth
from isocon_amd.edlib_alignment_module import _intern
from isocon_amd.store import SeqStore
accs, seqs, iso = synth.make_reads(8192, 2500, 10, 30001)
pairs = [(iso[int(a.split("_")[-1])], s) for a, s in zip(accs, seqs)]
sq, a, b = _intern(pairs)
st = SeqStore(sq)
ed = st.ed_pairs(a, b, None)
mm = np.full(len(pairs), -2, np.int8)
cells = float((st.lens[a].astype(np.int64) * st.lens[b]).sum())
for name, hint in (("full", None), ("banded", ed)):
    st.sg_trace(a[:64], b[:64], mm[:64], ed_upper=None if hint is None else hint[:64])
    t = time.time(); ops, ptr, res, ms = st.sg_trace(a, b, mm, return_ms=True, ed_upper=hint); dt = time.time() - t
    print("%-6s kernels %.1f ms, wall %.1f ms, %.3g matrix cells/s (kernel), median ed %d, checksum %d" %
          (name, ms, dt * 1e3, cells / (ms / 1e3), int(np.median(ed)), int(res[:, 0].astype(np.int64).sum() + ops.astype(np.int64).sum())))
