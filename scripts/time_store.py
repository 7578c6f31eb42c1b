"""Wall time of SeqStore(...) (host join + H2D + device packing) at C3 size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isocon_amd import synth
from isocon_amd.store import SeqStore
accs, seqs, _ = synth.make_reads(50000, 2500, 10, 30001)
seqs = sorted(dict.fromkeys(seqs), key=len)
SeqStore(seqs[:100]).close()
for _ in range(3):
    t = time.perf_counter(); st = SeqStore(seqs); dt = time.perf_counter() - t
    t = time.perf_counter(); fp = st.fingerprint; dfp = time.perf_counter() - t
    print("SeqStore(50k x 2.5kb): %.1f ms; digest %.2f ms (%d)" % (dt * 1e3, dfp * 1e3, fp), flush=True)
    st.close()
