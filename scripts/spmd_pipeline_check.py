"""One process per GPU through the WHOLE pipeline: every rank runs find_candidate_transcripts + stat_filter_candidates on the
same reads, the nearest-neighbour searches are shared between the ranks (isocon_amd.nearest_neighbor_graph picks up the
initialised process group).  Checks that every rank writes the files a single process writes.
  ISOCON_DIST_BACKEND=gloo ISOCON_GPU_DEVICE=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \\
      --master-addr 127.0.0.1 --master-port 29519 scripts/spmd_pipeline_check.py [n_reads]        (two ranks on one GPU)"""
import hashlib, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
local_rank = int(os.environ.get("LOCAL_RANK", "0"))
os.environ.setdefault("ISOCON_GPU_DEVICE", str(local_rank))
import torch
import torch.distributed as dist
from isocon_amd import isocon_get_candidates as IGC
from isocon_amd import isocon_statistical_test as IST
from isocon_amd import nearest_neighbor_graph as NNG
from isocon_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
accs, seqs, isoforms = synth.make_reads(n, 1200, 5, 4242)


def run(tag):
    with tempfile.TemporaryDirectory() as tmp:
        rf = os.path.join(tmp, "reads.fa")
        with open(rf, "w") as fh:
            for a, s in zip(accs, seqs):
                fh.write(">%s\n%s\n" % (a, s))

        class P:
            nr_cores = 1; neighbor_search_depth = 2 ** 32; verbose = False; develop_logfile = None; logfile = None; min_exon_diff = 20
            ignore_ends_len = 15; min_candidate_support = 2; is_fastq = False; ccs = None; outfolder = tmp
            p_value_threshold = 0.01; min_test_ratio = 5; max_phred_q_trusted = 43
        t = time.time()
        cand_file, rp, to_realign = IGC.find_candidate_transcripts(rf, P)
        IST.stat_filter_candidates(rf, cand_file, rp, to_realign, P)
        dt = time.time() - t
        digest = hashlib.sha1()
        for f in ("candidates_converged.fa", "final_candidates.fa", "cluster_info.tsv"):
            digest.update(open(os.path.join(tmp, f), "rb").read())
        return digest.hexdigest(), dt


alone, t_alone = run("single")          # no process group yet: the single-GPU path
backend = os.environ.get("ISOCON_DIST_BACKEND", "nccl")
if backend == "nccl":
    torch.cuda.set_device(local_rank)
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
else:
    dist.init_process_group(backend=backend)
assert NNG._process_group() is not None or dist.get_world_size() == 1
shared, t_shared = run("sharded")
same = [None] * dist.get_world_size()
dist.all_gather_object(same, (alone, shared))
if dist.get_rank() == 0:
    ok = all(a == same[0][0] and b == same[0][0] for a, b in same)
    print("spmd pipeline check: world %d, %d reads, outputs identical on every rank and to the single-process run: %s (%.1f s alone, %.1f s shared)"
          % (dist.get_world_size(), n, ok, t_alone, t_shared))
    assert ok
dist.barrier()
dist.destroy_process_group()
