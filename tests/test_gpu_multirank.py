"""GPU: BASELINE.json configs[3]'s code path -- one process per rank, pairs sharded by ownership, min-reductions and the
all-gather of attaining edges -- driven exactly as the driver drives it: `python bench.py --gpus 2` (which starts its two
ranks itself).  No second GPU exists on the test box, so the two ranks share the one GPU and talk over gloo
(ISOCON_DIST_BACKEND=gloo); the kernels, the sharding protocol and bench.py's launcher are the real ones.  The graph must be
the one of the single-process run (digest over bounds, row pointers and neighbour order)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(n_gpus, extra_env):
    env = dict(os.environ)
    env.update(extra_env)
    env.pop("RANK", None); env.pop("WORLD_SIZE", None); env.pop("LOCAL_RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n_gpus), "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-extras",
           "--reads", "6000", "--length", "1200", "--isoforms", "4", "--seed", "40001"]
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


@pytest.mark.timeout(900)
def test_bench_two_ranks_equal_one_rank():
    one = _bench(1, {})
    two = _bench(2, {"ISOCON_DIST_BACKEND": "gloo"})
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["rccl_ranks"] == 2 and len(two["per_rank_kernel_ms"]) == 2
    for key in ("alignments_per_step", "edges", "median_nn_distance", "graph_digest"):
        assert one["config"][key] == two["config"][key], key
    assert two["scaling"] == "strong" and two["value"] > 0
