for v in ${VARIANTS:-"" _c1}; do
  ISOCON_LIB=isocon_amd/lib/libisocon_hip$v.so timeout -k 10 200 python bench.py --gpus 1 --steps 15 --warmup 4 --no-cpu-baseline > gpurun_out/bench_sw$v.json 2> gpurun_out/bench_sw$v.err
  python - "$v" <<'P'
import json,sys
v=sys.argv[1]
try:
    d=json.load(open("gpurun_out/bench_sw%s.json"%v))
    k=d["roofline"]["step_kernels_ms"]
    print("variant %-8s step %.3f ms  mm %.3f  lists %.3f  filter %.3f  lanes %.3f  seeds %.3f  rejected_by_filter %d aligned %d" % (v or "base", d["ms_per_step"], k["bound matrix (k_qgram_mm)"], k["survivor lists (k_nn_entry_meta, k_nn_survivors)"], k["block filter (k_nn_block_filter)"], k["pair per lane (k_ed_lanes)"], k["seeds (k_qgram_seed_pairs, k_ed_lanes)"], d["roofline"]["pairs_rejected_by_block_filter"], d["roofline"]["pairs_aligned"]))
except Exception as e:
    print("variant", v, "failed:", e, open("gpurun_out/bench_sw%s.err"%v).read()[-300:])
P
done
