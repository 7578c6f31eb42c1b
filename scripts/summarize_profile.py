"""Turn gpurun_out/{prof,pmc_sq,pmc_fetch,pmc_write}_<TAG> (written by scripts/profile_bench.sh) into the tracked files
profiles/<TAG>_kernel_stats.csv, profiles/<TAG>_pmc_summary.txt, profiles/<TAG>_sq_summary.txt, profiles/<TAG>_bench.json and
profiles/counters.json (read by bench.py: per-dispatch SQ_INSTS_VALU and HBM bytes of the kernels its line reports)."""
import csv, glob, json, os, shutil, sys

if __name__ != "__main__" or len(sys.argv) < 2:          # a script that rewrites profiles/counters.json: never by import, never without its tag
    raise SystemExit("usage: python scripts/summarize_profile.py <TAG>")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
out = os.path.join(ROOT, "gpurun_out")
prof = os.path.join(ROOT, "profiles")


def newest(pattern):
    """the most recent match (a tag's directory may hold the files of several runs)"""
    return sorted(glob.glob(pattern), key=os.path.getmtime)[-1:]


ks = newest(os.path.join(out, "prof_" + tag, "*", "*_kernel_stats.csv"))
if ks:
    shutil.copy(ks[0], os.path.join(prof, tag + "_kernel_stats.csv"))
b = os.path.join(out, "bench_%s.json" % tag)
if os.path.exists(b):
    shutil.copy(b, os.path.join(prof, tag + "_bench.json"))


def dispatches(kind):
    """[(dispatch id, kernel name, grid, dur_ns, {counter: value})] in dispatch order, isocon kernels only"""
    f = newest(os.path.join(out, "pmc_%s_%s" % (kind, tag), "*", "*_counter_collection.csv"))
    if not f:
        return []
    d = {}
    for r in csv.DictReader(open(f[0])):
        if "isocon::k_" not in r["Kernel_Name"]:
            continue
        key = int(r["Dispatch_Id"])
        e = d.setdefault(key, [key, r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Grid_Size"]),
                               int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), {}])
        e[4][r["Counter_Name"]] = e[4].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return [tuple(d[k]) for k in sorted(d)]


def classify(rows):
    """dispatch classes bench.py reports: first main pass, first seed pass, the two 4096-pair SW batches, the two infix batches"""
    cls = {}
    # bench.py's two 4096-pair alignment batches: the full matrices (k_sg_forward), then the same pairs with band hints
    # (k_sg_band, plus a k_sg_forward launch right before it when some bands are wider than 256 diagonals)
    sg_full = [r for r in rows if "k_sg_forward" in r[1] and r[2] == 4096 * 64][:1]
    sg_banded = []
    for x, r in enumerate(rows):
        if "k_sg_band" in r[1] and r[2] == 4096 * 64:
            sg_banded = [r]
            for y in range(x + 1, len(rows)):          # the other band class of the same call, the launch for the pairs that run again
                if "k_sg_band" in rows[y][1] and rows[y][2] == 4096 * 64:
                    sg_banded.append(rows[y])
                elif "k_sg_recheck" not in rows[y][1] and "fillBuffer" not in rows[y][1]:
                    break
            if x and "k_sg_forward" in rows[x - 1][1] and rows[x - 1][2] == 4096 * 64 and rows[x - 1] not in sg_full:
                sg_banded.insert(0, rows[x - 1])
            break
    # isocon_hw_pairs = k_hw_locate launches (one per band class) followed by k_hw_finish launches: a call starts at the first
    # locate after a finish; bench.py's first call is its 64-pair warm-up
    calls, prev = [], "finish"
    for r in rows:
        if "k_hw_" not in r[1]:
            continue
        kind = "locate" if "locate" in r[1] else "finish"
        if kind == "locate" and prev == "finish":
            calls.append([])
        if calls:
            calls[-1].append(r)
        prev = kind
    # one NN-graph step: k_qgram_profile4, k_qgram_mm, k_qgram_seed_pairs, k_ed_lanes<true> (seeds), k_nn_entry_meta, k_nn_survivors,
    # k_nn_block_filter, [k_nn_scan_refill (tables) when enough chunks are left,] k_ed_lanes<true> (what the filter leaves + entries with few pairs)
    lanes = [r for r in rows if "k_ed_lanes<true>" in r[1]]
    for r in rows:
        # the two table launches of a step: the 64-row class (<.., false>) and the 32-row class (<.., true>)
        if "k_nn_scan_refill" in r[1] and "true>" not in r[1] and "nn_main" not in cls:
            cls["nn_main"] = [r]
        if "k_nn_scan_refill" in r[1] and "true>" in r[1] and "nn_main_narrow" not in cls:
            cls["nn_main_narrow"] = [r]
        if "k_qgram_mm" in r[1] and "nn_bound" not in cls:
            cls["nn_bound"] = [r]
        if "k_nn_survivors" in r[1] and "nn_lists" not in cls:
            cls["nn_lists"] = [r]
        if "k_nn_block_filter" in r[1] and "nn_filter" not in cls:
            cls["nn_filter"] = [r]
    # the 2-set leg of bench.py (reads x the g19 candidates: 51 030 entries): its bound kernel is the k_qgram_mm dispatch over the widest grid
    mms = [r for r in rows if "k_qgram_mm" in r[1]]
    if mms and max(r[2] for r in mms) > mms[0][2]:
        cls["nn_2set"] = [max(mms, key=lambda r: r[2])]
    if lanes:
        cls["nn_seed"] = lanes[:1]
    if len(lanes) > 1:
        cls["nn_lanes"] = lanes[1:2]
    if sg_full:
        cls["sg_full"] = sg_full
    if sg_banded:
        cls["sg_banded"] = sg_banded
    # the alignment dispatch of the wrappers leg: k_sg_band over ALL partition pairs of the workload (the widest k_sg_band launch of the run;
    # bench.py's own isolated call of it is the last one)
    bands = [r for r in rows if "k_sg_band" in r[1]]
    if bands and max(r[2] for r in bands) > 4096 * 64:
        # the call launches one k_sg_band per band class (<.., 4>: up to 256 diagonals, <.., 2>: up to 128) over the same grid: the last
        # dispatch of each class, plus the launch for the pairs that run again (narrow tries that did not certify themselves): the trailing
        # run of such launches, (a fill kernel for the counter between them)
        wide_grid = max(x[2] for x in bands)
        picked = []
        for r in reversed(rows):
            if "k_sg_band" in r[1] and r[2] == wide_grid:
                picked.append(r)
            elif picked and ("k_sg_recheck" in r[1] or "fillBuffer" in r[1]):
                continue
            elif picked:
                break
        cls["sg_partition"] = picked[::-1]
    if len(calls) >= 3:
        cls["hw_k25"], cls["hw_k63"] = calls[1], calls[2]
    if len(calls) >= 5:
        cls["hw_graph"] = calls[4]         # (calls[3] = the graph's own 4096-pair warm-up)
    return cls


sq, fetch, write = dispatches("sq"), dispatches("fetch"), dispatches("write")
sys.path.insert(0, ROOT)
import bench as _bench                      # source_digest(): the kernel sources these counters belong to
counters = {"round": tag, "source_digest": _bench.source_digest(), "command": "rocprofv3 --pmc <counters> --kernel-trace -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline "
                                     "(three passes: SQ_*, FETCH_SIZE, WRITE_SIZE)", "workload": "C3 (50k x 2.5kb, seed 30001)"}
lines = [counters["command"], "dispatches of one NN-graph step at C3 (50k x 2.5kb) and of bench.py's untimed extras"]
for kind, rows in (("sq", sq), ("fetch", fetch), ("write", write)):
    for did, name, grid, dur, c in rows:
        lines.append("%s grid=%d dur_ns=%d %s" % (name, grid, dur, " ".join("%s=%.6g" % kv for kv in sorted(c.items()))))
    for key, rr in classify(rows).items():
        e = counters.setdefault(key, {"kernel": rr[0][1], "grid": rr[0][2], "launches": len(rr)})
        for did, name, grid, dur, c in rr:
            for cn, v in c.items():
                e[cn] = e.get(cn, 0.0) + v
        e["dur_ns_%s_pass" % kind] = sum(r[3] for r in rr)
for key, e in counters.items():
    if isinstance(e, dict) and "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        # KiB -> bytes; FETCH_SIZE doubled per the gfx950 correction (MI355X_MICROARCH.md, HBM section)
        e["hbm_bytes"] = (2.0 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024.0
bj = os.path.join(out, "bench_sq_%s.json" % tag)
if os.path.exists(bj) and "nn_main" in counters:
    try:
        line = [ln for ln in open(bj) if ln.startswith("{")][-1]
        rl = json.loads(line)["roofline"].get("table_pass") or {}          # (only when table launches ran: the block filter usually leaves them nothing)
        counters["nn_main"]["wave_columns"] = rl["wave_columns_this_run"]
        if "nn_main_narrow" in counters:
            counters["nn_main_narrow"]["wave_columns"] = rl["narrow_wave_columns_this_run"]
    except Exception as ex:
        print("no wave_columns:", ex)
lines.append("HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (FETCH_SIZE doubled: gfx950 tallies 128-B read requests at 64 B, MI355X_MICROARCH.md)")
open(os.path.join(prof, tag + "_pmc_summary.txt"), "w").write("\n".join(lines) + "\n")

PEAK = 256 * 4 * 2.4e9 / 2
sql = ["SQ counters per dispatch class (%s); VALU peak = 256 CU x 4 SIMD x 2.4 GHz / 2 cycles = %.4g wave-instr/s" % (tag, PEAK)]
for key, e in counters.items():
    if isinstance(e, dict) and "SQ_INSTS_VALU" in e:
        dur = e.get("dur_ns_sq_pass", 0) or 1
        sql.append("%-10s %s grid=%d launches=%d dur_ms=%.3f" % (key, e["kernel"], e["grid"], e["launches"], dur / 1e6))
        sql.append("           " + " ".join("%s=%.6g" % (k, v) for k, v in sorted(e.items()) if k.startswith("SQ_")))
        sql.append("           VALU wave-instr/s = %.4g = %.3f of peak (profiled pass; bench.py divides by the un-profiled live time)"
                   % (e["SQ_INSTS_VALU"] / (dur / 1e9), e["SQ_INSTS_VALU"] / (dur / 1e9) / PEAK))
        if e.get("hbm_bytes"):
            sql.append("           HBM bytes = %.4g -> %.1f GB/s = %.4f of 8 TB/s" % (e["hbm_bytes"], e["hbm_bytes"] / dur, e["hbm_bytes"] / dur / 8000.0))
open(os.path.join(prof, tag + "_sq_summary.txt"), "w").write("\n".join(sql) + "\n")
json.dump(counters, open(os.path.join(prof, "counters.json"), "w"), indent=1)
print("\n".join(sql))
