"""Turn the rocprofv3 (rocpd sqlite) outputs under gpurun_out/prof into the small text
summaries that are committed under profiles/ (developer aid; run in the build container
after tools/profile_rocprof.sh has run on the GPU box)."""
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof")       # (c3; other configs: gpurun_out/prof_<cfg>, set below)
DST = os.environ.get("OBE_PROFILE_DST", os.path.join(ROOT, "profiles"))    # (on the GPU box: under gpurun_out/)
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
cfg = sys.argv[2] if len(sys.argv) > 2 else "c3"
n_p, n_s, d = {"c3": (1048576, 65536, 3), "c2": (262144, 4096, 3), "c5": (524288, 16384, 10)}[cfg]
if cfg != "c3":
    SRC = os.path.join(ROOT, "gpurun_out", f"prof_{cfg}")
n_read = d + 1 if cfg != "c5" else 10 + 1        # rows pass A of the update reads (c5: the noise row is one of the 10)


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


# 1. kernel-trace --stats summary
# (from the per-dispatch table, not rocprofv3's own top_kernels view: a sweep enqueued behind an update that
# then resampled returns at once — pdf_update()'s speculative sweep, DESIGN.md §3 — and such dispatches of a few
# microseconds must not be averaged with the ones that swept; they get a row of their own)
c = sqlite3.connect(os.path.join(SRC, "trace", "bench_results.db"))
per = {}
rows_all = list(c.execute("select name, start, end from kernels order by start"))
# (round 5) bench.py runs the sweep kernel back to back for ~150 ms before its warm-up steps (bench.warm_clocks: a chip
# that has idled needs ~35 ms of load to reach its sustained clocks): those launches — everything before the first
# update kernel — get a row of their own, so that the sweep kernel's average is that of the cycles
first_update = next((st for nm, st, _ in rows_all if "update_model_kernel" in nm), None)
for name, start, end in rows_all:
    if first_update is not None and start < first_update and "sweep_kernel" in name:
        name = name + " [clock warm-up]"
    per.setdefault(name, []).append((end - start) / 1e3)
split = {}
for name, durs in per.items():
    longest = max(durs)
    if ("sweep_" in name or "argmax_fold" in name) and "[clock warm-up]" not in name:
        ran = [v for v in durs if v >= 0.05 * longest]
        idle = [v for v in durs if v < 0.05 * longest]
        split[name] = ran
        if idle:
            split[name + " [returned at once: behind an update that resampled]"] = idle
    else:
        split[name] = durs
grand = sum(sum(v) for v in split.values())
rows = sorted(((n, len(v), sum(v), sum(v) / len(v), 100.0 * sum(v) / grand) for n, v in split.items()),
              key=lambda r: -r[2])
with open(os.path.join(DST, f"{tag}_kernel_stats_{cfg}.txt"), "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs ({cfg})\n")
    f.write("# durations in microseconds, from the trace's per-dispatch table; bench.py's own JSON line for this profiled run follows the table\n")
    f.write(f"{'kernel':70s} {'calls':>6s} {'total_us':>12s} {'avg_us':>12s} {'pct':>7s}\n")
    for name, calls, total, avg, pct in rows:
        label = short(name) + (name[name.index(" [returned"):] if " [returned" in name else "") + \
            (" [before the warm-up steps: bench.warm_clocks]" if "[clock warm-up]" in name else "")
        f.write(f"{label[:70]:70s} {calls:6d} {total:12.1f} {avg:12.2f} {pct:7.2f}\n")
        if len(label) > 70 and " [returned" in label:
            f.write(f"    ({label[label.index('[returned') + 1:].rstrip(']')})\n")
    log = os.path.join(SRC, "trace_stdout.log")
    if os.path.exists(log):
        for line in open(log):
            if line.startswith("{"):
                f.write("\n# bench.py output under the profiler:\n" + line)
print(open(os.path.join(DST, f"{tag}_kernel_stats_{cfg}.txt")).read()[:3000])


# 2. PMC passes: FETCH_SIZE / WRITE_SIZE are in KiB per dispatch
def per_kernel(db, counter):
    c = sqlite3.connect(db)
    vals = {}
    for name, value in c.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
        vals.setdefault(short(name), []).append(value)
    out = {}
    for name, v in vals.items():
        skipped = 0
        if "sweep_" in name:          # (dispatches that returned at once behind a resampling update moved nothing)
            ran = [x for x in v if x >= 0.05 * max(v)]
            skipped, v = len(v) - len(ran), ran
        out[name] = dict(dispatches=len(v), avg_kib=sum(v) / len(v), min_kib=min(v), max_kib=max(v))
        if skipped:
            out[name]["dispatches_that_returned_at_once"] = skipped
    return out


fetch = per_kernel(os.path.join(SRC, "pmc_fetch", "bench_results.db"), "FETCH_SIZE")
write = per_kernel(os.path.join(SRC, "pmc_write", "bench_results.db"), "WRITE_SIZE")
summary = {"config": cfg, "units": "KiB per dispatch as reported by rocprofv3 (uncorrected)", "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    if k.startswith("obe::"):
        summary["kernels"][k] = {"FETCH_SIZE": fetch.get(k), "WRITE_SIZE": write.get(k)}

# Calibration on a stream with a known byte count and the same access width (8 B per lane):
# pass A of the Bayes update reads (D+1) rows and writes one row of N_p doubles.
upd = [k for k in summary["kernels"] if "update_model_kernel" in k]
corr = None
if upd:
    u = summary["kernels"][upd[0]]
    known_read = 8 * n_read * n_p
    known_write = 8 * n_p
    corr = {"kernel": upd[0], "known_read_bytes": known_read, "known_write_bytes": known_write,
            # (min over dispatches: bench.py also runs this kernel on a 16 x tiled cloud)
            "fetch_reported_bytes": u["FETCH_SIZE"]["min_kib"] * 1024,
            "write_reported_bytes": u["WRITE_SIZE"]["min_kib"] * 1024}
    corr["fetch_factor"] = known_read / corr["fetch_reported_bytes"]
    corr["write_factor"] = known_write / corr["write_reported_bytes"]
summary["calibration"] = corr
sw = [k for k in summary["kernels"] if "sweep_kernel" in k]
if sw and corr:
    s = summary["kernels"][sw[0]]
    # The sweep kernel reads through the scalar path (s_load_dwordx16): FETCH_SIZE counts those
    # requests at their true size (tools/microbench_sload: 1.0000 of a known 1 GiB; the same bytes read
    # with 8-byte-per-lane vector loads report 0.5000, the factor the update kernel calibrates below).
    rd = s["FETCH_SIZE"]["avg_kib"] * 1024 * 1.0
    wr = s["WRITE_SIZE"]["avg_kib"] * 1024 * corr["write_factor"]
    summary["sweep_kernel_hbm_bytes_per_launch"] = rd + wr
    summary["sweep_kernel_read_bytes"], summary["sweep_kernel_written_bytes"] = rd, wr
    summary["sweep_kernel_note"] = ("FETCH (scalar loads, factor 1.0) + calibrated WRITE per launch; compulsory bytes are "
                                    f"{8 * (d + 1) * n_p + 16 * n_s} (cloud + settings + utility); the reads are the packed "
                                    "cloud once, the writes are the chunk partials (n_chunks x N_s x 16 B)")
path = os.path.join(DST, f"{tag}_pmc_hbm_{cfg}.json")
json.dump(summary, open(path, "w"), indent=1)
print(json.dumps(summary, indent=1)[:4000])
# bench.py reads this file for roofline.traffic
tpath = os.path.join(DST, "pmc_traffic.json")
allcfg = json.load(open(tpath)) if os.path.exists(tpath) else {}
allcfg[cfg] = {"sweep_kernel_hbm_bytes_per_launch": summary.get("sweep_kernel_hbm_bytes_per_launch"),
               "source": os.path.basename(path)}
json.dump(allcfg, open(tpath, "w"), indent=1)

# 3. SQ issue counters of the sweep kernel (tools/profile_sq.sh), if collected
sq_db = os.path.join(ROOT, "gpurun_out", "prof_sq", "sq") if cfg == "c3" else ""      # (collected for c3 only)
dbs = [os.path.join(dp, f) for dp, _, fs in os.walk(sq_db) for f in fs if f.endswith(".db")] if os.path.isdir(sq_db) else []
if dbs:
    c = sqlite3.connect(dbs[0])
    q = ("select kernel_name, counter_name, value from counters_collection where kernel_name like '%sweep_kernel%'")
    vals = {}
    for name, counter, value in c.execute(q):
        vals.setdefault((short(name), counter), []).append(value)
    out = {}
    for (name, counter), v in vals.items():
        ran = [x for x in v if x >= 0.05 * max(v)]       # (not the dispatches that returned at once, see above)
        out.setdefault(name, {"dispatches": len(ran)})[counter] = sum(ran) / len(ran)
    for k, v in out.items():
        if "GRBM_GUI_ACTIVE" in v and "SQ_ACTIVE_INST_VALU" in v:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_* counters are in quad-cycles summed over
            # all 1024 SIMDs (MI355X_MICROARCH.md: PMC units)
            cycles = v["GRBM_GUI_ACTIVE"] / 8.0
            v["derived"] = {"shader_cycles_per_launch": cycles,
                            "valu_busy_fraction": v["SQ_ACTIVE_INST_VALU"] / (cycles / 4.0 * 1024),
                            "mean_waves_per_simd": v["SQ_WAVE_CYCLES"] / (cycles / 4.0 * 1024),
                            "note": "effective clock = shader_cycles_per_launch / launch duration of the same run"}
    json.dump({"config": cfg, "source": "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU "
               "SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE (own pass)",
               "kernels": out}, open(os.path.join(DST, f"{tag}_sq_issue_{cfg}.json"), "w"), indent=1)
    print(json.dumps(out, indent=1)[:3000])
