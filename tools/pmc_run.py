"""rocprofv3 --pmc passes (two SQ counters per pass, each pass its own process) over a micro-driver, averaged per
kernel.    python tools/pmc_run.py <kernel-regex> <out.json> -- python3 tools/prof_cnn.py cnn 2
Run from the repo root on the GPU box (TMPDIR=/tmp).  No trace domains are combined with --pmc."""
import collections, csv, glob, json, os, re, shutil, subprocess, sys
regex, out = sys.argv[1], sys.argv[2]
cmd = [os.path.abspath(a) if os.path.exists(a) else a for a in sys.argv[sys.argv.index("--") + 1:]]   # rocprofv3 runs from /tmp
PASSES = ["SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_MFMA SQ_INSTS_VALU", "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE",
          "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY",
          "SQ_WAVE_CYCLES SQ_BUSY_CYCLES", "SQ_INSTS_LDS SQ_INSTS_SALU",
          # clock (GRBM_GUI_ACTIVE / kernel duration) and HBM bytes: FETCH_SIZE and WRITE_SIZE in their own passes
          # (MI355X_MICROARCH.md: TCC slots; FETCH_SIZE counts 64 B per 128-B request of wide streaming reads -> doubled below)
          "GRBM_GUI_ACTIVE", "FETCH_SIZE", "WRITE_SIZE"]
if os.environ.get("PMC_PASSES"):                      # e.g. PMC_PASSES="GRBM_GUI_ACTIVE;FETCH_SIZE;WRITE_SIZE"
    PASSES = [p.replace(",", " ") for p in os.environ["PMC_PASSES"].split(";")]
SKIP = int(os.environ.get("PMC_SKIP", "0"))
env = dict(os.environ, TMPDIR="/tmp")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for i, counters in enumerate(PASSES):
    d = "/tmp/pmc_run_%d" % i
    shutil.rmtree(d, ignore_errors=True)
    r = subprocess.run(["rocprofv3", "--pmc"] + counters.split() + ["--kernel-include-regex", regex, "--output-format", "csv",
                        "-d", d, "-o", "p", "--"] + cmd, env=env, cwd="/tmp", stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not files:
        print("pass %d produced no counters:\n%s" % (i, r.stdout[-1500:]))
        continue
    for f in files:
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r.get("Dispatch_Id") or 0))
        seen = collections.defaultdict(set)
        for row in rows:
            m = re.search(r"([A-Za-z_0-9]+_k)(<[^>]*>)?", row["Kernel_Name"])
            key = m.group(0) if m else row["Kernel_Name"]
            # PMC_SKIP: the first launches of every kernel are left out.  A process that has just started runs its first
            # tens of milliseconds of kernels at ~2.15 GHz instead of ~2.4 (conv backward: 3.21 ms against 2.89 ms, with or
            # without counters), so a two-launch driver measures durations and clocks of a GPU that has not ramped up yet.
            seen[key].add(row.get("Dispatch_Id"))
            if len(seen[key]) <= SKIP:
                continue
            acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
            if row["Counter_Name"] == "GRBM_GUI_ACTIVE" and row.get("Start_Timestamp") and row.get("End_Timestamp"):
                acc[key]["_duration_ns_grbm_pass"].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
res = {"how": "rocprofv3 --pmc <2 counters per pass> --kernel-include-regex %s -- %s ; per-launch averages (first %d launches of a kernel skipped), chip totals" % (regex, " ".join(cmd), SKIP),
       "kernels": {}}
for k, cs in acc.items():
    c = {n: sum(v) / len(v) for n, v in cs.items()}
    d = {"launches_averaged": max(len(v) for v in cs.values())}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CU_CYCLES" in c and c["SQ_BUSY_CU_CYCLES"]:
        d["mfma_pipe_busy_frac_of_simd_cycles"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * c["SQ_BUSY_CU_CYCLES"]), 4)
    if "SQ_INSTS_MFMA" in c and "SQ_INSTS_VALU" in c and c["SQ_INSTS_MFMA"]:
        d["valu_non_mfma_per_mfma"] = round((c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"]) / c["SQ_INSTS_MFMA"], 3)   # SQ_INSTS_VALU counts the MFMAs too
    if "SQ_LDS_IDX_ACTIVE" in c and "SQ_BUSY_CU_CYCLES" in c and c["SQ_BUSY_CU_CYCLES"]:
        d["lds_active_frac_of_cu_cycles"] = round(c["SQ_LDS_IDX_ACTIVE"] / c["SQ_BUSY_CU_CYCLES"], 4)
        d["lds_bank_conflict_frac_of_cu_cycles"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_BUSY_CU_CYCLES"], 4)
    if "GRBM_GUI_ACTIVE" in c and c.get("_duration_ns_grbm_pass", 0) > 5e4:      # not for launches of a few us: the counter window is wider than the kernel
        d["effective_clock_GHz"] = round(c["GRBM_GUI_ACTIVE"] / 8. / c["_duration_ns_grbm_pass"], 3)   # summed over the 8 XCDs
    if "FETCH_SIZE" in c:                             # KB units
        d["hbm_fetch_GB_doubled"] = round(2 * c["FETCH_SIZE"] * 1024 / 1e9, 3)
    if "WRITE_SIZE" in c:
        d["hbm_write_GB"] = round(c["WRITE_SIZE"] * 1024 / 1e9, 3)
    res["kernels"][k] = {"counters": {n: round(v, 1) for n, v in c.items()}, "derived": d}
os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: v["derived"] for k, v in res["kernels"].items()}, indent=1))
