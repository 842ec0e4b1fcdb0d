"""profiles/r06_bench_inputs.json: the PMC-derived figures bench.py prints next to its live HIP-event timings (counters
cannot be read from inside the benchmark process).  Per hand-written kernel of the headline step at the cfg4 per-GPU size
(78 400 masked images / Monotonic elements): MFMA instructions issued per image (element), other VALU instructions per MFMA,
HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, separate --pmc passes; MI355X_MICROARCH.md: FETCH_SIZE counts 64 B per
128-B request of wide streaming reads), MFMA-pipe busy fraction, LDS bank-conflict share, cycles per launch
(GRBM_GUI_ACTIVE / 8 XCDs) and the clock of the counter pass (cycles / kernel duration in that pass).      python tools/make_bench_inputs.py [out.json]      (GPU box, from the repo root)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r06_bench_inputs.json")
PASSES = ("SQ_INSTS_MFMA,SQ_INSTS_VALU;FETCH_SIZE;WRITE_SIZE;SQ_BUSY_CU_CYCLES,SQ_VALU_MFMA_BUSY_CYCLES;"
          "SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE;GRBM_GUI_ACTIVE")
env = dict(os.environ, TMPDIR="/tmp", PMC_PASSES=PASSES, PMC_SKIP="20")      # 40 launches per pass, the first 20 (clock ramp) left out
n = 78400
res = {"how": "", "n_images": n, "kernels": {}}
hows = []
# MFMA flop per instruction: v_mfma_f32_16x16x4_f32 and v_mfma_f32_16x16x1_4b_f32 both 2048
for which, regex, entries in (("cnn", "cnn_", (("gnf_mnistcnn_conv_bwd", "cnn_bwd_wino_k"), ("gnf_mnistcnn_conv_fwd", "cnn_fwd_wino_k"))),
                              ("mono", "mono_", (("gnf_monotonic_bwd", "mono_bwd_pair_x_k"), ("gnf_monotonic_fwd", "mono_fwd_x_k")))):
    tmp = "/tmp/bench_inputs_pmc_%s.json" % which
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_run.py"), regex, tmp, "--", "python3",
                    os.path.join(ROOT, "tools", "prof_cnn.py"), which, "40"], env=env, check=True, cwd=ROOT)
    pmc = json.load(open(tmp))
    hows.append(pmc["how"])
    for entry, kname in entries:
        cand = [(name, v) for name, v in pmc["kernels"].items() if name.startswith(kname)]
        if not cand:
            continue
        full_name, k = cand[0]
        c = k["counters"]
        # flop per MFMA instruction: 2048 for v_mfma_f32_16x16x4_f32 / 16x16x1_4b_f32; 16384 for the v_mfma_f32_16x16x32_bf16 of the
        # split-bf16 Monotonic forward (mono_fwd_x_k<3, EX, 1, false, true>: last template argument)
        split = kname == "mono_fwd_x_k" and full_name.replace(" ", "").endswith("true>")
        # the split-bf16 chain of the peeled backward (mono_bwd_pair_x_k<3, NH, EX, RP, true>): recompute + data gradient of the
        # 48 x 48 main blocks = 2 x 72 bf16 MFMAs per (16 elements, node pair, hidden->hidden layer) -- at S = 20 (11 node pairs) and
        # two such layers 11 x 2 x 144 / 16 = 198 per element; the rest of the counted instructions (weight gradients, W1h) are fp32
        mixed = kname == "mono_bwd_pair_x_k" and full_name.replace(" ", "").endswith("true,true>")
        res["kernels"][entry] = {
            "kernel": kname, "kernel_full_name": full_name, "mfma_flop": "mixed" if mixed else (16384 if split else 2048),
            "mfma_bf16_per_image": 198. if mixed else None,
            "mfma_per_image": c["SQ_INSTS_MFMA"] / n,
            "valu_per_mfma": (c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"]) / c["SQ_INSTS_MFMA"],     # SQ_INSTS_VALU counts the MFMAs too
            "hbm_bytes_per_launch": 2 * c["FETCH_SIZE"] * 1024 + c["WRITE_SIZE"] * 1024,
            "hbm_fetch_KB_raw": c["FETCH_SIZE"], "hbm_write_KB_raw": c["WRITE_SIZE"],
            "mfma_pipe_busy_frac_of_simd_cycles": k["derived"].get("mfma_pipe_busy_frac_of_simd_cycles"),
            "lds_bank_conflict_frac_of_lds_cycles": (c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]) if c.get("SQ_LDS_IDX_ACTIVE") else None,
            # The clock IN THE COUNTER PASS (pmc_run.py: GRBM_GUI_ACTIVE / 8 XCDs / duration of the launch in that pass) and the
            # cycle count of a launch.  Under rocprofv3 --pmc the HBM-streaming kernels run ~10 % longer at a ~10 % lower clock
            # than in an ordinary run (conv backward 3.20 ms at 2.15 GHz against 2.89 ms; rocm-smi shows 2.39 GHz during
            # bench.py): the cycle count of a launch is the same, so bench.py derives the clock of ITS run from it.
            "effective_clock_GHz_in_pmc_pass": k["derived"].get("effective_clock_GHz"),
            "duration_ns_in_pmc_pass": c.get("_duration_ns_grbm_pass"),
            "cycles_per_launch": (c["GRBM_GUI_ACTIVE"] / 8.) if c.get("GRBM_GUI_ACTIVE") else None,
        }
res["how"] = "python tools/make_bench_inputs.py  (" + " | ".join(hows) + ")"
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res["kernels"], indent=1))
