"""profiles/r03_bench_inputs.json: the PMC-derived figures bench.py prints next to its live HIP-event timings (counters
cannot be read from inside the benchmark process).  Per kernel of the conv pair at the cfg4 per-GPU size (78 400 masked
images): MFMA instructions issued per image, other VALU instructions per MFMA, HBM bytes per launch (2 x FETCH_SIZE +
WRITE_SIZE, separate --pmc passes; MI355X_MICROARCH.md: FETCH_SIZE counts 64 B per 128-B request of wide streaming reads),
MFMA-pipe busy fraction.      python tools/make_bench_inputs.py [out.json]      (GPU box, from the repo root)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r03_bench_inputs.json")
tmp = "/tmp/bench_inputs_pmc.json"
env = dict(os.environ, TMPDIR="/tmp",
           PMC_PASSES="SQ_INSTS_MFMA,SQ_INSTS_VALU;FETCH_SIZE;WRITE_SIZE;SQ_BUSY_CU_CYCLES,SQ_VALU_MFMA_BUSY_CYCLES")
subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_run.py"), "cnn_", tmp, "--", "python3",
                os.path.join(ROOT, "tools", "prof_cnn.py"), "cnn", "2"], env=env, check=True, cwd=ROOT)
pmc = json.load(open(tmp))
n = 78400
res = {"how": "python tools/make_bench_inputs.py  (" + pmc["how"] + ")", "n_images": n, "kernels": {}}
for entry, kname in (("gnf_mnistcnn_conv_bwd", "cnn_bwd_wino_k"), ("gnf_mnistcnn_conv_fwd", "cnn_fwd_wino_k")):
    k = next(v for name, v in pmc["kernels"].items() if name.startswith(kname))
    c = k["counters"]
    res["kernels"][entry] = {
        "kernel": kname,
        "mfma_per_image": c["SQ_INSTS_MFMA"] / n,
        "valu_per_mfma": (c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"]) / c["SQ_INSTS_MFMA"],     # SQ_INSTS_VALU counts the MFMAs too
        "hbm_bytes_per_launch": 2 * c["FETCH_SIZE"] * 1024 + c["WRITE_SIZE"] * 1024,
        "hbm_fetch_KB_raw": c["FETCH_SIZE"], "hbm_write_KB_raw": c["WRITE_SIZE"],
        "mfma_pipe_busy_frac_of_simd_cycles": k["derived"].get("mfma_pipe_busy_frac_of_simd_cycles"),
    }
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res["kernels"], indent=1))
