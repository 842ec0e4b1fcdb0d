"""bench.py's evidence plumbing (CPU): PMC-derived figures are read from a committed profiles/ file, not written into
the source; the configuration builders tests and tools share live in the package (no absolute repo path anywhere)."""
import json
import os
import re

from conftest import ROOT


def test_bench_reads_pmc_figures_from_profiles():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "r03_bench_inputs.json" in src
    # the literals of round 2 (HBM bytes per launch, VALU per MFMA, MFMAs per image) are gone
    for lit in ("1427.2e6", "1149.3e6", "4.30", "3.77", "8 * 6 * 3 + 36 * 16"):
        assert lit not in src, lit
    f = os.path.join(ROOT, "profiles", "r03_bench_inputs.json")
    assert os.path.isfile(f), "profiles/r03_bench_inputs.json (tools/make_bench_inputs.py) must be committed"
    d = json.load(open(f))
    assert d["n_images"] == 78400
    for k in ("gnf_mnistcnn_conv_bwd", "gnf_mnistcnn_conv_fwd"):
        e = d["kernels"][k]
        assert e["mfma_per_image"] > 100 and e["valu_per_mfma"] > 0 and e["hbm_bytes_per_launch"] > 1e8


def test_no_absolute_repo_path_in_tools_tests_or_bench():
    pat = re.compile(r"/root/repo")
    bad = []
    for d in ("tools", "tests"):
        for name in sorted(os.listdir(os.path.join(ROOT, d))):
            if name.endswith((".py", ".sh")) and name != "test_bench_contract.py":
                if pat.search(open(os.path.join(ROOT, d, name)).read()):
                    bad.append(d + "/" + name)
    if pat.search(open(os.path.join(ROOT, "bench.py")).read()):
        bad.append("bench.py")
    assert not bad, bad


def test_baseline_config_builders_live_in_the_package():
    from gnf_hip import configs
    flow, x = configs.baseline_config("cfg1", device="cpu")
    assert tuple(x.shape) == (512, 2) and len(flow.steps) == 1
