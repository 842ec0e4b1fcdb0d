"""Build libgnf_hip.so (the C-ABI library of hand-written gfx950 kernels) in-tree with hipcc.

    python -m gnf_hip.build            # from graphical-normalizing-flows_amd/
    python __graft_entry__.py build    # from the repo root

hipcc cross-compiles for gfx950 without a GPU.  The .so stays next to this file (it is
git-ignored but travels to the GPU box with the tree)."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(os.path.dirname(HERE))
INCLUDE = os.path.join(ROOT, "include")
LIB = os.path.join(HERE, "libgnf_hip.so")
OBJ = os.path.join(HERE, "_obj")
ARCH = "gfx950"
SOURCES = ["gnf_rowwise.hip", "gnf_dag_gate.hip", "gnf_gemm.hip", "gnf_gemm_split.hip", "gnf_linear.hip", "gnf_linear_tall.hip", "gnf_monotonic.hip", "gnf_monotonic_wide.hip", "gnf_mnistcnn_fwd.hip",
           "gnf_mnistcnn.hip", "gnf_mnistcnn_sparse.hip", "gnf_probe.hip"]
# per-file extra flags, each with the measurement that justifies it (tools/bench_cnn.py, cfg4 size)
EXTRA_FLAGS = {"gnf_mnistcnn_fwd.hip": ["-fno-slp-vectorize",     # conv forward 1.44 -> 1.40 ms (see the file header)
                                        # round 4: without the post-RA machine scheduler 1.390 / 1.392 / 1.420 -> 1.372 / 1.382 /
                                        # 1.388 ms (alternating on one box; the backward LOSES 2 % with it: 2.86 -> 2.92)
                                        "-mllvm", "-enable-post-misched=0"],
               # round 4: the restructured backward is faster WITHOUT the SLP vectoriser too (2.886 / 2.888 / 2.891 ->
               # 2.858 / 2.859 / 2.859 ms, three alternating runs on one box; the round-3 kernel was faster with it)
               "gnf_mnistcnn.hip": ["-fno-slp-vectorize"]}
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-I" + INCLUDE, "-I" + CSRC,
         "-Wno-unused-value"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def _newer(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def build_library(force=False, verbose=True):
    """Compile every .hip source for gfx950 and link libgnf_hip.so.  Returns its path."""
    hipcc = _hipcc()
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(INCLUDE, "gnf_hip.h"))
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        objs.append(o)
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-c", s, "-o", o]
        stamp = o + ".cmd"                      # an object built with other flags is stale too
        same_cmd = os.path.exists(stamp) and open(stamp).read() == " ".join(cmd)
        if force or not same_cmd or not _newer(o, [s] + headers):
            for stale in (stamp, o):                # a failed compile must not leave an object that looks up to date
                if os.path.exists(stale):
                    os.remove(stale)
            if verbose:
                print("[gnf_hip.build]", " ".join(cmd), flush=True)
            jobs.append((src, stamp, " ".join(cmd),
                         subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = []
    for src, stamp, line, p in jobs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed.append("hipcc failed on %s:\n%s" % (src, out))
        else:
            with open(stamp, "w") as f:             # the flags an object was built with, written once it exists
                f.write(line)
    if failed:
        raise RuntimeError("\n".join(failed))
    if force or jobs or not _newer(LIB, objs):
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print("[gnf_hip.build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv))
