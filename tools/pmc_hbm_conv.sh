# HBM traffic of the conv kernels (dense Winograd pair, sparse crop pair) at the cfg4 size: FETCH_SIZE and WRITE_SIZE in
# separate --pmc passes (MI355X_MICROARCH.md: they do not fit one pass), values in KiB per dispatch.
R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "cnn_|sparse_crop" --output-format csv -d /tmp/hb1 -o p -- python3 $R/tools/prof_cnn.py cnn 2 >/dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "cnn_|sparse_crop" --output-format csv -d /tmp/hb2 -o p -- python3 $R/tools/prof_cnn.py cnn 2 >/dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "cnn_|sparse_crop" --output-format csv -d /tmp/hb3 -o p -- python3 $R/tools/bench_sparse_front.py 100 >/dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "cnn_|sparse_crop" --output-format csv -d /tmp/hb4 -o p -- python3 $R/tools/bench_sparse_front.py 100 >/dev/null 2>&1
python3 - <<PY
import csv,glob,json,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("hb1","hb2","hb3","hb4"):
    for f in glob.glob("/tmp/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            n=r["Kernel_Name"]
            name="cnn_bwd_wino_k" if "cnn_bwd" in n else "cnn_fwd_wino_k" if "cnn_fwd" in n else "sparse_crop_bwd_k" if "crop_bwd" in n else "sparse_crop_k"
            if d in ("hb3","hb4") and name.startswith("cnn"): continue
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
print(json.dumps({k:{c:sum(v)/len(v) for c,v in d.items()} for k,d in acc.items()}))
PY
