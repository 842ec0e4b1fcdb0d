R=$PWD; cd /tmp && export TMPDIR=/tmp
i=0
for pair in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $pair --kernel-include-regex sparse_crop --output-format csv -d /tmp/pmc$i -o p -- python3 $R/tools/bench_sparse_front.py 100 >/dev/null 2>&1
done
python3 - <<PY
import csv,glob,json,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name="sparse_crop_bwd_k" if "bwd" in r["Kernel_Name"] else ("sparse_crop_k<true>" if "true" in r["Kernel_Name"] or "Lb1" in r["Kernel_Name"] else "sparse_crop_k<false>")
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out={k:{c:sum(v)/len(v) for c,v in d.items()} for k,d in acc.items()}
print(json.dumps(out))
PY
