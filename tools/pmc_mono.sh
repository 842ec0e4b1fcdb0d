R=$PWD; cd /tmp && export TMPDIR=/tmp
i=0
for pair in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $pair --kernel-include-regex "mono_bwd" --output-format csv -d /tmp/pm$i -o p -- python3 $R/tools/prof_cnn.py mono 2 >/dev/null 2>&1
done
python3 - <<PY
import csv,glob,json,collections
acc=collections.defaultdict(list)
for f in glob.glob("/tmp/pm*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(json.dumps({c:sum(v)/len(v) for c,v in acc.items()}))
PY
