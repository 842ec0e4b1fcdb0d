#!/bin/bash
# Round evidence in one GPU call, AFTER the last kernel commit: bash tools/collect_evidence.sh r06   (writes gpurun_out/<tag>_*;
# copy what is kept to profiles/)
tag=${1:-r06}; out=gpurun_out; mkdir -p $out
export TMPDIR=/tmp
python bench.py --steps 20 --warmup 3 2> $out/${tag}_bench.err | tail -1 > $out/${tag}_bench.json
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/${tag}_kt -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1 )
cp $(find /tmp/${tag}_kt -name "*kernel_stats.csv" | head -1) $out/${tag}_bench_kernel_stats.csv
# the conv pair: the PMC-derived inputs bench.py reads (incl. the Monotonic kernels), the full counter set, the per-phase cycle account
python tools/make_bench_inputs.py $out/${tag}_bench_inputs.json > /dev/null 2>&1
PMC_SKIP=20 python tools/pmc_run.py "cnn_" $out/${tag}_cnn_pmc.json -- python3 tools/prof_cnn.py cnn 40 > /dev/null 2>&1
python tools/time_cnn_phases.py > $out/${tag}_cnn_phases.txt 2>&1
python tools/bench_configs.py --graph 2>/dev/null | grep '^{' > $out/${tag}_all_configs.jsonl
python tools/bench_configs.py cfg4det cfg4dag --graph 2>/dev/null | grep '^{' >> $out/${tag}_all_configs.jsonl
python tools/bench_kernels.py --json $out/${tag}_kernel_roofline_table.json > /dev/null 2>&1
GNF_MONO_SHAPE=50000,63,30,20 python tools/pmc_run.py "mono_" $out/${tag}_mono_cfg5_pmc.json -- python3 tools/bench_mono.py 150 > /dev/null 2>&1
GNF_MONO_SHAPE=10000,6,30,20 PMC_SKIP=5 python tools/pmc_run.py "mono_" $out/${tag}_mono_cfg2_pmc.json -- python3 tools/bench_mono.py 100 > /dev/null 2>&1
PMC_SKIP=20 PMC_PASSES="GRBM_GUI_ACTIVE;SQ_BUSY_CU_CYCLES,SQ_VALU_MFMA_BUSY_CYCLES" python tools/pmc_run.py "gemm_" $out/${tag}_gemm_clock_pmc.json -- python3 tools/prof_gemm.py 40 > /dev/null 2>&1
( bash tools/kstats.sh ${tag}_lin_kt tools/bench_linear.py | grep "lin_" ) > $out/${tag}_linear_kernels.txt 2>&1
python tools/bench_linear.py >> $out/${tag}_linear_kernels.txt 2>&1
( GNF_MONO_SHAPE=10000,6,30,20 python tools/bench_mono.py 100; GNF_MONO_SHAPE=50000,63,30,20 python tools/bench_mono.py 150 ) 2>/dev/null | grep "H=" > $out/${tag}_mono_wide.txt
python tools/bench_sampling.py 2>/dev/null | grep -v amdgpu > $out/${tag}_sampling_run.txt
python tools/bench_fc1_split.py 2>/dev/null | grep ms > $out/${tag}_fc1_gemms.txt
ls -la $out/${tag}_*
