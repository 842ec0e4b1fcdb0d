#!/bin/bash
# Evidence for the split-bf16 Monotonic kernels of wide nets, one GPU call:  bash tools/collect_mono_split.sh r06
tag=${1:-r06}; out=gpurun_out; mkdir -p $out
export TMPDIR=/tmp
python tools/bench_mono_split.py 2>/dev/null | grep -v amdgpu > $out/${tag}_mono_split_ab.txt
python -m pytest tests/test_gpu_mono_split.py -q -s 2>&1 | grep -E "mono split|passed|failed" > $out/${tag}_mono_split_error.txt
hipcc --offload-arch=gfx950 -O3 -o /tmp/wsp tools/wide_split_probe.hip 2>/dev/null && /tmp/wsp > $out/${tag}_wide_split_probe.txt 2>&1
GNF_MONO_SHAPE=50000,63,30,20 python tools/pmc_run.py "mono_" $out/${tag}_mono_cfg5_pmc.json -- python3 tools/bench_mono.py 150 > /dev/null 2>&1
GNF_MONO_SHAPE=10000,6,30,20 PMC_SKIP=5 python tools/pmc_run.py "mono_" $out/${tag}_mono_cfg2_pmc.json -- python3 tools/bench_mono.py 100 > /dev/null 2>&1
( GNF_MONO_SHAPE=10000,6,30,20 python tools/bench_mono.py 100; GNF_MONO_SHAPE=50000,63,30,20 python tools/bench_mono.py 150 ) 2>/dev/null | grep "H=" > $out/${tag}_mono_wide.txt
( GNF_TRUE_F32=1 GNF_MONO_SHAPE=50000,63,30,20 python tools/bench_mono.py 150 ) 2>/dev/null | grep "H=" | sed 's/^/GNF_TRUE_F32=1  /' >> $out/${tag}_mono_wide.txt
bash tools/power_sample.sh $out/${tag}_mono_split_power.txt env GNF_MONO_SHAPE=50000,63,30,20 python tools/bench_mono.py 150 150 150
bash tools/power_sample.sh $out/${tag}_mono_f32_power.txt env GNF_TRUE_F32=1 GNF_MONO_SHAPE=50000,63,30,20 python tools/bench_mono.py 150 150 150
python tools/bench_configs.py --graph 2>/dev/null | grep '^{' > $out/${tag}_all_configs.jsonl
python tools/bench_configs.py cfg4det cfg4dag --graph 2>/dev/null | grep '^{' >> $out/${tag}_all_configs.jsonl
python tools/bench_kernels.py --json $out/${tag}_kernel_roofline_table.json > $out/${tag}_kernel_roofline_table.log 2>&1
ls -la $out/${tag}_*
