#!/bin/bash
# A/B build of the library with extra -D flags on ONE translation unit:  bash tools/build_variant.sh <name> <unit.hip> "<flags>"
# -> tools/<name>.bin (git-ignored, travels to the GPU box); use with GNF_AB_LIB=tools/<name>.bin (tools/bench_mono.py, ...)
set -e
root=$(cd "$(dirname "$0")/.." && pwd); pkg=$root/graphical-normalizing-flows_amd/gnf_hip
name=$1; unit=$2; flags=$3
python3 "$root/__graft_entry__.py" build > /dev/null
extra=""; case "$unit" in gnf_mnistcnn_fwd.hip) extra="-fno-slp-vectorize -mllvm -enable-post-misched=0";; gnf_mnistcnn.hip) extra="-fno-slp-vectorize";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$root/include -I$pkg/csrc -Wno-unused-value $extra $flags \
  -c $pkg/csrc/$unit -o /tmp/variant_$name.o
objs=$(ls $pkg/_obj/*.o | grep -v "/${unit%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/tools/$name.bin $objs /tmp/variant_$name.o
echo "tools/$name.bin"
