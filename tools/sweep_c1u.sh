#!/bin/bash
# conv backward: the deal of the next image's conv1 units over the wavefronts (GNF_BWD_C1U, see gnf_mnistcnn.hip), dense and
# compact-de variants timed by tools/bench_cnn.py -- one build per candidate:  bash tools/sweep_c1u.sh [deal ...] > gpurun_out/c1u.txt
deals=("$@")
if [ ${#deals[@]} -eq 0 ]; then
  deals=("0,0,0,3,2,2,1,3" "0,0,0,3,2,2,2,2" "0,0,0,2,2,2,2,3" "1,0,0,2,2,2,2,2" "0,0,1,3,2,2,1,2" "1,1,0,2,2,2,1,2" "1,1,1,2,2,2,1,1" "0,0,0,3,2,2,1,3")
fi
for c in "${deals[@]}"; do
  python tools/bench_cnn.py "-DGNF_BWD_C1U={$c}" --label "C1U $c" 2>&1 | grep "conv bwd"
done
