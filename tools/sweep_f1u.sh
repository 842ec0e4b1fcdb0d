#!/bin/bash
# conv forward: the deal of the next image's conv1 units over the wavefronts (GNF_FWD_F1U, see gnf_mnistcnn_fwd.hip; at most 3 per
# wavefront), timed by tools/bench_cnn.py -- one build per candidate:  bash tools/sweep_f1u.sh [deal ...] > gpurun_out/f1u.txt
deals=("$@")
if [ ${#deals[@]} -eq 0 ]; then
  deals=("0,0,3,2,1,1,2,2" "0,0,2,2,1,1,3,2" "0,0,3,3,1,1,2,1" "0,0,3,2,0,1,3,2" "0,0,3,3,0,0,3,2" "1,0,2,2,1,1,2,2" "0,0,2,3,1,1,2,2" "0,0,3,2,1,1,2,2")
fi
for c in "${deals[@]}"; do
  python tools/bench_cnn.py "-DGNF_FWD_F1U={$c}" --label "F1U $c" 2>&1 | grep "conv fwd" | sed 's/conv bwd.*//'
done
