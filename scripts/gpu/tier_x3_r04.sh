#!/bin/bash
# the full GPU tier three consecutive times on one box (round 4: 198 tests incl. the deferred / per-edge-weight repeat tests)
O=gpurun_out/tier_x3_r04
mkdir -p $O
for k in 1 2 3; do
  ( time timeout 1500 python -m pytest tests -q -m gpu ) > $O/run$k.log 2>&1 < /dev/null
  echo "run $k: $(grep -E 'passed|failed' $O/run$k.log | tail -1)  $(grep real $O/run$k.log | tail -1)"
done
