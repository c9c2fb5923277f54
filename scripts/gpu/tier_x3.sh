# the full GPU tier three times in a row (fresh process each): the backward repeat / skew tests must be green every time
O=gpurun_out/tier_x3
mkdir -p $O
for k in 1 2 3; do
  timeout 1500 python -m pytest tests -q -m gpu > $O/run$k.log 2>&1; echo "run $k rc=$?: $(grep -v amdgpu.ids $O/run$k.log | tail -1)"
done
