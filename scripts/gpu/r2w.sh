mkdir -p gpurun_out/r2w
T="tests/test_gpu_mgkn.py::test_mgkn_forward_calls_match_oracle"
i=0
for pre in "" "tests/test_gpu_hidden.py" "tests/test_gpu_bwd.py" "tests/test_gpu_headline.py" "tests/test_gpu_boundary.py tests/test_gpu_graph.py"; do
  i=$((i+1))
  timeout 600 python -m pytest $pre $T -q -m gpu > gpurun_out/r2w/p$i.log 2>&1 < /dev/null
  echo "[$pre] rc=$? $(grep -v amdgpu.ids gpurun_out/r2w/p$i.log | grep -E 'passed|failed' | tail -1) $(grep -o "AssertionError: (.*" gpurun_out/r2w/p$i.log | head -2 | tr '\n' ' ')"
done
