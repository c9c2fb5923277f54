cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r6_budget_check; mkdir -p $O
MODES=auto timeout 500 python scripts/time_deferred.py g241 6 5 2>&1 | grep -v amdgpu.ids | tail -1 | sed 's/losses.*stats/stats/' | cut -c1-420
timeout 1500 python -m pytest tests/test_gpu_headline_train.py tests/test_gpu_headline.py tests/test_gpu_deferred.py tests/test_gpu_keep_hidden.py tests/test_gpu_repeat.py tests/test_gpu_hidden.py -q -m gpu 2>&1 | grep -v amdgpu.ids | grep "passed\|failed\|^FAILED" | tail -4
( time timeout 900 python bench.py --detail-out $O/bench_detail.json > $O/bench.json 2> $O/bench.err < /dev/null ); echo "bench rc=$?"
grep "\[bench\]" $O/bench.err | tail -8 | cut -c1-200; wc -c $O/bench.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6_budget_check/bench.json"))
print({k: d["summary"][k] for k in ("g241_depth6_train_step", "depth6_g241_fwd_ms", "mgkn_train_ms", "bwd_g121")})
PY
