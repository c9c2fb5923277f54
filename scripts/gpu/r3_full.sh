O=gpurun_out/r3_05
mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -v amdgpu.ids $O/pytest.log | tail -6
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; grep "\[bench\]" $O/bench.err | tail -12
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3_05/bench.json'))
print(d['value'], d['roofline']['frac'], d['node_table_attributes'])
print({k:(v['ms_per_forward'], v['best_ms_per_forward']) for k,v in d['mgkn'].items()})
print(d['backward'])
PY
