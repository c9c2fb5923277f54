O=gpurun_out/r3_02
mkdir -p $O
GPDE_LIB=$PWD/graph-pde_amd/libgpde_T6.so timeout 300 python scripts/v6_timing.py g241 2>&1 | grep -v amdgpu.ids | tail -3 > $O/v6_timing.log
GPDE_LIB=$PWD/graph-pde_amd/libgpde_T6.so timeout 300 python scripts/v6_timing.py g121 2>&1 | grep -v amdgpu.ids | tail -3 >> $O/v6_timing.log
cat $O/v6_timing.log
timeout 600 python bench.py --no-cpu-baseline --no-alt --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; grep "\[bench\]" $O/bench.err | tail -8
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3_02/bench.json'))
print(d['value'], d['roofline']['frac'], d['mgkn'] and {k:(v.get('ms_per_forward') if isinstance(v,dict) else v) for k,v in d['mgkn'].items()}, d['backward'] and d['backward']['ms'])
PY
