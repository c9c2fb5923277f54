O=gpurun_out/r3_03
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_edgeweights.py tests/test_gpu_mgkn.py tests/test_gpu_boundary.py tests/test_gpu_models.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -v amdgpu.ids $O/pytest.log | tail -15
timeout 300 ./scripts/ubench/kloop_model_v6 > $O/kloop_model.txt 2>&1; tail -12 $O/kloop_model.txt
timeout 600 python - > $O/mgkn.json 2> $O/mgkn.err <<'PY'
import json, torch, bench
out = bench.mgkn_probe(torch.device("cuda:0"))
print(json.dumps(out))
PY
echo "mgkn rc=$?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/r3_03/mgkn.json'))
for k,v in d.items():
    print(k, {a:v[a] for a in ('ms_per_forward','ms_per_forward_fused_glue','ms_per_forward_grouped','max_rel_l2_vs_oracle','grouped_rel_l2_vs_fused_glue','edge_weight_cache')})
PY
tail -3 $O/mgkn.err
