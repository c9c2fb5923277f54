for rep in 1 2; do for v in "" _K1 _K2 _K16 _K24; do
  r=$(GPDE_LIB=$GRAFT_REPO_ROOT/scripts/ubench/lib/libgpde$v.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline --precision f16split 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print(j['value'], j['roofline']['avg_launch_ms'])")
  echo "rep=$rep variant=$v M-edges/s,avg_fused_ms: $r"
done; done
