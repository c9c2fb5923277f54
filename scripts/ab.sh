for rep in 1 2; do for v in "" _A _B _AB; do
  r=$(GPDE_LIB=$GRAFT_REPO_ROOT/scripts/ubench/lib/libgpde$v.so python bench.py --config ${CFG:-g121} --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print(j['value'], j['ms_per_step'])")
  echo "rep=$rep variant=$v : $r"
done; done
