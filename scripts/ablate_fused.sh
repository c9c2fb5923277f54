# timing-only ablations of the fused kernel (results are wrong by construction); see DESIGN.md
for v in "" _NOSTAGE _NOBARRIER _NOGEMM2 _NOCONV _ALL; do
  r=$(GPDE_LIB=$GRAFT_REPO_ROOT/scripts/ubench/lib/libgpde$v.so python bench.py --config ${CFG:-g121} --steps 5 --warmup 2 --no-cpu-baseline --precision ${PREC:-f16split} 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print(j['value'], j['ms_per_step'])")
  echo "variant=$v M-edges/s,ms_per_step: $r"
done
