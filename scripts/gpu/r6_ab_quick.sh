cd $GRAFT_REPO_ROOT
python scripts/dbg_dw1b.py 2>&1 | grep -v amdgpu.ids | head -2
for rep in 1 2 3; do
  for env in "" "GPDE_BWD_H1_IMAGE=1"; do
    echo "[$env] $(env $env GPDE_HIDDEN_CACHE=off timeout 300 python scripts/time_bwd.py g121 2>&1 | grep 'bwd M-edges' | tail -1)"
  done
done
