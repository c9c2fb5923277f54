cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r6_g241_budget; mkdir -p $O
run() { echo "== $*"; env "$@" MODES=auto timeout 500 python scripts/time_deferred.py g241 6 3 2>&1 | grep -v amdgpu.ids | tail -1 | sed 's/losses.*stats/stats/' | cut -c1-420; }
run GPDE_X=0 | tee $O/a.txt
run GPDE_SAVE_Z_GB=0 | tee $O/b.txt
run GPDE_SAVE_Z_GB=0 GPDE_HIDDEN_CACHE_GB=222 | tee $O/c.txt
run GPDE_SAVE_Z_GB=0 GPDE_HIDDEN_CACHE_GB=232 | tee $O/d.txt
run GPDE_HIDDEN_CACHE_GB=215 | tee $O/e.txt
