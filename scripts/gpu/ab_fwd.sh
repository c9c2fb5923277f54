# A/B of two builds of the library on ONE box (box-to-box spread is +-3 %): alternating runs of the headline forward.
# Arguments: paths relative to the repo root (developer builds live under scripts/ubench/lib/, graph-pde_amd/build.py).
B="python bench.py --no-cpu-baseline --no-alt --no-mgkn --no-reuse-probe --no-backward-probe --steps 6 --warmup 2"
for rep in 1 2 3; do
  for lib in ${1:-scripts/ubench/lib/libgpde_base.so} ${2:-graph-pde_amd/libgpde.so}; do
    GPDE_LIB=$PWD/$lib timeout 300 $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
  done
done
