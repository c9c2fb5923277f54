GPDE_LIB=$PWD/graph-pde_amd/libgpde_T6.so timeout 300 python scripts/v6_timing.py g241 2>&1 | grep -v amdgpu.ids | tail -3
