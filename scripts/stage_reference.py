#!/usr/bin/env python3
"""Stage / unstage the reference's own script files for a GPU-box run of the UNMODIFIED scripts.

    python scripts/stage_reference.py [--all]    # copy the needed (all) /root/reference/*/*.py -> oracle/_ref/ (git-ignored)
    python scripts/stage_reference.py --remove   # delete the staged copies again

The GPU box has no /root/reference; the gpurun snapshot (and the driver's round-end snapshot) carries git-ignored
files, so the byte-identical copies travel with it exactly like the built libgpde.so does.  They are never committed:
the repo holds no reference source.  `__graft_entry__.build()` calls `stage()` on the build container, where
/root/reference exists, so that tests/test_gpu_reference_scripts.py runs on the GPU box instead of skipping."""
import os
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DST = os.path.join(REPO, "oracle", "_ref")
SRC = "/root/reference"
PROJECTS = ("graph-neural-operator", "multipole-graph-neural-operator")
# what the unmodified-script tests execute (round 6: every NNConv script of the reference; neurips4_GCN.py is another operator): the scripts themselves and the `utilities` module each imports
# (nn_conv / torch_geometric / h5py resolve to graph-pde_amd/shims and are NOT staged)
NEEDED = {"graph-neural-operator": ("UAI1_full_resolution.py", "UAI2_full_equation.py", "UAI3_resolution.py", "UAI4_equation_sample.py",
                                    "UAI5_sample_generalize.py", "UAI6_sample_radius.py", "UAI7_evaluate.py", "UAI7_evaluate2.py",
                                    "UAI8_kernel.py", "utilities.py"),
          "multipole-graph-neural-operator": ("MGKN_general_darcy2d.py", "MGKN_orthogonal_burgers1d.py", "neurips1_GKN.py", "neurips5_GKN.py",
                                              "neurips1_MGKN.py", "neurips2_MGKN.py", "neurips3_MGKN.py", "utilities.py")}


def stage(src_root: str = SRC, dst: str = DST, everything: bool = False) -> int:
    """Copy the reference's *.py files (both projects) under oracle/_ref/.  Returns the number of files staged;
    0 when the reference is not on this machine (the GPU box: nothing to do, the snapshot brought them)."""
    if not os.path.isdir(src_root):
        return 0
    n = 0
    for proj in PROJECTS:
        src = os.path.join(src_root, proj)
        if not os.path.isdir(src):
            continue
        os.makedirs(os.path.join(dst, proj), exist_ok=True)
        for f in sorted(os.listdir(src)):
            if f.endswith(".py") and (everything or f in NEEDED[proj]):
                shutil.copyfile(os.path.join(src, f), os.path.join(dst, proj, f))
                n += 1
    if n:
        with open(os.path.join(dst, "STAGED_REFERENCE_FILES.txt"), "w") as fh:
            fh.write("Byte-identical copies of files of the reference (neuraloperator/graph-pde), staged by scripts/stage_reference.py from\n"
                     "/root/reference so that tests/test_gpu_reference_scripts.py can EXECUTE the unmodified scripts on a GPU box that has no\n"
                     "/root/reference.  Test input only: git-ignored (never committed), never imported by graph-pde_amd/, bench.py or the\n"
                     "oracle.  Remove with `python scripts/stage_reference.py --remove`.\n")
    return n


if __name__ == "__main__":
    if "--remove" in sys.argv:
        shutil.rmtree(DST, ignore_errors=True)
        print("removed", DST)
        sys.exit(0)
    print(f"staged {stage(everything='--all' in sys.argv)} files under {DST}")
