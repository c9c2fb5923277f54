#!/usr/bin/env python3
"""Stage / unstage the reference's own script files for a GPU-box run of the UNMODIFIED scripts.

    python scripts/stage_reference.py            # copy /root/reference/*/*.py -> oracle/_ref/ (git-ignored)
    python scripts/stage_reference.py --remove   # delete the staged copies again

The GPU box has no /root/reference; the gpurun snapshot carries git-ignored files, so the byte-identical
copies travel with it.  They are never committed, and they are removed after the run: the repo holds no
reference source (the logs of the runs are kept under profiles/)."""
import os
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DST = os.path.join(REPO, "oracle", "_ref")
if "--remove" in sys.argv:
    shutil.rmtree(DST, ignore_errors=True)
    print("removed", DST)
    sys.exit(0)
n = 0
for proj in ("graph-neural-operator", "multipole-graph-neural-operator"):
    src = os.path.join("/root/reference", proj)
    os.makedirs(os.path.join(DST, proj), exist_ok=True)
    for f in sorted(os.listdir(src)):
        if f.endswith(".py"):
            shutil.copyfile(os.path.join(src, f), os.path.join(DST, proj, f))
            n += 1
print(f"staged {n} files under {DST}")
