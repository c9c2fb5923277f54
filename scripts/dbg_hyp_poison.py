"""Poison the caching allocator's free blocks (NaN / large values), then run the tier's hypothesis examples: an uninitialised read
in any kernel then shows up as NaN / a large error instead of depending on what earlier tests left in memory."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
d = torch.device("cuda:0")
val = float(sys.argv[1]) if len(sys.argv) > 1 else float("nan")
blocks = []
for sz in [1 << 30] * 24 + [64 << 20] * 64 + [1 << 20] * 512 + [64 << 10] * 1024 + [4096] * 4096 + [512] * 4096:
    blocks.append(torch.full((sz // 4,), val, device=d))
torch.cuda.synchronize()
del blocks
import pytest
sys.exit(pytest.main(["tests/test_gpu_hypothesis.py", "-q", "-m", "gpu", "-x", "-p", "no:cacheprovider"]))
