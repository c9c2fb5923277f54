"""Import stub: the reference imports h5py at module top (utilities.py:4) but only uses it when
scipy.io.loadmat fails on a v7.3 .mat file (utilities.py:31-37)."""


class File:
    def __init__(self, *a, **k):
        raise ImportError("h5py is not available in this image; save .mat files in <= v7.2 format")
