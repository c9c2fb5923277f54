"""Compile the C part of the oracle (test infrastructure): oracle/radius_oracle.c -> oracle/libradius_oracle.so.
-ffp-contract=off: the restatement spells out where the reference's arithmetic fuses (fma) and where it rounds."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "radius_oracle.c")
LIB = os.path.join(HERE, "libradius_oracle.so")


def build(force: bool = False) -> str:
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", LIB, SRC, "-lm"], check=True)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
