#!/usr/bin/env python3
"""A/B harness (GPU box, tools only): run a tool / bench script against ANOTHER build of the library, e.g. the previous
round's kernels kept as tools/ab/lib_<tag>.so, so that two kernel versions are timed on the SAME box in one call (boxes of
the pool differ by several per cent).  The product loader is untouched: this script redirects it from the outside.

    python tools/ab_run.py tools/ab/lib_v1.so tools/lstm_bench.py
    python tools/ab_run.py tools/ab/lib_v1.so bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline
"""
import ctypes
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib_path, script = os.path.abspath(sys.argv[1]), sys.argv[2]
import mmbidaf_amd._lib as L
import mmbidaf_amd.build as Bd

L.LIB_PATH = lib_path
_h = ctypes.CDLL(lib_path)
_h.mmb_build_hash.restype = ctypes.c_char_p
_h.mmb_version.restype = ctypes.c_int
Bd.source_hash = lambda: _h.mmb_build_hash().decode()      # the A/B build is older than the sources beside it, by design
L.ABI_VERSION = _h.mmb_version()
print(f"[ab_run] library {lib_path} (hash {_h.mmb_build_hash().decode()}, ABI {_h.mmb_version()})", file=sys.stderr)
sys.argv = [script] + sys.argv[3:]
runpy.run_path(script, run_name="__main__")
