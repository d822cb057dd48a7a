#!/usr/bin/env python3
"""Developer tool: run another tool script against a variant build of the library.
  python3 tools/run_with_lib.py <path/to/lib.so> tools/perf.py [args...]"""
import os
import runpy
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from pse_amd import _lib   # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
script = sys.argv[2]
sys.argv = [script] + sys.argv[3:]
runpy.run_path(script, run_name="__main__")
