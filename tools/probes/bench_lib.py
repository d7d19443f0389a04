"""bench.py on a variant build of the library (development A/B).  Usage (GPU box): python tools/probes/bench_lib.py <libuvl_X.so> [bench.py flags]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401
from uvltrack_amd import _native
_native.LIB_PATH = os.path.abspath(sys.argv[1])
del sys.argv[1]
import bench
bench.main()
