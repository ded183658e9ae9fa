#!/usr/bin/env python3
"""bench.py on an experiment build of the library: QK_LIB=<path to .so> python lab/tools/bench_lib.py [bench.py arguments]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import qml_cutensornet_amd  # noqa: F401
from qml_cutensornet_amd import engine

if "QK_LIB" in os.environ or "QK_VARIANT" in os.environ:  # before anything calls engine.lib()
    engine.use_lab_library()
import bench

bench.main()
