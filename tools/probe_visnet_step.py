"""Probe for tools/ab.py: the BACE B=64 ViSNet training step (eager), blocks of 5 steps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PROBE_MODEL", "visnet"); os.environ.setdefault("PROBE_SHAPE", "bace"); os.environ.setdefault("PROBE_BATCH", "64")
import runpy
sys.argv = [sys.argv[0]] + sys.argv[1:3]
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe_step.py"), run_name="__main__")
