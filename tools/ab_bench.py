import sys, runpy
import bwgr_amd.build as B
B.LIB = sys.argv[1]
sys.argv = ["bench.py"] + sys.argv[2:]
runpy.run_path("bench.py", run_name="__main__")
