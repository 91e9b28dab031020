"""Development A/B: run bench.py's BA leg against another build of the library (vins-mono_amd/lib/<name>) in the same
process environment.  usage: python tests/manual/bench_with_lib.py libvinsgpu_A.so [bench args]"""
import os, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
pkg.LIB_PATH = os.path.join(os.path.dirname(pkg.LIB_PATH), sys.argv[1])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
