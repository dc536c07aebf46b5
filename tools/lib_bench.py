"""bench.py against another build of the library (SSDN_LIB=path): same-box A/B of a kernel change (measurement aid)."""
import os, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "selfsupervised-denoising_amd"), ROOT]
from ssdn.hip import lib as L
if os.environ.get("SSDN_LIB"):
    L.LIB_PATH = os.environ["SSDN_LIB"]
sys.argv = ["bench.py"] + (sys.argv[1:] or ["--steps", "300", "--warmup", "30", "--no-cpu-baseline"])
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
