"""One configuration of the plane-streaming FV kernel for counter passes -- development aid."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scripts.quick_bench_fv import run
from exahype_amd import solvers as exa
run(3, 15, 1, 5, 0, 8192, exa.PDE_EULER, exa.FV_RUSANOV, steps=3)
