"""Timing of the plane-streaming FV kernel (cfg 4's limiter patch, 3-D P = 15) -- development aid."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scripts.quick_bench_fv import run
from exahype_amd import solvers as exa
for _ in range(2):
    run(3, 15, 1, 5, 0, 8192, exa.PDE_EULER, exa.FV_RUSANOV)
    run(3, 15, 1, 5, 0, 8192, exa.PDE_EULER, exa.FV_FAITHFUL)
