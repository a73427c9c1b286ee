#!/usr/bin/env python3
"""Run the GRU sequence kernels of the B=4096 step a few times (for rocprofv3 --pmc runs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import microbench
microbench.REPS = 3
microbench.bench_gru(256, 4096, 30, 2)
