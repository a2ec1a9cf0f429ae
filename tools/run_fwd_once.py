#!/usr/bin/env python3
"""Developer tool: a few launches of the fine-pass forward / backward-chain kernels for counter collection.  Not a test."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MODES", "2,1"); os.environ.setdefault("WHAT", "fwd,chain"); os.environ.setdefault("REPS", "2")
exec(open(os.path.join(ROOT, "tests", "bench_mlp.py")).read())
