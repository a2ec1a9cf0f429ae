#!/usr/bin/env python3
"""Developer tool: build a VARIANT of liblush_march.so next to the product (profiling / timing-ablation builds).

  python tools/build_variant.py --out build/wide_prof.so --flags=-DLUSH_PROF            (note the "=": the value starts with a dash)
  python tools/build_variant.py --out build/noconv.so "--flags=-DLUSH_ABL_NOCONV -DX=1" [--no-audit]

The product build (lib.build(), __graft_entry__.build()) takes no flags from anywhere; this is the only way to compile the
-DLUSH_ABL_* branches (wrong results by construction: timing only) and it never writes the product's path.  The ISA audit
(lush_nerf_amd/isa_check.py) runs on variants too; --no-audit is for ablations that remove the instructions a rule looks at.
Load a variant from a tool with lib.use_library(path) before lib.load() (tools/bench_mlp.py: SO=...).
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lush_nerf_amd import lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--out", required=True)
ap.add_argument("--flags", default="")
ap.add_argument("--no-audit", action="store_true")
a = ap.parse_args()
os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
path = lib.build(force=True, extra_flags=a.flags.split(), out=a.out, audit=not a.no_audit)
print(path)
for k, v in sorted(lib.LAST_BUILD_USAGE.items()):
    if v["sgpr_spills"] or v["vgpr_spills"] or v["scratch_bytes_per_lane"]:
        print(f"  {k}: {v}")
