#!/usr/bin/env python3
"""Summarise gpurun_out/parity_allowances.jsonl (written by tests/conftest.py::check_parity during `pytest -m gpu`): how many of
the checks that carry a noise allowance passed on the plain 1e-5 bound, and the factor on the reference's own fp32 noise the
others would have needed.      python tools/parity_allowances_summary.py > profiles/r04_parity_allowances.txt"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'parity_allowances.jsonl')
rows = [json.loads(l) for l in open(path) if l.strip()]
plain = [r for r in rows if r['plain']]
rest = sorted((r for r in rows if not r['plain']), key=lambda r: -min(r['factor_needed'], 1e9))
print("# tests/conftest.py::check_parity, one line per check of a `pytest tests -m gpu` run: elementwise |gpu - ref| / max(1, |ref|) against the fp64")
print("# evaluation of the reference op sequence on the identical fp32 (S, X); allowance = 1e-5 + factor x noise, noise = the largest of three")
print("# witnesses of the reference's own fp32 scatter on those inputs (numpy fp32, PyTorch-CPU fp32, fp64 on inputs moved by one fp32 rounding).")
print("%d checks; %d on the plain 1e-5 bound; %d needed an allowance (factor allowed: %s)" % (
    len(rows), len(plain), len(rest), sorted(set(r['factor_allowed'] for r in rows))))
print("worst plain-bound error: %.3g" % max([r['err'] for r in plain] or [0.0]))
print("factor_needed  err        noise      test / what")
for r in rest:
    print("%-13.2f %.3e  %.3e  %s | %s" % (r['factor_needed'], r['err'], r['noise'], r['test'].split('::')[-1].replace(' (call)', ''), r['what']))
