#!/usr/bin/env python3
"""Convert an ECOS-style problem header (the reference's data_*.hpp / test/*.h layout) to the EPB1 container that
examples/run_demo.cpp, tests/ and bench.py read.  usage: tools/ecos_header_to_epb.py problem.h out.epb [prefix]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eicos_amd.problem_io import read_ecos_header, write_epb  # noqa: E402

pat, sets = read_ecos_header(sys.argv[1], sys.argv[3] if len(sys.argv) > 3 else None)
write_epb(sys.argv[2], pat, sets)
print(f"{sys.argv[2]}: n={pat.n} m={pat.m} p={pat.p} l={pat.l} cones={pat.ncones} nnzG={pat.nnzG} nnzA={pat.nnzA}")
