#!/usr/bin/env python3
"""Copy a rocprofv3 csv keeping the header and the rows of this project's kernels (drprg::, rocprim::, runtime fills/copies).

usage: python tools/trim_csv.py <in.csv> <out.csv>
"""
import sys

with open(sys.argv[1]) as fi, open(sys.argv[2], "w") as fo:
    for i, line in enumerate(fi):
        if i == 0 or "drprg" in line or "rocprim" in line or "rocclr" in line:
            fo.write(line)
