#!/usr/bin/env python3
"""tools/r05_table_into_design.py -- analysis only: put the table tools/r05_table.py makes from profiles/r05_*_kernel_stats_pmc.txt between the markers of DESIGN.md section 5."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
t = subprocess.run([sys.executable, os.path.join(root, "tools", "r05_table.py"), os.path.join(root, "profiles")], capture_output=True, text=True, check=True).stdout.rstrip("\n")
p = os.path.join(root, "DESIGN.md"); s = open(p).read()
a = s.index("<!-- r05 table begin -->") + len("<!-- r05 table begin -->\n"); b = s.index("<!-- r05 table end -->")
open(p, "w").write(s[:a] + t + "\n" + s[b:])
print(t)
