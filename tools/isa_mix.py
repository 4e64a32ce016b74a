#!/usr/bin/env python3
"""Instruction histogram of one kernel in a hipcc -S (device-only) listing.
usage: isa_mix.py file.s kernel_substring"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
pat = sys.argv[2]
lines = s.split('\n')
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\w*' + re.escape(pat) + r'\w*:', l))
end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
body = [l.strip() for l in lines[start + 1:end] if l.strip() and not l.strip().startswith(('.', ';'))]
c = Counter(l.split()[0] for l in body)
print(len(body), "instructions")
for k, v in c.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 25):
    print(f"{v:6d} {k}")
