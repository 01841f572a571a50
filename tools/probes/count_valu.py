#!/usr/bin/env python3
"""VALU / SALU instruction count of the largest loop of each kernel whose mangled name contains <filter>, read from the
gfx950 assembly hipcc leaves with --save-temps (developer tool of valu_probe).  usage: count_valu.py <file.s> <filter>"""
import collections
import re
import sys


def main():
    s = open(sys.argv[1]).read()
    flt = sys.argv[2]
    for name in re.findall(r"^(_Z\w+):", s, re.M):
        if flt not in name:
            continue
        body = re.split(r"^" + re.escape(name) + r":\s*(?:;.*)?$", s, flags=re.M)[1].split(".Lfunc_end")[0]
        lines = body.split("\n")
        labels = {}
        for n, l in enumerate(lines):
            m = re.match(r"(\.LBB\d+_\d+):", l.strip())
            if m:
                labels[m.group(1)] = n
        best = None
        for n, l in enumerate(lines):
            m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < n:
                lo = labels[m.group(1)]
                valu = sum(1 for x in lines[lo:n] if re.match(r"\s+v_", x))
                if best is None or valu > best[0]:
                    best = (valu, sum(1 for x in lines[lo:n] if re.match(r"\s+s_", x)), lo, n)
        if best is None:
            continue
        ops = collections.Counter(re.match(r"\s+(v_\w+)", x).group(1) for x in lines[best[2]:best[3]] if re.match(r"\s+v_", x))
        print(f"{name} loop_valu {best[0]} loop_salu {best[1]} top {ops.most_common(12)}")


if __name__ == "__main__":
    main()
