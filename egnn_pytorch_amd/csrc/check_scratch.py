"""csrc/build.sh: which kernels may touch scratch memory.

Register spills are allowed only OUTSIDE loops nested two or more deep -- i.e. setup values parked once and read back by the
epilogue -- never in a kernel's hot loops.  Input: device assembly files (hipcc -S --cuda-device-only).  Exit code 1 and a
list of offending (kernel, line) pairs otherwise."""
import re
import sys


def check(path):
    bad, kernel, depth_of, cur_depth = [], None, {}, 0
    for n, line in enumerate(open(path), 1):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kernel, cur_depth = m.group(1), 0
            continue
        m = re.search(r"Loop Header: Depth=(\d+)", line)
        if m:
            cur_depth = int(m.group(1))
            continue
        if re.match(r"^\.LBB\d+_\d+:", line) and "in Loop" not in line and "Loop Header" not in line:
            # a block label without a loop annotation on the same line: depth is given by the following comment lines, if any
            cur_depth = 0
        m = re.search(r"in Loop: Header=\S+ Depth=(\d+)", line)
        if m:
            cur_depth = int(m.group(1))
        if "scratch_" in line and cur_depth >= 2:
            bad.append((kernel, n, line.strip()))
    return bad


if __name__ == "__main__":
    bad = [b for p in sys.argv[1:] for b in check(p)]
    for k, n, l in bad:
        print(f"scratch access inside a nested loop: {k} line {n}: {l}")
    sys.exit(1 if bad else 0)
