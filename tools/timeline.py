"""Gaps between the kernels of one forward step, from a rocprofv3 --kernel-trace CSV:   python tools/timeline.py <kernel_trace.csv> [steps] [from=<index>]
Prints, for the last `steps` steps (a step starts at node_prep_hl_kernel) -- or for `steps` steps from step number `from=` on, counted from
the start of the trace: bench.py's timed region lies behind its priming + warm-up steps, its per-kernel timing (events between the
launches) at the end -- every kernel's start relative to the step, its duration and the idle gap before it on the union of all streams."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows))
starts = [i for i, e in enumerate(ev) if "node_prep_hl_kernel" in e[2]]
frm = [int(a.split("=")[1]) for a in sys.argv[3:] if a.startswith("from=")]
spans = list(zip(starts, starts[1:] + [len(ev)]))
print(f"{len(spans)} steps in the trace")
for s0, s1 in (spans[frm[0]:frm[0] + steps] if frm else spans[-steps - 1:-1]):
    t0 = ev[s0][0]
    busy_end = t0
    print(f"--- step: {len(ev[s0:s1])} kernels, {(ev[s1][0] - t0) / 1e3 if s1 < len(ev) else 0:.1f} us to the next step's first kernel")
    for a, b, name, q in ev[s0:s1]:
        gap = max(0, a - busy_end)
        print(f"  +{(a - t0) / 1e3:8.1f} us  dur {(b - a) / 1e3:8.1f}  idle-before {gap / 1e3:6.1f}  q={q}  {name}")
        busy_end = max(busy_end, b)
