import ast
import sys
tag = None
for ln in sys.stdin:
    ln = ln.rstrip()
    if ln and not ln.startswith(" ") and not ln.startswith("{") and "_" in ln and " " not in ln:
        tag = ln
    elif ln.strip().startswith("{"):
        d = ast.literal_eval(ln.strip())
        print("%-45s by_src %s by_dest %s rows_gather_sum %s dest_lists %s" % (tag, d.get("edge_bwd_by_src"), d.get("edge_bwd_by_dest"), d.get("rows_gather_sum"), d.get("dest_lists")))
    elif ln.strip().startswith("step"):
        print("     " + ln.strip())
