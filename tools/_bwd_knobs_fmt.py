import ast
import sys
tag = None
for ln in sys.stdin:
    ln = ln.rstrip()
    if ln.startswith("edge_bwd"):
        tag = ln
    elif ln.strip().startswith("{"):
        d = ast.literal_eval(ln.strip())
        print("%-45s by_src %s by_dest %s rows_gather_sum %s" % (tag, d.get("edge_bwd_by_src"), d.get("edge_bwd_by_dest"), d.get("rows_gather_sum")))
    elif ln.strip().startswith("step"):
        print("     " + ln.strip())
