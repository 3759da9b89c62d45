"""The tables of DESIGN.md section 5 from the bench lines of a measurement run:   python tools/design_tables.py profiles/r04_final"""
import json
import os
import sys

d = sys.argv[1] if len(sys.argv) > 1 else "profiles/r04_final"
ld = lambda n: json.load(open(os.path.join(d, n)))          # noqa: E731
b = ld("bench_line.json")
print(f"head {b.get('head')}  value {b['value']:.0f} graphs/s  {b['ms_per_step']:.3f} ms  deferred-mode value {b.get('value_range_check_deferred')}")
print(f"cpu_baseline {b['cpu_baseline']['value']}  eager {b.get('reference_gpu_eager', {}).get('value')}  train {b.get('train_step')}")
for k in b["kernels"]:
    print("  ", k.get("kernel"), f"{k['avg_ms']:.4f} ms", k.get("bound"), f"{k.get('achieved', 0):.1f} {k.get('unit')}", "frac", k.get("frac"))
print("traffic", b["roofline"].get("traffic"), b["roofline"].get("traffic_head"))
for w in ("bench_ragged_mask.json", "bench_c2_dense.json", "bench_c3_network.json", "bench_c4_sparse.json", "bench_c5_shard.json"):
    x = ld(w)
    ks = {k.get("kernel"): k for k in x["kernels"]}
    e = ks.get("edge_fused", {})
    print(w, f"{x['value']:.0f} graphs/s {x['ms_per_step']:.3f} ms | edge {e.get('avg_ms', 0):.3f} frac {e.get('frac')} | proj/mlp0/mlp1",
          " / ".join(f"{ks.get(n, {}).get('avg_ms', 0):.3f}" for n in ("node_proj", "node_mlp0", "node_mlp1")),
          "| knn", f"{ks.get('knn_select', {}).get('avg_ms', 0):.3f}")
