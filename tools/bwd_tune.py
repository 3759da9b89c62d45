"""A/B of compile-time variants of the edge kernel on one training step (forward + backward) at the north-star shape:
    python tools/bwd_tune.py "EDGE_BWD_NT=0" "EDGE_BWD_NT=1" ...
Builds each variant into /tmp like tools/edge_tune.py and runs tools/train_step_probe.py against it (EGNN_HIP_LIB)."""
import os
import subprocess
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import edge_tune

if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = "edge_fused"
    for spec in sys.argv[1:]:
        if spec.startswith("src="):
            src = spec[4:]
            continue
        defs = dict(kv.split("=") for kv in spec.split(",") if kv)
        tag = "bwd_" + spec.replace("=", "").replace(",", "_")
        lib = edge_tune.build(tag, defs, src, tuning=src == "edge_fused")
        env = dict(os.environ, EGNN_HIP_LIB=lib)
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "train_step_probe.py"), "2"], env=env, capture_output=True, text=True, timeout=600)
        lines = r.stdout.strip().splitlines()
        print(f"{spec:32s} {lines[-2] if len(lines) > 1 else r.stderr[-300:]}\n{'':32s} {lines[-1] if lines else ''}", flush=True)
