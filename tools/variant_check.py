"""Build variants of a kernel source (like edge_tune.py) and run a subset of the GPU parity tests with each.
   python tools/variant_check.py [src=edge_fused] [k=<pytest -k expr>] "EDGE_RING=0" "EDGE_RING=1" ..."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from edge_tune import ROOT, build

if __name__ == "__main__":
    src, kexpr = "edge_fused", "golden or layer_vs_oracle"
    for spec in sys.argv[1:]:
        if spec.startswith("src="):
            src = spec[4:]
            continue
        if spec.startswith("k="):
            kexpr = spec[2:]
            continue
        defs = dict(kv.split("=") for kv in spec.split(",") if kv)
        tag = "vc_" + (spec.replace("=", "").replace(",", "_") or "default")
        lib = build(tag, defs, src, tuning=False)
        env = dict(os.environ, EGNN_HIP_LIB=lib)
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-m", "gpu", "-q",
                            "-k", kexpr, "-p", "no:cacheprovider", "--tb=no"], env=env, capture_output=True, text=True, cwd=ROOT)
        tail = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]
        print(f"{spec:50s} {tail}", flush=True)
