"""Build variants of the edge kernel (compile-time knobs) on the GPU box and time them at the north-star shape.
   python tools/edge_tune.py "THREADS=512,HC=256,MINW=2" "THREADS=256,HC=256,MINW=4" ...
Each variant gets its own .so (EGNN_HIP_LIB) and is timed in a subprocess with events on the launch stream."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "egnn_pytorch_amd", "csrc")

TIMER = r'''
import sys, json, torch
torch.set_grad_enabled(False)
sys.path.insert(0, %r)
from egnn_pytorch_amd import EGNN, phase_timer
torch.manual_seed(0)
layer = EGNN(dim=512, num_nearest_neighbors=32).cuda().eval()
g = torch.Generator().manual_seed(1)
feats = torch.randn(64, 1024, 512, generator=g).cuda(); coors = torch.randn(64, 1024, 3, generator=g).cuda()
mask = torch.ones(64, 1024, dtype=torch.bool).cuda()
for _ in range(3): layer(feats, coors, mask=mask)
with phase_timer() as pt:
    for _ in range(10): layer(feats, coors, mask=mask)
s = pt.summary()
print(json.dumps({k: round(min(v), 4) for k, v in s.items()}))
''' % ROOT


def build(tag, defs, src="edge_fused", tuning=True):
    out = f"/tmp/egnn_{tag}"
    os.makedirs(out, exist_ok=True)
    names = ("knn_select", "spatial_order", "adj_expand", "linear_hl", "node_ops", "edge_fused", "edge_fused_c", "edge_bwd", "edge_tail", "layer_api", "segment_sum", "global_attn")
    objs = [os.path.join(CSRC, "obj", f + ".o") for f in names if f != src]
    o = f"{out}/{src}.o"
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-pass-failed", "-Wno-inline-asm",
           ] + (["-DEGNN_EDGE_TUNING_BUILD"] if tuning else []) + [f"-DEGNN_{k}={v}" for k, v in defs.items()] + \
          ["-c", os.path.join(CSRC, src + ".hip"), "-o", o]
    subprocess.run(cmd, check=True)
    lib = f"{out}/libegnn_hip.so"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, o] + objs, check=True)
    return lib


if __name__ == "__main__":
    src = "edge_fused"
    for spec in sys.argv[1:]:
        if spec.startswith("src="):
            src = spec[4:]
            continue
        defs = dict(kv.split("=") for kv in spec.split(",") if kv)
        tag = spec.replace("=", "").replace(",", "_") or "default"
        try:
            lib = build(tag, defs, src)
            env = dict(os.environ, EGNN_HIP_LIB=lib)
            r = subprocess.run([sys.executable, "-c", TIMER], env=env, capture_output=True, text=True, timeout=300)
            line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:]
            print(f"{spec:44s} {line}", flush=True)
        except Exception as exc:  # noqa: BLE001
            print(f"{spec:44s} FAILED {exc}", flush=True)
