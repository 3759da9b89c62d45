"""A/B of compile-time variants of one kernel source, split in two halves so that no GPU-box time goes into compiling:

   python tools/variants.py build [src=edge_fused] [full=1] "EDGE_GDMA=0,EDGE_LO_MFMA=0" "EDGE_GDMA=1" ...      (dev container)
   python tools/variants.py run [shapes=ns,c3] [reps=10] [probe=train]                                (MI355X, via gpurun)

`build` compiles <src>.hip once per variant (-DEGNN_<K>=<V>; the tuning build of the edge pass unless full=1), links it with
the other objects of csrc/obj into build_variants/<tag>/libegnn_hip.so (git-ignored; travels with the gpurun snapshot) and
records the list in build_variants/index.json.  `run` times every kernel of a layer forward with each library
(EGNN_HIP_LIB, HIP events on the launch stream, min over reps) and prints a digest of the outputs: variants that are meant
to be bit-identical (LDS-DMA gathers, the residual on the matrix cores) must show the same digest.  probe=train: one training
step of the north-star layer instead (tools/train_step_probe.py: forward / backward wall time and every kernel of the step)."""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "egnn_pytorch_amd", "csrc")
VDIR = os.path.join(ROOT, "build_variants")
PROD = ("knn_select", "spatial_order", "adj_expand", "linear_hl", "node_mlp_fused", "node_ops", "edge_fused", "edge_pw", "edge_exact", "edge_exact_bwd", "edge_fused_d", "edge_fused_cd", "linear_f32", "node_prep_f32", "fp64", "edge_fused_c", "edge_bwd", "edge_tail",
        "layer_api", "segment_sum", "entry_lists", "global_attn")

TIMER = r'''
import sys, json, hashlib, torch
torch.set_grad_enabled(False)
sys.path.insert(0, %(root)r)
from egnn_pytorch_amd import EGNN, phase_timer
shape = %(shape)r
reps = %(reps)d
torch.manual_seed(0)
if shape == "ns":
    layer, B, N, D = EGNN(dim=512, num_nearest_neighbors=32), 64, 1024, 512
elif shape == "c3":
    layer, B, N, D = EGNN(dim=128, num_nearest_neighbors=32, norm_feats=True), 64, 1024, 128
elif shape == "c5":
    layer, B, N, D = EGNN(dim=256, num_nearest_neighbors=32, norm_feats=True, norm_coors=True), 64, 1024, 256
elif shape == "c2":
    layer, B, N, D = EGNN(dim=512), 8, 256, 512
elif shape == "c4":
    layer, B, N, D = EGNN(dim=512, edge_dim=4, only_sparse_neighbors=True), 32, 2048, 512
else:
    raise SystemExit("unknown shape " + shape)
for m in layer.modules():
    if isinstance(m, torch.nn.Linear):
        torch.nn.init.xavier_normal_(m.weight)
layer = layer.cuda().eval()
g = torch.Generator().manual_seed(1)
feats = torch.randn(B, N, D, generator=g).cuda(); coors = torch.randn(B, N, 3, generator=g).cuda()
mask = torch.ones(B, N, dtype=torch.bool).cuda()
kw = dict(mask=mask)
if shape == "c4":
    i = torch.arange(N)
    kw["adj_mat"] = ((i[:, None] - i[None, :]).abs() <= 1).cuda()
    kw["edges"] = torch.randn(B, N, N, 4, generator=g).cuda()
for _ in range(3): out = layer(feats, coors, **kw)
with phase_timer() as pt:
    for _ in range(reps): out = layer(feats, coors, **kw)
s = pt.summary()
h = hashlib.sha256()
for o in out: h.update(o.float().cpu().numpy().tobytes())
res = {k: round(min(v), 4) for k, v in s.items()}
res["_avg_edge"] = round(sum(s["edge_fused"]) / len(s["edge_fused"]), 4)
res["_digest"] = h.hexdigest()[:12]
print(json.dumps(res))
'''


def build(src, spec, full):
    defs = dict(kv.split("=") for kv in spec.split(",") if kv)
    tag = (src + "_" + spec.replace("=", "").replace(",", "_")) if spec else src + "_default"
    out = os.path.join(VDIR, tag)
    os.makedirs(out, exist_ok=True)
    objs = [os.path.join(CSRC, "obj", f + ".o") for f in PROD if f != src]
    for o in objs:
        if not os.path.exists(o):
            raise SystemExit(f"{o} missing: run csrc/build.sh first")
    o = os.path.join(out, src + ".o")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-pass-failed", "-Wno-inline-asm",
           "-Rpass-analysis=kernel-resource-usage"]
    if src == "edge_fused" and not full:
        cmd.append("-DEGNN_EDGE_TUNING_BUILD")
    if src == "knn_select":
        cmd.append("-ffp-contract=off")                  # (as csrc/build.sh: the ranking must reproduce the reference's un-fused sums)
    cmd += [f"-DEGNN_{k}={v}" for k, v in defs.items()] + ["-c", os.path.join(CSRC, src + ".hip"), "-o", o]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        print(r.stderr[-3000:])
        raise SystemExit(f"variant {tag} does not compile")
    spills = [ln for ln in r.stderr.splitlines() if "ScratchSize [bytes/lane]: " in ln and "ScratchSize [bytes/lane]: 0 " not in ln]
    lib = os.path.join(out, "libegnn_hip.so")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, o] + objs, check=True)
    os.remove(o)
    print(f"built {tag}" + (f"   ({len(spills)} kernels spill)" if spills else ""), flush=True)
    return tag


def main():
    mode = sys.argv[1]
    args = sys.argv[2:]
    keys = ("src", "full", "shapes", "reps", "only", "probe")
    opts = dict(a.split("=", 1) for a in args if a.split("=")[0] in keys)
    specs = [a for a in args if a.split("=")[0] not in keys]
    if mode == "build":
        src = opts.get("src", "edge_fused")
        tags = [build(src, s if s != "default" else "", opts.get("full", "0") == "1") for s in specs]
        idx = os.path.join(VDIR, "index.json")
        old = json.load(open(idx)) if os.path.exists(idx) else []
        json.dump([t for t in old if t not in tags] + tags, open(idx, "w"))
    elif mode == "run":
        tags = json.load(open(os.path.join(VDIR, "index.json")))
        if "only" in opts:
            tags = [t for t in tags if any(o in t for o in opts["only"].split("+"))]
        shapes = opts.get("shapes", "ns").split(",")
        reps = int(opts.get("reps", "10"))
        if opts.get("probe") == "train":
            for tag in tags:
                env = dict(os.environ, EGNN_HIP_LIB=os.path.join(VDIR, tag, "libegnn_hip.so"), EGNN_RANGE_CHECK="off")
                r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "train_step_probe.py"), "2"], env=env, capture_output=True, text=True, timeout=600)
                lines = r.stdout.strip().splitlines()
                print(f"{tag}\n   {lines[-2] if len(lines) > 1 else r.stderr[-600:]}\n   {lines[-1] if lines else ''}", flush=True)
            return
        for shape in shapes:
            for tag in tags:
                env = dict(os.environ, EGNN_HIP_LIB=os.path.join(VDIR, tag, "libegnn_hip.so"), EGNN_RANGE_CHECK="off")
                code = TIMER % {"root": ROOT, "shape": shape, "reps": reps}
                try:
                    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
                    line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "FAILED " + r.stderr[-600:]
                except subprocess.TimeoutExpired:
                    line = "TIMEOUT"
                print(f"{shape:3s} {tag:60s} {line}", flush=True)
    else:
        raise SystemExit(__doc__)


if __name__ == "__main__":
    main()
