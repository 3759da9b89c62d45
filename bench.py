#!/usr/bin/env python
"""bench.py -- EGNN.forward throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1 without a launcher: spawns the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one EGNN.forward over one batch of synthetic graphs: the north-star workload
EGNN(dim=512, num_nearest_neighbors=32), B=64 graphs x N=1024 nodes per GPU, fp32, inference
(eval / no_grad), inputs resident in HBM before the timed region.  Multi-GPU = independent graphs
sharded over ranks (weak scaling: 64 graphs per GPU), no collective in the data path; the timed
region is bracketed by barrier + synchronize and the max over ranks is taken.

Rank 0 prints ONE JSON line: the driver contract plus
  "roofline"     -- the dominant kernel against its roofline, duration measured live with events on
                    the launch stream (the kernels are launched on torch's current stream);
  "kernels"      -- the same for every kernel of the step;
  "cpu_baseline" -- the reference module itself (oracle/_ref, torch CPU, all host cores) on a bounded
                    sample of the same workload on this box's host cores; the numpy port if the
                    artefact is absent (`kind` says which).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# chip peaks: /opt/skills/guides/MI355X_MICROARCH.md (chip-level parameters)
HBM_PEAK_GBS = 8000.0          # HBM3E spec
MFMA_F16_PEAK_TFLOPS = 2500.0  # v_mfma_f32_*_f16 dense (the GEMMs issue 3 f16 MFMAs per fp32 product: split-f16x3)
SPLIT_TERMS = 3

WORKLOADS = {
    # name: (layer kwargs, graphs per GPU, nodes)  -- the default is the configuration BASELINE.json's metric is quoted on
    "north_star": (dict(dim=512, num_nearest_neighbors=32), 64, 1024),
    # the other BASELINE.json configs (parity-test cases; timed on request with --workload, not part of the default line)
    "c2_dense": (dict(dim=512), 8, 256),
    "c3_network": (dict(depth=3, dim=128, num_nearest_neighbors=32), 64, 1024),
    "c4_sparse": (dict(dim=512, edge_dim=4, only_sparse_neighbors=True), 32, 2048),
    "c5_shard": (dict(depth=6, dim=256, num_nearest_neighbors=32, norm_coors=True), 64, 1024),   # 1/8 of B=512
}


def neighbours_of(kwargs, n):
    """K of the workload: num_nearest_neighbors, the maximum adjacency row sum for only_sparse_neighbors (3 for the README chain
    |i - j| <= 1 this file builds, diagonal included: egnn_pytorch.py:249), N on the dense all-pairs path."""
    if kwargs.get("only_sparse_neighbors"):
        return min(3, n)
    return kwargs.get("num_nearest_neighbors", 0) or n


def model_counts(kwargs, b, n):
    """Algorithmic bytes / executed flops per LAUNCH of every kernel of one layer-forward;
    a network step launches each of them `depth` times."""
    dim = kwargs["dim"]
    k = neighbours_of(kwargs, n)
    edge_dim = kwargs.get("edge_dim", 0)
    m = kwargs.get("m_dim", 16)
    din = 2 * dim + 1 + edge_dim
    h = 2 * din
    hp = (h + 31) // 32 * 32
    bn, e = b * n, b * n * k
    weights = 4 * (h * din + h + m * h + m + (dim + m) * 2 * dim + 2 * dim + 2 * dim * dim + dim + 4 * m * m + 9 * m + 1)
    return {
        # every edge reads its neighbour's feature row once, every node row read + written once
        "edge_fused": dict(bound="hbm", bytes=4 * dim * (e + 2 * bn) + 4 * e + 24 * bn + bn + 4 * edge_dim * e + weights,
                           flops=2.0 * e * hp * 16 + 2.0 * e * (16 * 64 + 64)),
        # GEMMs: algorithmic (fp32-equivalent) flops; the kernels issue SPLIT_TERMS x that on the f16 matrix cores
        "node_proj": dict(bound="mfma", flops=2.0 * bn * dim * 2 * hp, bytes=4 * (bn * dim + 2 * hp * dim + bn * 2 * hp)),
        "node_mlp0": dict(bound="mfma", flops=2.0 * bn * (dim + m) * 2 * dim,
                          bytes=4 * (bn * (dim + m) + 2 * dim * (dim + m) + bn * 2 * dim)),
        "node_mlp1": dict(bound="mfma", flops=2.0 * bn * 2 * dim * dim, bytes=4 * (bn * 2 * dim + 2 * dim * dim + 2 * bn * dim)),
        # narrow layers (dim <= 256): both Linears in one launch (csrc/node_mlp_fused.hip) -- the hidden activation never reaches memory:
        # reads [LayerNorm(h) | m_i] and the residual, writes the output rows
        "node_mlp": dict(bound="mfma", flops=2.0 * bn * ((dim + m) * 2 * dim + 2 * dim * dim),
                         bytes=4 * (bn * (dim + m) + 2 * bn * dim + 2 * dim * (dim + m) + 2 * dim * dim)),
        # one pass over feats: read fp32 rows once, write [node_norm(feats) | 0] as a (hi, lo) pair -- and the raw rows as a second pair
        # only when node_norm is a LayerNorm (the projection reads the one image otherwise: egnn_linear_hl_lda_f32)
        "node_prep": dict(bound="hbm", bytes=4 * bn * dim + (4 * bn * dim if kwargs.get("norm_feats") else 0) + 4 * bn * (dim + m),
                          flops=10.0 * bn * dim),
        "split_f16": dict(bound="hbm", bytes=4 * 2 * bn * dim, flops=2.0 * bn * dim),
        "spatial_order": dict(bound="hbm", bytes=16 * bn, flops=0.0),
        # per-slot records for the edge pass's setup: reads the neighbour list, rank and two coordinate rows, writes 16 bytes per slot
        "slot_prep": dict(bound="hbm", bytes=e * (4 + 4 + 16) + 24 * bn, flops=6.0 * e),
        # the K selected pairs' edge features out of the (B,N,N,edge_dim) tensor (network front-end / c4)
        "edge_features": dict(bound="hbm", bytes=e * (4 + 2 * 4 * max(edge_dim, 1)), flops=0.0),
        # fused pairwise distance + ranking + top-K: its compulsory traffic is tiny (B N 13 bytes in, B N K 8 out) and nothing of
        # size N^2 exists -- a VALU / scalar-unit kernel, reported as ordered pairs per second
        "knn_select": dict(bound="valu", pairs=float(b) * n * n, bytes=13 * bn + 8 * e, flops=8.0 * b * n * n),
    }, dict(E=e, K=k, H=h, Hp=hp)


def roofline_entry(name, counts, ms, launches=1):
    """One kernel against its roofline: `ms` = average duration of ONE launch (events on the launch stream), `launches` per step."""
    c = counts[name]
    sec = ms * 1e-3
    if c["bound"] == "valu":
        return dict(kernel=name, bound="valu", achieved=round(c["pairs"] / sec / 1e9, 2), peak=None, unit="Gpairs/s", frac=None,
                    avg_ms=round(ms, 4), launches_per_step=launches, ordered_pairs=int(c["pairs"]),
                    note="pairwise distance + ranking + top-K in registers: VALU / scalar-unit bound (VALUBusy: profiles/), compulsory "
                         f"HBM traffic {int(c['bytes'])} bytes")
    if c["bound"] == "hbm":
        ach = c["bytes"] / sec / 1e9
        return dict(kernel=name, bound="hbm", achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(ach / HBM_PEAK_GBS, 4), avg_ms=round(ms, 4), launches_per_step=launches, algorithmic_bytes=int(c["bytes"]))
    ach = SPLIT_TERMS * c["flops"] / sec / 1e12          # MFMA-issued flops (3 f16 MFMAs per fp32 product)
    # a GEMM with a short contraction (dim 128 / 256: K = 128 ... 544) is bound by reading A and writing C, not by the matrix cores:
    # both fractions are reported, `binding` says which roofline is the nearer one (c3's 0.14 - 0.22 of the MFMA peak
    # is 0.3 - 0.4 of the HBM roofline on the compulsory bytes)
    hbm = c["bytes"] / sec / 1e9
    return dict(kernel=name, bound="mfma", achieved=round(ach, 2), peak=MFMA_F16_PEAK_TFLOPS, unit="TFLOP/s",
                frac=round(ach / MFMA_F16_PEAK_TFLOPS, 4), avg_ms=round(ms, 4), launches_per_step=launches,
                mfma_issued_flops=SPLIT_TERMS * c["flops"],
                algorithmic_tflops=round(c["flops"] / sec / 1e12, 2), mfma_dtype="f16 (split x3, fp32 accumulate)",
                compulsory_bytes=int(c["bytes"]), hbm_GBs=round(hbm, 1), hbm_frac=round(hbm / HBM_PEAK_GBS, 4),
                binding="hbm" if hbm / HBM_PEAK_GBS > ach / MFMA_F16_PEAK_TFLOPS else "mfma")


def workload_label(kwargs, b, n, k):
    net = f"EGNN_Network(depth={kwargs['depth']}, " if "depth" in kwargs else "EGNN("
    opts = "".join(f", {o}" for o in ("norm_coors", "only_sparse_neighbors") if kwargs.get(o))
    if kwargs.get("edge_dim"):
        opts += f", edge_dim={kwargs['edge_dim']}"
    if kwargs.get("only_sparse_neighbors"):
        path = f"adjacency neighbours (K={k})"
    elif kwargs.get("num_nearest_neighbors"):
        path = f"masked k-NN (k={k})"
    else:
        path = "dense all-pairs"
    return f"{net}dim={kwargs['dim']}{opts}) {path} B={b}/GPU N={n} fp32"


# rocprofv3 leaves kernels with _Float16 parameters mangled: the dominant kernels by the name fragments that identify them
_TRAFFIC_KERNELS = {
    "edge_fused": (r"edge_pw_kernel", r"edge_kernelI", r"edge_kernel<"),
    "node_proj": (r"linear_hl_kernelILi\d+ELi0ELb0E", r"linear_hl_kernel<\d+, 0, false"),
    "node_mlp0": (r"linear_hl_kernelILi\d+ELi1ELb0E", r"linear_hl_kernel<\d+, 1, false"),
    "node_mlp1": (r"linear_hl_kernelILi\d+ELi0ELb1E", r"linear_hl_kernel<\d+, 0, true"),
    "node_mlp": (r"node_mlp_fused_kernel",),
    "knn_select": (r"knn_select_kernel",),
    "node_prep": (r"node_prep_hl_kernel",),
    "slot_prep": (r"slot_prep_kernel",),
    "spatial_order": (r"spatial_order_kernel",),
}


def _pmc_child(workload, counter, timeout_s):
    """One child run of this script under `rocprofv3 --kernel-trace --pmc <counter>` (PMC collection cannot run inside the timed process):
    the rows of its counter CSV as (kernel name, value, dispatch ns), or a string saying why there are none."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3")
    if rp is None:
        return "rocprofv3 not on PATH"
    # (a plain single-process child: nothing of a launcher's rendezvous may leak into it)
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT")
           and not k.startswith("TORCHELASTIC")}
    env.update(TMPDIR="/tmp", EGNN_BENCH_TRAFFIC_CHILD="1")
    rows = []
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        cmd = [rp, "--kernel-trace", "--pmc", counter, "-d", tmp, "-o", "pmc", "--output-format", "csv", "--", sys.executable,
               os.path.abspath(__file__), "--workload", workload, "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-train-step",
               "--no-live-traffic"]
        try:
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s, check=True)
        except Exception as ex:                      # (never fatal: the line falls back to the committed profile)
            return f"rocprofv3 --pmc {counter}: {type(ex).__name__}"
        for f in glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    if row["Counter_Name"] == counter:
                        rows.append((row["Kernel_Name"], float(row["Counter_Value"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
    return rows or f"no {counter} rows"


def live_traffic(workload, kernel, timeout_s=150):
    """HBM traffic of `kernel` per launch, measured NOW: two child runs of this script under `rocprofv3 --pmc` (FETCH_SIZE and WRITE_SIZE
    cannot share a pass), collected and corrected as MI355X_MICROARCH.md's HBM section prescribes -- both counters in KiB, FETCH_SIZE
    doubled on gfx950 (it reports half the bytes of 16-byte-per-lane coalesced reads).  Returns (bytes per launch, note) or (None, why not)."""
    import re
    pats = _TRAFFIC_KERNELS.get(kernel)
    if pats is None:
        return None, f"no kernel pattern for {kernel}"
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        rows = _pmc_child(workload, counter, timeout_s)
        if isinstance(rows, str):
            return None, rows
        got = [v for name, v, _ in rows if any(re.search(p, name) for p in pats)]
        if not got:
            return None, f"no {counter} rows for {kernel}"
        vals[counter] = sum(got) / len(got)
    return int((2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024), \
        "measured in this run: two child runs of this command under rocprofv3 --kernel-trace --pmc (FETCH_SIZE, WRITE_SIZE: separate " \
        "passes), (2 FETCH_SIZE + WRITE_SIZE) * 1024 per launch (MI355X_MICROARCH.md: KiB units, gfx950 FETCH_SIZE correction)"


def live_clocks(workload, timeout_s=150):
    """Effective shader clock of each kernel of the step, measured NOW (MI355X_MICROARCH.md, "DVFS give-back": the chip clocks to its
    power budget; effective clock = GRBM_GUI_ACTIVE / kernel wall time): one more child run under `rocprofv3 --pmc GRBM_GUI_ACTIVE`.
    rocprofv3 sums the counter over the 8 XCDs and its sampling window is wider than the dispatch: the smallest value of the pass (a
    few-microsecond copy kernel) is subtracted as that fixed part; wall time = the dispatch's own timestamps in that pass.
    Returns ({kernel: GHz}, note) or (None, why not).  The roofline peaks (2.5 PFLOP/s dense f16) assume 2.4 GHz."""
    import re
    rows = _pmc_child(workload, "GRBM_GUI_ACTIVE", timeout_s)
    if isinstance(rows, str):
        return None, rows
    floor = min(v for _, v, _ in rows)
    out = {}
    for kernel, pats in _TRAFFIC_KERNELS.items():
        got = [(v - floor) / 8.0 / ns for name, v, ns in rows if ns > 20000 and any(re.search(p, name) for p in pats)]
        if got:
            out[kernel] = round(sum(got) / len(got), 3)
    return (out or None), "GRBM_GUI_ACTIVE (summed over 8 XCDs; the pass's smallest value subtracted as the sampling window's fixed part) " \
        "/ 8 / dispatch wall time, one child run of this command under rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE"


def child_value(args, extra_flags, env_extra=None, timeout_s=120):
    """`value` of a child run of this command (same workload / steps / warm-up, no CPU baseline, training step or PMC passes) with
    `extra_flags`: (graphs/s, None) or (None, why not).  For secondary figures of the line; never fatal."""
    import subprocess
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT")
           and not k.startswith("TORCHELASTIC")}
    env.update(EGNN_BENCH_TRAFFIC_CHILD="1", **(env_extra or {}))
    cmd = [sys.executable, os.path.abspath(__file__), "--workload", args.workload, "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--no-cpu-baseline", "--no-train-step", "--no-live-traffic"] + list(extra_flags)
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout_s, check=True)
        return float(json.loads(r.stdout.strip().splitlines()[-1])["value"]), None
    except Exception as ex:
        return None, f"{type(ex).__name__}"


def unprimed_value(args, timeout_s=120):
    """`value` of a FRESH process that skips the priming steps (EGNN_BENCH_PRIME=0: only the contract's W warm-up steps in front of the
    timed region), as BENCH_r01 ... r04 were taken: a child run of this command without the CPU baseline, the training step and the PMC
    passes.  Returns (graphs/s, None) or (None, why not).  Secondary figure: what the priming changes is visible in the line."""
    import subprocess
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT")
           and not k.startswith("TORCHELASTIC")}
    env.update(EGNN_BENCH_PRIME="0", EGNN_BENCH_TRAFFIC_CHILD="1")
    cmd = [sys.executable, os.path.abspath(__file__), "--workload", args.workload, "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--no-cpu-baseline", "--no-train-step", "--no-live-traffic"] + (["--ragged-mask"] if args.ragged_mask else [])
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout_s, check=True)
        return float(json.loads(r.stdout.strip().splitlines()[-1])["value"]), None
    except Exception as ex:                                  # (never fatal)
        return None, f"{type(ex).__name__}"


def git_head():
    """Commit of the tree that printed the line: `git rev-parse`, or -- the GPU boxes get a snapshot without .git -- the file `.head`
    that tools/stamp_head.sh writes before a gpurun call."""
    try:
        import subprocess
        h = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True, timeout=5).stdout.strip()
        if h:
            return h
    except Exception:
        pass
    try:
        return open(os.path.join(ROOT, ".head")).read().strip() or None
    except Exception:
        return None


def make_inputs(kwargs, b, n, device, seed):
    g = torch.Generator().manual_seed(seed)
    feats = torch.randn(b, n, kwargs["dim"], generator=g)
    coors = torch.randn(b, n, 3, generator=g)
    mask = torch.ones(b, n, dtype=torch.bool)          # throughput runs: all-True mask
    return feats.to(device), coors.to(device), mask.to(device)


LAST_LOCAL_ELAPSED = [None]


def timed_region(step, steps, warmup, sync, barrier, reduce_max):
    """W untimed steps, then EXACTLY K steps bracketed by barrier + synchronize on both sides;
    returns the max over ranks of the elapsed seconds.  LAST_LOCAL_ELAPSED[0] = this rank's own K steps (its synchronize, before the
    closing barrier): what the N-rank line reports per rank (min / max) so that a straggler shows as such, not only through the max."""
    for _ in range(warmup):
        step()
    sync()
    barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    LAST_LOCAL_ELAPSED[0] = time.perf_counter() - t0
    barrier()
    sync()
    return reduce_max(time.perf_counter() - t0)


def _reference_module(kwargs):
    """The reference itself (oracle/_ref: bytecode of /root/reference built by oracle/build_ref.py), default init."""
    from oracle.build_ref import import_reference
    ref = import_reference()
    torch.manual_seed(0)
    return (ref.EGNN_Network(**kwargs) if "depth" in kwargs else ref.EGNN(**kwargs)).eval()


# RAM-safe batch for the reference on the host (it materialises E*(Din+2H)*4 bytes of edge activations)
B_CPU = {"north_star": 16, "c2_dense": 8, "c3_network": 16, "c4_sparse": 2, "c5_shard": 4}


def cpu_baseline(workload, kwargs, n, budget_s=25.0):
    """The reference CPU path on this box's host cores: the unmodified reference module (oracle/_ref,
    `kind: "reference"`), fp32, eval, no_grad, torch intra-op threads = all host cores, B_cpu graphs, 1 warm-up + min of
    up to 3 runs inside the time budget.  Falls back to the numpy port (oracle/egnn_oracle.py, `kind: "port"`) when the
    artefact is absent."""
    try:
        layer = _reference_module(kwargs)
    except ImportError:
        return cpu_baseline_port(kwargs, n, budget_s)
    b_cpu = B_CPU.get(workload, 2)
    g = torch.Generator().manual_seed(1000)
    feats = torch.randn(b_cpu, n, kwargs["dim"], generator=g)
    coors = torch.randn(b_cpu, n, 3, generator=g)
    mask = torch.ones(b_cpu, n, dtype=torch.bool)
    edges = adj = None
    if kwargs.get("edge_dim", 0) > 0:
        edges = torch.randn(b_cpu, n, n, kwargs["edge_dim"], generator=g)
    if kwargs.get("only_sparse_neighbors"):
        i = torch.arange(n)
        adj = (i[:, None] - i[None, :]).abs() <= 1
    is_net = "depth" in kwargs

    def run():
        with torch.no_grad():
            if is_net:
                layer(feats, coors, adj_mat=adj, edges=edges, mask=mask)
            else:
                layer(feats, coors, edges, mask, adj)

    t0 = time.perf_counter()
    run()                                                   # warm-up (thread pool, allocator)
    spent = time.perf_counter() - t0
    best, reps = None, 0
    while reps < 3 and (reps == 0 or spent + best < budget_s):
        t0 = time.perf_counter()
        run()
        dt = time.perf_counter() - t0
        spent += dt
        reps += 1
        best = dt if best is None else min(best, dt)
    cpu = ""
    try:
        cpu = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except Exception:
        pass
    return dict(value=round(b_cpu / best, 4), unit="graphs/s", cores=int(torch.get_num_threads()), kind="reference",
                sample=f"the reference module itself (oracle/_ref, torch {torch.__version__} CPU fp32, eval, no_grad), "
                       f"{b_cpu} graphs x N={n}, 1 warm-up + min of {reps} runs, {best:.2f} s per forward; "
                       f"host: {os.cpu_count()} logical cores ({cpu}), torch threads {torch.get_num_threads()}")


def train_step(layer, feats, coors, mask, edges=None, adj=None, steps=3):
    """One training step of the same workload (forward under autograd + backward of a scalar loss of both outputs), after the
    timed inference region: SURVEY.md §8f rank 2.  Reported next to the metric, never part of `value`."""
    import time
    f = feats.detach().clone().requires_grad_(True)
    c = coors.detach().clone().requires_grad_(True)
    fwd, bwd = [], []
    torch.cuda.reset_peak_memory_stats()
    for it in range(steps + 1):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        of, oc = layer(f, c, edges, mask=mask, adj_mat=adj)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        (of.square().mean() + oc.square().mean()).backward()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        if it:                                              # (the first step carries one-time allocations)
            fwd.append(1e3 * (t1 - t0)); bwd.append(1e3 * (t2 - t1))
        layer.zero_grad(); f.grad = None; c.grad = None
    return {"forward_ms": round(min(fwd), 3), "backward_ms": round(min(bwd), 3), "steps": steps,
            "peak_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2), "backward": "egnn_pytorch_amd.autograd (DESIGN.md section 10)"}


def reference_gpu_eager(kwargs, b, n, device, steps=3):
    """Secondary baseline: the reference module run on the MI355X through PyTorch-ROCm eager, timed with
    events on the current stream.  Context only -- never `value`."""
    layer = _reference_module(kwargs).to(device)
    feats, coors, mask = make_inputs(kwargs, b, n, device, seed=1000)
    with torch.no_grad():
        layer(feats, coors, mask=mask)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            layer(feats, coors, mask=mask)
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    return dict(value=round(b / ms * 1e3, 2), unit="graphs/s", ms_per_step=round(ms, 3),
                sample=f"reference module (oracle/_ref) .cuda(), PyTorch-ROCm eager, B={b} N={n}, mean of {steps} forwards",
                peak_mem_gb=round(torch.cuda.max_memory_allocated() / 2 ** 30, 1))


def cpu_baseline_port(kwargs, n, budget_s=25.0):
    """The oracle (numpy port of the reference's unfactorised algorithm) on a bounded sample of the same
    workload: B_cpu graphs of N nodes, default-scale weights, on this box's host cores."""
    import numpy as np
    from oracle import egnn_oracle as O
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    cfg = O.EGNNConfig(**kwargs)
    params = O.random_params(cfg, seed=0)
    rng = np.random.default_rng(0)
    b_cpu = 1
    feats = rng.standard_normal((b_cpu, n, kwargs["dim"])).astype(np.float32)
    coors = rng.standard_normal((b_cpu, n, 3)).astype(np.float32)
    mask = np.ones((b_cpu, n), bool)
    best, reps, spent = None, 0, 0.0
    while reps < 5 and (spent < budget_s * 0.6 or reps < 2):
        t0 = time.perf_counter()
        O.egnn_forward(cfg, params, feats, coors, mask=mask)
        dt = time.perf_counter() - t0
        spent += dt
        reps += 1
        best = dt if best is None else min(best, dt)
    return dict(value=round(b_cpu / best, 4), unit="graphs/s", cores=int(threads), kind="port",
                sample=f"oracle/egnn_oracle.py (numpy fp32), {b_cpu} graph x N={n}, min of {reps} runs, "
                       f"{best:.2f} s per forward; host {os.cpu_count()} logical cores")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="north_star", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--standin-backend", default=None, choices=["gloo"],
                    help="TEST AID (tests/test_sharding_gloo.py): run the rank logic -- env parsing, process group, broadcast-free "
                         "barrier / max-over-ranks timing, rank-0 JSON -- on CPU with this backend and a dummy step; no GPU work, "
                         "no numbers of any meaning")
    ap.add_argument("--ragged-mask", action="store_true", help="ragged masks (len ~ U{N/2..N}) instead of all-True")
    ap.add_argument("--train-step", action="store_true", help="also time forward + backward of the same workload (not part of `value`); "
                                                               "on by default for the single-layer workloads at N = 1")
    ap.add_argument("--no-train-step", action="store_true", help="skip the forward + backward timing")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not run the two rocprofv3 --pmc passes that measure the dominant kernel's HBM traffic (roofline.traffic then "
                         "comes from profiles/pmc_traffic.json, stamped with the commit of that profile run)")
    ap.add_argument("--hipgraph", action="store_true",
                    help="also time the step as a captured HIP graph (egnn_pytorch_amd.graphed), after everything else (secondary figure)")
    ap.add_argument("--reference-eager", action="store_true",
                    help="also time the reference module on the MI355X through PyTorch eager (secondary baseline)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves -- one process per GPU under
        # torch.distributed.run on this node, exactly the command line of the module docstring -- and hand its exit code on.
        # Rank 0 of the child job prints the ONE JSON line on the inherited stdout.
        import socket
        import subprocess
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world                                  # (under a launcher the launcher's world size is authoritative)
    standin = args.standin_backend is not None
    if standin:
        device = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X (no GPU visible); there is no CPU path to time")
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)

    dist = None
    if world > 1 or "WORLD_SIZE" in os.environ:            # under a launcher the process group is real even at world size 1 (RCCL smoke)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if standin:
            dist.init_process_group(args.standin_backend, rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)   # nccl == RCCL on ROCm
            if world > 1:
                from egnn_pytorch_amd.sharding import pin_to_local_numa_node
                pin_to_local_numa_node(local_rank)          # (best effort: the rank's host threads next to its GPU)

    def barrier():
        if dist is not None:
            dist.barrier()

    def reduce_max(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def per_rank_ms(steps):
        """{"min", "max", "ranks": [...]}: every rank's own ms per step over the timed region (all_gather of LAST_LOCAL_ELAPSED)."""
        mine = LAST_LOCAL_ELAPSED[0] / steps * 1e3
        if dist is None:
            vals = [mine]
        else:
            t = torch.tensor([mine], dtype=torch.float64, device=device)
            parts = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(parts, t)
            vals = [float(p.item()) for p in parts]
        return {"min": round(min(vals), 4), "max": round(max(vals), 4), "ranks": [round(v, 4) for v in vals]}

    if standin:
        # the rank logic only (CPU, gloo): the module's REAL forward code above the kernels -- a small layer of the workload's kind on this
        # rank's shard, rank-dependent input seeds, parameters broadcast from rank 0 -- with the kernel layer replaced by the torch
        # restatement of tests/_cpu_stub.py; the same timed region, the same rank-0 line (marked as a stand-in)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import _cpu_stub
        _cpu_stub.install()
        from egnn_pytorch_amd import EGNN, EGNN_Network
        kwargs, b, _ = WORKLOADS[args.workload]
        small = dict(kwargs, dim=16)
        small["num_nearest_neighbors"] = min(kwargs.get("num_nearest_neighbors", 0), 4)
        small.pop("edge_dim", None)
        small.pop("only_sparse_neighbors", None)
        torch.manual_seed(100 + rank)                       # ranks start with different weights; the broadcast makes them rank 0's
        layer = (EGNN_Network(**small) if "depth" in small else EGNN(**small)).eval()
        if dist is not None:
            from egnn_pytorch_amd.sharding import broadcast_parameters
            broadcast_parameters(layer)
        feats, coors, mask = make_inputs(small, 2, 12, device, seed=1000 + rank)

        @torch.no_grad()
        def standin_step():
            out = layer(feats, coors, mask=mask)
            return float(out[0].sum())
        elapsed = timed_region(standin_step, args.steps, args.warmup, lambda: None, barrier, reduce_max)
        rank_ms = per_rank_ms(args.steps)
        if rank == 0:
            print(json.dumps({"metric": "EGNN.forward graphs/sec", "value": round(world * b * args.steps / elapsed, 2),
                              "unit": "graphs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": round(elapsed / args.steps * 1e3, 4), "ms_per_step_by_rank": rank_ms, "higher_is_better": True,
                              "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                              "data": f"STAND-IN ({args.standin_backend}, CPU, kernels stubbed by tests/_cpu_stub.py): rank logic only, not a measurement",
                              "config": {"workload": args.workload, "graphs_per_gpu": b, "global_batch": world * b}}), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    from egnn_pytorch_amd import EGNN, EGNN_Network, phase_timer, check_range, _ops
    # Range status word (include/egnn_hip.h: EGNN_RANGE_*): the timed region runs the library's DEFAULT -- `sync`: every forward ends with
    # one 8-byte device -> host read, which is also what arms the automatic plain-fp32 re-run of out-of-range calls -- so that `value` is
    # the configuration a user gets (rounds 1 - 4 timed the deferred mode).  The deferred mode (no synchronisation:
    # the word is copied to pinned memory after each forward and examined at the next call) is timed beside it: `value_range_check_deferred`.
    _ops.RANGE_CHECK = os.environ.get("EGNN_RANGE_CHECK", "sync")

    kwargs, b, n = WORKLOADS[args.workload]
    torch.manual_seed(0)
    is_net = "depth" in kwargs
    layer = (EGNN_Network(**kwargs) if is_net else EGNN(**kwargs)).to(device).eval()
    if dist is not None:
        from egnn_pytorch_amd.sharding import broadcast_parameters
        broadcast_parameters(layer)                       # the only collective: one-off weight replication
    feats, coors, mask = make_inputs(kwargs, b, n, device, seed=1000 + rank)   # this rank's shard of the batch
    if args.ragged_mask:                                                        # parity-style ragged batch
        g = torch.Generator().manual_seed(2000 + rank)
        lens = torch.randint(n // 2, n + 1, (b,), generator=g)
        mask = (torch.arange(n)[None, :] < lens[:, None]).to(device)
    edges = adj = None
    if kwargs.get("edge_dim", 0) > 0:
        edges = torch.randn(b, n, n, kwargs["edge_dim"], device=device)
    if kwargs.get("only_sparse_neighbors"):
        i = torch.arange(n, device=device)
        adj = (i[:, None] - i[None, :]).abs() <= 1                              # README chain adjacency incl. diagonal

    @torch.no_grad()                                     # the metric is inference (eval, no_grad): BASELINE.md §2
    def step():
        if is_net:
            layer(feats, coors, adj_mat=adj, edges=edges, mask=mask)
        else:
            layer(feats, coors, edges, mask, adj)

    # Priming, before the W warm-up steps of the contract (untimed, reported as `priming_steps`): on most boxes a one-off stall of ~35 ms
    # lands somewhere in a process's first few dozen forwards of the network workloads, and with 20 short steps it dominated whichever
    # timed region came first -- c3: 15 - 18 k instead of 27 - 30 k graphs/s (profiles/r05_experiments/bench_first_region.txt).  The
    # north-star line is not affected (22.4 - 22.6 k with and without).
    mode0 = _ops.RANGE_CHECK
    PRIME_STEPS = int(os.environ.get("EGNN_BENCH_PRIME", "24"))   # (0: no priming)
    PRIME_SYNC = 3 if PRIME_STEPS > 0 else 0
    for m_, cnt in (("deferred", PRIME_STEPS), ("sync", PRIME_SYNC)):
        if mode0 == "off":
            break
        _ops.RANGE_CHECK = m_
        for _ in range(cnt):
            step()
        torch.cuda.synchronize()
        check_range()
    _ops.RANGE_CHECK = mode0
    elapsed = timed_region(step, args.steps, args.warmup, torch.cuda.synchronize, barrier, reduce_max)
    rank_ms = per_rank_ms(args.steps)                    # (a collective: every rank calls it)
    check_range()                                        # raises if any timed step left the representable range

    # the same K steps in the other mode of the range check, so that the line shows what the per-forward synchronisation costs
    value_other, other_mode = None, ("deferred" if _ops.RANGE_CHECK == "sync" else "sync")
    if world == 1 and _ops.RANGE_CHECK in ("sync", "deferred"):
        mode = _ops.RANGE_CHECK
        _ops.RANGE_CHECK = other_mode
        try:
            el2 = timed_region(step, args.steps, 1, torch.cuda.synchronize, barrier, reduce_max)
            check_range()
            value_other = world * b * args.steps / el2
        finally:
            _ops.RANGE_CHECK = mode

    # ---- per-kernel durations (events on the launch stream), outside the timed region
    PROBE_STEPS = 5
    with phase_timer() as pt:
        for _ in range(PROBE_STEPS):
            step()
    summary = pt.summary()
    per_kernel = {k: sum(v) / len(v) for k, v in summary.items()}              # average duration of one launch
    launches = {k: len(v) / PROBE_STEPS for k, v in summary.items()}           # launches per step (network: one per layer)

    if rank == 0:
        counts, shp = model_counts(kwargs, b, n)
        depth = kwargs.get("depth", 1)
        # kernels ordered by their share of the step; the roofline object is the dominant one
        order = sorted(per_kernel.items(), key=lambda kv: -kv[1] * launches[kv[0]])
        kernels = [roofline_entry(k, counts, ms, launches[k]) for k, ms in order if k in counts]
        other = {k: round(ms * launches[k], 4) for k, ms in order if k not in counts}
        dominant = dict(next(kr for kr in kernels if kr["bound"] in ("hbm", "mfma")))
        traffic = thead = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                per_wl = tj.get(args.workload, tj if args.workload == "north_star" and "north_star" not in tj else {})
                traffic = per_wl.get(dominant["kernel"])
                thead = per_wl.get("_head", tj.get("_head"))
            except Exception:
                traffic = None
        dominant["traffic_profile"] = traffic                # (the committed profile's figure, with the commit it was taken at)
        dominant["traffic_head"] = thead
        src = "profiles/pmc_traffic.json: (2 FETCH_SIZE + WRITE_SIZE) * 1024 per launch, separate rocprofv3 --pmc passes " \
              "of this command (tools/profile.sh)" if traffic is not None else None
        under_profiler = "rocprofiler" in os.environ.get("LD_PRELOAD", "") or any(k.startswith("ROCPROF") for k in os.environ)
        if world == 1 and not args.no_live_traffic and os.environ.get("EGNN_BENCH_TRAFFIC_CHILD") != "1" and not under_profiler:
            live, note = live_traffic(args.workload, dominant["kernel"])
            if live is not None:
                traffic, src = live, note
                dominant["traffic_head"] = git_head()
            else:
                src = (src or "none") + f" [live measurement unavailable: {note}]"
        dominant["traffic"] = traffic
        dominant["traffic_source"] = src
        # the clock each kernel actually ran at (the chip clocks to its power budget: the MFMA peak of 2.5 PFLOP/s assumes 2.4 GHz)
        if world == 1 and not args.no_live_traffic and os.environ.get("EGNN_BENCH_TRAFFIC_CHILD") != "1" and not under_profiler:
            clocks, cnote = live_clocks(args.workload)
            if clocks:
                for kr in kernels + [dominant]:
                    ghz = clocks.get(kr["kernel"])
                    if ghz:
                        kr["effective_clock_ghz"] = ghz
                        if kr["bound"] == "mfma":
                            kr["frac_at_clock"] = round(kr["frac"] * 2.4 / ghz, 4)
                dominant["clock_source"] = cnote
            else:
                dominant["clock_source"] = f"unavailable: {cnote}"
        graphs = world * b * args.steps
        value = graphs / elapsed
        out = {
            "metric": "EGNN.forward graphs/sec", "value": round(value, 2), "unit": "graphs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "ms_per_step_by_rank": rank_ms, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "edges_per_s": round(value * n * shp["K"] * depth, 1), "range_check": _ops.RANGE_CHECK, "priming_steps": PRIME_STEPS + PRIME_SYNC,
            f"value_range_check_{other_mode}": None if value_other is None else round(value_other, 2),
            "config": {"workload": workload_label(kwargs, b, n, shp["K"]), "name": args.workload,
                       "graphs_per_gpu": b, "nodes": n, "neighbors": shp["K"], "layers": depth, "global_batch": world * b,
                       "mask": "ragged" if args.ragged_mask else "all-true",
                       "parallelism": f"batch-shard x{world} (no data-path collective)"},
            "head": git_head(),
            "roofline": dominant,
            "kernels": kernels,
            "other_kernels_ms_per_step": other,
            "sum_kernel_ms": round(sum(per_kernel[k] * launches[k] for k in per_kernel), 4),
        }
        if world == 1 and PRIME_STEPS > 0 and not args.no_live_traffic and os.environ.get("EGNN_BENCH_TRAFFIC_CHILD") != "1" and not under_profiler:
            uv, why = unprimed_value(args)
            out["value_unprimed"] = round(uv, 2) if uv is not None else None
            if uv is None:
                out["value_unprimed_note"] = f"child run failed: {why}"
        if (world == 1 and not args.ragged_mask and not args.no_live_traffic and os.environ.get("EGNN_BENCH_TRAFFIC_CHILD") != "1"
                and not under_profiler and not kwargs.get("only_sparse_neighbors")):
            # a padded batch (the parity protocol's ragged masks, SURVEY.md section 8d: len ~ U{N/2..N}): the edge pass and the projection
            # skip the padded nodes.  Secondary figure; `value` is the all-true batch the metric is defined on.
            rv, why = child_value(args, ["--ragged-mask"])
            out["value_ragged_masks"] = round(rv, 2) if rv is not None else None
            if rv is None:
                out["value_ragged_masks_note"] = f"child run failed: {why}"
        if dist is not None:
            out["process_group"] = {"backend": dist.get_backend(), "world_size": world}
        if not args.no_cpu_baseline and world == 1:          # the CPU leg is timed at N = 1 only
            out["cpu_baseline"] = cpu_baseline(args.workload, kwargs, n)
        if args.reference_eager and world == 1:
            out["reference_gpu_eager"] = reference_gpu_eager(kwargs, b, n, device)
        default_train = args.workload == "north_star" and not args.no_train_step
        if (args.train_step or default_train) and world == 1 and not is_net:
            # after the timed inference region (never part of `value`): must not cost the line if it fails
            try:
                out["train_step"] = train_step(layer, feats, coors, mask, edges, adj)
            except Exception as exc:                           # noqa: BLE001
                out["train_step"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
        if args.hipgraph and world == 1 and not kwargs.get("only_sparse_neighbors"):
            # the same K steps as ONE captured HIP graph per step (egnn_pytorch_amd.graphed: the launch sequence replayed, the range word
            # read after every replay): what the host-side launch path costs at this size.  Secondary figure, never `value`.
            try:
                from egnn_pytorch_amd import graphed
                with torch.no_grad():
                    if is_net:
                        run = graphed(layer, feats, coors, adj_mat=adj, edges=edges, mask=mask, range_check="sync")
                        gstep = lambda: run(feats, coors, adj_mat=adj, edges=edges, mask=mask)     # noqa: E731
                    else:
                        run = graphed(layer, feats, coors, edges, mask, adj, range_check="sync")
                        gstep = lambda: run(feats, coors, edges, mask, adj)                        # noqa: E731
                    elg = timed_region(gstep, args.steps, args.warmup, torch.cuda.synchronize, barrier, reduce_max)
                out["hipgraph_replay"] = {"value": round(b * args.steps / elg, 2), "ms_per_step": round(elg / args.steps * 1e3, 4),
                                          "range_check": "sync", "includes": "copy of the inputs into the graph's buffers"}
            except Exception as exc:                           # noqa: BLE001
                out["hipgraph_replay"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
        print(json.dumps(out), flush=True)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
