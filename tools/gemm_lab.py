"""The split-f16 GEMM (csrc/linear_hl.hip) in isolation: every library under build_variants/ (tools/variants.py build src=linear_hl ...)
loaded into ONE process and timed in interleaved rounds on the layer's real GEMM shapes -- the projection (through the `_lda` entry:
the [feats | m_i] image), node_mlp.0 (SiLU, packed (hi, lo) output) and node_mlp.3 (residual) of the north star, c3 and c5 -- with a
digest of the outputs (variants that only re-schedule must be bit-identical to the production library).

   python tools/gemm_lab.py [shapes=ns,c3,c5] [rounds=5] [n=10] [only=tagA+tagB] [prod=0] [gm=1,2,4,8,16]

One line per (shape, variant): min / median over the rounds of the mean of n back-to-back launches (HIP events on the launch stream)."""
import ctypes
import hashlib
import zlib
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

torch.set_grad_enabled(False)
from egnn_pytorch_amd import _abi, _ops, _weights  # noqa: E402

VDIR = os.path.join(ROOT, "build_variants")
c_void_p, c_int, c_int64, c_float = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float


def bind(path):
    lib = ctypes.CDLL(path)
    lib.egnn_linear_hl_f32.restype = c_int
    lib.egnn_linear_hl_f32.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int64, c_void_p,
                                       c_int64, c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]
    lib.egnn_linear_hl_lda_f32.restype = c_int
    lib.egnn_linear_hl_lda_f32.argtypes = [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int64, c_void_p,
                                           c_int64, c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]
    return lib


def layer_shapes(tag, dim, m_rows=65536, m_dim=16):
    h = 2 * (2 * dim + 1)
    hp = _weights.padded_hidden(h)
    return {
        tag + "_proj": dict(M=m_rows, N=2 * hp, K=dim, lda=_ops._kpad(dim + m_dim), split_cols=hp, bias=True, act=0, res=False, out="f32"),
        tag + "_mlp0": dict(M=m_rows, N=2 * dim, K=dim + m_dim, lda=0, split_cols=0, bias=True, act=1, res=False, out="hl"),
        tag + "_mlp1": dict(M=m_rows, N=dim, K=2 * dim, lda=0, split_cols=0, bias=True, act=0, res=True, out="f32"),
    }


class Case:
    def __init__(self, name, s, dev):
        g = torch.Generator(device="cpu").manual_seed(zlib.crc32(name.encode()) & 0xffff)
        self.name, self.s = name, s
        m, n, k = s["M"], s["N"], s["K"]
        a = torch.randn(m, k, generator=g).to(dev)
        kp = _ops._kpad(k)
        if s["lda"]:
            wide = torch.zeros(m, s["lda"], device=dev)
            wide[:, :k] = a
            self.a = _ops.split_f16(wide)
        else:
            self.a = _ops.split_f16(a)
        self.kp = kp
        w = (torch.randn(n, k, generator=g) / k ** 0.5).to(dev)
        self.whi, self.wlo, self.inv, self.w_rows = _weights.split_f16(w)
        self.bias = torch.randn(n, generator=g).to(dev) if s["bias"] else None
        self.res = torch.randn(m, n, generator=g).to(dev) if s["res"] else None
        self.c = torch.empty(m, n, device=dev) if s["out"] == "f32" else None
        self.kp_out = _ops._kpad(n) if s["out"] == "hl" else 0
        self.chi = _ops._packed_empty(m, self.kp_out, dev, True) if s["out"] == "hl" else None
        self.clo = _ops._packed_empty(m, self.kp_out, dev, True) if s["out"] == "hl" else None
        self.status = torch.zeros(4, dtype=torch.int32, device=dev)
        # flops issued on the matrix cores: three MFMA terms per algorithmic multiply-add
        self.flops = 3 * 2.0 * m * n * kp

    def launch(self, lib, stream):
        if isinstance(lib, tuple):
            os.environ["EGNN_HL_GROUP_M"] = lib[1]
            lib = lib[0]
        s = self
        p = lambda t: None if t is None else t.data_ptr()
        if s.s["lda"]:
            rc = lib.egnn_linear_hl_lda_f32(p(s.a.hi), p(s.a.lo), s.a.kp, p(s.whi), p(s.wlo), float(s.inv), p(s.bias), p(s.res),
                                            s.s["N"] if s.res is not None else 0, p(s.c), s.s["N"], p(s.chi), p(s.clo), s.kp_out,
                                            s.s["M"], s.s["N"], s.kp, s.w_rows, s.s["act"], s.s["split_cols"], p(s.status), stream)
        else:
            rc = lib.egnn_linear_hl_f32(p(s.a.hi), p(s.a.lo), p(s.whi), p(s.wlo), float(s.inv), p(s.bias), p(s.res),
                                        s.s["N"] if s.res is not None else 0, p(s.c), s.s["N"], p(s.chi), p(s.clo), s.kp_out,
                                        s.s["M"], s.s["N"], s.kp, s.w_rows, s.s["act"], s.s["split_cols"], p(s.status), stream)
        if rc != 0:
            raise RuntimeError(f"{s.name}: rc {rc}")

    def digest(self):
        h = hashlib.sha256()
        for t in (self.c, self.chi, self.clo):
            if t is not None:
                h.update(t.cpu().numpy().tobytes())
        return h.hexdigest()[:10]

    def poison(self):
        for t in (self.c, self.chi, self.clo):
            if t is not None:
                t.view(torch.uint8).fill_(0x7F) if t.dtype != torch.float32 else t.fill_(float("nan"))


def main():
    opts = dict(a.split("=", 1) for a in sys.argv[1:] if "=" in a)
    shapes = opts.get("shapes", "ns").split(",")
    rounds, n = int(opts.get("rounds", "5")), int(opts.get("n", "10"))
    libs = []
    if opts.get("prod", "1") != "0":
        libs.append(("production", _abi.lib_path()))
    idx = os.path.join(VDIR, "index.json")
    if os.path.exists(idx):
        for tag in json.load(open(idx)):
            if not tag.startswith("linear_hl"):
                continue
            if "only" in opts and not any(o in tag for o in opts["only"].split("+")):
                continue
            libs.append((tag, os.path.join(VDIR, tag, "libegnn_hip.so")))
    bound = [(tag, bind(path)) for tag, path in libs]
    if "gm" in opts:
        # sweep the run-time group size of the block -> tile map (EGNN_HL_GROUP_M, read by the library at every launch) on the first library
        tag0, lib0 = bound[0]
        bound = [(f"{tag0} GROUP_M={g}", (lib0, g)) for g in opts["gm"].split(",")]
    dev = torch.device("cuda", 0)
    all_shapes = {}
    for sh in shapes:
        dim = {"ns": 512, "c3": 128, "c5": 256}[sh]
        all_shapes.update(layer_shapes(sh, dim))
    stream = torch.cuda.current_stream().cuda_stream
    totals = {tag: 0.0 for tag, _ in bound}
    for name, s in all_shapes.items():
        case = Case(name, s, dev)
        times = {tag: [] for tag, _ in bound}
        digests = {}
        for tag, lib in bound:                       # warm-up + digest
            case.poison()
            case.launch(lib, stream)
            torch.cuda.synchronize()
            digests[tag] = case.digest()
            for _ in range(2):
                case.launch(lib, stream)
        torch.cuda.synchronize()
        for r in range(rounds):
            order = bound if r % 2 == 0 else bound[::-1]
            for tag, lib in order:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(n):
                    case.launch(lib, stream)
                e1.record()
                torch.cuda.synchronize()
                times[tag].append(e0.elapsed_time(e1) / n)
        ref = digests[bound[0][0]]
        for tag, _ in bound:
            mn, md = min(times[tag]), statistics.median(times[tag])
            totals[tag] += mn
            print(f"{name:9s} {tag:58s} min {mn:.4f} med {md:.4f} ms  {case.flops / mn / 1e9:7.0f} TF issued = {case.flops / mn / 1e9 / 2500:.3f}"
                  f"  {digests[tag]}{'' if digests[tag] == ref else '  != ' + bound[0][0]}", flush=True)
        del case
        torch.cuda.empty_cache()
    print("sum of minima over the shapes:")
    for tag, _ in bound:
        print(f"   {tag:58s} {totals[tag]:.4f} ms")


if __name__ == "__main__":
    main()
