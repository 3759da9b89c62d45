"""Build recipe for oracle/_ref/: the REFERENCE ITSELF as a test / baseline artefact (test infrastructure only).

The reference (lucidrains/egnn-pytorch, /root/reference) is pure Python, so its "compiled" form is CPython bytecode:
this script byte-compiles the package's modules, from the sources where they lie under /root/reference, into
`oracle/_ref/egnn_pytorch/*.pyc` (sourceless modules, importable with `oracle/_ref` on sys.path under the same
interpreter version).  No reference source is copied; `oracle/_ref/` is git-ignored (outputs only) but travels to the
GPU box with the snapshot, where /root/reference does not exist.

Consumers (and only these): `bench.py`'s `cpu_baseline` / `--reference-eager` legs (the reference timed on the box's
host cores and, for context, on the MI355X through PyTorch-ROCm eager) and `tests/` (`tests/test_ref_artifact.py`).
Nothing under `egnn_pytorch_amd/` may import it.

    python oracle/build_ref.py            # no-op (exit 0) when /root/reference is absent and the artefact exists
"""
import os
import py_compile
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = os.environ.get("EGNN_REFERENCE_SRC", "/root/reference")
OUT = os.path.join(HERE, "_ref")
MODULES = ("__init__", "egnn_pytorch", "egnn_pytorch_geometric", "utils")


def built() -> bool:
    return all(os.path.exists(os.path.join(OUT, "egnn_pytorch", m + ".pyc")) for m in MODULES)


def build() -> bool:
    """Returns True when oracle/_ref/ is usable afterwards."""
    pkg = os.path.join(REF_SRC, "egnn_pytorch")
    if not os.path.isdir(pkg):
        return built()
    os.makedirs(os.path.join(OUT, "egnn_pytorch"), exist_ok=True)
    for m in MODULES:
        py_compile.compile(os.path.join(pkg, m + ".py"), cfile=os.path.join(OUT, "egnn_pytorch", m + ".pyc"),
                           dfile=f"egnn_pytorch/{m}.py", doraise=True, optimize=0)
    with open(os.path.join(OUT, "BUILD_INFO"), "w") as f:
        f.write(f"byte-compiled from {pkg} with CPython {sys.version.split()[0]} (magic {py_compile.importlib.util.MAGIC_NUMBER.hex()})\n")
    return True


_cached = None


def import_reference():
    """Import the reference package from oracle/_ref (bytecode) and return the package module.  It is loaded in
    isolation: an `egnn_pytorch` some other test already imported from /root/reference keeps its place in sys.modules.
    Raises ImportError if the artefact is missing or was built by another interpreter version."""
    global _cached
    if _cached is not None:
        return _cached
    if not built():
        raise ImportError("oracle/_ref is not built (run `python oracle/build_ref.py` where /root/reference exists)")
    mine = lambda k: k == "egnn_pytorch" or k.startswith("egnn_pytorch.")
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if mine(k)}
    sys.path.insert(0, OUT)
    try:
        import egnn_pytorch as ref                          # the reference, not egnn_pytorch_amd
    finally:
        sys.path.remove(OUT)
        for k in [k for k in sys.modules if mine(k)]:
            sys.modules.pop(k)
        sys.modules.update(saved)
    if not os.path.abspath(ref.__file__).startswith(OUT):
        raise ImportError(f"`egnn_pytorch` resolved to {ref.__file__}, not to oracle/_ref")
    _cached = ref
    return ref


if __name__ == "__main__":
    ok = build()
    print("oracle/_ref:", "ready" if ok else "NOT built (no /root/reference here and no prebuilt artefact)")
    sys.exit(0)
