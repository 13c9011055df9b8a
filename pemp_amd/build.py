"""Build libpemp_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

Every ``csrc/*.hip`` is compiled to an object under ``pemp_amd/_obj/`` (git-ignored) -- in parallel, and only when
the source, a header or the flags changed -- then linked into ``pemp_amd/libpemp_hip.so``."""
import hashlib
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
OUT = os.path.join(HERE, "libpemp_hip.so")
SOURCES = ("conv_igemm.hip", "conv_dma.hip", "conv_dma2.hip", "conv_wgrad.hip", "train_ops.hip", "pool_misc.hip", "head.hip", "head_bwd.hip", "cedt.hip",
           "episode_io.hip", "dropout.hip", "cm_linear.hip")
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC"]


def _headers():
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".h", ".inc"))] + \
           [os.path.join(HERE, "..", "include", "pemp_hip.h")]


def csrc_digest():
    """Short content hash of every kernel source + the C-ABI header: profiles/*.json record it, so that a PMC figure is only
    ever quoted next to the code it was measured on."""
    h = hashlib.sha1()
    for f in sorted(os.listdir(CSRC)):
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    with open(os.path.join(HERE, "..", "include", "pemp_hip.h"), "rb") as fh:
        h.update(fh.read())
    return h.hexdigest()[:12]


CONV_ENGINE = ("common.h", "conv_common.h", "conv_igemm.hip", "conv_dma.hip", "conv_dma2.hip")


def conv_digest():
    """Content hash of the conv engine's sources only: the PMC figures bench.py quotes for the eval step (HBM bytes per conv
    launch, MFMA pipe utilisation) belong to these files, a change elsewhere (head, training kernels) does not unpin them."""
    h = hashlib.sha1()
    for f in CONV_ENGINE:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:12]


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "pemp_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def _stamp(src, flags):
    h = hashlib.sha1(" ".join(flags).encode())
    for f in [src] + _headers():
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _compile(hipcc, name, flags, force, verbose):
    src = os.path.join(CSRC, name)
    obj = os.path.join(OBJ, name.replace(".hip", ".o"))
    stamp_file = obj + ".stamp"
    stamp = _stamp(src, flags)
    if not force and os.path.exists(obj) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
        return obj
    cmd = [hipcc] + flags + ["-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    with open(stamp_file, "w") as f:
        f.write(stamp)
    return obj


def build(force=False, verbose=False, extra_flags=(), out=None):
    out = out or OUT
    if not force and out == OUT and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    flags = FLAGS + list(extra_flags)
    jobs = max(1, min(len(SOURCES), int(os.environ.get("PEMP_BUILD_JOBS", str(min(os.cpu_count() or 1, 8))))))
    with ThreadPoolExecutor(jobs) as ex:
        objs = list(ex.map(lambda n: _compile(hipcc, n, flags, force, verbose), SOURCES))
    tmp = out + f".tmp{os.getpid()}"          # link aside, then rename: concurrent loaders never see a torn file
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(tmp, out)
    return out


if __name__ == "__main__":
    print(build(force=True, verbose=True))
