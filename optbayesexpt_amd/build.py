"""Build recipe for libobe_hip.so (gfx950 only).

``python -m optbayesexpt_amd.build`` compiles every ``csrc/*.hip`` with hipcc for
gfx950 and links them into ``optbayesexpt_amd/lib/libobe_hip.so`` (in-tree, so the
binary travels to the GPU box with the repository snapshot).  hipcc cross-compiles
without a GPU, so this also is the CPU-side "does it build" check.

Flags: -ffp-contract=off makes every ``a*b+c`` in the parity-critical kernels two
correctly rounded operations, like NumPy; the sweep kernel requests its FMAs
explicitly with fma().
"""
import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "lib")
OBJ_DIR = os.path.join(OUT_DIR, "obj")
LIB = os.path.join(OUT_DIR, "libobe_hip.so")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
         "-Wall", "-Wno-unused-function", f"-I{INCLUDE}"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    return hs + [os.path.join(INCLUDE, "obe_hip.h"), os.path.abspath(__file__)]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _stamp():
    """obj/obe_fingerprint.h: the hash of the kernel sources, compiled into the library
    (obe_source_fingerprint) so that a stale .so is recognised when it is loaded.  Rewritten only
    when the hash changes."""
    path = os.path.join(OBJ_DIR, "obe_fingerprint.h")
    text = f'#define OBE_SOURCE_FINGERPRINT "{_source_fingerprint()}"\n'
    if not os.path.exists(path) or open(path).read() != text:
        with open(path, "w") as f:
            f.write(text)
    return path


def _compile(src):
    obj = os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o")
    stamp = _stamp()
    deps = [src] + headers() + ([stamp] if src.endswith("obe_capi.hip") else [])
    if _stale(obj, deps):
        cmd = [HIPCC] + FLAGS + ["-include", stamp, "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    return obj


def build(force=False, verbose=False):
    """Compile (if stale) and return the path of libobe_hip.so."""
    os.makedirs(OBJ_DIR, exist_ok=True)
    if force:
        for f in os.listdir(OBJ_DIR):
            os.remove(os.path.join(OBJ_DIR, f))
    srcs = sources()
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        objs = list(ex.map(_compile, srcs))
    if force or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    with open(LIB + ".fingerprint", "w") as f:          # what library_is_stale() reads without loading the .so
        f.write(_source_fingerprint())
    if verbose:
        print(f"built {LIB} ({os.path.getsize(LIB) / 1024:.0f} KiB)")
    return LIB


def library_is_stale():
    """True if libobe_hip.so is missing or was built from other kernel sources than the ones
    next to it (sidecar file written by build(); the library also carries the hash itself)."""
    side = LIB + ".fingerprint"
    if not (os.path.exists(LIB) and os.path.exists(side)):
        return True
    return open(side).read().strip() != _source_fingerprint()


#: where generated models are compiled to; OBE_PLUGIN_DIR overrides (e.g. a read-only installation)
PLUGIN_DIR = os.environ.get("OBE_PLUGIN_DIR", os.path.join(OUT_DIR, "plugins"))
PLUGIN_SOURCES = ["obe_capi.hip", "obe_update.hip", "obe_sweep.hip", "obe_yspace.hip"]   # model-dependent


def _source_fingerprint():
    import hashlib
    h = hashlib.sha1()
    for f in sorted(os.listdir(CSRC)) + [os.path.join(INCLUDE, "obe_hip.h")]:
        path = f if os.path.isabs(f) else os.path.join(CSRC, f)
        h.update(open(path, "rb").read())
    h.update(" ".join(FLAGS).replace(INCLUDE, "<include>").encode())      # not the checkout's absolute path
    h.update(open(os.path.join(HERE, "_exprmodel.py"), "rb").read())     # the header generator
    return h.hexdigest()[:12]


def plugin_path(model_digest):
    return os.path.join(PLUGIN_DIR, f"libobe_model_{model_digest}_{_source_fingerprint()}.so")


def build_plugin(header_text, model_digest, verbose=False):
    """Compile the model-dependent kernel sources for ONE generated model
    (``OBE_PLUGIN_MODEL_HEADER``) into a plugin library with the same entry points as
    libobe_hip.so.  Cached by the hash of the model header and of the kernel sources."""
    lib = plugin_path(model_digest)
    os.makedirs(PLUGIN_DIR, exist_ok=True)
    fp = _source_fingerprint()
    for f in os.listdir(PLUGIN_DIR):          # plugins of older kernel sources are dead weight
        if not f.rsplit(".", 1)[0].endswith(fp):
            os.remove(os.path.join(PLUGIN_DIR, f))
    if os.path.exists(lib):
        return lib
    if not os.path.exists(HIPCC):
        raise RuntimeError(f"{lib} is not built and hipcc ({HIPCC}) is not available to build it")
    stem = os.path.basename(lib)[3:-3]
    header = os.path.join(PLUGIN_DIR, stem + ".h")
    with open(header, "w") as f:
        f.write(header_text)
    define = f'-DOBE_PLUGIN_MODEL_HEADER="{header}"'
    os.makedirs(OBJ_DIR, exist_ok=True)
    stamp = _stamp()

    def one(src):
        obj = os.path.join(PLUGIN_DIR, f"{stem}_{src[:-4]}.o")
        r = subprocess.run([HIPCC] + FLAGS + [define, "-include", stamp, "-c", os.path.join(CSRC, src), "-o", obj],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for the generated model ({src}):\n{r.stdout}\n{r.stderr}")
        return obj

    with concurrent.futures.ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(one, PLUGIN_SOURCES))
    # -Bsymbolic: calls between the plugin's own entry points must not be interposed by
    # the same-named symbols of libobe_hip.so already loaded in the process
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic"] + objs + ["-o", lib],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"plugin link failed:\n{r.stdout}\n{r.stderr}")
    for o in objs:
        os.remove(o)
    if verbose:
        print(f"built {lib}")
    return lib


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
