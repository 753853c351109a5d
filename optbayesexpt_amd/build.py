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
import contextlib
import fcntl
import os
import re
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "lib")
OBJ_DIR = os.path.join(OUT_DIR, "obj")
LIB = os.path.join(OUT_DIR, "libobe_hip.so")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fvisibility=hidden: the only dynamic symbols of the library (and of a plugin) are the entry points
# include/obe_hip.h marks OBE_API; the C++ helpers and the kernels' host stubs stay internal
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden",
         "-fvisibility-inlines-hidden", "-Wall", "-Wno-unused-function", f"-I{INCLUDE}"]


def _export_map(directory):
    """Linker version script: the dynamic symbol table holds the obe_* entry points and nothing else.
    (-fvisibility=hidden covers the C++ helpers; the host-side handle objects clang emits for every
    __global__ kernel keep default visibility whatever the flag says, hence the list at link time.)"""
    path = os.path.join(directory, "obe_exports.map")
    text = "{ global: obe_*; local: *; };\n"
    if not os.path.exists(path) or open(path).read() != text:
        with open(path, "w") as f:
            f.write(text)
    return path


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


@contextlib.contextmanager
def _locked(directory):
    """Exclusive inter-process lock on ``directory/.lock``: with one process per GPU every rank may
    find the same library or plugin missing at the same moment; one of them builds, the others wait
    and then see the finished file."""
    os.makedirs(directory, exist_ok=True)
    with open(os.path.join(directory, ".lock"), "a+") as fh:
        fcntl.flock(fh, fcntl.LOCK_EX)
        try:
            yield
        finally:
            fcntl.flock(fh, fcntl.LOCK_UN)


def _write_atomically(path, text):
    tmp = f"{path}.tmp{os.getpid()}"
    with open(tmp, "w") as f:
        f.write(text)
    os.replace(tmp, path)


def build(force=False, verbose=False, only_if_stale=False):
    """Compile (if stale) and return the path of libobe_hip.so.  Serialised across processes by a
    file lock; the library is linked under a temporary name and renamed into place, so a concurrent
    dlopen never sees a half-written file.  ``only_if_stale`` re-checks the source fingerprint under
    the lock (the caller's own check may be out of date by the time the lock is granted).  Messages
    go to stderr: stdout belongs to the caller (bench.py prints one JSON line there)."""
    with _locked(OUT_DIR):
        if only_if_stale and not library_is_stale():
            return LIB
        os.makedirs(OBJ_DIR, exist_ok=True)
        if force:
            for f in os.listdir(OBJ_DIR):
                if f.endswith(".o") and os.path.isfile(os.path.join(OBJ_DIR, f)):
                    os.remove(os.path.join(OBJ_DIR, f))
        srcs = sources()
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
            objs = list(ex.map(_compile, srcs))
        if force or _stale(LIB, objs):
            tmp = f"{LIB}.tmp{os.getpid()}"
            cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC",
                   f"-Wl,--version-script={_export_map(OBJ_DIR)}"] + objs + ["-o", tmp]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                if os.path.exists(tmp):
                    os.remove(tmp)
                raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
            os.replace(tmp, LIB)
        # what library_is_stale() reads without loading the .so
        _write_atomically(LIB + ".fingerprint", _source_fingerprint())
        if verbose:
            print(f"built {LIB} ({os.path.getsize(LIB) / 1024:.0f} KiB)", file=sys.stderr)
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
#: linked into every plugin as well, but the same for every model: the object the LIBRARY build left in lib/obj is
#: reused (its ~45 kernel instantiations took 9 of the 10 s of a formula's first use when each plugin recompiled them)
PLUGIN_COMMON_SOURCES = ["obe_update_common.hip"]


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


_PLUGIN_FILE = re.compile(r"^(?:lib)?obe_model_[0-9a-f]{16}_([0-9a-f]{12})\.(?:so|h)$")


def _remove_outdated_plugins(fp):
    """Plugins (and their generated headers) built from older kernel sources are dead weight.  Only
    regular files carrying this package's own naming pattern are touched: PLUGIN_DIR may be a
    shared directory (OBE_PLUGIN_DIR=/tmp, ~/.cache, ...) holding files that are not ours."""
    for f in os.listdir(PLUGIN_DIR):
        m = _PLUGIN_FILE.match(f)
        path = os.path.join(PLUGIN_DIR, f)
        if m and m.group(1) != fp and os.path.isfile(path) and not os.path.islink(path):
            try:
                os.remove(path)
            except OSError:
                pass


def build_plugin(header_text, model_digest, verbose=False):
    """Compile the model-dependent kernel sources for ONE generated model
    (``OBE_PLUGIN_MODEL_HEADER``) into a plugin library with the same entry points as
    libobe_hip.so.  Cached by the hash of the model header and of the kernel sources.  Safe with
    one process per GPU: the build happens under a file lock in a private scratch directory and the
    finished library is renamed into place; ranks that lose the race find it there."""
    lib = plugin_path(model_digest)
    if os.path.exists(lib):
        return lib
    with _locked(PLUGIN_DIR):
        fp = _source_fingerprint()
        _remove_outdated_plugins(fp)
        if os.path.exists(lib):               # another process built it while we waited for the lock
            return lib
        if not os.path.exists(HIPCC):
            raise RuntimeError(f"{lib} is not built and hipcc ({HIPCC}) is not available to build it")
        stem = os.path.basename(lib)[3:-3]
        scratch = tempfile.mkdtemp(prefix=f".build_{os.getpid()}_", dir=PLUGIN_DIR)
        try:
            header = os.path.join(scratch, stem + ".h")
            with open(header, "w") as f:
                f.write(header_text)
            stamp = os.path.join(scratch, "obe_fingerprint.h")
            with open(stamp, "w") as f:
                f.write(f'#define OBE_SOURCE_FINGERPRINT "{fp}"\n')
            define = f'-DOBE_PLUGIN_MODEL_HEADER="{header}"'

            def one(src):
                obj = os.path.join(scratch, src[:-4] + ".o")
                r = subprocess.run([HIPCC] + FLAGS + [define, "-include", stamp, "-c", os.path.join(CSRC, src),
                                                      "-o", obj], capture_output=True, text=True)
                if r.returncode != 0:
                    raise RuntimeError(f"hipcc failed for the generated model ({src}):\n{r.stdout}\n{r.stderr}")
                return obj

            prebuilt, to_compile = [], list(PLUGIN_SOURCES)
            for src in PLUGIN_COMMON_SOURCES:
                obj = os.path.join(OBJ_DIR, src[:-4] + ".o")
                if os.path.exists(obj) and not library_is_stale() and not _stale(obj, [os.path.join(CSRC, src)] + headers()):
                    prebuilt.append(obj)
                else:                        # (a library that was not built on this box / with these sources)
                    to_compile.append(src)
            with concurrent.futures.ThreadPoolExecutor(max_workers=len(to_compile)) as ex:
                objs = list(ex.map(one, to_compile)) + prebuilt
            # -Bsymbolic: the entry points call one another (obe_model_validate from every model-dependent
            # call, ...) and those calls must stay inside the plugin: libobe_hip.so, loaded RTLD_GLOBAL
            # before it, exports the same names.  (Everything that is not an entry point is hidden.)
            tmp = os.path.join(scratch, os.path.basename(lib))
            r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic",
                                f"-Wl,--version-script={_export_map(scratch)}"] + objs
                               + ["-o", tmp], capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"plugin link failed:\n{r.stdout}\n{r.stderr}")
            shutil.copyfile(header, os.path.join(PLUGIN_DIR, stem + ".h"))     # kept for inspection
            os.replace(tmp, lib)
        finally:
            shutil.rmtree(scratch, ignore_errors=True)
        if verbose:
            print(f"built {lib}", file=sys.stderr)
    return lib


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
