"""Build libobe_hip from the tree's sources with extra hipcc flags into tools/_variants/<name>.so
(developer aid for same-box A/B measurements; the product library is untouched).
    python tools/build_variant.py w4 -DOBE_SWEEP_WAVES_PER_EU=4
Load it with OBE_VARIANT=w4 in tools/measure_sweep_launch.py."""
import concurrent.futures, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from optbayesexpt_amd import build as b

name, extra = sys.argv[1], sys.argv[2:]
out = os.path.join(ROOT, "tools", "_variants")
obj = os.path.join(out, "obj_" + name)
os.makedirs(obj, exist_ok=True)
stamp = os.path.join(obj, "obe_fingerprint.h")
# the extra flags are part of the stamp: a variant can never pass for the product library
# (the loader compares obe_source_fingerprint() with the hash of the sources alone)
flag_tag = "+" + "".join(c if c.isalnum() or c in "-_=" else "_" for c in " ".join(extra)) if extra else ""
open(stamp, "w").write(f'#define OBE_SOURCE_FINGERPRINT "{b._source_fingerprint()}{flag_tag}"\n')


def one(src):
    o = os.path.join(obj, os.path.basename(src)[:-4] + ".o")
    r = subprocess.run([b.HIPCC] + b.FLAGS + extra + ["-include", stamp, "-c", src, "-o", o], capture_output=True, text=True)
    if r.returncode:
        raise SystemExit(r.stderr)
    return o


with concurrent.futures.ThreadPoolExecutor(4) as ex:
    objs = list(ex.map(one, b.sources()))
lib = os.path.join(out, f"libobe_hip_{name}.so")
r = subprocess.run([b.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib], capture_output=True, text=True)
if r.returncode:
    raise SystemExit(r.stderr)
print(lib)
