"""Cold build time of a per-model plugin library, per source and in total.

    python tools/time_plugin_build.py            (on the box whose time matters: the GPU box has ~128 cores)

Compiles the kernel sources of the Lorentzian expression model one at a time (what each costs), then builds the
whole plugin the way models.from_expression does on first use (sources in parallel, link) into a scratch
directory, so that nothing cached is reused; finally, when libhiprtc is present, the same model through the
run-time compiler (boxes without hipcc)."""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from optbayesexpt_amd import _exprmodel, build        # noqa: E402


def main():
    header, _, digest = _exprmodel.translate("b + a / (((x - x0) / d)**2 + 1)", ("x",), ("x0", "a", "b"), ("d",))
    with tempfile.TemporaryDirectory() as tmp:
        hpath = os.path.join(tmp, "model.h")
        open(hpath, "w").write(header)
        stamp = os.path.join(tmp, "fp.h")
        open(stamp, "w").write('#define OBE_SOURCE_FINGERPRINT "timing"\n')
        print(f"host cpus: {os.cpu_count()}; hipcc: {build.HIPCC}")
        total = 0.0
        for src in build.PLUGIN_SOURCES:
            t0 = time.perf_counter()
            r = subprocess.run([build.HIPCC] + build.FLAGS + [f'-DOBE_PLUGIN_MODEL_HEADER="{hpath}"', "-include", stamp, "-c",
                                                              os.path.join(build.CSRC, src), "-o",
                                                              os.path.join(tmp, src + ".o")], capture_output=True, text=True)
            dt = time.perf_counter() - t0
            total += dt
            print(f"  {src:18s} {dt:6.2f} s" + ("" if r.returncode == 0 else "   FAILED: " + r.stderr[-300:]))
        print(f"  sum of the four, one after the other: {total:.2f} s")
        os.environ["OBE_PLUGIN_DIR"] = os.path.join(tmp, "plugins")
        build.PLUGIN_DIR = os.environ["OBE_PLUGIN_DIR"]
        t0 = time.perf_counter()
        lib = build.build_plugin(header, digest)
        print(f"  build_plugin (cold, as from_expression runs it: sources in parallel + link): {time.perf_counter() - t0:.2f} s "
              f"-> {os.path.getsize(lib) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
