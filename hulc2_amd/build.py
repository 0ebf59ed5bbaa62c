"""Build libhulc2_amd.so (gfx950 HIP kernels + C ABI) in-tree with hipcc.

`python -m hulc2_amd.build` cross-compiles without a GPU.  The .so lands next to this file so it
travels with the repo snapshot to the GPU box; nothing is JIT-compiled at run time.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
OBJ = CSRC / "_obj"
LIB = HERE / "libhulc2_amd.so"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wno-unused-result"]
FLAGS += os.environ.get("HULC_BUILD_FLAGS", "").split()      # (A/B builds of compile-time switches, e.g. -DHULC_NT_FRAMES=0; part of the object digest)


def _digest(src: Path) -> str:
    h = hashlib.sha1()
    h.update(" ".join(FLAGS).encode())
    for dep in [src] + sorted(CSRC.glob("*.h")) + [HERE.parent / "include" / "hulc2_amd.h"]:
        h.update(dep.read_bytes())
    return h.hexdigest()


def _compile(src: Path) -> Path:
    obj = OBJ / (src.stem + ".o")
    stamp = OBJ / (src.stem + ".sha1")
    dig = _digest(src)
    if obj.exists() and stamp.exists() and stamp.read_text() == dig:
        return obj
    cmd = [HIPCC, *FLAGS, "-c", str(src), "-o", str(obj)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stdout}\n{r.stderr}")
    stamp.write_text(dig)
    return obj


def build(verbose: bool = True) -> Path:
    OBJ.mkdir(exist_ok=True)
    srcs = sorted(CSRC.glob("*.hip"))
    if not srcs:
        raise RuntimeError("no HIP sources found")
    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        objs = list(ex.map(_compile, srcs))
    newest = max(o.stat().st_mtime for o in objs)
    if not LIB.exists() or LIB.stat().st_mtime < newest:
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB), *map(str, objs)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"[hulc2_amd.build] {LIB} ({LIB.stat().st_size >> 10} KiB, {len(objs)} objects)")
    return LIB


if __name__ == "__main__":
    build()
    sys.exit(0)
