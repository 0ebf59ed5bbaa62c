"""Build a probe from its .hip next to this file when the binary is missing or older than the source (hipcc --offload-arch=gfx950).
The binaries are not committed: `ensure("tr_probe.so")` / `ensure("overlap_probe", shared=False)` return the path to load / run."""
import os
import subprocess
from pathlib import Path

HERE = Path(__file__).resolve().parent
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def ensure(name: str, shared: bool = True) -> str:
    out = HERE / name
    src = HERE / (Path(name).stem + ".hip")
    if not out.exists() or out.stat().st_mtime < src.stat().st_mtime:
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", str(src), "-o", str(out)] + (["-shared", "-fPIC"] if shared else [])
        subprocess.run(cmd, check=True)
    return os.fspath(out)


if __name__ == "__main__":          # python tools/probe/_build.py overlap_probe overlap_probe2  -> builds the executables
    import sys
    for n in sys.argv[1:]:
        print(ensure(n, shared=n.endswith(".so")))
