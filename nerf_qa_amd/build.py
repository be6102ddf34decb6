"""Build libnqa_hip.so in-tree with hipcc for gfx950 (no torch headers, plain C ABI)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libnqa_hip.so")
SOURCES = ["nqa_api.hip", "nqa_conv.hip", "nqa_pool_stats.hip", "nqa_adists.hip", "nqa_prep.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# nqa_adists.hip: the window kernels' tap arithmetic is written as scalar float FMAs with literal-constant weights
# (v_fmac_f32 with a 32-bit immediate); the SLP vectorizer would pair them into v_pk_*_f32, which issue at half
# rate on gfx950 and need a {w, w} register pair built per tap
FILE_FLAGS = {"nqa_adists.hip": ["-fno-slp-vectorize"]}


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "nqa.h"), __file__]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, extra_flags=(), out: str | None = None) -> str:
    """Compile every HIP source and link the shared library; returns its path.

    extra_flags/out build a development variant (e.g. -DNQA_ABLATE_NO_DMA for timing-only
    ablations) next to the product library; NQA_LIB selects it at load time.
    """
    lib = out or LIB
    variant = out is not None
    if not force and not extra_flags and not _stale():
        return LIB
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".o" if not out else "." + os.path.basename(out) + ".o"))
        cmd = [HIPCC, *FLAGS, *FILE_FLAGS.get(src, []), *extra_flags, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"hipcc failed on {src}:\n{out}\n")
        elif verbose and out.strip():
            print(out)
    if failed:
        raise RuntimeError("libnqa_hip.so: compilation failed")
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", lib]
    subprocess.run(cmd, check=True)
    if variant:  # a development variant's objects are not reused: do not let them pile up
        for o in objs:
            os.remove(o)
    return lib


if __name__ == "__main__":
    flags = [a for a in sys.argv[1:] if a.startswith("-D")]
    outs = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--out=")]
    print(build(force="--force" in sys.argv, verbose=True, extra_flags=flags,
                out=os.path.join(HERE, outs[0]) if outs else None))
