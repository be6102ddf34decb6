"""Build libnqa_hip.so in-tree with hipcc for gfx950 (no torch headers, plain C ABI)."""
from __future__ import annotations

import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libnqa_hip.so")
SOURCES = ["nqa_api.hip", "nqa_conv.hip", "nqa_conv_pool.hip", "nqa_conv1_pool.hip", "nqa_pool_stats.hip", "nqa_adists.hip", "nqa_prep.hip",
           "nqa_backward.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Rpass-analysis=kernel-resource-usage"]  # the remarks are parsed below: no hand-scheduled kernel may spill
RESOURCES = os.path.join(HERE, "kernel_resources.json")            # tracked: the report of the committed sources
RESOURCES_BUILD = os.path.join(HERE, "build", "kernel_resources.json")  # what the last build in this tree measured
# Kernels whose schedule is written against an exact register budget and counted vmcnt waits (scratch traffic shares
# the vmcnt counter with the LDS-DMA rings, so a spill makes every counted wait over-wait): a build in which one of
# these uses scratch FAILS.  Matched against the demangled-ish kernel name in the compiler's remark.
NO_SCRATCH = ("conv3x3_igemm_kernel", "conv3x3_regw_kernel", "conv3x3_regw128_kernel", "conv3x3_regw128_pool_kernel", "conv1_pool_kernel", "conv1_regw_kernel",
              "conv1_fused_kernel", "conv1_tile_kernel", "conv1_split_kernel", "conv1_regw_split_kernel", "conv3x3_regw_split_kernel", "pool_stats_kernel",
              "adists_window_lds_kernel", "adists_window_planar_kernel", "l2pool_kernel", "stats_nhwc_kernel")


def parse_resource_remarks(text: str) -> dict:
    """hipcc -Rpass-analysis=kernel-resource-usage -> {mangled kernel name: {"vgprs", "agprs", "sgprs", "scratch",
    "occupancy", "lds"}} (one block of remarks per kernel, `Function Name:` first)."""
    out, cur = {}, None
    keys = {"VGPRs": "vgprs", "AGPRs": "agprs", "SGPRs": "sgprs", "ScratchSize [bytes/lane]": "scratch",
            "Occupancy [waves/SIMD]": "occupancy", "LDS Size [bytes/block]": "lds"}
    for line in text.splitlines():
        m = re.search(r"remark: [^:]*:\d+:\d+: +(.*?) \[-Rpass-analysis", line) or \
            re.search(r"remark: +(.*?) \[-Rpass-analysis", line)
        if not m:
            continue
        body = m.group(1).strip()
        if body.startswith("Function Name:"):
            cur = out.setdefault(body.split(":", 1)[1].strip(), {})
        elif cur is not None and ":" in body:
            k, v = body.rsplit(":", 1)
            if k.strip() in keys:
                try:
                    cur[keys[k.strip()]] = int(v)
                except ValueError:
                    pass
    return out


# Analysed exceptions, bytes per lane: adists_window_lds_kernel<float, C >= 128> keeps one loop-invariant value of its
# channel-block loop in scratch (one store before the loop, one reload per channel block, none inside the row loop: ISA
# checked); the alternatives cost occupancy (nqa_adists.hip).  Anything beyond this fails the build.
SCRATCH_ALLOWED = {"adists_window_lds_kernel": 8}


def check_no_scratch(res: dict) -> list:
    bad = []
    for name, r in res.items():
        sc = r.get("scratch", 0)
        if sc and any(k in name for k in NO_SCRATCH):
            allowed = max([v for k, v in SCRATCH_ALLOWED.items() if k in name] or [0])
            if sc > allowed:
                bad.append(f"{name}: {sc} bytes/lane of scratch ({r.get('vgprs')} VGPRs; allowed {allowed})")
    return sorted(bad)

# nqa_adists.hip: the window kernels' tap arithmetic is written as scalar float FMAs with literal-constant weights
# (v_fmac_f32 with a 32-bit immediate); the SLP vectorizer would pair them into v_pk_*_f32, which issue at half
# rate on gfx950 and need a {w, w} register pair built per tap
# nqa_conv_pool.hip: the fused epilogue is written as slices of a few scalar float instructions per k-step; SLP would
# pair instructions of DIFFERENT slices (packed f32 ops, packed conversions) and drag them out of the MFMAs' shadow
FILE_FLAGS = {"nqa_adists.hip": ["-fno-slp-vectorize"], "nqa_conv_pool.hip": ["-fno-slp-vectorize"],
              "nqa_conv1_pool.hip": ["-fno-slp-vectorize"]}


def source_hash() -> str:
    """sha256 (16 hex digits) over the HIP sources, headers and compile flags: names the library build that the
    committed rocprofv3 summaries (profiles/r03_traffic.json) were measured on, independently of the machine."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    for path in files + [os.path.join(HERE, "..", "include", "nqa.h")]:
        h.update(os.path.basename(path).encode())
        h.update(open(path, "rb").read())
    h.update(" ".join(FLAGS + sorted(f"{k}:{' '.join(v)}" for k, v in FILE_FLAGS.items())).encode())
    return h.hexdigest()[:16]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "nqa.h"), __file__]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, extra_flags=(), out: str | None = None,
          emit_resources: bool = False) -> str:
    """Compile every HIP source and link the shared library; returns its path.

    The per-kernel register / LDS / scratch report of a product build goes to nerf_qa_amd/build/kernel_resources.json
    (git-ignored); the TRACKED nerf_qa_amd/kernel_resources.json is rewritten only with emit_resources=True
    (`python -m nerf_qa_amd.build --force --emit-resources`), so a plain rebuild leaves the tree clean;
    tests/test_cpu_lib.py compares the two when a fresh report exists.

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
    resources = {}
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"hipcc failed on {src}:\n" + "\n".join(
                ln for ln in out.splitlines() if "-Rpass-analysis" not in ln) + "\n")
            continue
        resources.update(parse_resource_remarks(out))
        rest = "\n".join(ln for ln in out.splitlines() if "-Rpass-analysis" not in ln)
        if verbose and rest.strip():
            print(rest)
    if failed:
        raise RuntimeError("libnqa_hip.so: compilation failed")
    spills = check_no_scratch(resources)
    if spills and not extra_flags:  # (timing-only ablation builds may do what they like)
        raise RuntimeError("libnqa_hip.so: hand-scheduled kernels spill to scratch:\n  " + "\n  ".join(spills))
    if not variant:
        os.makedirs(os.path.dirname(RESOURCES_BUILD), exist_ok=True)
        for path in [RESOURCES_BUILD] + ([RESOURCES] if emit_resources else []):
            with open(path, "w") as f:
                json.dump({k: resources[k] for k in sorted(resources)}, f, indent=0)
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
                out=os.path.join(HERE, outs[0]) if outs else None, emit_resources="--emit-resources" in sys.argv))
