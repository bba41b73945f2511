#!/usr/bin/env python3
"""Compiles pt_kernels.hip for gfx950 (device code only, same flags as the library) with
-Rpass-analysis=kernel-resource-usage and prints one line per kernel: registers, spills, private
segment, LDS, occupancy.  `check()` is what the build gate calls: it raises if any megakernel
variant spills a register or uses a private segment."""
from __future__ import annotations

import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "path-tracer_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

# template instantiations the library launches (launch_items: 2 x 2 x 5; launch_pool: 2 x 5)
EXPECTED_TRACE_VARIANTS = 20
EXPECTED_POOL_VARIANTS = 10

FIELDS = {
    "TotalSGPRs": "sgpr", "VGPRs": "vgpr", "AGPRs": "agpr", "ScratchSize [bytes/lane]": "scratch",
    "Occupancy [waves/SIMD]": "occupancy", "SGPRs Spill": "sgpr_spill", "VGPRs Spill": "vgpr_spill",
    "LDS Size [bytes/block]": "lds",
}


def device_flags() -> list[str]:
    """The FLAGS line of the library's Makefile, minus what only matters when linking."""
    flags = subprocess.check_output(["make", "-s", "-C", CSRC, "print-flags"], text=True).split()
    return [f for f in flags if f not in ("-fPIC", "-shared")]


def demangle(names: list[str]) -> list[str]:
    try:
        out = subprocess.check_output(["c++filt"], input="\n".join(names), text=True)
        return out.splitlines()
    except Exception:
        return names


def usage(extra: list[str] | None = None, out_asm: str | None = None) -> list[dict]:
    """out_asm: where to keep the assembly (default: a temporary file of this call's own, deleted afterwards)."""
    import tempfile
    tmp = None
    if out_asm is None:
        fd, tmp = tempfile.mkstemp(prefix="pt_kernels_usage_", suffix=".s")
        os.close(fd)
        out_asm = tmp
    cmd = [HIPCC] + device_flags() + ["-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage",
                                      os.path.join(CSRC, "pt_kernels.hip"), "-o", out_asm] + (extra or [])
    try:
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    finally:
        if tmp is not None and os.path.exists(tmp):
            os.unlink(tmp)
    if p.returncode != 0:
        sys.stderr.write(p.stderr)
        raise RuntimeError("hipcc failed")
    kernels, cur = [], None
    for line in p.stderr.splitlines():
        m = re.search(r"remark: (?:\S+: )?\s*Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            kernels.append(cur)
            continue
        m = re.search(r"remark: (?:\S+: )?\s*([A-Za-z][^:]*): (\d+)", line)
        if m and cur is not None and m.group(1).strip() in FIELDS:
            cur[FIELDS[m.group(1).strip()]] = int(m.group(2))
    for k, d in zip(kernels, demangle([k["name"] for k in kernels])):
        d = re.sub(r"\(anonymous namespace\)::", "", d)
        d = re.sub(r"^void ", "", d)
        k["pretty"] = re.sub(r">\(.*$", ">", d) if "<" in d else d.split("(")[0]
        k["pretty"] = k["pretty"].replace("true", "1").replace("false", "0").replace("(bool)", "")
    return kernels


def check(kernels: list[dict]) -> None:
    """The gate: the megakernels (trace_kernel, pool_kernel) keep everything in registers.  It must not pass by having
    seen nothing: every variant the library launches has to be in the compiler's report, each with its spill and
    private-segment fields."""
    mega = [k for k in kernels if re.search(r"(trace|pool)_kernel", k["pretty"])]
    n_trace = sum("trace_kernel" in k["pretty"] for k in mega)
    n_pool = sum("pool_kernel" in k["pretty"] for k in mega)
    if n_trace < EXPECTED_TRACE_VARIANTS or n_pool < EXPECTED_POOL_VARIANTS:
        raise RuntimeError(f"the compiler's resource report lists {n_trace} trace_kernel and {n_pool} pool_kernel variants, expected at least "
                           f"{EXPECTED_TRACE_VARIANTS} and {EXPECTED_POOL_VARIANTS}: has the remark format changed?")
    for k in mega:
        missing = [f for f in ("sgpr_spill", "vgpr_spill", "scratch", "vgpr") if f not in k]
        if missing:
            raise RuntimeError(f"{k['pretty']}: the resource report lacks {missing}: has the remark format changed?")
    bad = [k for k in kernels if re.search(r"(trace|pool)_kernel", k["pretty"])
           and (k.get("sgpr_spill", 0) or k.get("vgpr_spill", 0) or k.get("scratch", 0))]
    if bad:
        lines = "\n".join(f"  {k['pretty']}: {k.get('sgpr_spill', 0)} SGPRs and {k.get('vgpr_spill', 0)} VGPRs spilled, "
                          f"{k.get('scratch', 0)} B of private segment per lane" for k in bad)
        raise RuntimeError("megakernel variants that spill registers (a spill at the register limit is where the "
                           "compiler's lane-spill code has failed before):\n" + lines)


def table(ks: list[dict]) -> str:
    lines = [f"{'kernel':<58}{'sgpr':>5}{'vgpr':>5}{'agpr':>5}{'sspill':>7}{'vspill':>7}{'scratch':>8}{'lds':>7}{'occ':>4}"]
    for k in ks:
        name = k["pretty"] if len(k["pretty"]) < 58 else k["pretty"].split("(")[0]
        lines.append(f"{name:<58}{k.get('sgpr', 0):>5}{k.get('vgpr', 0):>5}{k.get('agpr', 0):>5}{k.get('sgpr_spill', 0):>7}"
                     f"{k.get('vgpr_spill', 0):>7}{k.get('scratch', 0):>8}{k.get('lds', 0):>7}{k.get('occupancy', 0):>4}")
    return "\n".join(lines) + "\n"


def write_record(ks: list[dict], path: str) -> None:
    """The table of the build at hand, kept under profiles/ (the build step rewrites it, so it cannot go stale)."""
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        f.write("# registers, spills, private segment, LDS and occupancy of every kernel of pt_kernels.hip as hipcc reports them for\n"
                "# gfx950 with the library's flags (tools/resource_usage.py; rewritten by __graft_entry__.build())\n")
        f.write(table(ks))


def main():
    ks = usage(sys.argv[1:])
    sys.stdout.write(table(ks))
    try:
        check(ks)
    except RuntimeError as e:
        print(e)
        sys.exit(1)


if __name__ == "__main__":
    main()
