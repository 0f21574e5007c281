#!/usr/bin/env python3
"""Static instruction mix of one kernel in the gfx950 ISA of pla_kernels.hip (no GPU needed).

usage: python tools/isa_stats.py [kernel-name-substring] [extra hipcc flags...]
Prints VALU / SALU / LDS / VMEM counts, SGPR-spill traffic (v_writelane / v_readlane), scratch
traffic, and the register budget the compiler reports.  If the source carries
`asm volatile("; PLA_PHASE n")` markers the counts are also split by phase.
"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def compile_isa(extra=(), out="/tmp/pla_isa.s"):
    """One device-only compile of pla_kernels.hip to gfx950 assembly; returns the lines."""
    cmd = ["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-mllvm", "-disable-machine-licm",
           "--offload-device-only", "-S", "-o", out, os.path.join(ROOT, "pyloo_amd/csrc/pla_kernels.hip")] + list(extra)
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return open(out).read().split("\n")


def kernel_stats(lines, pat):
    """Instruction mix, per-phase mix and resource lines of the first kernel whose mangled name contains `pat`."""
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(pat) + r"\w*:", l))
    # (the end of the function, not its first s_endpgm: a kernel may leave early and go on for a thousand lines)
    end = next(i for i in range(start, len(lines)) if re.match(r"^\.Lfunc_end\d+:", lines[i]))
    total = collections.Counter()
    phases = collections.defaultdict(collections.Counter)
    ph = 0
    for l in lines[start:end]:
        l = l.strip()
        m = re.match(r"; PLA_PHASE (\d+)", l)
        if m:
            ph = int(m.group(1))
            continue
        if not l or l[0] in ";." or l.endswith(":"):
            continue
        op = l.split()[0]
        kind = "VALU" if op.startswith("v_") else "SALU" if op.startswith("s_") else "LDS" if op.startswith("ds_") else "VMEM"
        for c in (total, phases[ph]):
            c[kind] += 1
            if op in ("v_writelane_b32", "v_readlane_b32", "s_nop", "s_waitcnt") or op.startswith("scratch_"):
                c[op] += 1
            if op.startswith("s_cbranch") or op == "s_branch":
                c["branch"] += 1
    res = {}
    for l in lines[end:end + 600]:  # the resource comments follow the kernel descriptor
        m = re.search(r"; (NumVgprs|NumAgprs|ScratchSize|Occupancy|LDSByteSize|NumSgprs): (\d+)", l)
        if m and m.group(1) not in res:
            res[m.group(1)] = int(m.group(2))
        if len(res) == 6:
            break
    return lines[start].split(":")[0], total, phases, res


def main():
    pat = sys.argv[1] if len(sys.argv) > 1 else "wave_loo_kernelIdLi2"
    lines = compile_isa(sys.argv[2:])
    name, total, phases, res = kernel_stats(lines, pat)
    print(name)
    print(" total", dict(total))
    if len(phases) > 1:
        for k in sorted(phases):
            print(f"  phase {k:2d}", dict(phases[k]))
    for k, v in res.items():
        print(f"  ; {k}: {v}")


if __name__ == "__main__":
    main()
