#!/usr/bin/env python3
"""Static instruction mix of one kernel in the gfx950 ISA of the kernel units pyloo_amd/csrc/pla_k_*.hip (no GPU needed).

usage: python tools/isa_stats.py [kernel-name-substring] [extra hipcc flags...]
       python tools/isa_stats.py --table      (every kernel with its registers / scratch / LDS, markdown)
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


KERNEL_UNITS = ["pla_k_general.hip", "pla_k_wave_f64.hip", "pla_k_wave_f32.hip", "pla_k_chunked_f64.hip", "pla_k_chunked_f32.hip",
                "pla_k_fit.hip", "pla_k_lwout.hip", "pla_k_waic.hip", "pla_k_col.hip", "pla_k_eloo.hip"]


def compile_isa(extra=(), out="/tmp/pla_isa.s", units=None):
    """Device-only compiles of the kernel units (pyloo_amd/csrc/pla_k_*.hip) to gfx950 assembly, in parallel; returns the
    lines of all of them, one after the other (also written to `out`).  Same flags as pyloo_amd/build.py."""
    import concurrent.futures

    units = list(units or KERNEL_UNITS)

    def one(u):
        dst = out + "." + u[:-4] + ".s"
        cmd = ["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-mllvm", "-disable-machine-licm",
               "--offload-device-only", "-S", "-o", dst, os.path.join(ROOT, "pyloo_amd/csrc", u)] + list(extra)
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        return open(dst).read()

    with concurrent.futures.ThreadPoolExecutor(max_workers=min(len(units), os.cpu_count() or 1)) as pool:
        text = "\n".join(pool.map(one, units))
    with open(out, "w") as f:
        f.write(text)
    return text.split("\n")


def unit_of(pat):
    """The unit that holds a kernel (saves compiling the others when one kernel is asked for)."""
    f32 = "If" in pat
    if pat.startswith("wave_loo_chunked"):
        return ["pla_k_chunked_f32.hip" if f32 else "pla_k_chunked_f64.hip"]
    if pat.startswith(("wave_loo_kernel", "is_wave")):
        return ["pla_k_wave_f32.hip" if f32 else "pla_k_wave_f64.hip"]
    if pat.startswith("fit_rows"):
        return ["pla_k_fit.hip"]
    if pat.startswith("lw_output"):
        return ["pla_k_lwout.hip"]
    if pat.startswith(("col_", "tile_")):
        return ["pla_k_col.hip"]
    if pat.startswith("e_loo"):
        return ["pla_k_eloo.hip"]
    if pat.startswith("waic"):
        return ["pla_k_waic.hip"]
    return None


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
        m = re.search(r"; (NumVgprs|NumAgprs|ScratchSize|Occupancy|LDSByteSize|TotalNumSgprs): (\d+)", l)
        if m and m.group(1) not in res:
            res[m.group(1)] = int(m.group(2))
        if len(res) == 6:
            break
    return lines[start].split(":")[0], total, phases, res


def masked_spills(lines, pat):
    """SGPR spills of one kernel that are WRITTEN inside an exec-masked region (between an s_*_saveexec_b64 into a scalar pair and
    the s_or_b64 exec that restores it).  A wave whose lanes all skip such a region -- the compiler jumps over it with
    s_cbranch_execz -- never executes the v_writelane, and the v_readlane behind the region then returns whatever the vector
    register held: the failure of round 4's streamed fit kernel.  Spills written in uniform control flow are harmless."""
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(pat) + r"\w*:", l))
    end = next(i for i in range(start, len(lines)) if re.match(r"^\.Lfunc_end\d+:", lines[i]))
    open_regions, bad = [], []
    for l in lines[start:end]:
        t = l.strip()
        m = re.match(r"s_(?:and|andn2|or|xor|nand|nor|xnor)_saveexec_b64 (s\[\d+:\d+\]|vcc)", t)
        if m:
            if m.group(1) not in open_regions:
                open_regions.append(m.group(1))
            continue
        m = re.match(r"s_or_b64 exec, exec, (s\[\d+:\d+\]|vcc)", t)
        if m and m.group(1) in open_regions:
            open_regions.remove(m.group(1))
            continue
        if t.startswith("v_writelane_b32") and open_regions:
            bad.append(t)
    return bad


def all_kernels(lines):
    names = [m.group(1) for l in lines for m in [re.match(r"^(_Z\w+):\s*(;.*)?$", l)] if m and "pla" in m.group(1)]
    return [n for n in names if any(l.strip() == ".amdhsa_kernel " + n for l in lines)]


def demangle(names):
    """c++filt over the mangled kernel names (the LLVM copy under /opt/rocm when binutils is absent)."""
    for tool in ("c++filt", "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"):
        try:
            out = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
            return dict(zip(names, out))
        except (OSError, subprocess.CalledProcessError):
            continue
    return {n: n for n in names}


def table(extra=()):
    """Every kernel of the library with the resources its code object declares (markdown): `python tools/isa_stats.py --table`."""
    lines = compile_isa(extra)
    names = [m.group(1) for l in lines for m in [re.match(r"^(_Z\w+):\s*(;.*)?$", l)] if m and "pla" in m.group(1)]
    kernels = []
    for n in names:  # (kernels only: device functions have no .amdhsa_kernel descriptor)
        if any(l.strip() == ".amdhsa_kernel " + n for l in lines):
            kernels.append(n)
    pretty = demangle(kernels)
    rows = []
    for n in kernels:
        _, total, _, res = kernel_stats(lines, n[2:])
        short = re.sub(r"\(.*", "", pretty[n]).replace("void ", "").replace("pla::", "")
        rows.append((short, res.get("NumVgprs", 0), res.get("NumAgprs", 0), res.get("TotalNumSgprs", 0), res.get("ScratchSize", 0),
                     res.get("LDSByteSize", 0), res.get("Occupancy", 0), total.get("v_writelane_b32", 0), total.get("VALU", 0)))
    print("| kernel | VGPRs | AGPRs | SGPRs | scratch B | static LDS B | waves/SIMD (registers) | v_writelane | VALU instructions (static) |")
    print("|---|---|---|---|---|---|---|---|---|")
    for r in sorted(rows):
        print("| `" + r[0] + "` | " + " | ".join(str(x) for x in r[1:]) + " |")


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--table":
        return table(sys.argv[2:])
    pat = sys.argv[1] if len(sys.argv) > 1 else "wave_loo_kernelIdLi2"
    lines = compile_isa(sys.argv[2:], units=unit_of(pat))
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
