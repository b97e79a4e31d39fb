"""Build lint for the prefill GEMM kernels (csrc/woq_gemm_f16.hip, woq_gemm_f16p.h): no kernel-argument load may sit
behind a kernel's K loop. hipcc once re-loaded `M` there and recycled the segment pointer's register right behind the
load; on cold process starts single waves then saw garbage and stored nothing. The epilogue arguments
are pinned in SGPRs since (WOQ_PIN_EPILOGUE_ARGS); this keeps it that way. Also reports VGPR counts / spills of the
ring kernels, which have to stay within 168 registers (three waves per SIMD) without spilling.
usage: python tools/check_kernarg_reloads.py [isa.s]   (compiles the file itself when no ISA is given)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "intel_extension_for_transformers_amd", "csrc")


def compile_isa(out):
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-mllvm",
           "-amdgpu-kernarg-preload-count=14", "-S", "--cuda-device-only", os.path.join(CSRC, "woq_gemm_f16.hip"), "-o", out]
    subprocess.run(cmd, check=True, cwd=CSRC, stderr=subprocess.DEVNULL)


def check(isa_path):
    text = open(isa_path).read()
    problems, kernels = [], 0
    for m in re.finditer(r"^(_ZN3woq(?:16gemm_f16[pst]|19gemm_f16frag)_kernel\w+):.*?\n(.*?)s_endpgm", text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        lines = [l.strip() for l in body.splitlines() if l.strip() and not l.strip().startswith(";")]
        loop_labels = [i for i, l in enumerate(lines) if re.match(r"\.LBB\d+_\d+:", l)]
        # the K loop = the largest backward branch span
        best = None
        for i, l in enumerate(lines):
            b = re.match(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
            if not b:
                continue
            tgt = [j for j in loop_labels if lines[j].startswith(b.group(1) + ":")]
            if tgt and tgt[0] < i and (best is None or i - tgt[0] > best[1] - best[0]):
                best = (tgt[0], i)
        if best is None:
            problems.append("%s: no loop found" % name)
            continue
        kernels += 1
        late = [l for l in lines[best[1]:] if l.startswith("s_load_")]
        if late:
            problems.append("%s: %d kernel-argument load(s) behind the K loop, e.g. '%s'" % (name, len(late), late[0]))
    ring = []
    for m in re.finditer(r"\.name:\s+(_ZN3woq16gemm_f16p_kernelILi\dELb\dELi\dELb\dELb1E\w+)", text):
        blk = text[m.start():m.start() + 1500]
        v = int(re.search(r"\.vgpr_count:\s+(\d+)", blk).group(1))
        sp = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", blk).group(1))
        # the launcher never picks the ring form for group-32 asymmetric blobs with fp32 scales (SMODE 1, ASYM, ST 2)
        excluded = re.search(r"kernelILi1ELb1ELi2E", m.group(1)) is not None
        ring.append((m.group(1), v, sp, excluded))
        if not excluded and (v > 168 or sp > 0):  # (gemm_f16p ring kernels)
            problems.append("%s: ring kernel at %d VGPRs, %d spilled (limit 168 / 0)" % (m.group(1), v, sp))
    # round 6: the 256-row kernels (woq_gemm_f16t.h) run two workgroups per CU: <= 256 registers, no spills
    for m in re.finditer(r"\.name:\s+(_ZN3woq16gemm_f16t_kernel\w+)", text):
        blk = text[m.start():m.start() + 1500]
        v = int(re.search(r"\.vgpr_count:\s+(\d+)", blk).group(1))
        ag = int((re.search(r"\.agpr_count:\s+(\d+)", blk) or re.search(r"(0)", "0")).group(1))
        sp = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", blk).group(1))
        ring.append((m.group(1), v + ag, sp, False))
        if v + ag > 256 or sp > 0:
            problems.append("%s: 256-row kernel at %d registers, %d spilled (limit 256 / 0)" % (m.group(1), v + ag, sp))
    return kernels, ring, problems


def main():
    if len(sys.argv) > 1 and sys.argv[1] != "--compile-to":
        isa = sys.argv[1]
    elif len(sys.argv) > 2:
        isa = sys.argv[2]
        compile_isa(isa)
    else:
        isa = os.path.join(tempfile.mkdtemp(prefix="woq_isa_"), "woq_gemm_f16.s")
        compile_isa(isa)
    kernels, ring, problems = check(isa)
    print("%d GEMM kernels checked, %d ring kernels (max %d VGPRs among the selectable ones)"
          % (kernels, len(ring), max([v for _, v, _, ex in ring if not ex] or [0])))
    for p in problems:
        print("PROBLEM:", p)
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
