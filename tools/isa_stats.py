#!/usr/bin/env python3
"""Static instruction mix per kernel of the HIP sources, from the compiler's assembly.

  python tools/isa_stats.py [--check] [extra hipcc flags...]

--check: exit with status 1 if a kernel that walks the tree as a wave packet (k_extend<3, *>, k_shadow<3, *>,
k_query<3, *>) uses scratch memory.  Their traversal stack lives in the lanes of a vector register
(v_writelane / v_readlane); a spill of that register under a partial EXEC mask would silently lose entries,
so those kernels must compile without any spill.  The build runs this.

Compiles raytracer_amd/csrc/rt_kernels.hip for gfx950 to assembly (device only) and
prints, per kernel: VGPRs, SGPRs, scratch bytes, occupancy, and static counts of vector ALU,
vector moves, scalar ALU, branches, scalar loads, global loads / stores."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "raytracer_amd", "csrc")


def main():
    check = "--check" in sys.argv
    if check:
        sys.argv.remove("--check")
    out = os.path.join(tempfile.gettempdir(), "rt_kernels.s")
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
           "-fno-slp-vectorize", "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S", "-o", out,
           os.path.join(SRC, "rt_kernels.hip")] + sys.argv[1:]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    name, rows, cur = None, [], None
    for line in open(out):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name = m.group(1)
            cur = dict(valu=0, vmov=0, salu=0, branch=0, smem=0, gload=0, gstore=0, scratch_ops=0)
            continue
        if cur is None:
            continue
        t = line.strip()
        if t.startswith("v_mov") or t.startswith("v_pk_mov"):
            cur["vmov"] += 1; cur["valu"] += 1
        elif t.startswith("v_"):
            cur["valu"] += 1
        elif t.startswith("s_cbranch") or t.startswith("s_branch"):
            cur["branch"] += 1
        elif t.startswith("s_load") or t.startswith("s_buffer_load"):
            cur["smem"] += 1
        elif t.startswith("s_") and not t.startswith(("s_waitcnt", "s_nop", "s_endpgm")):
            cur["salu"] += 1
        elif t.startswith("global_load") or t.startswith("flat_load"):
            cur["gload"] += 1
        elif t.startswith("global_store") or t.startswith("flat_store") or t.startswith("global_atomic"):
            cur["gstore"] += 1
        elif t.startswith("scratch_"):
            cur["scratch_ops"] += 1
        for key, pat in (("vgpr", r"; NumVgprs: (\d+)"), ("sgpr", r"; TotalNumSgprs: (\d+)"),
                         ("scratch", r"; ScratchSize: (\d+)"), ("occ", r"; Occupancy: (\d+)")):
            m = re.match(pat, t)
            if m:
                cur[key] = int(m.group(1))
                if key == "occ":
                    rows.append((name, cur)); cur = None
                break
    print(f"{'kernel':44s} vgpr sgpr scr occ | valu vmov salu  br smem gld gst scr_ops")
    bad = []
    for n, c in rows:
        short = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void ", "")
        if re.match(r"k_(extend|shadow|query)<3,", short) and (c.get("scratch", 0) or c["scratch_ops"]):
            bad.append(short)
        print(f"{short:44s} {c.get('vgpr', 0):4d} {c.get('sgpr', 0):4d} {c.get('scratch', 0):3d} {c.get('occ', 0):3d} | "
              f"{c['valu']:4d} {c['vmov']:4d} {c['salu']:4d} {c['branch']:3d} {c['smem']:4d} {c['gload']:3d} {c['gstore']:3d} {c['scratch_ops']:4d}")


    if check and bad:
        print("packet-walk kernels must not use scratch memory (lane-resident traversal stack): " + ", ".join(bad), file=sys.stderr)
        sys.exit(1)


if __name__ == "__main__":
    main()
