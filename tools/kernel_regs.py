#!/usr/bin/env python3
"""Register / spill / LDS metadata of the kernels in a device assembly listing.
   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on -S --cuda-device-only -o /tmp/x.s openslam_g2o_amd/csrc/<file>.hip
   python tools/kernel_regs.py /tmp/x.s [regex]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
md = txt[txt.index("amdhsa.kernels:"):]
for blk in md.split("  - .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\.%s:\s*(\S+)" % k, blk) or [None, "?"])[1]
    name = g("name")
    try:
        dn = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
    except Exception:
        dn = name
    dn = dn.replace("g2ohip::(anonymous namespace)::", "").replace("void ", "")
    dn = re.sub(r"\(.*", "", dn)
    if pat and not pat.search(dn):
        continue
    print("%-60s vgpr %3s agpr %3s sgpr %3s spill %3s scratch %4s lds %6s" % (dn[:60], g("vgpr_count"), blk.split()[0], g("sgpr_count"), g("vgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
