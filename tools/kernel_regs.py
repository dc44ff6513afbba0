#!/usr/bin/env python3
"""VGPR / SGPR / scratch of every kernel of one HIP source (compiles it to gfx950 assembly): tools/kernel_regs.py gemm.hip [filter]"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "regt-gcn_amd", "csrc", sys.argv[1])
out = "/tmp/_kregs.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out, src],
                      stderr=subprocess.DEVNULL)
s = open(out).read()
filt = sys.argv[2] if len(sys.argv) > 2 else ""
blocks = s.split("  - .agpr_count:")[1:]
for b in blocks:
    g = lambda k: re.search(r"\." + k + r":\s+(\S+)", b)
    name = g("name").group(1)
    try:
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
    except OSError:
        pass
    if filt and filt not in name:
        continue
    print(f"vgpr {g('vgpr_count').group(1):>4} agpr {b.split()[0]:>3} sgpr {g('sgpr_count').group(1):>3} scratch {g('private_segment_fixed_size').group(1):>5} "
          f"lds {g('group_segment_fixed_size').group(1):>6}  {name[:120]}")
