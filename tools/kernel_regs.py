"""Register / LDS / scratch usage of every kernel in a saved-temps gfx950 assembly file (hipcc -save-temps=obj).
usage: python tools/kernel_regs.py file.s [filter]"""
import re
import subprocess
import sys

s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
meta = s[s.index("amdhsa.kernels:"):]
for blk in meta.split("  - .agpr_count:")[1:]:
    def g(k):
        m = re.search(r"\." + k + r":\s+(\S+)", blk)
        return m.group(1) if m else "?"
    name = g("name")
    try:
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
    except FileNotFoundError:
        pass
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    if flt in name:
        print(f"{name[:110]:110s} vgpr {g('vgpr_count'):>4s} spill {g('vgpr_spill_count'):>3s} sgpr {g('sgpr_count'):>4s} "
              f"lds {g('group_segment_fixed_size'):>6s} scratch {g('private_segment_fixed_size'):>5s}")
