"""Registers, scratch and static LDS of every kernel in an ISA dump (hipcc -S --cuda-device-only): python scripts/isa_regs.py file.s [filter]"""
import re, subprocess, sys
t = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for b in t.split('  - .agpr_count')[1:]:
    name = re.search(r'\.name:\s+(\S+)', b).group(1)
    v = re.search(r'\.vgpr_count:\s+(\d+)', b).group(1)
    sg = re.search(r'\.sgpr_count:\s+(\d+)', b).group(1)
    sp = re.search(r'\.private_segment_fixed_size:\s+(\d+)', b).group(1)
    lds = re.search(r'\.group_segment_fixed_size:\s+(\d+)', b).group(1)
    d = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    d = re.sub(r'\(.*', '', d)
    if flt in d:
        print(f"vgpr {v:>3} sgpr {sg:>3} scratch {sp:>4} lds {lds:>5}  {d}")
