"""Register / LDS / scratch figures of every kernel in an AMDGPU assembly file (hipcc --cuda-device-only -S): the .amdhsa metadata, one line
per kernel.  usage: kernel_regs.py file.s [name filter]"""
import re, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in txt.split("  - .agpr_count:")[1:]:
    blk = ".agpr_count:" + blk
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
    name = g("name")
    if flt in name:
        print(f"{name[:70]:70s} vgpr {g('vgpr_count'):>4s} agpr {g('agpr_count'):>4s} sgpr {g('sgpr_count'):>4s} spill v/s {g('vgpr_spill_count')}/{g('sgpr_spill_count')} lds {g('group_segment_fixed_size'):>6s} scratch {g('private_segment_fixed_size')}")
