"""Instruction-class counts per basic block of one kernel in a gfx950 assembly listing.
    hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off --cuda-device-only -S pseg_mfma.hip -o /tmp/m.s
    python tools/isa_blocks.py /tmp/m.s Li8ELi2ELi5ELi1ELi3ELi0ELi33E [min_block_size]"""
import collections, re, sys

txt = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
minb = int(sys.argv[3]) if len(sys.argv) > 3 else 16
start = next(i for i, l in enumerate(txt) if l.startswith("_ZN") and key in l and l.rstrip().split(":")[0].endswith("E") and ":" in l)
blocks = collections.OrderedDict()
cur = "entry"
blocks[cur] = []
for l in txt[start + 1:]:
    s = l.strip()
    if s.startswith("s_endpgm"):
        break
    if not s or s.startswith((";", ".")) and not re.match(r"\.LBB\d+_\d+:", s):
        continue
    m = re.match(r"(\.LBB\d+_\d+):", s)
    if m:
        cur = m.group(1)
        blocks[cur] = []
        continue
    blocks[cur].append(s.split()[0])


def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("ds_"): return "ds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith(("v_accvgpr", "v_mov")): return "vmov"
    if op.startswith("v_"): return "valu"
    if op.startswith(("s_waitcnt", "s_barrier", "s_nop")): return "wait"
    if op.startswith(("s_cbranch", "s_branch")): return "br"
    if op.startswith("s_load"): return "smem"
    if op.startswith("s_"): return "salu"
    return "other"


tot = collections.Counter()
for k, b in blocks.items():
    c = collections.Counter(cls(o) for o in b)
    tot.update(c)
    if len(b) >= minb:
        print("%-12s n=%4d  %s" % (k, len(b), "  ".join("%s %d" % kv for kv in sorted(c.items()))))
print("TOTAL", sum(tot.values()), dict(tot))
