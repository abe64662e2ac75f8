"""Instruction mix per basic block of one kernel in a hipcc --save-temps .s file.
usage: isa_mix.py file.s kernel_substring"""
import re, sys, collections
path, kern = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*%s\S*:" % kern, l))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")): return "vmem"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_"): return "salu"
    return "other"
blocks = []; cur = ["entry", collections.Counter(), start]
for i in range(start + 1, end + 1):
    l = lines[i].strip()
    if not l or l.startswith((";", ".")) and not l.startswith(".LBB"): continue
    m = re.match(r"^(\.LBB\S+):", l)
    if m:
        blocks.append(cur); cur = [m.group(1), collections.Counter(), i]; continue
    op = l.split()[0]
    cur[1][cls(op)] += 1
    if op.startswith(("s_cbranch", "s_branch")): cur[1]["->" + l.split()[1]] += 0
blocks.append(cur)
tot = collections.Counter()
for name, c, ln in blocks:
    n = sum(v for k, v in c.items() if not k.startswith("->"))
    if n >= int(sys.argv[3]) if len(sys.argv) > 3 else 20:
        print(f"{name:12s} line {ln:6d} n={n:5d} " + " ".join(f"{k}={v}" for k, v in sorted(c.items()) if not k.startswith("->")),
              " ".join(k for k in c if k.startswith("->")))
    tot.update({k: v for k, v in c.items() if not k.startswith("->")})
print("total", dict(tot))
