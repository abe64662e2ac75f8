#!/usr/bin/env python3
"""List every gfx950 kernel of the built library with its register count and scratch bytes (from the code objects' metadata):
tools/scratch_check.py [lib.so].  A kernel with scratch > 0 spills; the hot kernels must show 0."""
import os, re, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin/"
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "honk2_amd", "libkws_hip.so")
tmp = tempfile.mkdtemp()
fat = os.path.join(tmp, "fat.bin")
subprocess.run([LLVM + "llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
data = open(fat, "rb").read()
# the section is a concatenation of clang offload bundles; each embeds ELF code objects: carve them out by their ELF headers
rows = []
pos = 0
idx = 0
while True:
    pos = data.find(b"\x7fELF", pos)
    if pos < 0:
        break
    # e_shoff + e_shnum * e_shentsize bounds the object
    import struct
    e_shoff = struct.unpack_from("<Q", data, pos + 0x28)[0]
    e_shentsize, e_shnum = struct.unpack_from("<HH", data, pos + 0x3A)
    end = pos + e_shoff + e_shentsize * e_shnum
    co = os.path.join(tmp, f"co{idx}.elf")
    open(co, "wb").write(data[pos:end])
    md = subprocess.run([LLVM + "llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    for blk in md.split("- .agpr_count")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        scr = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk).group(1))
        vg = int(re.search(r"\.vgpr_count:\s+(\d+)", blk).group(1))
        rows.append((name, vg, scr))
    idx += 1
    pos = end
dem = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
for (name, vg, scr), d in zip(rows, dem):
    print(f"{vg:4d} vgpr {scr:5d} scratch  {d.split('(')[0][:110]}")
print(len(rows), "kernels;", sum(1 for r in rows if r[2] > 0), "with scratch")
