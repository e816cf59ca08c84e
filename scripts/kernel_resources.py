#!/usr/bin/env python3
"""Register / scratch / LDS usage of the kernels in libmi_icp.so, read from the gfx950 code object's
metadata notes (no GPU needed).  usage: scripts/kernel_resources.py [name-substring ...]"""
import os, re, struct, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.environ.get("LIB", os.path.join(ROOT, "cupoch_amd", "lib", "libmi_icp.so"))
blob = open(lib, "rb").read()
magic = b"__CLANG_OFFLOAD_BUNDLE__"
dem = lambda s: subprocess.run(["c++filt", s], capture_output=True, text=True).stdout.strip()
pats = sys.argv[1:]
at = blob.find(magic)
assert at >= 0, "no offload bundle in " + lib
seen = set()
while at >= 0:      # (one bundle per translation unit of the library: csrc/ctx.h)
    n = struct.unpack_from("<Q", blob, at + 24)[0]
    pos = at + 32
    co = None
    for _ in range(n):
        off, size, tl = struct.unpack_from("<QQQ", blob, pos)
        triple = blob[pos + 24:pos + 24 + tl].decode()
        pos += 24 + tl
        if "gfx950" in triple:
            co = blob[at + off:at + off + size]
    at = blob.find(magic, at + 1)
    if not co:
        continue
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(co); f.flush()
        txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout
    for blk in re.split(r"\n  - \.agpr_count:", txt)[1:]:
        g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "?"])[1]
        name = dem(g("name"))
        short = re.sub(r"\(.*", "", name).replace("mi::", "")
        if pats and not any(p in short for p in pats):
            continue
        line = "%-60s vgpr %3s agpr %3s sgpr %3s scratch %5s lds %6s" % (short[:60], g("vgpr_count"), blk.split()[0], g("sgpr_count"),
                                                                  g("private_segment_fixed_size"), g("group_segment_fixed_size"))
        if line not in seen:      # (a static kernel included by several units appears once per unit)
            seen.add(line)
            print(line)
