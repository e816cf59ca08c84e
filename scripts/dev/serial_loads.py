"""Which kernels wait for their loads one by one?  Compiles every translation unit of libmi_icp to gfx950 assembly and
counts, per kernel, the global loads that are followed at once by `s_waitcnt vmcnt(0)` -- what a load under a condition
(`if (i < n) x = p[i]`) or behind a possibly-aliasing store compiles to.  CPU only (hipcc cross-compiles)."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
units = sys.argv[1:] or ["mi_icp", "mi_build", "mi_geometry", "mi_knn"]
seen = set()
for u in units:
    out = os.path.join(tempfile.gettempdir(), u + ".s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I/opt/rocm/include", "-S",
                    "--cuda-device-only", os.path.join(ROOT, "cupoch_amd", "csrc", u + ".hip"), "-o", out], check=True, stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
    name, stats = None, {}
    for i, l in enumerate(lines):
        m = re.match(r'^(_Z\S+):\s+; @', l)
        if m:
            name = m.group(1)
            stats[name] = [0, 0]
            continue
        if name is None:
            continue
        if l.startswith(".Lfunc_end"):
            name = None
            continue
        if re.search(r'\b(global|buffer|flat)_load', l):
            stats[name][0] += 1
            nxt = [x for x in lines[i + 1:i + 4] if x.strip() and not x.strip().startswith(';')][:2]
            if any('s_waitcnt vmcnt(0)' in x for x in nxt):
                stats[name][1] += 1
    for k, (a, b) in sorted(stats.items(), key=lambda kv: -kv[1][1]):
        if b >= 4 and k not in seen:
            seen.add(k)
            d = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
            print("%-12s %-90s loads %4d  waited for at once %3d" % (u, d[:90], a, b))
