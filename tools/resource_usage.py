"""Per-kernel register / scratch / occupancy table of the library's translation units (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/resource_usage.py [file.hip ...] [--spills]      # default: every csrc/*.hip

Used after every kernel edit: a kernel that starts spilling accumulators shows up here, not in a test."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "multimodal-baby_amd", "csrc")


def table(src):
    cmd = ["/opt/rocm/bin/hipcc", "-x", "hip", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
           "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + os.environ.get("CVCL_EXTRA_FLAGS", "").split()
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = subprocess.run(["/usr/bin/c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            cur = {"name": re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0]}
            rows.append(cur)
            continue
        for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    return rows


if __name__ == "__main__":
    files = [a for a in sys.argv[1:] if not a.startswith("--")] or sorted(
        os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    only_spills = "--spills" in sys.argv
    for f in files:
        for r in table(f):
            if only_spills and not r.get("scratch"):
                continue
            print(f"{os.path.basename(f):18s} {r['name'][:70]:70s} vgpr {r.get('vgpr', 0):3d} agpr {r.get('agpr', 0):3d} "
                  f"scratch {r.get('scratch', 0):4d} occ {r.get('occ', 0)}")
