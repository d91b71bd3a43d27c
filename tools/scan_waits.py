"""ISA audit (round 5; CPU only -- hipcc cross-compiles): for every kernel of the given .hip files, the innermost loops that hold MFMAs or
global / LDS-DMA loads and, inside them, `s_waitcnt vmcnt(0)`, scratch accesses, barriers.  A drain or a scratch reload INSIDE a
pipelined loop is how this round's two silent slow-downs looked: the e4m3 8-wave kernel's MX kinds (two counters in scratch, reloaded
behind vmcnt(0) every other K stage) and bn_relu_maxpool (nine tap loads, each sunk next to its use: nine dependent round trips).
    python tools/scan_waits.py [multimodal-baby_amd/csrc/gemm_fp8.hip ...]        (default: every .hip of the library; a .s listing is audited as it is)"""
import glob, os, re, subprocess, sys, tempfile
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "multimodal-baby_amd", "csrc", "*.hip")))
tmp = tempfile.mkdtemp()
asm = []
for f in files:
    if f.endswith(".s"):                  # an assembly listing somebody already made (-save-temps): audited as it is
        asm.append(f)
        continue
    out = os.path.join(tmp, os.path.basename(f) + ".s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out, f],
                   check=True, stderr=subprocess.DEVNULL)
    asm.append(out)
for path in asm:
    lines = open(path).read().split("\n")
    # split into functions
    funcs, cur, name = [], None, None
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+|\w+):\s*(;.*)?$", l)
        if m and not l.startswith(".") and (i + 1 < len(lines)):
            if cur is not None: funcs.append((name, cur))
            name, cur = m.group(1), []
        elif cur is not None:
            cur.append(l)
    if cur is not None: funcs.append((name, cur))
    for name, body in funcs:
        if not any("v_mfma" in l or "global_load" in l for l in body): continue
        labels = {}
        for i, l in enumerate(body):
            m = re.match(r"^(\.LBB\d+_\d+):", l)
            if m: labels[m.group(1)] = i
        loops = []
        for i, l in enumerate(body):
            m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", l)
            if m:
                t = m.group(1) or m.group(2)
                if t in labels and labels[t] < i: loops.append((labels[t], i))
        # innermost loops only
        inner = [lp for lp in loops if not any(o != lp and lp[0] <= o[0] and o[1] <= lp[1] for o in loops)]
        rep = []
        for a, b in inner:
            seg = body[a:b + 1]
            mf = sum("v_mfma" in l for l in seg)
            gl = sum(("global_load" in l or "buffer_load" in l) for l in seg)
            if mf == 0 and gl == 0: continue
            w0 = sum(bool(re.search(r"s_waitcnt.*vmcnt\(0\)", l)) for l in seg)
            sc = sum("scratch_" in l for l in seg)
            bar = sum("s_barrier" in l for l in seg)
            if w0 or sc:
                rep.append(f"    loop@{a}-{b} ({b - a} lines): mfma {mf} gload {gl} barrier {bar} vmcnt(0) {w0} scratch {sc}")
        if rep:
            import subprocess
            dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()[:110]
            print(dn); print("\n".join(rep))
