#!/bin/bash
# L2 <-> fabric bytes of GEMM lab variants (FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE x2 on gfx950)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/pmc_lab_hbm
rm -rf $OUT; mkdir -p $OUT
i=0
for cfg in "$@"; do
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  (cd $ROOT && rocprofv3 --pmc $grp --kernel-trace -d $OUT/g$i -o pmc --output-format csv -- $ROOT/tools/gemm_lab/lab $cfg 3 0 0 > $OUT/g$i.log 2>&1)
  echo "$cfg" > $OUT/g$i.cfg
done
done
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/pmc_lab_hbm"
res = collections.OrderedDict()
for d in sorted(glob.glob(out + "/g*/"), key=lambda p: int(p.rstrip("/").split("g")[-1])):
    cfg = open(d.rstrip("/") + ".cfg").read().strip()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "gemm" in r["Kernel_Name"] and "ref_rows" not in r["Kernel_Name"]]
        agg = collections.defaultdict(list)
        for r in rows:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        res.setdefault(cfg, {}).update({c: sum(v) / len(v) for c, v in agg.items()})
for cfg, c in res.items():
    v, M, N, K = cfg.split()
    M, N, K = int(M), int(N), int(K)
    alg = 2 * (M * K + N * K + M * N)
    rd = c.get("FETCH_SIZE", 0) * 1024 * 2
    wr = c.get("WRITE_SIZE", 0) * 1024
    hit, miss = c.get("TCC_HIT_sum", 0), c.get("TCC_MISS_sum", 0)
    print(f"{cfg}: read {rd/1e6:.1f} MB (x2 corrected), write {wr/1e6:.1f} MB, algorithmic A {2*M*K/1e6:.1f} + W {2*N*K/1e6:.1f} + C {2*M*N/1e6:.1f} = {alg/1e6:.1f} MB; L2 hit rate {hit/(hit+miss+1e-9):.3f}")
PY
