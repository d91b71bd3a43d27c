#!/bin/bash
# phase ablation of the Gram kernel (bn_gram.hip, -DCVCL_GRAM_ABLATE=<bits> libraries built as lib/libcvcl_hip_abl<bits>.so):
# kernel durations from a rocprofv3 kernel trace of tools/gram_bench.py per library
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for a in "" _abl1 _abl2 _abl4 _abl8 _abl3 _abl6 _abl7; do
  [ -f $R/multimodal-baby_amd/lib/libcvcl_hip$a.so ] || continue
  rm -rf /tmp/ga
  CVCL_HIP_LIB=$R/multimodal-baby_amd/lib/libcvcl_hip$a.so rocprofv3 --kernel-trace -d /tmp/ga -o ga --output-format csv -- python3 $R/tools/gram_bench.py > /dev/null 2>&1
  python3 - "$a" <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/ga/**/*kernel_trace.csv", recursive=True)[0]
agg = {}
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "gram" not in n: continue
    import re
    m = re.search(r"(gram_pro_kernel<\d>|gram_reduce_kernel|bn_from_gram_kernel)", n)
    key = (m.group(1) if m else n[:30]) + " g" + r["Grid_Size_X"]
    agg.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000)
print("lib", sys.argv[1] or "(product)", " | ".join(f"{k} {sorted(v)[len(v)//2]:.1f}" for k, v in sorted(agg.items())))
PY
done
