cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_ft
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_ft -o ft --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_finetune.py --batch ${FT_BATCH:-256} --steps 2 > $GRAFT_REPO_ROOT/gpurun_out/prof_ft.log 2>&1
tail -3 $GRAFT_REPO_ROOT/gpurun_out/prof_ft.log
python3 - <<'PY'
import csv, glob, os
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof_ft/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    print(f'{float(r["TotalDurationNs"])/1e6/6:9.2f} ms/step {int(r["Calls"])//6:5d} calls/step {100*float(r["TotalDurationNs"])/tot:5.1f}%  {r["Name"][:110]}')
PY
