"""Summarise the rocprofv3 --pmc passes of tools/pmc_bench.sh into profiles/<round>_pmc_hbm_traffic.json: L2 <-> fabric bytes of
the last bench step, per kernel.  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide
coalesced reads (MI355X_MICROARCH.md, HBM section) -> x2 on the read side."""
import collections, csv, glob, json, sys
rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3          # bench steps (+warmup) in the profiled command
def load(c):
    f = glob.glob(f"gpurun_out/pmc/{c}/**/*counter_collection.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    return rows
F, W = load("FETCH_SIZE"), load("WRITE_SIZE")
# trunk passes in the profiled command = launches of the stem kernel (one per pass): bench steps + warm-up + the calibration pass of the
# storage centres + (round 5) the pass that fills the cache of the tail-only measurement -- counted, not assumed
n_stem = sum(1 for r in F if "stem_mfma" in r["Kernel_Name"] or "stem_pool_mfma" in r["Kernel_Name"])
if n_stem:
    steps = n_stem
KERNELS = (("gemm_glds_kernel", "gemm_glds"), ("gemm8w_kernel", "gemm8w"), ("gemm_pro_kernel", "gemm_pro"), ("gconv_mfma_kernel", "gconv_mfma"),
           ("bn_add_relu_kernel", "bn_add_relu"), ("bn_relu_apply_kernel", "bn_relu_apply"), ("stem_mfma_kernel", "stem_mfma"),
           ("stem_pool_mfma_kernel", "stem_pool_mfma"),
           ("bn_relu_maxpool_kernel", "bn_relu_maxpool"), ("bn_finalize_kernel", "bn_finalize"), ("avgpool_kernel", "avgpool"),
           ("gram_pro_kernel", "gram_pro"), ("gram_reduce_kernel", "gram_reduce"), ("bn_from_gram_kernel", "bn_from_gram"))
out, tot_r, tot_w = collections.OrderedDict(), 0.0, 0.0
for name, pat in KERNELS:
    f = [r for r in F if pat in r["Kernel_Name"]]
    w = [r for r in W if pat in r["Kernel_Name"]]
    if not f:
        continue
    nper = len(f) // steps
    f, w = f[-nper:], w[-nper:]
    fk = sum(float(r["Counter_Value"]) for r in f)
    wk = sum(float(r["Counter_Value"]) for r in w)
    out[name] = {"launches_per_step": nper, "FETCH_SIZE_KB_sum": fk, "WRITE_SIZE_KB_sum": wk,
                 "hbm_read_bytes_per_step_corrected_x2": fk * 1024 * 2, "hbm_write_bytes_per_step": wk * 1024}
    tot_r += fk * 2048; tot_w += wk * 1024
    print(f"{name}: {nper} launches/step, read {fk*2048/1e9:.3f} GB (x2 corrected), write {wk*1024/1e9:.3f} GB")
gemm = [out[k] for k in ("gemm_glds_kernel", "gemm8w_kernel", "gemm_pro_kernel") if k in out]
out["conv_gemm_all"] = {"launches_per_step": sum(g["launches_per_step"] for g in gemm),
                        "hbm_read_bytes_per_step_corrected_x2": sum(g["hbm_read_bytes_per_step_corrected_x2"] for g in gemm),
                        "hbm_write_bytes_per_step": sum(g["hbm_write_bytes_per_step"] for g in gemm)}
out["trunk_total"] = {"hbm_read_bytes_per_step_corrected_x2": tot_r, "hbm_write_bytes_per_step": tot_w, "total_GB": (tot_r + tot_w) / 1e9}
print(f"trunk total: {(tot_r + tot_w)/1e9:.2f} GB per step (read {tot_r/1e9:.2f} x2-corrected + write {tot_w/1e9:.2f})")
json.dump(out, open(f"profiles/{rnd}_pmc_hbm_traffic.json", "w"), indent=1)
