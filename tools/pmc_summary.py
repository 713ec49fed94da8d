"""Summarise rocprofv3 --pmc passes (one directory per pass, each holding *_counter_collection.csv) for one
kernel: mean counter value per launch, plus the derived HBM traffic / MFMA figures DESIGN.md quotes.
FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x
(MI355X_MICROARCH.md, HBM / rocprofv3 section), so the derived traffic doubles it.
Usage: python tools/pmc_summary.py <dir-with-pass-subdirs> <kernel-substring> [skip_first_n] > summary.json"""
import csv
import glob
import json
import os
import sys


def main():
    root, kern = sys.argv[1], sys.argv[2]
    skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    out = {}
    for path in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        per_dispatch = {}
        with open(path) as f:
            for row in csv.DictReader(f):
                if kern not in row["Kernel_Name"]:
                    continue
                key = (row["Counter_Name"], int(row["Dispatch_Id"]))
                per_dispatch[key] = per_dispatch.get(key, 0.0) + float(row["Counter_Value"])
        names = sorted({k[0] for k in per_dispatch})
        for name in names:
            vals = [v for (n, d), v in sorted(per_dispatch.items(), key=lambda kv: kv[0][1]) if n == name][skip:]
            if vals:
                out[name] = {"mean_per_launch": sum(vals) / len(vals), "launches": len(vals)}
    d = {"kernel": kern}
    g = lambda n: out.get(n, {}).get("mean_per_launch")  # noqa: E731
    if g("FETCH_SIZE") is not None:
        d["hbm_read_bytes_per_launch"] = 2.0 * g("FETCH_SIZE") * 1024
    if g("WRITE_SIZE") is not None:
        d["hbm_write_bytes_per_launch"] = g("WRITE_SIZE") * 1024
    if "hbm_read_bytes_per_launch" in d and "hbm_write_bytes_per_launch" in d:
        d["hbm_bytes_per_launch"] = d["hbm_read_bytes_per_launch"] + d["hbm_write_bytes_per_launch"]
    if g("SQ_INSTS_VALU_MFMA_MOPS_F32") is not None:
        d["mfma_flops_per_launch"] = g("SQ_INSTS_VALU_MFMA_MOPS_F32") * 512   # 1 MOP = 512 flops
    if g("SQ_INSTS_VALU_MFMA_MOPS_BF16") is not None:
        d["mfma_bf16_flops_per_launch"] = g("SQ_INSTS_VALU_MFMA_MOPS_BF16") * 512
    if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None:
        d["l2_hit_rate"] = g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))
    if g("SQ_WAIT_ANY") is not None and g("SQ_WAVE_CYCLES"):
        d["wait_frac_of_wave_cycles"] = g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")
    if g("SQ_VALU_MFMA_BUSY_CYCLES") is not None and g("SQ_WAVE_CYCLES"):
        # busy cycles are counted per SIMD (x4 per CU); wave cycles = 4 waves per workgroup resident
        d["mfma_busy_frac_of_wave_cycles"] = g("SQ_VALU_MFMA_BUSY_CYCLES") / 4.0 / g("SQ_WAVE_CYCLES")
    if g("SQ_LDS_BANK_CONFLICT") is not None and g("SQ_LDS_IDX_ACTIVE"):
        d["lds_conflict_frac"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")
    out["_derived"] = d
    try:                                     # the build these counters were taken on (bench.py reports traffic only for its own)
        here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        with open(os.path.join(here, "fab_torch_amd", "libfabhip.so.srchash")) as f:
            out["lib_srchash"] = f.read().split()[0]
    except (OSError, IndexError):
        out["lib_srchash"] = None
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
