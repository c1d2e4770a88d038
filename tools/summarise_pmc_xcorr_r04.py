"""gpurun_out/r4/pmc_xcorr (tools/collect_pmc_xcorr_r04.sh) -> profiles/r04_xcorr32_pmc_traffic.json: HBM-side bytes per launch of the two
correlation kernels on the config-3 shape.  FETCH_SIZE is scaled by the factor the round-2 calibration copy kernel gave (reads of this access
pattern are tallied at half their bytes on gfx950; profiles/r02_xcorr32_pmc_traffic.json), WRITE_SIZE is exact."""
import collections, csv, glob, json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r4", "pmc_xcorr")
cal = json.load(open(os.path.join(ROOT, "profiles", "r02_xcorr32_pmc_traffic.json")))["calibration"]


def counters(sub):
    d = collections.defaultdict(list)
    for f in glob.glob(os.path.join(SRC, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "es_xcorr32_kernel" in r["Kernel_Name"]:
                d[("fused" if "Lb1" in r["Kernel_Name"] or "true" in r["Kernel_Name"] else "screen", r["Counter_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in d.items()}


c = {}
for sub in ("fetch", "write", "sq"):
    c.update(counters(sub))
B = 65536
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, one counter per run, no trace flags (tools/collect_pmc_xcorr_r04.sh); MI355X, round 4; "
                 "means over the warm launches of tools/xcorr32_c3_launch.py",
       "calibration": cal}
for kind, name, alg in (("screen", "c3_launch", (4 * 2048 + 4 * 1986) * B), ("fused", "c3_launch_fused", (4 * 2048 + 150) * B)):
    f, w = c.get((kind, "FETCH_SIZE")), c.get((kind, "WRITE_SIZE"))
    if f is None or w is None:
        continue
    out[name] = {"kernel": "es_xcorr32_kernel<17,2048,%s>" % ("FUSED" if kind == "fused" else "false"), "records": B, "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w,
                 "hbm_bytes_per_launch": round(f * 1024 * cal["bytes_per_FETCH_SIZE_byte"] + w * 1024 * cal["bytes_per_WRITE_SIZE_byte"]),
                 "algorithmic_bytes": alg}
    g = c.get((kind, "GRBM_GUI_ACTIVE"))
    if g:
        out[name]["kernel_ms_at_2.4GHz"] = g / 8.0 / 2.4e6
        out[name]["valu_active_fraction_per_simd"] = 4.0 * c[(kind, "SQ_ACTIVE_INST_VALU")] / 1024.0 / (g / 8.0)
json.dump(out, open(os.path.join(ROOT, "profiles", "r04_xcorr32_pmc_traffic.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "calibration"}, indent=1))
