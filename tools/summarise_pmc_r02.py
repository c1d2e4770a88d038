"""Turn the rocprofv3 output of tools/collect_pmc_r02.sh (gpurun_out/r2/pmc/) into the committed summaries under profiles/:
r02_xcorr32_pmc_traffic.json, r02_scl_pmc.json, r02_bench_kernel_stats.csv, r02_bench.json."""
import collections, csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r2", "pmc")
DST = os.path.join(ROOT, "profiles")


def counters(sub, pat):
    d = collections.defaultdict(list)
    files = glob.glob(os.path.join(SRC, sub, "**", "*counter_collection.csv"), recursive=True)
    files = sorted(files, key=os.path.getmtime)[-1:]          # gpurun merges into gpurun_out/: files of earlier collections stay behind
    for f in files:
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                d[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in d.items()}


def pick(d, name_part, counter):
    v = [val for (k, c), val in d.items() if name_part in k and c == counter]
    assert len(v) == 1, (name_part, counter, [k for k in d])
    return v[0]


B = 65536
# ---- FETCH_SIZE / WRITE_SIZE calibration on the arithmetic-free row copy (4 bytes per lane, the kernel's access pattern)
rc_f = counters("rowcopy_FETCH_SIZE", "rowcopy<2>"); rc_w = counters("rowcopy_WRITE_SIZE", "rowcopy<2>")
fetch_kib = sum(rc_f.values()) / len(rc_f); write_kib = sum(rc_w.values()) / len(rc_w)
read_bytes, written_bytes = B * 4860, B * 4612
cal_f = read_bytes / (fetch_kib * 1024); cal_w = written_bytes / (write_kib * 1024)
xf = counters("x32_FETCH_SIZE", "es_xcorr32_kernel"); xw = counters("x32_WRITE_SIZE", "es_xcorr32_kernel")
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, one counter per run, no trace flags (tools/collect_pmc_r02.sh); MI355X, round 2",
       "calibration": {"kernel": "tools/ub/ub_rowcopy.hip rowcopy<2>: 65 536 rows of 1215 floats in, 1153 floats out, 4 bytes per lane, no arithmetic",
                       "bytes_read": read_bytes, "FETCH_SIZE_KiB": fetch_kib, "bytes_per_FETCH_SIZE_byte": cal_f,
                       "bytes_written": written_bytes, "WRITE_SIZE_KiB": write_kib, "bytes_per_WRITE_SIZE_byte": cal_w,
                       "reading": "FETCH_SIZE counts half of the bytes of 4-byte-per-lane coalesced reads as well (the guide states it for 16-byte lanes): x2; "
                                  "WRITE_SIZE is exact.  The factors come from this copy kernel, not from the kernel being measured."}}
for key, part, bytes_alg in (("c3_launch", "<17, 2048, false>", B * (4 * 2048 + 4 * 1986)), ("c3_launch_fused", "<17, 2048, true>", B * (4 * 2048 + 150))):
    f = pick(xf, part, "FETCH_SIZE"); w = pick(xw, part, "WRITE_SIZE")
    out[key] = {"kernel": "es_xcorr32_kernel<17,2048," + ("FUSED>" if "true" in part else "false>"), "records": B,
                "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "hbm_bytes_per_launch": round(f * 1024 * cal_f + w * 1024 * cal_w),
                "algorithmic_bytes": bytes_alg}
out["c3_launch_fused"]["note"] = "reads also cover the float64 samples of the exact re-evaluations (~80 doubles per evaluated lag) and the tables"
json.dump(out, open(os.path.join(DST, "r02_xcorr32_pmc_traffic.json"), "w"), indent=1)

# ---- list decoder (both mappings of the multi-frame kernel)
sa = counters("scl_a", "es_scl_"); sb = counters("scl_b", "es_scl_")
scl = {"source": "rocprofv3 --pmc (two passes of 8 SQ/GRBM counters, counters only) -- python3 tools/scl_pmc2.py '' 65536; MI355X, round 2; "
                 "SQ_* cycle counters are in quad-cycles, GRBM_GUI_ACTIVE sums the 8 XCDs"}
for tag, part, what in (("es_scl_multi_kernel<8>  B=65536, 16 paths x 4 lanes per wave (the mapping of the pipelined headline)", "<8, 4>", "two frames per wave"),
                        ("es_scl_multi_kernel<8,2 lanes>  B=65536, 32 paths x 2 lanes per wave", "<8, 2>", "four frames per wave"),
                        ("es_scl_wide_kernel<64,8>  B=65536, 64 paths x 1 lane per wave (the grouped pipeline of the headline; batches of tens of thousands of frames)", "<64, 8>", "eight frames per wave")):
    g = lambda d, c: pick(d, part, c)
    valu, act = g(sa, "SQ_INSTS_VALU"), g(sa, "SQ_ACTIVE_INST_VALU")
    gui = g(sb, "GRBM_GUI_ACTIVE") / 8.0
    scl[tag] = {
        "SQ_WAVES": g(sa, "SQ_WAVES"), "GRBM_GUI_ACTIVE": g(sb, "GRBM_GUI_ACTIVE"), "SQ_WAVE_CYCLES": g(sa, "SQ_WAVE_CYCLES"), "SQ_INSTS_VALU": valu,
        "SQ_ACTIVE_INST_VALU": act, "SQ_WAIT_INST_ANY": g(sa, "SQ_WAIT_INST_ANY"), "SQ_WAIT_ANY": g(sa, "SQ_WAIT_ANY"),
        "SQ_INSTS_SALU": g(sb, "SQ_INSTS_SALU"), "SQ_INSTS_LDS": g(sb, "SQ_INSTS_LDS"), "SQ_INSTS_VMEM_RD": g(sb, "SQ_INSTS_VMEM_RD"),
        "SQ_INSTS_VMEM_WR": g(sb, "SQ_INSTS_VMEM_WR"), "SQ_LDS_BANK_CONFLICT": g(sb, "SQ_LDS_BANK_CONFLICT"),
        "per_frame": {"valu_instructions": round(valu / B), "salu_instructions": round(g(sb, "SQ_INSTS_SALU") / B), "lds_instructions": round(g(sb, "SQ_INSTS_LDS") / B)},
        "cycles_per_valu_instruction": 4.0 * act / valu,
        "valu_active_fraction_per_simd": 4.0 * act / 1024.0 / gui,
        "kernel_ms_at_2.35GHz": gui / 2.35e6,
        "reading": f"three waves per SIMD, {what}; every vector instruction holds its SIMD's vector unit for ~4 cycles (FP64 and 32-bit alike in "
                   "this mix), so the issue peak is one wave-instruction per 4 cycles and SIMD"}
json.dump(scl, open(os.path.join(DST, "r02_scl_pmc.json"), "w"), indent=1)

for f in sorted(glob.glob(os.path.join(SRC, "bench_stats", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)[-1:]:
    shutil.copy(f, os.path.join(DST, "r02_bench_kernel_stats.csv"))
shutil.copy(os.path.join(SRC, "bench.json"), os.path.join(DST, "r02_bench.json"))
print(json.dumps(out, indent=1)[:1500]); print(json.dumps(scl, indent=1)[:1800])
