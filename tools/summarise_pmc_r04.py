"""Turn the rocprofv3 output of tools/collect_pmc_r04.sh (gpurun_out/r4/pmc_TAG/) into profiles/r04_scl_pmc[_TAG].json:
per-frame instruction counts by class, HBM-side bytes and the wait / activity split of es_scl_wide_kernel<64,8> at B = 65 536.
    python tools/summarise_pmc_r04.py [TAG] [fp64_cycles other_cycles trans_cycles]   (issue costs from tools/ub/ub_issue.hip)"""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "base"
SRC = os.path.join(ROOT, "gpurun_out", "r4", f"pmc_{TAG}")
B = 65536


def counters(sub):
    d = collections.defaultdict(list)
    for f in glob.glob(os.path.join(SRC, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "es_scl_wide_kernel" in r["Kernel_Name"]:
                d[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in d.items()}


c = {}
for sub in ("fetch", "write", "sq_a", "sq_b", "f64", "f32", "waits", "tcc", "tcc2", "tcp"):
    c.update(counters(sub))
fp64 = c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + c["SQ_INSTS_VALU_FMA_F64"]
trans = c["SQ_INSTS_VALU_TRANS_F64"]
valu = c["SQ_INSTS_VALU"]
ints = c["SQ_INSTS_VALU_INT32"] + c["SQ_INSTS_VALU_INT64"]
other = valu - fp64 - trans
cyc = {"fp64": float(sys.argv[2]) if len(sys.argv) > 2 else 4.0, "other": float(sys.argv[3]) if len(sys.argv) > 3 else 4.0,
       "trans_f64": float(sys.argv[4]) if len(sys.argv) > 4 else 16.0}
gui = c["GRBM_GUI_ACTIVE"] / 8.0                     # cycles (the counter sums the 8 XCDs)
fetch_b = c["FETCH_SIZE"] * 1024 * 2                 # 128-byte read requests are tallied at 64 B on gfx950 (MI355X_MICROARCH.md, HBM; here: TCP_TCC_READ_REQ x 128 B matches)
write_b = c["WRITE_SIZE"] * 1024
wave_cyc = c["SQ_WAVE_CYCLES"] * 4
out = {
    "source": f"rocprofv3 --pmc, one pass per counter group, counters only (tools/collect_pmc_r04.sh {TAG}) -- python3 tools/scl_pmc3.py '' 65536 8; MI355X, round 4; "
              "SQ_* cycle counters are in quad-cycles, GRBM_GUI_ACTIVE sums the 8 XCDs; means over 3 warm launches",
    "kernel": "es_scl_wide_kernel<64,8>, B = 65 536 frames, 8 192 one-wave blocks, three waves per SIMD"
              + {"base": " -- the kernel as it stood at the START of round 4 (= round 3's, built as libechoseal_hip_r3wide.so)", "final": " -- the kernel as shipped at the end of round 4 (tree depth 7 in LDS, ds_bpermute gather)"}.get(TAG, f" -- build {TAG}"),
    "raw": {k: c[k] for k in sorted(c)},
    "per_frame": {"valu_instructions": round(valu / B), "fp64_instructions": round(fp64 / B), "trans_f64_instructions": round(trans / B),
                  "int_instructions": round(ints / B), "cvt_instructions": round(c["SQ_INSTS_VALU_CVT"] / B),
                  "other_valu_instructions (moves, selects, compares, lane permutes)": round((other - ints - c["SQ_INSTS_VALU_CVT"]) / B),
                  "salu_instructions": round(c["SQ_INSTS_SALU"] / B), "lds_instructions": round(c["SQ_INSTS_LDS"] / B),
                  "vmem_read_instructions": round(c["SQ_INSTS_VMEM_RD"] / B), "vmem_write_instructions": round(c["SQ_INSTS_VMEM_WR"] / B),
                  "fetch_bytes (FETCH_SIZE x 2)": round(fetch_b / B), "write_bytes (WRITE_SIZE)": round(write_b / B),
                  "hbm_side_bytes": round((fetch_b + write_b) / B), "algorithmic_bytes": 4096 + 520},
    "l2": {"TCP_TCC_READ_REQ": c.get("TCP_TCC_READ_REQ_sum"), "TCP_TCC_WRITE_REQ": c.get("TCP_TCC_WRITE_REQ_sum"), "TCC_EA0_RDREQ": c.get("TCC_EA0_RDREQ_sum"),
           "TCC_EA0_WRREQ": c.get("TCC_EA0_WRREQ_sum"), "TCC_HIT": c.get("TCC_HIT_sum"), "TCC_MISS": c.get("TCC_MISS_sum"),
           "read_requests_that_leave_the_l2": c.get("TCC_EA0_RDREQ_sum", 0) / max(1.0, c.get("TCP_TCC_READ_REQ_sum", 1.0)),
           "bytes_per_read_request": fetch_b / max(1.0, c.get("TCP_TCC_READ_REQ_sum", 1.0)), "bytes_per_write_request": write_b / max(1.0, c.get("TCP_TCC_WRITE_REQ_sum", 1.0))},
    "kernel_cycles": gui, "kernel_ms_at_2.4GHz": gui / 2.4e6,
    "wave_lifetime_split": {"active_any": c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], "wait_any (s_waitcnt, barriers)": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
                            "wait_inst_any (issue stalls)": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]},
    "mean_resident_waves_per_simd": wave_cyc / gui / 1024.0,
    "valu_active_fraction_per_simd": 4.0 * c["SQ_ACTIVE_INST_VALU"] / 1024.0 / gui,
    "valu_active_fraction_while_three_waves_are_resident": 4.0 * c["SQ_ACTIVE_INST_VALU"] / 1024.0 / gui / (wave_cyc / gui / 1024.0 / 3.0),
    "issue_cycles": cyc,
    "fp64_pipe_fraction": (fp64 * 4.0 + trans * 16.0) / 1024.0 / gui,
    "issue_slot_fraction_mixed_ceiling": (fp64 * cyc["fp64"] + trans * cyc["trans_f64"] + other * cyc["other"]) / 1024.0 / gui,
    "hbm_side_GBps_at_this_launch": (fetch_b + write_b) / (gui / 2.4e9) / 1e9,
}
name = {"final": "r04_scl_pmc.json", "base": "r04_scl_pmc_before.json"}.get(TAG, f"r04_scl_pmc_{TAG}.json")
json.dump(out, open(os.path.join(ROOT, "profiles", name), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "raw"}, indent=1))
