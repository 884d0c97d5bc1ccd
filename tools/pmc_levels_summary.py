"""Summarise a tools/pmc_levels.sh run -> JSON on stdout (profiles/r03_pmc_levels.json; bench.py reads it for roofline.levels[*].pmc).
    python tools/pmc_levels_summary.py gpurun_out/pmc_levels <git commit>
Per level and kernel of the block loop: launches averaged, duration (us, from the dispatch timestamps of the counter passes — kernels
run serialised and a little slower under counter collection), counters per launch, and derived fractions:
  mfma_busy   = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024 SIMDs), kernel cycles = GRBM_GUI_ACTIVE / 8 (sum over the 8 XCDs)
  valu_active = 4 x SQ_ACTIVE_INST_VALU / (kernel cycles x 1024)   (SQ_ACTIVE_* count quad-cycles, MI355X_MICROARCH.md)
  lds_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE"""
import collections, csv, glob, json, sys

root, commit = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
out = {"commit": commit, "method": "rocprofv3 --pmc passes over tools/profile_block.py (one shifted cross BasicBlock of the level in a loop, B=16 256x256, "
                                   "encoder widths), separate passes per counter group (tools/pmc_levels.sh); averages per launch, whole chip", "levels": {}}
for lvl in range(5):
    ker = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for g in sorted(glob.glob(f"{root}/l{lvl}/g*/**/p_counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(g)):
            name = r["Kernel_Name"]
            if "swf::" not in name or "pack" in name:
                continue
            short = name.replace("void ", "").replace("swf::(anonymous namespace)::", "").replace("swf::", "").split("(")[0]
            ker[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Counter_Name"] in ("SQ_WAVE_CYCLES", "SQ_INSTS_VALU"):
                dur[short].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    lev = {}
    for k, cs in ker.items():
        avg = {c: (sum(v[3:]) / len(v[3:]) if len(v) > 4 else sum(v) / len(v)) for c, v in cs.items()}
        d = dur.get(k, [])
        d = d[3:] if len(d) > 4 else d
        e = {"launches": len(next(iter(cs.values()))), "us_under_pmc": round(sum(d) / len(d), 1) if d else None,
             "counters": {c: round(v) for c, v in sorted(avg.items())}}
        cyc = avg.get("GRBM_GUI_ACTIVE", 0) / 8
        if cyc:
            e["kernel_cycles"] = round(cyc)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in avg:
                e["mfma_busy"] = round(avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024), 3)
            if "SQ_ACTIVE_INST_VALU" in avg:
                e["valu_active"] = round(4 * avg["SQ_ACTIVE_INST_VALU"] / (cyc * 1024), 3)
        if avg.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_conflict"] = round(avg.get("SQ_LDS_BANK_CONFLICT", 0) / avg["SQ_LDS_IDX_ACTIVE"], 3)
        lev[k] = e
    out["levels"][str(lvl)] = lev
print(json.dumps(out, indent=1))
