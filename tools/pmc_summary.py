"""Average the counters of one kernel over the launches of a tools/pmc_block.sh run.
    python tools/pmc_summary.py gpurun_out/pmc_xxx <kernel-name-substring>"""
import collections, csv, glob, sys
root, pat = sys.argv[1], sys.argv[2]
meta = None
for g in sorted(glob.glob(f"{root}/g*/p_counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(g)):
        if pat in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = (r["Kernel_Name"][:70], "VGPR", r.get("VGPR_Count"), "AGPR", r.get("Accum_VGPR_Count"), "LDS", r.get("LDS_Block_Size"),
                    "scratch", r.get("Scratch_Size"), "grid", r.get("Grid_Size"), "wg", r.get("Workgroup_Size"))
    for k, v in acc.items():
        v = v[3:] if len(v) > 4 else v
        print(f"  {k:32s} {sum(v) / len(v):16.0f}  (n={len(v)})")
print(meta)
