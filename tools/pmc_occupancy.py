"""Per-kernel occupancy / LDS / wave-state summary from rocprofv3 --pmc passes (one CSV per pass; counters of a kernel
are averaged per launch and summed over XCDs/SEs by rocprofv3 already).  Usage:
    python tools/pmc_occupancy.py out.txt pass1.csv pass2.csv ...
Derived figures (MI355X_MICROARCH.md section 'rocprofv3 PMC slots'):
    waves/launch            = SQ_WAVES
    mean resident waves/CU  = SQ_LEVEL_WAVES / SQ_BUSY_CU_CYCLES   (waves summed over busy CU-cycles) when both are present,
                              else SQ_WAVE_CYCLES / SQ_BUSY_CYCLES / 4 ... per SIMD: / 4 again
    wave-cycle split        = SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY as fractions of SQ_WAVE_CYCLES
    LDS bank conflict rate  = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (conflict cycles per LDS-active cycle)
VGPR / LDS / scratch per kernel come from the dispatch records of the same CSVs."""
import csv
import sys
from collections import defaultdict

out_path, passes = sys.argv[1], sys.argv[2:]
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
meta = {}
for path in passes:
    with open(path) as fh:
        for r in csv.DictReader(fh):
            k = r["Kernel_Name"].split("(")[0].replace("void expv_mi::dev::", "").replace("expv_mi::", "")
            if not k.startswith(("k_", "dev::k_")):
                continue
            a = acc[k][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
            meta[k] = (r["Workgroup_Size"], r["Grid_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"])
lines = []
for k in sorted(acc):
    c = {name: v[1] / max(v[0], 1) for name, v in acc[k].items()}
    n = max(v[0] for v in acc[k].values())
    wg, grid, vgpr, agpr, sgpr, lds, scr = meta[k]
    lines.append("%s" % k)
    lines.append("    launches %d | workgroup %s, grid(last) %s | VGPR %s (+%s acc) SGPR %s LDS %s B scratch %s B/lane" % (n, wg, grid, vgpr, agpr, sgpr, lds, scr))
    g = lambda x: c.get(x)
    if g("SQ_WAVES") is not None:
        lines.append("    SQ_WAVES/launch %.0f" % g("SQ_WAVES"))
    if g("SQ_LEVEL_WAVES") and g("SQ_BUSY_CU_CYCLES"):
        lines.append("    mean resident waves per busy CU %.2f  (= %.2f per SIMD; capacity 8 per SIMD)" % (
            g("SQ_LEVEL_WAVES") / g("SQ_BUSY_CU_CYCLES"), g("SQ_LEVEL_WAVES") / g("SQ_BUSY_CU_CYCLES") / 4))
    for d in ("MeanOccupancyPerCU", "MeanOccupancyPerActiveCU", "OccupancyPercent"):
        if g(d) is not None:
            lines.append("    %s (rocprofv3 derived, gfx94x formula) %.2f" % (d, g(d)))
    if g("SQ_WAVE_CYCLES"):
        wc = g("SQ_WAVE_CYCLES")
        parts = ["%s %.1f%%" % (nm, 100 * g(nm) / wc) for nm in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY") if g(nm) is not None]
        lines.append("    wave cycles: " + ", ".join(parts))
        for nm in ("SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA"):
            if g(nm) is not None:
                lines.append("        %s %.1f%% of wave cycles" % (nm, 100 * g(nm) / wc))
    if g("GRBM_GUI_ACTIVE") and g("SQ_BUSY_CYCLES"):
        lines.append("    SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE = %.3f (summed over SEs / XCDs: ratio to the 1-instance GUI counter)" % (g("SQ_BUSY_CYCLES") / g("GRBM_GUI_ACTIVE")))
    if g("SQ_INSTS_LDS") is not None:
        lines.append("    LDS instructions/launch %.0f" % g("SQ_INSTS_LDS"))
    if g("SQ_LDS_BANK_CONFLICT") is not None and g("SQ_LDS_IDX_ACTIVE"):
        lines.append("    LDS bank conflicts: %.0f conflict cycles / %.0f LDS-active cycles = %.2f%%" % (
            g("SQ_LDS_BANK_CONFLICT"), g("SQ_LDS_IDX_ACTIVE"), 100 * g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")))
    if g("LDSBankConflict") is not None:
        lines.append("    LDSBankConflict (rocprofv3 derived) %.2f%%" % g("LDSBankConflict"))
    if g("SQ_INSTS_VMEM_RD") is not None:
        lines.append("    VMEM instructions/launch: %.0f reads, %.0f writes" % (g("SQ_INSTS_VMEM_RD"), g("SQ_INSTS_VMEM_WR") or 0))
open(out_path, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
