"""Summary of scripts/archive/r04_kmat_pmc.sh: kmat_kernel's counters per problem size (the launches are told apart by grid size),
with the derived figures the question needs — VALU issue occupancy, instructions per element, store bytes per launch."""
import collections
import csv
import glob
import json
import os
import sys


def rows(src, sub):
    out = []
    for path in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
        out += list(csv.DictReader(open(path)))
    return out


def main(src, dst):
    per = collections.defaultdict(lambda: collections.defaultdict(list))      # grid -> counter -> values
    for sub in ("pmc_sq", "pmc_sq2", "pmc_write", "pmc_fetch"):
        for r in rows(src, sub):
            if "kmat_kernel" not in r["Kernel_Name"]:
                continue
            per[int(r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = collections.defaultdict(list)
    for path in glob.glob(os.path.join(src, "trace", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if "kmat_kernel" in r["Kernel_Name"]:
                grid = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
                dur[grid].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    out, lines = {}, [f"# kmat_kernel counters from {src} (per-launch averages)"]
    for grid, cs in sorted(per.items()):
        tiles = grid // 256                     # 256 threads per lower 64x64 tile
        nt = int(((8 * tiles + 1) ** 0.5 - 1) / 2)
        NP = nt * 64
        e = {c: sum(v) / len(v) for c, v in cs.items()}
        e["NP"] = NP
        e["elements_written"] = tiles * 4096
        if grid in dur:
            e["avg_us_unprofiled_trace"] = sum(dur[grid]) / len(dur[grid]) / 1e3
        gui = e.get("GRBM_GUI_ACTIVE")
        if gui:
            cyc = gui / 8.0                     # GRBM_GUI_ACTIVE is summed over the 8 XCDs
            e["kernel_cycles_est"] = cyc
            if "SQ_ACTIVE_INST_VALU" in e:      # quad-cycles summed over the 1024 SIMDs
                e["valu_issue_occupancy"] = e["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / cyc
            if "SQ_BUSY_CYCLES" in e:
                e["sq_busy_frac"] = e["SQ_BUSY_CYCLES"] / 8.0 / cyc if e["SQ_BUSY_CYCLES"] / 8.0 <= cyc * 1.5 else None
            if "SQ_WAVE_CYCLES" in e:
                e["waves_resident_per_simd"] = e["SQ_WAVE_CYCLES"] * 4.0 / 1024.0 / cyc
        if "SQ_INSTS_VALU" in e:
            e["valu_insts_per_element"] = e["SQ_INSTS_VALU"] * 64.0 / e["elements_written"]
            # one fp64 VALU instruction occupies its SIMD 4 cycles per wave64 (scripts/archive/r03_latency_probe.py): the floor of the
            # kernel if nothing but this stream ran, at the clock of `kernel_cycles_est`
            e["valu_floor_cycles"] = e["SQ_INSTS_VALU"] * 4.0 / 1024.0
            if gui:
                e["valu_floor_frac_of_kernel"] = e["valu_floor_cycles"] / (gui / 8.0)
        if "WRITE_SIZE" in e:
            e["write_bytes"] = e["WRITE_SIZE"] * 1024
            e["write_bytes_algorithmic"] = tiles * 4096 * 8
        if "FETCH_SIZE" in e:
            e["fetch_bytes_corrected_x2"] = 2 * e["FETCH_SIZE"] * 1024
        if all(k in e for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES")):
            tot = e["SQ_WAVE_CYCLES"]
            e["wave_time_split"] = {"parked_waitcnt_barrier": e["SQ_WAIT_ANY"] / tot, "issue_stall": e["SQ_WAIT_INST_ANY"] / tot,
                                    "issuing": e["SQ_ACTIVE_INST_ANY"] / tot}
        out[f"NP={NP}"] = e
        lines.append(f"NP = {NP} (grid {grid})")
        for c in sorted(e):
            lines.append(f"    {c:32s} {e[c]}")
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from bayesianoptimization_amd.build import _fingerprint
    out["_meta"] = {"source_fingerprint": _fingerprint(), "src": src}
    open(dst + ".txt", "w").write("\n".join(lines) + "\n")
    json.dump(out, open(dst + ".json", "w"), indent=1, sort_keys=True)
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
