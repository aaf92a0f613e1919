"""Summary of scripts/r05_pmc_C2_posterior.sh: per-launch averages of the SQ counters for the posterior kernel of BASELINE config 2
(posterior_kernel_v2<8, 1, 1, 32, 16>: fused k* generation + MFMA contraction, 512-row chunks) and what they say about its ceiling.

Units (MI355X_MICROARCH.md, rocprofv3 PMC): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves;
SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the 1024 SIMDs; GRBM_GUI_ACTIVE cycles summed over 8 XCDs.
WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES (disjoint buckets of a wave's life).
"""
import collections
import csv
import json
import os
import sys


def load(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    if os.path.exists(path):
        for r in csv.DictReader(open(path)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def main(src, dst):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bayesianoptimization_amd.build import _fingerprint
    merged = collections.defaultdict(dict)
    for sub in ("pmc_a", "pmc_b"):
        for k, v in load(os.path.join(src, sub, "p_counter_collection.csv")).items():
            for c, xs in v.items():
                merged[k][c if c not in merged[k] else c + "_" + sub] = sum(xs) / len(xs)
            merged[k]["launches_" + sub] = max(len(xs) for xs in v.values())
    dur = {}
    tpath = os.path.join(src, "trace", "t_kernel_stats.csv")
    if os.path.exists(tpath):
        for r in csv.DictReader(open(tpath)):
            dur[r["Name"]] = (int(r["Calls"]), float(r["AverageNs"]))
    out, lines = {}, [f"# SQ counters per launch, C2 bench command ({src})"]
    for k, e in merged.items():
        if "posterior_kernel" not in k and "kstar" not in k:
            continue
        e = dict(e)
        cyc = e.get("GRBM_GUI_ACTIVE", 0.0) / 8.0                      # shader cycles of the launch
        if k in dur:
            e["avg_ns"] = dur[k][1]
            e["shader_clock_mhz_est"] = cyc / dur[k][1] * 1e3 if dur[k][1] else None
        if cyc > 0:
            e["mfma_pipe_busy_frac"] = e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0 / cyc
            wc = e.get("SQ_WAVE_CYCLES", 0.0)
            if wc > 0:
                # a wave's life in buckets (fractions of its resident quad-cycles)
                e["wave_frac_issuing_any"] = e.get("SQ_ACTIVE_INST_ANY", 0.0) / wc
                e["wave_frac_issuing_valu_incl_mfma"] = e.get("SQ_ACTIVE_INST_VALU", 0.0) / wc
                e["wave_frac_issue_stalled"] = e.get("SQ_WAIT_INST_ANY", 0.0) / wc
                e["wave_frac_parked_waitcnt_or_barrier"] = e.get("SQ_WAIT_ANY", 0.0) / wc
                e["wave_frac_issue_stalled_on_lds"] = e.get("SQ_WAIT_INST_LDS", 0.0) / wc
                e["waves_resident_per_simd_avg"] = wc * 4.0 / 1024.0 / cyc
            # issue slots: one VALU-class instruction (incl. MFMA) occupies its SIMD's issue port for >= 4 cycles
            e["valu_non_mfma_insts_per_simd"] = (e.get("SQ_INSTS_VALU", 0.0) - e.get("SQ_INSTS_MFMA", 0.0)) / 1024.0
            e["valu_non_mfma_issue_frac_at_4_cycles"] = e["valu_non_mfma_insts_per_simd"] * 4.0 / cyc
            e["valu_non_mfma_issue_frac_at_8_cycles_fp64"] = e["valu_non_mfma_insts_per_simd"] * 8.0 / cyc
            e["lds_array_busy_frac"] = e.get("SQ_LDS_IDX_ACTIVE", 0.0) / 256.0 / cyc if "SQ_LDS_IDX_ACTIVE" in e else None
            e["lds_bank_conflict_frac_of_lds_cycles"] = (e.get("SQ_LDS_BANK_CONFLICT", 0.0) / e["SQ_LDS_IDX_ACTIVE"]
                                                         if e.get("SQ_LDS_IDX_ACTIVE") else None)
        out[k] = e
        lines.append(k)
        for c in sorted(e):
            if e[c] is not None:
                lines.append(f"    {c:44s} {e[c]:.6g}")
    out["_meta"] = {"source_fingerprint": _fingerprint(), "src": src, "command": "python bench.py --config C2 --steps 20 --warmup 3 --no-cpu-baseline --no-suggest"}
    open(dst + ".txt", "w").write("\n".join(lines) + "\n")
    json.dump(out, open(dst + ".json", "w"), indent=1, sort_keys=True)
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
