"""Summarise rocprofv3 --pmc CSV passes (scripts/profile_pmc.sh) per kernel: averages per launch, the
gfx950 FETCH_SIZE x2 correction for wide coalesced reads (MI355X_MICROARCH.md §HBM), MFMA pipe utilisation.

    python scripts/pmc_summary.py gpurun_out/pmc_r2 profiles/r02_pmc_C3   -> .txt and .json (run ON THE GPU BOX, right
    after scripts/profile_pmc.sh, so that the stamped source fingerprint is the one of the profiled library)
"""
import collections
import csv
import json
import os
import sys


def load(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    if not os.path.exists(path):
        return agg
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def main(src, dst, n_steps=2):
    """n_steps = bench steps + warm-ups the profiled command ran (profile_pmc.sh: 1 + 1): per-kernel launch counts are
    divided by it so that bench.py can turn per-launch averages into bytes per posterior pass (a pass is several slab
    launches)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bayesianoptimization_amd.build import _fingerprint
    merged = collections.defaultdict(dict)
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
        for k, v in load(os.path.join(src, sub, "p_counter_collection.csv")).items():
            for c, xs in v.items():
                merged[k][c] = sum(xs) / len(xs)
                merged[k]["launches_" + sub] = len(xs)
    dur = {}
    tpath = os.path.join(src, "trace", "t_kernel_stats.csv")
    if os.path.exists(tpath):
        for r in csv.DictReader(open(tpath)):
            dur[r["Name"]] = (int(r["Calls"]), float(r["AverageNs"]))
    out = {}
    lines = [f"# PMC summary of {src} (per-launch averages; FETCH/WRITE_SIZE in KiB as reported)"]
    for k, v in sorted(merged.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
        if k.startswith("__amd"):
            continue
        e = dict(v)
        if "FETCH_SIZE" in e:
            e["fetch_bytes_reported"] = e["FETCH_SIZE"] * 1024
            e["fetch_bytes_corrected_x2"] = 2 * e["FETCH_SIZE"] * 1024
        if "WRITE_SIZE" in e:
            e["write_bytes"] = e["WRITE_SIZE"] * 1024
        if "SQ_VALU_MFMA_BUSY_CYCLES" in e and e.get("GRBM_GUI_ACTIVE"):
            # busy cycles are summed over 1024 SIMDs, GRBM_GUI_ACTIVE over 8 XCDs
            e["mfma_pipe_busy_frac"] = (e["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0) / (e["GRBM_GUI_ACTIVE"] / 8.0)
        n_l = max([v for c, v in e.items() if c.startswith("launches_")] or [0])
        e["launches_per_step"] = n_l / float(n_steps)
        if k in dur:
            e["avg_ns"] = dur[k][1]
            if e.get("GRBM_GUI_ACTIVE"):
                e["shader_clock_mhz_est"] = e["GRBM_GUI_ACTIVE"] / 8.0 / dur[k][1] * 1e3
        out[k] = e
        lines.append(k)
        for c in sorted(e):
            lines.append(f"    {c:34s} {e[c]:.6g}")
    # the kernel sources these counters belong to: bench.py reports the traffic only while the library is still built
    # from exactly these sources
    out["_meta"] = {"source_fingerprint": _fingerprint(), "n_steps_profiled": n_steps, "src": src}
    open(dst + ".txt", "w").write("\n".join(lines) + "\n")
    json.dump(out, open(dst + ".json", "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 2)
