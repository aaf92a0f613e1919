"""Copy the results of scripts/r06_final_run.sh (merged back under gpurun_out/r06f/) to their tracked names in profiles/."""
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = os.path.join(ROOT, "gpurun_out", "r06f")
P = os.path.join(ROOT, "profiles")
MAP = {
    "bench_default.json": "r06_bench_default_C3_with_configs.json",
    "bench_C4_group8_virtual.json": "r06_bench_C4_group_8_virtual_ranks_1gpu.json",
    "pmc_C3/summary.json": "r06_pmc_C3.json", "pmc_C3/summary.txt": "r06_pmc_C3.txt",
    "pmc_C3/trace/t_kernel_stats.csv": "r06_trace_C3_kernel_stats.csv",
    "pmc_C2/summary.json": "r06_pmc_C2.json", "pmc_C2/summary.txt": "r06_pmc_C2.txt",
    "pmc_C2/trace/t_kernel_stats.csv": "r06_trace_C2_one_step_kernel_stats.csv",
    "c2_trace/t_kernel_stats.csv": "r06_trace_C2_kernel_stats.csv",
    "lml/trace_lml_2048_kernel_stats.csv": "r06_trace_lml_2048_kernel_stats.csv",
    "lml/trace_lml_4096_kernel_stats.csv": "r06_trace_lml_4096_kernel_stats.csv",
    "lml/pmc_lml_4096.txt": "r06_pmc_lml_4096.txt", "lml/pmc_lml_4096.json": "r06_pmc_lml_4096.json",
    "lml_4096_timeline.txt": "r06_lml_4096_timeline.txt", "lml_2048_timeline.txt": "r06_lml_2048_timeline.txt",
    "theta_search_timing.json": "r06_theta_search_timing.json",
    "small_fit_timing.json": "r06_small_fit_timing.json", "maximize_loop.json": "r06_maximize_loop.json",
    "polish_fused_ab.json": "r06_polish_fused_ab.json", "suggest_host_profile.txt": "r06_suggest_host_profile.txt",
    "small_n_trace/t_kernel_stats.csv": "r06_trace_small_n_kernel_stats.csv", "n300_trace/t_kernel_stats.csv": "r06_trace_n300_kernel_stats.csv",
    "polish_sweep.json": "r06_polish_sweep.json", "conditioning.json": "r06_conditioning.json",
    "pytest.log": "r06_pytest_gpu.log", "smoke.log": "r06_smoke.log",
    "suggest_host_profile_n16_64.txt": "r06_suggest_host_profile_n16_64.txt", "small_step_breakdown.txt": "r06_small_step_breakdown.txt",
    "lanes_grouping.json": "r06_lanes_grouping.json", "gemm_bench.json": "r06_gemm_bench.json", "post_10k_ab.json": "r06_post_10k_ab.json",
}
for src, dst in MAP.items():
    s = os.path.join(F, src)
    if os.path.exists(s):
        if dst == "r06_polish_fused_ab.json" and os.path.exists(os.path.join(P, dst)):      # keep the notes written beside the table
            new, old = json.load(open(s)), json.load(open(os.path.join(P, dst)))
            new["notes"] = old.get("notes", "")
            json.dump(new, open(os.path.join(P, dst), "w"), indent=1)
        else:
            shutil.copyfile(s, os.path.join(P, dst))
        print("ok  ", dst)
    else:
        print("MISSING", src)
# the loop over four seeds: its time depends on the trajectory
seeds = {}
for s in (1, 2, 3, 4):
    p = os.path.join(F, "maximize_loop.json" if s == 1 else f"maximize_loop_seed{s}.json")
    if os.path.exists(p):
        d = json.load(open(p))["device"]
        seeds[str(s)] = {"total_s": d["total_s"], "warm_up_s": d.get("warm_up_s"), "spikes_over_3x_band_median": d.get("spikes_over_3x_band_median"),
                         "band_median_ms": {b["N"]: b.get("median_ms") for b in d["bands"]}}
if seeds:
    tot = [v["total_s"] for v in seeds.values()]
    json.dump({"what": "scripts/r06_maximize_loop.py --seed S (N = 16 -> 528, d = 4, the reference's defaults) on the final library: the loop's time depends on "
                       "the trajectory (how many rounds the theta searches of the later steps take), so four seeds",
               "mean_total_s": sum(tot) / len(tot), "min_total_s": min(tot), "max_total_s": max(tot), "seeds": seeds},
              open(os.path.join(P, "r06_maximize_loop_seeds.json"), "w"), indent=1)
    print("ok   r06_maximize_loop_seeds.json", tot)
rep = sorted(glob.glob(os.path.join(F, "transcript_replay_*.json")))
if rep:
    json.dump({"what": "tests/test_gpu_transcript.py on MI355X: every engine call the REAL bayes_opt driver made (recorded in the build container, "
                       "oracle/gen_transcript.py) replayed on libgpbo.so; worst error of each kind as a FRACTION OF ITS BAR (tests/transcript.Bars)",
               "transcripts": [json.load(open(f)) for f in rep]}, open(os.path.join(P, "r06_transcript_replay.json"), "w"), indent=1)
    print("ok   r06_transcript_replay.json")
