"""Copy the results of scripts/archive/r02_final_run.sh (merged back under gpurun_out/final/) to their tracked names in profiles/."""
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
F = os.path.join(ROOT, "gpurun_out", "final")
P = os.path.join(ROOT, "profiles")
MAP = {
    "bench_C3.json": "r02_bench_C3.json", "bench_C5.json": "r02_bench_C5_f32_shard.json",
    "bench_C4_1gpu.json": "r02_bench_C4_shard0_1gpu.json",
    "bench_C4_group2_virtual.json": "r02_bench_C4_group_2_virtual_ranks_1gpu.json",
    "pmc_C3/summary.json": "r02_pmc_C3.json", "pmc_C3/summary.txt": "r02_pmc_C3.txt",
    "pmc_C5/summary.json": "r02_pmc_C5.json", "pmc_C5/summary.txt": "r02_pmc_C5.txt",
    "pmc_C3/trace/t_kernel_stats.csv": "r02_trace_C3_kernel_stats.csv",
    "pmc_C5/trace/t_kernel_stats.csv": "r02_trace_C5_kernel_stats.csv",
    "config_table.json": "r02_config_table.json", "theta_search_timing.json": "r02_theta_search_timing.json",
    "append_latency.json": "r02_append_latency.json", "r02_suggest_modes.json": "r02_suggest_modes.json",
    "mt_timing.log": "r02_mt19937_timing.log", "r02_fit_probe.json": "r02_fit_probe_final.json",
}
for src, dst in MAP.items():
    s = os.path.join(F, src)
    if os.path.exists(s):
        shutil.copyfile(s, os.path.join(P, dst))
        print("ok  ", dst)
    else:
        print("MISSING", src)
