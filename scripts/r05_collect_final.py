"""Copy the results of scripts/r05_final_run.sh (merged back under gpurun_out/r05f/) to their tracked names in profiles/."""
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = os.path.join(ROOT, "gpurun_out", "r05f")
P = os.path.join(ROOT, "profiles")
MAP = {
    "bench_default.json": "r05_bench_default_C3_with_configs.json",
    "bench_C4_group2_virtual.json": "r05_bench_C4_group_2_virtual_ranks_1gpu.json",
    "pmc_C3/summary.json": "r05_pmc_C3.json", "pmc_C3/summary.txt": "r05_pmc_C3.txt",
    "pmc_C3/trace/t_kernel_stats.csv": "r05_trace_C3_kernel_stats.csv",
    "pmc_C2/summary.json": "r05_pmc_C2.json", "pmc_C2/summary.txt": "r05_pmc_C2.txt",
    "pmc_C2/trace/t_kernel_stats.csv": "r05_trace_C2_one_step_kernel_stats.csv",
    "c2_trace/t_kernel_stats.csv": "r05_trace_C2_kernel_stats.csv",
    "r05_pmc_C2_posterior_summary.txt": "r05_pmc_C2_posterior.txt", "r05_pmc_C2_posterior_summary.json": "r05_pmc_C2_posterior.json",
    "theta_search_timing.json": "r05_theta_search_timing.json",
    "small_fit_timing.json": "r05_small_fit_timing.json", "maximize_loop.json": "r05_maximize_loop.json",
    "tri_grid_ab.json": "r05_tri_grid_ab.json", "r04_chol_chain.json": "r05_chol_chain.json",
    "polish_fused_ab.json": "r05_polish_fused_ab.json", "suggest_host_profile.txt": "r05_suggest_host_profile.txt",
    "polish_sweep.json": "r05_polish_sweep.json",
    "pytest.log": "r05_pytest_gpu.log", "smoke.log": "r05_smoke.log",
}
for src, dst in MAP.items():
    s = os.path.join(F, src)
    if os.path.exists(s):
        shutil.copyfile(s, os.path.join(P, dst))
        print("ok  ", dst)
    else:
        print("MISSING", src)
