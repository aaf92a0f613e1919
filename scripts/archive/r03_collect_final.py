"""Copy the results of scripts/archive/r03_final_run.sh (merged back under gpurun_out/final3/) to their tracked names in profiles/."""
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
F = os.path.join(ROOT, "gpurun_out", "final3")
P = os.path.join(ROOT, "profiles")
MAP = {
    "bench_default.json": "r03_bench_default_C3_with_configs.json",
    "bench_C4_group2_virtual.json": "r03_bench_C4_group_2_virtual_ranks_1gpu.json",
    "pmc_C3/summary.json": "r03_pmc_C3.json", "pmc_C3/summary.txt": "r03_pmc_C3.txt",
    "pmc_C3/trace/t_kernel_stats.csv": "r03_trace_C3_kernel_stats.csv",
    "r03_chol_probe.json": "r03_chol_probe.json", "theta_search_timing.json": "r03_theta_search_timing.json",
    "r03_polish_modes.json": "r03_polish_modes.json", "r03_la_probe.json": "r03_la_probe.json",
    "col_stamps.log": "r03_col_stamps.log", "latency_probe.log": "r03_latency_probe.log",
    "chol_trace_4096_kernel_stats.txt": "r03_trace_cholesky_4096_kernel_stats.txt",
    "chol_trace_512_kernel_stats.txt": "r03_trace_cholesky_512_kernel_stats.txt",
    "lml_trace_kernel_stats.txt": "r03_trace_lml_4096_kernel_stats.txt",
    "pytest.log": "r03_pytest_gpu.log", "pytest_round2_forms.log": "r03_pytest_gpu_round2_forms.log",
    "suggest_C2_n_smart_0_reference_kernel_stats.txt": "r03_trace_C2_suggest_kernel_stats.txt",
    "suggest_C2_n_smart_10_device_kernel_stats.txt": "r03_trace_C2_suggest_device_local_search_kernel_stats.txt",
    "r03_select_probe.json": "r03_select_probe.json", "c2_trace_kernel_stats.txt": "r03_trace_C2_kernel_stats.txt",
}
for src, dst in MAP.items():
    s = os.path.join(F, src)
    if os.path.exists(s):
        shutil.copyfile(s, os.path.join(P, dst))
        print("ok  ", dst)
    else:
        print("MISSING", src)
