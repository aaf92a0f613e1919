"""Copy the results of scripts/archive/r04_final_run.sh (merged back under gpurun_out/r04f/) to their tracked names in profiles/."""
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
F = os.path.join(ROOT, "gpurun_out", "r04f")
P = os.path.join(ROOT, "profiles")
MAP = {
    "bench_default.json": "r04_bench_default_C3_with_configs.json",
    "bench_C4_group2_virtual.json": "r04_bench_C4_group_2_virtual_ranks_1gpu.json",
    "pmc_C3/summary.json": "r04_pmc_C3.json", "pmc_C3/summary.txt": "r04_pmc_C3.txt",
    "pmc_C3/trace/t_kernel_stats.csv": "r04_trace_C3_kernel_stats.csv",
    "pmc_C2/summary.json": "r04_pmc_C2.json", "pmc_C2/summary.txt": "r04_pmc_C2.txt",
    "pmc_C2/trace/t_kernel_stats.csv": "r04_trace_C2_one_step_kernel_stats.csv",
    "c2_trace/t_kernel_stats.csv": "r04_trace_C2_kernel_stats.csv",
    "theta_search_timing.json": "r04_theta_search_timing.json",
    "pytest.log": "r04_pytest_gpu.log", "smoke.log": "r04_smoke.log",
}
for src, dst in MAP.items():
    s = os.path.join(F, src)
    if os.path.exists(s):
        shutil.copyfile(s, os.path.join(P, dst))
        print("ok  ", dst)
    else:
        print("MISSING", src)
