#!/bin/bash
# One more PMC pass over the bench command: LDS bank conflicts / LDS activity / MFMA-VALU co-execution of the dominant kernels.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in C3 C5; do
  OUT=gpurun_out/pmc_extra_$c; rm -rf $OUT; mkdir -p $OUT
  CMD="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-suggest --config $c"
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/a -o p -- $CMD > /dev/null 2> $OUT/a.err
  rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH --kernel-trace --output-format csv -d $OUT/b -o p -- $CMD > /dev/null 2> $OUT/b.err
  python - "$OUT" <<'PY'
import collections, csv, glob, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(out + "/*/**/p_counter_collection.csv", recursive=True) + glob.glob(out + "/*/p_counter_collection.csv"):
    for r in csv.DictReader(open(p)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
lines = []
for k, v in agg.items():
    if not any(t in k for t in ("posterior_kernel", "kstar_gen", "gemm128", "chol_step")):
        continue
    lines.append(k[:100])
    for c, xs in sorted(v.items()):
        lines.append(f"    {c:34s} {sum(xs) / len(xs):.6g}   (launches {len(xs)})")
    d = {c: sum(xs) / len(xs) for c, xs in v.items()}
    if d.get("SQ_LDS_IDX_ACTIVE"):
        lines.append(f"    lds_bank_conflict_cycles / lds_active_cycles = {d.get('SQ_LDS_BANK_CONFLICT', 0) / d['SQ_LDS_IDX_ACTIVE']:.4f}")
    if d.get("SQ_BUSY_CYCLES"):
        lines.append(f"    mfma_busy / (4 simd x busy) = {d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (4 * d['SQ_BUSY_CYCLES']):.4f} ; coexec / mfma_busy = {d.get('SQ_VALU_MFMA_COEXEC_CYCLES', 0) / max(d.get('SQ_VALU_MFMA_BUSY_CYCLES', 1), 1):.4f}")
open(out + "/summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
done
