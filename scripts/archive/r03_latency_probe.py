"""Prints the single-wave latency probe (gpbo_debug_latency_probe): cycles per instruction pattern."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bayesianoptimization_amd.engine import GpEngine  # noqa: E402

NAMES = ["empty bracket", "dependent v_fma_f64", "dependent v_mul_f64", "4 independent v_fma_f64 chains (64 ops)",
         "8 independent v_fma_f64 chains (64 ops)", "dependent v_rsq_f64", "v_readlane_b32 (independent)",
         "dependent [2 readlane -> v_fma_f64 on the SGPR pair]", "dependent [readlane -> 32-bit VALU on the SGPR]",
         "LDS write -> read -> wait", "LDS read -> wait", "v_writelane_b32", "dependent v_cndmask_b32",
         "dependent v_fma_f64 + 2 independent v_mov", "2 independent v_fma_f64 chains (128 ops)", "(register set-up)",
         "dependent [2 readlane, s_nop 1, v_fma_f64 on the pair] (asm)", "[2 readlane, s_nop 1, v_fma_f64], fixed source lanes",
         "16 x [8 readlane, 4 v_fma_f64] (per fma)", "16 x [4 bcast ds_read_b64, wait, 4 v_fma_f64] (per fma)",
         "16 x [2 bcast ds_read_b128, wait, 4 v_fma_f64] (per fma)", "dependent [2 readlane, s_nop 1, v_rsq_f64, v_mul_f64]",
         "dependent v_mul_f64 + ds_write_b64 of the result", "(register set-up)",
         "16 x column chain (2 readlane, rsq, 2 Newton, l, ds_write2st64, next pivot) (x4 = per column), FIRST pass",
         "the same code, SECOND pass (x4 = per column)",
         "16 x column chain without the LDS write (x4 = per column)", "16 x column chain with one ds_write_b64 (x4 = per column)",
         "16 x column chain, v_mul for v_rsq, no LDS write (x4 = per column)"]
eng = GpEngine(0, debug=True)
for _ in range(2):
    out = eng.latency_probe(len(NAMES))
base = out[0]
for nme, v in zip(NAMES, out):
    print(f"{nme:58s} {int(v):7d} cycles / 64   -> {(v - base) / 64:7.2f} per copy")
