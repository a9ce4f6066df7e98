#!/bin/bash
# MFMA counters of the dense Cholesky path: bash profiles/pmc_mfma.sh <tag> <bench args...>
set -u
TAG=${1:-mfma}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 "$ROOT/bench.py" --no-cpu-baseline "$@" > "$OUT/trace_bench.json" 2> "$OUT/trace.err" || { echo "trace failed"; tail -5 "$OUT/trace.err"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc" -o mfma -- python3 "$ROOT/bench.py" --no-cpu-baseline --steps 1 --warmup 0 "$@" > "$OUT/pmc_bench.json" 2> "$OUT/pmc.err" || { echo "pmc failed"; tail -5 "$OUT/pmc.err"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
dur = defaultdict(list)
for r in csv.DictReader(open(glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True)[0])):
    dur[r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("sfmba::", "").replace("void ", "")].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(glob.glob(os.path.join(out, "pmc", "**", "*counter_collection.csv"), recursive=True)[0])):
    acc[r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("sfmba::", "").replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = ["SQ_INSTS_VALU_MFMA_F64", "SQ_INSTS_VALU_MFMA_MOPS_F64", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE"]
print("| kernel | launches | avg us | total ms | " + " | ".join(names) + " | MFMA util % (busy / (GUI_ACTIVE x 1024 SIMDs)) |")
print("|---|---|---|---|" + "---|" * (len(names) + 1))
for k in ("k_chol_flow", "k_chol_prepare", "k_chol_diag", "k_chol_panel", "k_chol_back", "k_bsr_to_dense", "k_schur_pairs"):
    if k not in dur:
        continue
    v = acc[k]
    tot = {c: sum(v.get(c, [0])) for c in names}
    util = 100 * tot["SQ_VALU_MFMA_BUSY_CYCLES"] / max(tot["GRBM_GUI_ACTIVE"] / 8 * 1024, 1)
    n = len(dur[k])
    print(f"| {k} | {n} | {sum(dur[k]) / n / 1e3:.2f} | {sum(dur[k]) / 1e6:.3f} | " + " | ".join(f"{tot[c]:.4g}" for c in names) + f" | {util:.2f} |")
PY
find "$OUT" -name "*kernel_trace.csv" -size +20M -delete
