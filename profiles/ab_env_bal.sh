#!/bin/bash
# ab_env.sh for the second camera model: bash profiles/ab_env_bal.sh <tag> VAR value [value ...]
set -u
TAG=$1; VAR=$2; shift 2
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
for v in "$@"; do
    env "$VAR=$v" timeout -k 10 200 python bench.py --no-cpu-baseline --camera-model bal > "$OUT/bal_$VAR$v.json" 2> "$OUT/bal_$VAR$v.err" || exit 1
done
python - "$OUT" "$VAR" "$@" <<'PY'
import json, sys
out, var, vals = sys.argv[1], sys.argv[2], sys.argv[3:]
for v in vals:
    d = json.loads(open(f"{out}/bal_{var}{v}.json").read().strip().splitlines()[-1])
    fam = {k: round(x["total_ms"], 3) for k, x in d["kernels"].items() if x["launches"]}
    print(f"bal {var}={v}: {d['ms_per_step']:.3f} ms/solve, cost {d['final_cost']!r}, families {fam}")
PY
