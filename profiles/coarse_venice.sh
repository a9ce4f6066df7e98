set -e
for c in -1 128 64; do
  echo "== venice pinhole stepwise coarse=$c"
  timeout -k 10 200 python -u bench.py --pcg-mode 1 --pcg-coarse $c --steps 3 --warmup 1 --no-cpu-baseline
done
