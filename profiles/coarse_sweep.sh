set -e
export SFMBA_PCG_COARSE_FROM=24
for c in 64 96 128 192 256; do
  echo "== final pinhole k=$c"
  timeout -k 10 200 python -u bench.py --shape final --pcg-coarse $c --steps 3 --warmup 1 --no-cpu-baseline
done
for c in 48 64 96 128; do
  echo "== venice bal k=$c"
  timeout -k 10 200 python -u bench.py --camera-model bal --pcg-coarse $c --steps 3 --warmup 1 --no-cpu-baseline
done
