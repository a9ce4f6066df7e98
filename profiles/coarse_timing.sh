# Timing of the two-level PCG preconditioner against block-Jacobi alone (SFMBA_PCG_COARSE:
# -1 = off, 0 = the default policy, k = clusters of k cameras from the first solve on).
set -e
for c in -1 0; do
  echo "== final pinhole coarse=$c"
  SFMBA_PCG_COARSE=$c timeout -k 10 200 python -u bench.py --shape final --steps 3 --warmup 1 --no-cpu-baseline
done
for c in -1 0; do
  echo "== venice bal coarse=$c"
  SFMBA_PCG_COARSE=$c timeout -k 10 200 python -u bench.py --camera-model bal --steps 3 --warmup 1 --no-cpu-baseline
done
