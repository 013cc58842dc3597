python -m pytest tests -m gpu -x -q 2>&1 | tail -5 || exit 1
for cfg in "16384 2 8" "16384 2 14" "16384 3 12" "16384 4 8" "32768 2 14" "4096 2 8" "1024 2 4" "256 2 2" "32 2 1" "1 1 1"; do
  set -- $cfg
  python bench.py --replicas $1 --groups $2 --threads $3 --steps 200 --warmup 20 --no-cpu || exit 1
done
