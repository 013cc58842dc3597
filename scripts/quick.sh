python -m pytest tests -m gpu -x -q 2>&1 | tail -5 || exit 1
for cfg in "256 1 0 1 1 0" "1024 1 0 1 1 0" "1024 4 0 4 1 0" "4096 8 0 8 1 0" "16384 16 0 8 1 0" "16384 16 2 8 1 0" "32 2 0 2 1 0" "1 1 0 1 1 0"; do
  set -- $cfg
  python bench.py --replicas $1 --groups $2 --parts $3 --threads $4 --kernel $5 --zero-copy-moves $6 --steps 300 --warmup 30 --no-cpu || exit 1
done
