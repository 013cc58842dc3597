set -x
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 || exit 1
for cfg in "1 1 0 1 1 0" "1 1 0 1 1 1" "32 2 0 2 1 0" "1024 4 0 1 1 0" "1024 4 0 4 1 0" "1024 4 0 4 0 0" "1024 8 0 8 1 0" "4096 8 0 8 1 0" "4096 16 0 8 1 0" "4096 8 0 8 1 1" "16384 16 0 8 1 0"; do
  set -- $cfg
  python bench.py --replicas $1 --groups $2 --parts $3 --threads $4 --kernel $5 --zero-copy-moves $6 --steps 300 --warmup 30 --no-cpu || exit 1
done
