set -x
python __graft_entry__.py --smoke || exit 1
for cfg in "1 1 0" "1 1 1" "32 1 0" "32 2 0" "32 4 0" "256 1 0" "256 4 0" "1024 1 0" "1024 4 0" "1024 4 1" "1024 8 0" "4096 4 0" "4096 8 1"; do
  set -- $cfg
  python bench.py --replicas $1 --groups $2 --parts $3 --steps 300 --warmup 30 --no-cpu || exit 1
done
python bench.py --replicas 1024 --groups 4 --steps 300 --warmup 30 --no-cpu --no-events
