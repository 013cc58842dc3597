python -m pytest tests -m gpu -x -q 2>&1 | tail -8 || exit 1
python bench.py --no-cpu --steps 200 --warmup 20 || exit 1
