R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/devmoves
mkdir -p $OUT
timeout -k 10 600 python -m pytest $R/tests -x -q -m gpu > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
( time python3 $R/bench.py > $OUT/default.json 2> $OUT/default.err ) 2>&1 | grep real
cat $OUT/default.json
