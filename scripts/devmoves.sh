# device-side move generation: tests, then host vs device proposals at several replica counts
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/devmoves
mkdir -p $OUT
timeout -k 10 600 python -m pytest $R/tests -x -q -m gpu > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
for dm in 0 1; do
  for th in 8 2; do
    python3 $R/bench.py --no-cpu --no-secondary --device-moves $dm --threads $th > $OUT/b_${dm}_${th}.json 2>$OUT/b_${dm}_${th}.err || exit 1
    python3 -c "import json,sys; d=json.load(open('$OUT/b_${dm}_${th}.json')); print('dm',$dm,'threads',$th,'value %.3e'%d['value'],'ms/step %.4f'%d['ms_per_step'],'kernel us %.1f'%d['roofline']['avg_launch_us'],'acc %.3f'%d['acceptance'],'drift %.1e'%d['energy_drift_rel'])"
  done
done
for dm in 0 1; do
  for R_ in 1 32; do
    python3 $R/bench.py --no-cpu --no-secondary --device-moves $dm --replicas $R_ --groups 1 --threads 1 --steps 3000 --warmup 300 > $OUT/s_${dm}_${R_}.json 2>$OUT/s_${dm}_${R_}.err || exit 1
    python3 -c "import json,sys; d=json.load(open('$OUT/s_${dm}_${R_}.json')); print('dm',$dm,'R',$R_,'value %.3e'%d['value'],'us/step %.2f'%(1e3*d['ms_per_step']))"
  done
done
