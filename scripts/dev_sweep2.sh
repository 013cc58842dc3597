# the default bench line under different driver shapes
R=$GRAFT_REPO_ROOT
run() {
  timeout -k 10 200 python3 $R/bench.py --no-cpu --no-secondary $1 > /tmp/x.json 2>/tmp/x.err < /dev/null || { cat /tmp/x.err | tail -5; exit 1; }
  python3 -c "import json; d=json.load(open('/tmp/x.json')); print('$1','value %.3e'%d['value'],'ms/step %.4f'%d['ms_per_step'],'kernel us %.1f'%d['roofline']['avg_launch_us'])"
}
for a in "$@"; do run "$a"; done
