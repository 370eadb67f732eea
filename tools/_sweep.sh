cd /root/repo
run() { echo "== $*"; env "$@" timeout -k 10 200 python3 bench.py --mode train_mixing --no-cpu-baseline --steps 200 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
run A=1
run FOV_WGRAD_STREAM=0
run FOV_WGRAD_SPLIT=0
run FOV_WGRAD_SPLIT=1
run FOV_WGRAD_ENC_SIDE=1
run FOV_WGRAD_ENC_SIDE=0
run FOV_SIDE_PRIORITY=normal
