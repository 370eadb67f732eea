set -e
cd /root/repo
out=gpurun_out/r5q; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_knobs.py tests/test_gpu_threads.py -x -q -m gpu -k "mixing or stacked or knob or threads or dx" > $out/t1.log 2>&1 || { tail -40 $out/t1.log; exit 1; }
tail -3 $out/t1.log
timeout -k 10 300 python3 bench.py --mode train_mixing --no-cpu-baseline --steps 200 > $out/tm.json 2> $out/tm.err
timeout -k 10 300 python3 bench.py --mode train_mixing --no-cpu-baseline --steps 100 --t-in 30 --t-out 30 > $out/tm30.json 2> $out/tm30.err
FOV_NO_DX_FUSION=1 timeout -k 10 300 python3 bench.py --mode train_mixing --no-cpu-baseline --steps 200 > $out/tm_nodx.json 2> $out/tm_nodx.err
python3 - <<'PY'
import json
for f in ("tm","tm30","tm_nodx"):
    d=json.load(open('gpurun_out/r5q/%s.json'%f)); print(f, d["ms_per_step"], d["roofline"]["frac"])
PY
