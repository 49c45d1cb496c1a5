export TMPDIR=/tmp; O=gpurun_out/r5r; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
emu() { # world mode env
  w=$1; mode=$2; shift; shift
  r=$(env "$@" python3 bench.py --emulate-world $w $mode --steps 60 --warmup 5 --no-cpu-baseline --no-verify 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step_rank0'],4), round(d['single_call_ms'],4))")
  echo "w=$w $mode $* $r" >> $O/emu.txt
}
for rep in 1 2; do
for w in 8 4 2; do
  emu $w "--bases-unchanged" X=1
  emu $w "--bases-unchanged" CURDLE_AUX_PRIO=0
  emu $w "" X=1
  emu $w "" CURDLE_AUX_PRIO=0
done
for v in X=1 CURDLE_AUX_PRIO=3 "CURDLE_AUX_PRIO=3 CURDLE_REDUCE_PRIO=0"; do
  r=$(env $v python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-verify 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), d['config']['single_call_ms'])")
  echo "whole $v $r" >> $O/emu.txt
done
done
cat $O/emu.txt
