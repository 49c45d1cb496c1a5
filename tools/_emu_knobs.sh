export TMPDIR=/tmp; O=gpurun_out/$1; mkdir -p $O
run() { # label, mode, env...
  lab=$1; mode=$2; shift; shift
  r=$(env "$@" python3 bench.py --emulate-world 8 $mode --steps 60 --warmup 5 --no-cpu-baseline --no-verify 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step_rank0'],4), round(d['single_call_ms'],4), d['kernel_ms_alone'])")
  echo "$lab $mode $r" >> $O/knobs.txt
}
