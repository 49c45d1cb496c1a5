# usage: bash tools/_emu_run.sh <outdir> ; 8-way rank step (gnark layout / bases unchanged / resident), whole MSM, sync calls, sweep
export TMPDIR=/tmp; O=gpurun_out/$1; mkdir -p $O
emu() { python3 bench.py --emulate-world $1 $2 --steps 60 --warmup 5 --no-cpu-baseline --no-verify 2>/dev/null | tail -1; }
for w in 8; do
  emu $w "" >> $O/emu.jsonl
  emu $w "--bases-unchanged" >> $O/emu.jsonl
  emu $w "--resident-bases" >> $O/emu.jsonl
done
python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-verify 2>/dev/null | tail -1 >> $O/emu.jsonl
python3 tools/sweep.py 308,1268,4096,16384,65536,131072,262144,1048576 > $O/sweep.log 2>&1
python3 tools/bench_sync_call.py --variants "X=1" 16 17 20 > $O/sync.jsonl 2>$O/sync.err
python3 - <<'E' $O
import json,sys
for l in open(sys.argv[1]+'/emu.jsonl'):
    d=json.loads(l)
    if 'emulated_world' in d: print(d['emulated_world'], 'unch' if d.get('bases_unchanged_flag') else ('res' if d['resident_bases'] else 'gnark'), round(d['ms_per_step_rank0'],4), round(d['single_call_ms'],4), d['kernel_ms_alone'])
    else: print('whole', round(d['ms_per_step'],4), d['config']['single_call_ms'], d['roofline']['kernel_ms_alone'])
E
cat $O/sweep.log | cut -c1-330; cat $O/sync.jsonl
