export TMPDIR=/tmp
run() { echo "$1: $(env $2 LRA_BENCH_DBG=1 python bench.py --steps 7 --warmup 2 --no-cpu-baseline $3 2>&1 | grep -o 'align [0-9]* ms' | tail -6 | tr '\n' ' ')"; }
run norec-ish "LRA_BENCH_TAIL=copy" ""
run base14 "X=1" ""
run leak14 "LRA_RECORD_LEAK=1" ""
run base128 "LRA_RECORD_THREADS=128" ""
run leak128 "LRA_RECORD_THREADS=128 LRA_RECORD_LEAK=1" ""
