#!/bin/bash
# Collects the files of profiles/ on the GPU box into gpurun_out/profiles_new/ (see profiles/README.md for what each one is).
# Every GPU step has its own time limit; the script stops at the first failure.
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/profiles_new
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
P=${ROUND:-r03}
say() { echo "[collect] $*"; }

say default bench
timeout -k 10 400 python3 -u bench.py > $O/${P}_bench_default.json 2> $O/default.err

prof() {   # name, bench arguments...
  name=$1; shift
  say rocprof $name
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tmp_$name -o p -- python3 bench.py "$@" > $O/tmp_$name.log 2>&1
  cp $O/tmp_$name/p_kernel_stats.csv $O/${P}_bench_${name}_kernel_stats.csv
  grep '^{' $O/tmp_$name.log | tail -n 1 > $O/${P}_bench_${name}_under_rocprof.json
  rm -rf $O/tmp_$name $O/tmp_$name.log
}
prof forward_b1 --steps 200 --warmup 20 --no-wavernn --no-cpu-baseline --no-in-flight
prof gst_b32 --workload gst --batch 32 --steps 20 --warmup 3 --no-cpu-baseline --no-wavernn
prof train_b32 --workload train --batch 32 --steps 10 --warmup 3 --no-cpu-baseline
prof train_b32_steps_only --workload train --batch 32 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline
prof predict --workload predict --steps 2 --frames 600 --no-cpu-baseline
prof wavernn_c4 --steps 20 --warmup 2 --no-cpu-baseline --no-roofline

table() {  # name, bench arguments...
  name=$1; shift
  say kernel table $name
  timeout -k 10 400 python3 -u bench.py "$@" --kernel-table --no-cpu-baseline > $O/${P}_bench_${name}_kernel_table.txt 2>&1
}
table forward_b1 --no-wavernn
table gst --workload gst --batch 32
table train --workload train --batch 32
table train_wavernn --workload train_wavernn --batch 32 --steps 5 --warmup 2
table train_ar --workload train_ar --batch 8 --steps 10 --warmup 3

pmc() {   # name, program and arguments...: one counter per pass, FETCH_SIZE then WRITE_SIZE
  name=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    say pmc $name $c
    timeout -s KILL 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/tmp_pmc -o p -- python3 "$@" > $O/tmp_pmc.log 2>&1
    cp $O/tmp_pmc/p_counter_collection.csv $O/${P}_pmc_${name}_$c.csv
    rm -rf $O/tmp_pmc $O/tmp_pmc.log
  done
}
pmc forward_b1 bench.py --steps 20 --warmup 2 --no-wavernn --no-cpu-baseline --no-roofline --no-in-flight
pmc gst_b32 bench.py --workload gst --batch 32 --steps 3 --warmup 1 --no-wavernn --no-cpu-baseline --no-roofline
pmc train_b32 bench.py --workload train --batch 32 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline
pmc wavernn_c4 tools/wavernn_steps.py 200
say pmc table '(copy into profiles/ together with the counter summaries)'
python3 tools/update_pmc_traffic.py $O/${P}_pmc_forward_b1_FETCH_SIZE.csv $O/${P}_pmc_forward_b1_WRITE_SIZE.csv \
    $O/${P}_pmc_gst_b32_FETCH_SIZE.csv $O/${P}_pmc_gst_b32_WRITE_SIZE.csv \
    $O/${P}_pmc_train_b32_FETCH_SIZE.csv $O/${P}_pmc_train_b32_WRITE_SIZE.csv \
    $O/${P}_pmc_wavernn_c4_FETCH_SIZE.csv $O/${P}_pmc_wavernn_c4_WRITE_SIZE.csv > /dev/null && cp profiles/pmc_traffic.json $O/pmc_traffic.json
# the raw counter files of the three larger workloads run to several MB: keep their per-kernel summaries instead
for n in gst_b32 train_b32 wavernn_c4; do
  python3 tools/pmc_summary.py $O/${P}_pmc_${n}_FETCH_SIZE.csv $O/${P}_pmc_${n}_WRITE_SIZE.csv > $O/${P}_pmc_${n}_summary.json
  rm -f $O/${P}_pmc_${n}_FETCH_SIZE.csv $O/${P}_pmc_${n}_WRITE_SIZE.csv
done
say precision sweep
timeout -k 10 400 python3 -u tests/precision_sweep.py --out $O/${P}_precision_sweep.json > $O/${P}_precision_sweep.txt 2>&1
say config 1
timeout -k 10 200 python3 -u bench.py --workload config1 --steps 3 > $O/${P}_bench_config1_cpu.json 2>/dev/null
say done
ls -la $O
