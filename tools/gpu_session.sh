#!/bin/bash
# One gpurun call of round 2 (the script is rewritten per session; results land in gpurun_out/r2_sN/).
# usage: tools/gpu_session.sh <N> [what...]      what: tests bench ab prof pmc pmcsq ...
set -u
S=${1:-1}; shift || true
WHAT=${*:-tests bench ab prof pmc}
OUT=gpurun_out/r2_s$S
mkdir -p $OUT
export TMPDIR=/tmp MIOPEN_FIND_MODE=${MIOPEN_FIND_MODE:-2}
for w in $WHAT; do
case $w in
tests)
  timeout 900 python -m pytest tests -m gpu -q --maxfail=25 -x --durations=15 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
  tail -40 $OUT/pytest_gpu.log ;;
tests_all)
  timeout 1200 python -m pytest tests -m gpu -q --maxfail=25 --durations=15 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
  tail -60 $OUT/pytest_gpu.log ;;
bench)
  timeout 600 python bench.py --steps 10 --warmup 3 --json-out $OUT/bench_default.json > $OUT/bench_default.log 2>&1; echo "bench rc=$?"
  tail -c 6000 $OUT/bench_default.log ;;
ab)
  for lib in ${AB_LIBS:-libirn_hip.so libirn_hip_prev.so}; do
    [ -f irn_amd/lib/$lib ] || continue
    for wl in walk coco walk_r5; do
      IRN_HIP_LIB=$PWD/irn_amd/lib/$lib timeout 300 python bench.py --workload $wl --steps 6 --warmup 2 --no-legs --no-cpu-baseline \
         --json-out $OUT/ab_${wl}_${lib%.so}.json > $OUT/ab_${wl}_${lib%.so}.log 2>&1
      python - <<PY
import json
try:
    r = json.load(open("$OUT/ab_${wl}_${lib%.so}.json"))
    print("$lib $wl: %.1f img/s, launch %.3f ms, frac %.4f" % (r["value"], r["roofline"]["avg_launch_ms"], r["roofline"]["frac"]))
except Exception as e:
    print("$lib $wl: FAILED", e)
PY
    done
  done ;;
prof)
  R=$PWD; cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_trace -o walk -f csv -- python $R/bench.py --steps 4 --warmup 1 --no-legs --no-cpu-baseline > $R/$OUT/prof_trace.log 2>&1
  cd $R
  find $OUT/prof_trace -name "*kernel_stats*" -exec cp {} $OUT/kernel_stats.csv \;
  head -12 $OUT/kernel_stats.csv ;;
pmc)
  R=$PWD; cd /tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c -d $R/$OUT/prof_$c -o walk -f csv -- python $R/bench.py --steps 1 --warmup 0 --no-legs --no-cpu-baseline > $R/$OUT/prof_$c.log 2>&1
  done
  cd $R
  python tools/reduce_prof.py $OUT > $OUT/prof_summary.txt 2>&1; grep -E "resident_k|affinity_k" $OUT/prof_summary.txt | cut -c1-24,96-200 | head -12
  find $OUT -name "walk_kernel_trace.csv" -delete; find $OUT -name "walk_counter_collection.csv" -delete ;;
pmcsq)
  # SQ counters of the walk kernels, one rocprofv3 pass per group of four (profiles/r02_s12_sq_counters.txt)
  R=$PWD; cd /tmp; i=0
  for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F64 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $set -d $R/$OUT/prof_sq$i -o walk -f csv -- python $R/bench.py --steps 1 --warmup 0 --no-legs --no-cpu-baseline ${SQ_BENCH_ARGS:-} > $R/$OUT/prof_sq$i.log 2>&1
  done
  cd $R
  python tools/reduce_prof.py $OUT > $OUT/sq_summary.txt 2>&1; grep -E "resident_kernel|affinity_kernel" $OUT/sq_summary.txt | cut -c1-40,96-200
  find $OUT -name "walk_counter_collection.csv" -delete ;;
ins)
  for r in 5 10; do timeout 200 python tools/ins_step_breakdown.py $r 64 5 > $OUT/ins_breakdown_r$r.txt 2>&1; cat $OUT/ins_breakdown_r$r.txt | grep -v Warning; done
  timeout 300 python tools/ins_step_bench.py 64 > $OUT/ins_step_bench.txt 2>&1; grep walk_batch $OUT/ins_step_bench.txt ;;
sweep)
  for d in ${SWEEP_PD:-6 8 12 14}; do
    timeout 200 python bench.py --steps ${SWEEP_STEPS:-5} --warmup 2 --no-legs --no-cpu-baseline --walk-option poll_delay=$d --json-out $OUT/sweep_pd$d.json > /dev/null 2>&1
    python -c "import json; r=json.load(open('$OUT/sweep_pd$d.json')); print('poll_delay $d: %.1f img/s launch %.3f ms' % (r['value'], r['roofline']['avg_launch_ms']))"
  done
  for d in ${SWEEP_PDP:-1 3 4}; do
    timeout 200 python bench.py --workload walk_r5 --steps 5 --warmup 2 --no-legs --no-cpu-baseline --walk-option poll_delay_plain=$d --json-out $OUT/sweep_r5_pdp$d.json > /dev/null 2>&1
    python -c "import json; r=json.load(open('$OUT/sweep_r5_pdp$d.json')); print('r5 poll_delay_plain $d: %.1f img/s launch %.3f ms' % (r['value'], r['roofline']['avg_launch_ms']))"
  done ;;
abalt)
  for lib in libirn_hip.so libirn_hip_alt.so; do
    for wl in walk_r5 ins; do
      IRN_HIP_LIB=$PWD/irn_amd/lib/$lib timeout 300 python bench.py --workload $wl --steps 6 --warmup 2 --no-legs --no-cpu-baseline \
         --json-out $OUT/abalt_${wl}_${lib%.so}.json > $OUT/abalt_${wl}_${lib%.so}.log 2>&1
      python -c "import json; r=json.load(open('$OUT/abalt_${wl}_${lib%.so}.json')); print('$lib $wl: %.1f img/s, %.3f ms/step' % (r['value'], r['ms_per_step']))" || tail -3 $OUT/abalt_${wl}_${lib%.so}.log
    done
  done ;;
stepsleg)
  for b in 64 128; do
    timeout 600 python bench.py --workload steps --steps 2 --warmup 1 --batch $b --json-out $OUT/steps_b$b.json > $OUT/steps_b$b.log 2>&1
    python -c "import json; r=json.load(open('$OUT/steps_b$b.json')); print('steps, $b images per pass: %.1f img/s' % r['value'], r['config'].get('last_pass_seconds'))" || tail -5 $OUT/steps_b$b.log
  done ;;
profile)
  for c in 1 2 3 4; do timeout 100 python tools/resident_profile.py 10 4 $c 2>&1 | grep "wg 0"; done
  for c in 1 2 3; do timeout 100 python tools/resident_profile.py 5 16 $c 2>&1 | grep "wg 0"; done ;;
stepsprof)
  timeout 600 python -m cProfile -s cumtime bench.py --workload steps --steps 2 --warmup 1 --batch 64 > $OUT/steps_cprofile.txt 2>&1
  grep -v "MIOpen" $OUT/steps_cprofile.txt | grep -E "^\{|cumtime|_work|_flush|edges_for|forward_batch|make_loader|msf_pack|cam_merge|label_epilogue|synchronize|\.cpu|numpy|save|result|acquire|sleep|__call__|sync" | head -50 ;;
camprof|e2eprof)
  # which MIOpen / PyTorch / irn kernels a backbone-bound leg runs, and their share (profiles/r02_s14_cam_kernel_stats_fused.csv)
  leg=${w%prof}
  R=$PWD; cd /tmp
  timeout 400 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_$leg -o $leg -f csv -- python $R/bench.py --workload $leg --steps 2 --warmup 1 --no-legs --no-cpu-baseline > $R/$OUT/prof_$leg.log 2>&1
  cd $R
  find $OUT/prof_$leg -name "*kernel_stats*" -exec cp {} $OUT/${leg}_kernel_stats.csv \;
  find $OUT/prof_$leg -name "${leg}_kernel_trace.csv" -delete
  python tools/kernel_classes.py $OUT/${leg}_kernel_stats.csv ;;
affab)
  R=$PWD
  for lib in ${AB_LIBS:-libirn_hip.so}; do
    [ -f irn_amd/lib/$lib ] || continue
    echo "== $lib"
    IRN_HIP_LIB=$R/irn_amd/lib/$lib timeout 300 python -m pytest tests/test_gpu_walk.py -q -m gpu -x -k "affinity" 2>&1 | tail -1
    for wl in walk walk_r5; do
      cd /tmp; IRN_HIP_LIB=$R/irn_amd/lib/$lib timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/aff_${lib%.so}_$wl -o t -f csv -- python $R/bench.py --workload $wl --steps 4 --warmup 1 --no-legs --no-cpu-baseline > $R/$OUT/aff_${lib%.so}_$wl.log 2>&1; cd $R
      f=$(find $OUT/aff_${lib%.so}_$wl -name '*kernel_stats*' | head -1)
      grep -E 'affinity_kernel|resident_kernel' $f | awk -F'",' '{print substr($1,2,60), $2, $4}' | sed "s/^/$wl /"
      find $OUT/aff_${lib%.so}_$wl -name 't_kernel_trace.csv' -delete
    done
  done ;;
insab)
  for r in ins ins_r10; do for b in "" --ins-blocking; do
    timeout 300 python bench.py --workload $r --steps 6 --warmup 2 --no-legs --no-cpu-baseline $b --json-out $OUT/insab_${r}_${b:-pipelined}.json > /dev/null 2>&1
    python -c "import json; r=json.load(open('$OUT/insab_${r}_${b:-pipelined}.json')); print('$r ${b:-pipelined}: %.1f img/s, %.2f ms/step' % (r['value'], r['ms_per_step']))"
  done; done ;;
epibench)
  for lib in ${AB_LIBS:-libirn_hip.so}; do
    [ -f irn_amd/lib/$lib ] || continue
    echo "== $lib"; IRN_HIP_LIB=$PWD/irn_amd/lib/$lib timeout 300 python tools/epilogue_bench.py 2>&1 | grep -v MIOpen | tee $OUT/epilogue_bench_${lib%.so}.txt
  done ;;
fusedab)
  # trunk epilogue fused (irn_bn_act) vs composed PyTorch ops, same run otherwise
  for w in cam e2e; do for f in 1 0; do
    IRN_FUSED_EPILOGUE=$f timeout 300 python bench.py --workload $w --steps 3 --warmup 1 --no-legs --no-cpu-baseline --json-out $OUT/fusedab_${w}_$f.json > /dev/null 2>&1
    python -c "import json; r=json.load(open('$OUT/fusedab_${w}_$f.json')); print('$w fused=$f: %.1f img/s, %.2f ms/step' % (r['value'], r['ms_per_step']))"
  done; done ;;
smoke)
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 2 --no-legs --no-cpu-baseline 2>&1 | grep -E "^\{" | cut -c1-400 ;;
legs)
  timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --json-out $OUT/bench_legs.json > $OUT/bench_legs.log 2>&1; tail -c 4000 $OUT/bench_legs.log ;;
esac
done
du -sh $OUT
