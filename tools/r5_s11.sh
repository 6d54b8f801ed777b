#!/bin/bash
# round 5 session 11: is the channels-last trunk bit-stable on a find database without split-K implicit GEMMs (tools/miopen_det_filter.py)? at what speed?
set -u
OUT=gpurun_out/r5_s11; mkdir -p $OUT
export TMPDIR=/tmp MIOPEN_FIND_MODE=2
KEY=gfx950-cu256-hip7.0.51831
mkdir -p /tmp/detroot && cp -r irn_amd/data/miopen/$KEY-det /tmp/detroot/$KEY
for p in a b; do IRN_MIOPEN_SEED_DIR=/tmp/detroot IRN_MIOPEN_CACHE=/tmp/mc_det$p timeout 300 python tools/determinism_probe.py $OUT/det_filtered_$p.json --sizes 512x512,375x500 --pairs 8 --scales 1.0,0.5,1.5,2.0 > $OUT/det_filtered_$p.log 2>&1; done
grep -E "repeat|miopen db" $OUT/det_filtered_a.log; python tools/determinism_probe.py --compare $OUT/det_filtered_a.json $OUT/det_filtered_b.json
for p in a; do IRN_MIOPEN_CACHE=/tmp/mc_ship$p timeout 300 python tools/determinism_probe.py $OUT/det_shipped_$p.json --sizes 512x512,375x500 --pairs 8 --scales 1.0,0.5,1.5,2.0 > $OUT/det_shipped_$p.log 2>&1; done
echo "-- shipped database for comparison"; grep -E "repeat" $OUT/det_shipped_a.log
for tag in shipped filtered; do
  if [ $tag = filtered ]; then export IRN_MIOPEN_SEED_DIR=/tmp/detroot; fi
  for wl in cam e2e; do IRN_MIOPEN_CACHE=/tmp/mc_b$tag timeout 300 python bench.py --workload $wl --steps 12 --warmup 3 --no-legs --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.readline()); print('$tag $wl %.1f images/s' % r['value'], r['config']['trunk']['layout'], r['config']['trunk']['tuned_nhwc_shapes'])"; done
done
