#!/bin/bash
# round 6 profile session: the default bench line (short form + detail), kernel traces (four-stream + single-stream) of mv-3ddet, PMC
# passes (separate passes, kernel trace only): HBM traffic, SQ / MFMA busy; single-stream kernel traces of the two other configurations
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
T=${TAG:-r6}
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
B="$GRAFT_REPO_ROOT/bench.py"
db () { find /tmp/prof_$1 -name '*.db' | head -1; }
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/${T}_bench_default_line.json 2> $OUT/${T}_bench_default.err; echo "bench rc $?"
cp bench_detail.json $OUT/${T}_bench_default_detail.json
tail -c 2200 $OUT/${T}_bench_default_line.json
CMD="python $B --no-cpu-baseline --no-other-configs --steps 4 --warmup 2"
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_ks -o p -- $CMD > /tmp/prof_ks.log 2>&1); echo "rc $?"
python tools/rocpd_stats.py "$(db ks)" $OUT/${T}_kernel_stats.txt > /dev/null
python tools/rocpd_critical.py "$(db ks)" > $OUT/${T}_critical_chain.txt 2>&1
(cd /tmp && ES_TWO_STREAMS=0 ES_WGRAD_ASYNC=0 timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_ss -o p -- $CMD > /tmp/prof_ss.log 2>&1); echo "rc $?"
python tools/rocpd_stats.py "$(db ss)" $OUT/${T}_single_stream_kernel_stats.txt > /dev/null
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum -d /tmp/prof_pf -o p -- $CMD > /tmp/prof_pf.log 2>&1); echo "rc $?"
python tools/rocpd_pmc.py "$(db pf)" $OUT/${T}_pmc_fetch.txt > /dev/null
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_MISS_sum -d /tmp/prof_pw -o p -- $CMD > /tmp/prof_pw.log 2>&1); echo "rc $?"
python tools/rocpd_pmc.py "$(db pw)" $OUT/${T}_pmc_write.txt > /dev/null
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY -d /tmp/prof_sq -o p -- $CMD > /tmp/prof_sq.log 2>&1); echo "rc $?"
python tools/rocpd_pmc.py "$(db sq)" $OUT/${T}_pmc_sq.txt > /dev/null
python tools/pmc_traffic.py --tag $T --config mv3ddet --steps 16 --fetch $OUT/${T}_pmc_fetch.txt --write $OUT/${T}_pmc_write.txt > $OUT/${T}_pmc_traffic_stdout.txt 2>&1
cp profiles/${T}_pmc_traffic.json $OUT/ 2>/dev/null
python tools/pmc_sq_util.py $OUT/${T}_pmc_sq.txt > $OUT/${T}_mfma_util.txt 2>&1
for kind in occupancy grounding; do
  C2="python $B --no-cpu-baseline --only $kind --steps 3 --warmup 1 --other-steps 3"
  (cd /tmp && ES_TWO_STREAMS=0 ES_WGRAD_ASYNC=0 timeout 250 rocprofv3 --kernel-trace --stats -d /tmp/prof_ks_$kind -o p -- $C2 > /tmp/prof_ks_$kind.log 2>&1); echo "rc $?"
  python tools/rocpd_stats.py "$(db ks_$kind)" $OUT/${T}_single_stream_kernel_stats_$kind.txt > /dev/null
  # HBM-side bytes of the configuration's roofline family (separate counter passes, kernel trace only)
  (cd /tmp && timeout 250 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum -d /tmp/prof_pf_$kind -o p -- $C2 > /tmp/prof_pf_$kind.log 2>&1); echo "rc $?"
  python tools/rocpd_pmc.py "$(db pf_$kind)" $OUT/${T}_pmc_fetch_$kind.txt > /dev/null
  (cd /tmp && timeout 250 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_MISS_sum -d /tmp/prof_pw_$kind -o p -- $C2 > /tmp/prof_pw_$kind.log 2>&1); echo "rc $?"
  python tools/rocpd_pmc.py "$(db pw_$kind)" $OUT/${T}_pmc_write_$kind.txt > /dev/null
  python tools/pmc_traffic.py --tag $T --config $kind --fetch $OUT/${T}_pmc_fetch_$kind.txt --write $OUT/${T}_pmc_write_$kind.txt --command "$C2" > $OUT/${T}_pmc_traffic_stdout_$kind.txt 2>&1
  cp profiles/${T}_pmc_traffic_$kind.json $OUT/ 2>/dev/null
done
timeout 300 python tools/bench_halo.py > $OUT/${T}_halo_ab.txt 2>&1
timeout 300 python tools/bench_imgwgrad.py > $OUT/${T}_imgwgrad_ab.txt 2>&1
timeout 300 python tools/bench_imgconv.py > $OUT/${T}_imgconv_ab.txt 2>&1
ls -la $OUT | grep "${T}_" | tail -24
