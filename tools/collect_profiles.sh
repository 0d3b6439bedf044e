#!/bin/bash
# Collects a round's profiles on the GPU box into gpurun_out/profiles_<tag>/ (copied into profiles/ afterwards; tag = $1, default r03):
# kernel-trace statistics and counter passes (each in a run of its own) of the headline bench command, counter passes
# of the analytic configurations, and the bench lines of every configuration.
set -e
R=$GRAFT_REPO_ROOT
T=${1:-r03}
OUT=$R/gpurun_out/profiles_$T
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp

stats() {   # <name> <bench args...>
  local name=$1; shift
  rm -rf /tmp/kt_$name
  timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d /tmp/kt_$name -o t --output-format csv -- python3 $R/bench.py --no-cpu-baseline "$@" > /tmp/kt_$name.log 2>&1
  cp $(find /tmp/kt_$name -name "*kernel_stats.csv" | head -1) $OUT/${T}_bench_${name}_kernel_stats.csv
  echo "stats $name done"
}
pmc() {     # <name> <counters...> -- <bench args...>
  local name=$1; shift
  local ctrs=()
  while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
  shift
  rm -rf /tmp/pmc_$name
  timeout -s KILL 400 rocprofv3 --pmc "${ctrs[@]}" -d /tmp/pmc_$name -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > /tmp/pmc_$name.log 2>&1
  python3 $R/tools/pmc_sum.py /tmp/pmc_$name $OUT/${T}_pmc_$name.csv "rocprofv3 --pmc ${ctrs[*]} -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline $*; sums over all dispatches of each kernel in the run (column dispatches): one frame, and on the analytic scenes a second one with the counting build (k_extend_pair<.., true, ..>); a frame of more paths than the arena holds is several batches, each with its own launches"
  echo "pmc $name done"
}

stats demo1080p --steps 3 --warmup 1
pmc sq_demo1080p_1024spp SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU --
pmc fetch_demo1080p_1024spp FETCH_SIZE --
pmc write_demo1080p_1024spp WRITE_SIZE --
python3 $R/tools/merge_hbm.py $OUT/${T}_pmc_fetch_demo1080p_1024spp.csv $OUT/${T}_pmc_write_demo1080p_1024spp.csv $OUT/${T}_pmc_hbm_demo1080p_1024spp.csv
pmc mfma_demo1080p_1024spp SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES --
# the bench lines quote the counter passes: the lines of this collection have to read this collection's passes
bench_reads() { cp "$@" $R/profiles/; }
bench_reads $OUT/${T}_pmc_sq_demo1080p_1024spp.csv $OUT/${T}_pmc_hbm_demo1080p_1024spp.csv
for sc in random100k mixed20k random1m; do
  case $sc in
    random100k) A="--scene random100k --spp 256"; N=random100k_256spp;;
    mixed20k)   A="--scene mixed20k --width 3840 --height 2160 --spp 64"; N=mixed20k_4k_64spp;;
    random1m)   A="--scene random1m --width 3840 --height 2160 --spp 32"; N=random1m_4k_32spp;;
  esac
  pmc sq_$N SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU -- $A
  pmc cache_$N TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum -- $A
  pmc fetch_$N FETCH_SIZE -- $A
  pmc write_$N WRITE_SIZE -- $A
  # the vector-memory pipeline in front of the caches (address unit and L1 of every CU), what the traversal is bound by
  pmc ta_$N TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum GRBM_GUI_ACTIVE -- $A
  bench_reads $OUT/${T}_pmc_fetch_$N.csv
  timeout -k 10 400 python3 $R/bench.py $A --steps 3 --warmup 1 > $OUT/${T}_bench_$N.json 2> /tmp/bench_$N.err
  echo "bench $N done"
done
timeout -k 10 400 python3 $R/bench.py --steps 10 --warmup 2 > $OUT/${T}_bench_demo1080p.json 2> /tmp/bench_demo.err
echo "bench demo done"
